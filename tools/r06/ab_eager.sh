#!/bin/bash
# Eager call sequences (drop_in block of bench.py) of two trees on ONE box, interleaved: has the host path of round 6 regressed
# against round 5's, or do the boxes differ?   bash tools/r06/ab_eager.sh <other tree>   (e.g. a worktree of 9236903 copied in)
OTHER=$1
FLAGS="--steps 200 --warmup 20 --no-cpu-baseline --no-contract --no-frames-block --no-second-order --no-list-refresh"
py() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); di=d['drop_in']
print(sys.argv[1], 'graph', round(d['ms_per_step'],5), 'drop_in', round(di['ms_per_step'],4), 'python_nodes', round(di.get('ms_per_step_python_nodes',0),4), 'cold_list', round(di.get('cold_list_ms',0),4))" $1; }
for rep in 1 2 3; do
  (cd $OTHER && python bench.py $FLAGS 2>/dev/null | py other)
  python bench.py $FLAGS 2>/dev/null | py head
done
