#!/bin/bash
# experiment: lanes per row of the pair kernels (standalone + co-scheduled), and graph-branch overlap
cd $GRAFT_REPO_ROOT
for v in default lanes8 lanes32 lanes64; do
  if [ $v = default ]; then L=$PWD/torch-pme_amd/libmipme.so; else L=$PWD/torch-pme_amd/libmipme_$v.so; fi
  echo "== $v"
  MIPME_LIB=$L python tools/time_fused.py 2>/dev/null | tail -5
  MIPME_LIB=$L python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-drop-in 2>/dev/null | python -c '
import json,sys; d=json.loads(sys.stdin.read()); print("bench ms/step", round(d["ms_per_step"],5), {k: v["ms_per_launch"] for k,v in d["kernels"].items()})'
done
echo "== overlap (graph branches)"
MIPME_OVERLAP=1 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-drop-in 2>/dev/null | python -c '
import json,sys; d=json.loads(sys.stdin.read()); print("bench ms/step", round(d["ms_per_step"],5))'
MIPME_COSCHEDULE=0 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-drop-in 2>/dev/null | python -c '
import json,sys; d=json.loads(sys.stdin.read()); print("no-coschedule bench ms/step", round(d["ms_per_step"],5), {k: v["ms_per_launch"] for k,v in d["kernels"].items()})'
