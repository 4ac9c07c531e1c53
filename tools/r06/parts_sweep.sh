#!/bin/bash
# plane-spread parts per plane with the streamed entries (round 6): cfg3 graph step, same box
FLAGS="--steps 500 --warmup 20 --no-cpu-baseline --no-drop-in --no-contract --no-frames-block --no-second-order --no-list-refresh"
for rep in 1 2; do
for p in 1 2 3 4; do
  echo "parts $p: $(MIPME_PLANE_PARTS=$p python bench.py $FLAGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],5), {k:round(x['ms_per_launch']*1e3,2) for k,x in d['kernels'].items()})")"
done
done
