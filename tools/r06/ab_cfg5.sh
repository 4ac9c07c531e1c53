#!/bin/bash
# cfg5 (262 144 atoms, 128^3, 1/r^6): planes spread in BANDS of 32 rows co-scheduled with the pair sum (+ a y-column launch)
# against the owner-computes bricks + forward plane launch (MIPME_PLANE_BANDS=0), same box, interleaved
FLAGS="--preset cfg5 --steps 100 --warmup 10 --no-cpu-baseline --no-drop-in --no-contract --no-frames-block --no-second-order --no-list-refresh"
py() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], round(d['ms_per_step'],5), d['roofline'].get('kernel_name'), {k:round(x['ms_per_launch']*1e3,1) for k,x in d['kernels'].items()}, d['accuracy'].get('rel_energy_error'), d['accuracy'].get('force_rel_l2_error_256_atoms'))" $1; }
for rep in 1 2 3; do
  MIPME_PLANE_BANDS=0 python bench.py $FLAGS 2>/dev/null | py bricks
  python bench.py $FLAGS 2>/dev/null | py bands
done
