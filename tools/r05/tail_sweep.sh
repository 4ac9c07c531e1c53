#!/bin/bash
# Round 5: A/B of the row tail (MIPME_ROWS_TAIL = 64-row blocks of the pair sum that ride on the convolution's inverse plane
# launch) and of the plane workgroups' issue priority (MIPME_PLANE_PRIO) -- one box, interleaved.  usage: tail_sweep.sh <out> [preset]
OUT=$1; PRESET=${2:-cfg3}
for rep in 1 2; do
for cfg in "0 0" "0 1" "56 0" "56 1" "96 1" "128 1" "160 1" "224 1"; do
  set -- $cfg
  MIPME_ROWS_TAIL=$1 MIPME_PLANE_PRIO=$2 python bench.py --preset $PRESET --steps 300 --warmup 20 --no-drop-in --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('tail $1 prio $2  step %.5f ms  launch %.2f us  conv %.2f  bin %.2f  gather %.2f   dE %.1e dF %.1e' % (d['ms_per_step'], 1e3*k['spread+rspace_forward']['ms_per_launch'], 1e3*k['convolve_xfused']['ms_per_launch'], 1e3*k['bin_atoms']['ms_per_launch'], 1e3*[v for n,v in k.items() if n.startswith('gather')][0]['ms_per_launch'], d['accuracy']['rel_energy_error'], d['accuracy']['force_rel_l2_error_256_atoms']))" >> $OUT
done; done
