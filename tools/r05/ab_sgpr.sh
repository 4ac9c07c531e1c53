#!/bin/bash
# Round 5, experiments item 8: spread_rows_kernel held to 80 scalar registers (4 workgroups per CU) against the compiler's own
# allocation (88-94: 3 per CU), one box, interleaved.  needs torch-pme_amd/libmipme_nocap.so (tools/build_variant.sh nocap -DMIPME_SGPR_CAP_OFF)
OUT=$1
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$1  step %.5f ms  launch %.2f us  dE %.1e' % (d['ms_per_step'], 1e3*k['spread+rspace_forward']['ms_per_launch'], d['accuracy']['rel_energy_error']))"; }
for rep in 1 2; do
  MIPME_LIB=$PWD/torch-pme_amd/libmipme_nocap.so python bench.py --preset cfg5 --steps 100 --warmup 10 --no-drop-in --no-cpu-baseline 2>/dev/null | line "cfg5 no cap  " >> $OUT
  python bench.py --preset cfg5 --steps 100 --warmup 10 --no-drop-in --no-cpu-baseline 2>/dev/null | line "cfg5 80 SGPRs" >> $OUT
  MIPME_PLANE_SPREAD=0 MIPME_LIB=$PWD/torch-pme_amd/libmipme_nocap.so python bench.py --steps 300 --warmup 20 --no-drop-in --no-cpu-baseline 2>/dev/null | line "cfg3 bricks no cap  " >> $OUT
  MIPME_PLANE_SPREAD=0 python bench.py --steps 300 --warmup 20 --no-drop-in --no-cpu-baseline 2>/dev/null | line "cfg3 bricks 80 SGPRs" >> $OUT
done
