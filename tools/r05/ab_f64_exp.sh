#!/bin/bash
# Round 5, experiments item 15: exp(-x) of the fp64 pair body from a 64-entry table + 5 FMAs (default) against the 13-FMA form
# (tools/build_variant.sh exp13 -DMIPME_F64_EXP_TABLE=0) -- one box, interleaved.   usage: ab_f64_exp.sh <out>
OUT=$1
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; a=d['accuracy']
print('$1  step %.5f ms  launch %.2f us  dE %.2e dF %.2e' % (d['ms_per_step'], 1e3*k['spread+rspace_forward']['ms_per_launch'], a['rel_energy_error'], a['force_rel_l2_error_256_atoms']))"; }
for rep in 1 2; do
  for cfg in cfg2 cfg4; do
    MIPME_LIB=$PWD/torch-pme_amd/libmipme_exp13.so python bench.py --preset $cfg --steps 300 --warmup 20 --no-drop-in --no-cpu-baseline 2>/dev/null | line "$cfg 13 FMAs      " >> $OUT
    python bench.py --preset $cfg --steps 300 --warmup 20 --no-drop-in --no-cpu-baseline 2>/dev/null | line "$cfg table + 5 FMAs" >> $OUT
  done
done
