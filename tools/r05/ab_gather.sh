#!/bin/bash
# Round 5, experiments item 10: the fp32 gather + tail kernels compiled for 6 waves per SIMD (80 vector registers: three 512-thread
# workgroups per CU; default) against the compiler's own 84-93 (two) -- one box, interleaved.
# needs torch-pme_amd/libmipme_gather1.so (tools/build_variant.sh gather1 -DMIPME_GATHER_TAIL_WAVES=1)    usage: ab_gather.sh <out>
OUT=$1
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; f=d.get('frames') or {}
fr=(f.get('one_launch_per_kernel') or {}).get('8', {}).get('ms_per_step')
print('$1  step %.5f ms  gather %.2f us  dE %.1e%s' % (d['ms_per_step'], 1e3*k['gather+energy+forces']['ms_per_launch'], d['accuracy']['rel_energy_error'], ('   8 frames in one launch %.4f ms' % fr) if fr else ''))"; }
for rep in 1 2; do
  for nb in list stream; do
    MIPME_LIB=$PWD/torch-pme_amd/libmipme_gather1.so python bench.py --preset cfg5 --neighbors $nb --steps 100 --warmup 10 --no-drop-in --no-cpu-baseline 2>/dev/null | line "cfg5 $nb own registers " >> $OUT
    python bench.py --preset cfg5 --neighbors $nb --steps 100 --warmup 10 --no-drop-in --no-cpu-baseline 2>/dev/null | line "cfg5 $nb 6 waves       " >> $OUT
  done
  MIPME_LIB=$PWD/torch-pme_amd/libmipme_gather1.so python bench.py --steps 300 --warmup 20 --no-drop-in --no-cpu-baseline 2>/dev/null | line "cfg3 own registers " >> $OUT
  python bench.py --steps 300 --warmup 20 --no-drop-in --no-cpu-baseline 2>/dev/null | line "cfg3 6 waves       " >> $OUT
done
# eight 31 944-atom fp32 frames in one launch per kernel (frames_gather_tail_kernel: 88 -> 80 registers + 20 bytes of scratch)
for rep in 1 2; do
  MIPME_LIB=$PWD/torch-pme_amd/libmipme_gather1.so python bench.py --frames-per-gpu 8 --steps 100 --warmup 10 --no-drop-in --no-cpu-baseline 2>/dev/null | line "8 water frames own registers " >> $OUT
  python bench.py --frames-per-gpu 8 --steps 100 --warmup 10 --no-drop-in --no-cpu-baseline 2>/dev/null | line "8 water frames 6 waves       " >> $OUT
done
