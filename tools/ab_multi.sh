# Several builds of the library on ONE box:  bash tools/ab_multi.sh old idx brk   (torch-pme_amd/libmipme_<name>.so each, then the
# in-tree libmipme.so as "head"); PRESETS / REPS as in tools/ab_lib.sh
mkdir -p gpurun_out/ab
py() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d['kernels']; s=[v for n,v in k.items() if 'rspace' in n or 'spread' in n]
print(sys.argv[2], round(d['ms_per_step'],5), round(d['ms_per_step_median'],5), round(s[0]['ms_per_launch']*1e3,2) if s else None,'us', d['accuracy'].get('rel_energy_error'), d['accuracy'].get('force_rel_l2_error_256_atoms', ''))
" $1 $2; }
for preset in ${PRESETS:-cfg3 cfg2}; do
  for rep in $(seq 1 ${REPS:-2}); do
    for name in "$@" head; do
      if [ $name = head ]; then L=$PWD/torch-pme_amd/libmipme.so; else L=$PWD/torch-pme_amd/libmipme_$name.so; fi
      MIPME_LIB=$L python bench.py --preset $preset --no-drop-in --no-cpu-baseline --no-list-refresh ${BENCH_ARGS:-} > gpurun_out/ab/${preset}_${name}_$rep.json 2>/dev/null; py gpurun_out/ab/${preset}_${name}_$rep.json ${preset}_${name}
    done
  done
done
