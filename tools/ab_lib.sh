# A/B of two builds of the library on ONE box: torch-pme_amd/libmipme_<name>.so (tools/build_variant.sh, or a build of another
# tree copied there) against the in-tree libmipme.so.   bash tools/ab_lib.sh old   [PRESETS="cfg3 cfg5" REPS=2]
NAME=$1
mkdir -p gpurun_out/ab
M=$PWD/torch-pme_amd/libmipme_$NAME.so
py() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d['kernels']; s=[v for n,v in k.items() if 'rspace' in n or 'spread' in n]
print(sys.argv[2], round(d['ms_per_step'],5), round(d['ms_per_step_median'],5), round(s[0]['ms_per_launch']*1e3,2) if s else None,'us', d['accuracy'].get('rel_energy_error'), d['accuracy'].get('force_rel_l2_error_256_atoms', ''))
" $1 $2; }
for preset in ${PRESETS:-cfg3 cfg5 cfg4 cfg2}; do
  for rep in $(seq 1 ${REPS:-2}); do
    MIPME_LIB=$M python bench.py --preset $preset --no-drop-in --no-cpu-baseline --no-list-refresh > gpurun_out/ab/${preset}_${NAME}_$rep.json 2>/dev/null; py gpurun_out/ab/${preset}_${NAME}_$rep.json ${preset}_${NAME}
    python bench.py --preset $preset --no-drop-in --no-cpu-baseline --no-list-refresh > gpurun_out/ab/${preset}_head_$rep.json 2>/dev/null; py gpurun_out/ab/${preset}_head_$rep.json ${preset}_head
  done
done
