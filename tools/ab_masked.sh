# A/B of the unmasked main loop of the pair bodies on ONE box: libmipme_masked.so = tools/build_variant.sh masked -DMIPME_ROWS_UNMASKED=0
mkdir -p gpurun_out/ab
M=$PWD/torch-pme_amd/libmipme_masked.so
py() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d['kernels']; s=[v for n,v in k.items() if 'rspace' in n or 'spread' in n]
print(sys.argv[2], round(d['ms_per_step'],5), round(d['ms_per_step_median'],5), round(s[0]['ms_per_launch']*1e3,2) if s else None,'us', d['accuracy'].get('rel_energy_error'))
" $1 $2; }
for preset in ${PRESETS:-cfg2 cfg4}; do
  for rep in 1 2; do
    MIPME_LIB=$M python bench.py --preset $preset --no-drop-in --no-cpu-baseline --no-list-refresh > gpurun_out/ab/${preset}_masked_$rep.json 2>/dev/null; py gpurun_out/ab/${preset}_masked_$rep.json ${preset}_masked
    python bench.py --preset $preset --no-drop-in --no-cpu-baseline --no-list-refresh > gpurun_out/ab/${preset}_unmasked_$rep.json 2>/dev/null; py gpurun_out/ab/${preset}_unmasked_$rep.json ${preset}_unmasked
  done
done
