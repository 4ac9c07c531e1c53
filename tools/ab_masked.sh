mkdir -p gpurun_out/r04_g
M=$PWD/torch-pme_amd/libmipme_masked.so
py() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['ms_per_step'],5), round(d['ms_per_step_median'],5), round(d['kernels']['spread+rspace_forward']['ms_per_launch']*1e3,2),'us')
" $1 $2; }
for rep in 1 2; do
  MIPME_LIB=$M python bench.py --no-drop-in --no-cpu-baseline --no-list-refresh > gpurun_out/r04_g/ab_masked_$rep.json 2>/dev/null; py gpurun_out/r04_g/ab_masked_$rep.json cfg3_masked
  python bench.py --no-drop-in --no-cpu-baseline --no-list-refresh > gpurun_out/r04_g/ab_unmasked_$rep.json 2>/dev/null; py gpurun_out/r04_g/ab_unmasked_$rep.json cfg3_unmasked
done
MIPME_LIB=$M python bench.py --preset cfg5 --no-drop-in --no-cpu-baseline --no-list-refresh > gpurun_out/r04_g/ab5_masked.json 2>/dev/null; py gpurun_out/r04_g/ab5_masked.json cfg5_masked
python bench.py --preset cfg5 --no-drop-in --no-cpu-baseline --no-list-refresh > gpurun_out/r04_g/ab5_unmasked.json 2>/dev/null; py gpurun_out/r04_g/ab5_unmasked.json cfg5_unmasked
MIPME_LIB=$M python bench.py --neighbors stream --no-drop-in --no-cpu-baseline --no-list-refresh > gpurun_out/r04_g/abs_masked.json 2>/dev/null; py gpurun_out/r04_g/abs_masked.json live_masked
python bench.py --neighbors stream --no-drop-in --no-cpu-baseline --no-list-refresh > gpurun_out/r04_g/abs_unmasked.json 2>/dev/null; py gpurun_out/r04_g/abs_unmasked.json live_unmasked
echo rows alone:; MIPME_LIB=$M python tools/time_rows.py 2>&1 | tail -4; python tools/time_rows.py 2>&1 | tail -4
