mkdir -p gpurun_out/ab
M=$PWD/torch-pme_amd/libmipme_masked.so
py() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d['kernels']; s=[v for n,v in k.items() if 'rspace' in n or 'spread' in n]
print(sys.argv[2], round(d['ms_per_step'],5), round(d['ms_per_step_median'],5), round(s[0]['ms_per_launch']*1e3,2) if s else None,'us', d['accuracy'].get('rel_energy_error'), d['accuracy'].get('force_rel_l2_error_256_atoms'))
" $1 $2; }
for args in "" "--preset cfg5" "--preset cfg4" "--preset cfg2"; do
  tag=$(echo "$args" | tr -d ' -')
  for rep in 1 2; do
    MIPME_LIB=$M python bench.py $args --no-drop-in --no-cpu-baseline --no-list-refresh > gpurun_out/ab/s_${tag}_old_$rep.json 2>/dev/null; py gpurun_out/ab/s_${tag}_old_$rep.json "${tag:-cfg3}_old"
    python bench.py $args --no-drop-in --no-cpu-baseline --no-list-refresh > gpurun_out/ab/s_${tag}_new_$rep.json 2>/dev/null; py gpurun_out/ab/s_${tag}_new_$rep.json "${tag:-cfg3}_new"
  done
done
echo rows alone; MIPME_LIB=$M python tools/time_rows.py 2>&1 | grep "4-byte"; python tools/time_rows.py 2>&1 | grep "4-byte"
