py() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d['kernels']; s=[v for n,v in k.items() if 'rspace' in n or 'spread' in n]
print(sys.argv[2], round(d['ms_per_step'],5), round(d['ms_per_step_median'],5), round(s[0]['ms_per_launch']*1e3,2) if s else None,'us', d['accuracy'].get('rel_energy_error'), d['accuracy'].get('force_rel_l2_error_256_atoms', ''))
" $1 $2; }
mkdir -p gpurun_out/ab
for preset in cfg5 cfg3; do for rep in 1 2; do for a in 0 1 2 3; do
  MIPME_BRICK_PATTERN=$a python bench.py --preset $preset --no-drop-in --no-cpu-baseline --no-list-refresh > gpurun_out/ab/${preset}_pat${a}_$rep.json 2>/dev/null; py gpurun_out/ab/${preset}_pat${a}_$rep.json ${preset}_pattern$a
done; done; done
