for v in default w4 uc3 default w4; do
  if [ $v = default ]; then unset MIPME_LIB; else export MIPME_LIB=$PWD/torch-pme_amd/libmipme_$v.so; fi
  echo "== $v"
  timeout 200 python bench.py --no-cpu-baseline --no-drop-in | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['energy'], d['roofline']['kernel_ms'])"
done
