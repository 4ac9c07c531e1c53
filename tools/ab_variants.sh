#!/bin/bash
# Run ON THE GPU BOX: default library against variant libraries built with tools/build_variant.sh (A/B on one box)
#   bash tools/ab_variants.sh uc4 uc2
for v in default "$@" default; do
  if [ $v = default ]; then unset MIPME_LIB; else export MIPME_LIB=$PWD/torch-pme_amd/libmipme_$v.so; fi
  echo "== $v"
  for p in cfg3 cfg2 cfg5; do
    timeout 200 python bench.py --no-cpu-baseline --no-drop-in --preset $p | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$p', round(d['ms_per_step'],5), d['energy'])"
  done
done
