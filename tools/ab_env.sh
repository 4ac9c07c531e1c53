#!/bin/bash
# Run ON THE GPU BOX: the size sweep's large cases and the BASELINE configurations with an environment switch off / on
#   bash tools/ab_env.sh MIPME_XCD_MAP
VAR=$1
for v in 0 1; do
  echo "== $VAR=$v"
  env $VAR=$v timeout 300 python tools/size_sweep.py 2>&1 | tail -5
  env $VAR=$v bash tools/bench_all.sh
done
