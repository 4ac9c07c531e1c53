#!/bin/bash
# usage (on the GPU box): bash tools/variant_bench.sh <lib suffixes...>   -> per-call ms of the pair kernels per variant
for v in "$@"; do
  MIPME_LIB=$PWD/torch-pme_amd/libmipme_$v.so python bench.py --launch eager --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | V=$v python -c '
import json,sys,os; d=json.loads(sys.stdin.read()); print(os.environ["V"], {k: round(v*1000,1) for k,v in d["kernel_ms"].items() if "kspace" not in k})'
done
