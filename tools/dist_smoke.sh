#!/bin/bash
# 1-rank RCCL smoke test of bench.py's distributed path (run on the GPU box): torchrun + nccl process group + HIP graph
run() {
  echo "== $*"
  env MIPME_FORCE_DIST=1 MIPME_BENCH_DEBUG=1 "$@" timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --launch ${LAUNCH:-graph} 2>gpurun_out/dist_tmp.err | grep '^{' | python -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t); print('OK', d['n_gpus'], d['config']['launch'], d['ms_per_step'], d['value'])
except Exception as e:
    print('FAILED', t[:200])"
  grep -E "fault|Error" gpurun_out/dist_tmp.err | head -4
}
LAUNCH=graph run A=1
LAUNCH=eager run A=1
