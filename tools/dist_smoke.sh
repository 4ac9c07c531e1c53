#!/bin/bash
# RCCL smoke test of bench.py's distributed path (run on a GPU box):  bash tools/dist_smoke.sh [N]   (default N = 1)
# torchrun + nccl process group + HIP graph replay + the energy-log exchange (default) or a per-step all-gather, N ranks on N GPUs of one node; with N = 8 and
# `--preset cfg4` this is BASELINE.json configs[3] (64 frames, 8 per GPU).  One line per run: OK n_gpus launch ms/step value
# weak_efficiency.
N=${1:-1}
mkdir -p gpurun_out
run() {
  echo "== N=$N $*"
  env MIPME_FORCE_DIST=1 MIPME_BENCH_DEBUG=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
      --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus $N --steps 100 --warmup 10 --no-cpu-baseline \
      --no-drop-in "$@" 2>gpurun_out/dist_tmp.err | grep '^{' | python -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads(t); print('OK', d['n_gpus'], d['config']['launch'], d['ms_per_step'], d['value'], (d.get('weak_efficiency') or {}).get('value'))
except Exception as e:
    print('FAILED', t[:200])"
  grep -E "fault|Error" gpurun_out/dist_tmp.err | head -4
}
run --launch graph
run --launch eager
if [ "$N" -gt 1 ]; then
  run --launch graph --preset cfg4
  run --launch graph --exchange pipelined
fi
