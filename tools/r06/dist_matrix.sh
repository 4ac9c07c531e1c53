#!/bin/bash
# One-rank RCCL pass of bench.py's distributed path on the GPU box (round-5 verdict item 1):
#   bash tools/r06/dist_matrix.sh [N]        (default N = 1; MIPME_FORCE_DIST=1 makes one rank take the nccl path)
# For {cfg3 single frame, --preset cfg4} it runs bench.py twice under torch.distributed.run with the nccl backend:
#   default exchange (log) -- the run also times one block each of none / per-step / pipelined / final (same clocks) --
#   and --exchange in-graph (the per-step collective captured into the step's HIP graph).
# Output: gpurun_out/r06_dist_${N}rank.txt (one table) + the raw JSON lines next to it.
N=${1:-1}
OUT=gpurun_out/r06_dist_${N}rank
mkdir -p gpurun_out
: > $OUT.jsonl
run() {
  env MIPME_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
      --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus $N --steps 500 --warmup 20 --no-cpu-baseline \
      --no-drop-in --no-contract --no-frames-block --no-second-order --no-list-refresh --exchange-sweep full "$@" 2>>$OUT.err | grep '^{' >> $OUT.jsonl
  echo "rc=$? $*" >> $OUT.err
}
run
run --exchange in-graph
run --preset cfg4
run --preset cfg4 --exchange in-graph
python - "$OUT" "$N" <<'PY'
import json, sys
out, n = sys.argv[1], sys.argv[2]
rows = [json.loads(l) for l in open(out + ".jsonl") if l.startswith("{")]
with open(out + ".txt", "w") as f:
    f.write(f"# bench.py under torch.distributed.run, backend nccl (RCCL), {n} rank(s); ms per step, MAX over ranks\n")
    f.write(f"# {'workload':<28}{'exchange':<12}{'ms/step':>10}   other protocols timed in the same invocation (one block each)\n")
    for r in rows:
        p = r["parallelism"]
        wl = f"{r['config']['workload'][:18]} x{r['config'].get('frames_per_gpu', 1)}"
        f.write(f"  {wl:<28}{p['exchange']:<12}{r['ms_per_step']:>10.5f}   {json.dumps(p.get('other_exchange_modes_ms_per_step'))}"
                f"  energies_sum={p.get('energies_sum')}\n")
print(open(out + ".txt").read())
PY
