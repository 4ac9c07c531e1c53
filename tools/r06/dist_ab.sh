#!/bin/bash
# Where do the 3 % between `python bench.py` and the same step under torch.distributed.run (one rank, nccl) come from?
# Same box, back to back: plain / bound to cores / process group without the launcher / launcher with and without binding.
FLAGS="--steps 500 --warmup 20 --no-cpu-baseline --no-drop-in --no-contract --no-frames-block --no-second-order --no-list-refresh --no-exchange-sweep"
ms() { grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],5), [round(x,5) for x in d['timing']['blocks_ms_per_step']], d['parallelism']['exchange'], (d['parallelism']['ranks'][0] or {}).get('affinity'))"; }
for rep in 1 2; do
echo "plain            $(python bench.py $FLAGS 2>/dev/null | ms)"
echo "plain bind       $(MIPME_BIND=1 python bench.py $FLAGS 2>/dev/null | ms)"
echo "pg no launcher   $(MIPME_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29711 python bench.py $FLAGS 2>/dev/null | ms)"
echo "pg nobind        $(MIPME_BIND=0 MIPME_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29712 python bench.py $FLAGS 2>/dev/null | ms)"
echo "pg final         $(MIPME_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29713 python bench.py $FLAGS --exchange final 2>/dev/null | ms)"
echo "torchrun         $(MIPME_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29714 bench.py $FLAGS 2>/dev/null | ms)"
echo "torchrun omp8    $(OMP_NUM_THREADS=8 MIPME_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29715 bench.py $FLAGS 2>/dev/null | ms)"
done
