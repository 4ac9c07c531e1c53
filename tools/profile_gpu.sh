#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash tools/profile_gpu.sh <tag> [bench args...]
# Collects a kernel trace and two separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950),
# and writes text summaries to gpurun_out/<tag>_*.txt (copy the ones to keep into profiles/).
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o t -- python $ROOT/bench.py --no-cpu-baseline --no-drop-in "$@" > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o f -- python $ROOT/bench.py --no-cpu-baseline --no-drop-in "$@" --steps 5 --warmup 2 > /dev/null 2> $OUT/${TAG}_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_write -o w -- python $ROOT/bench.py --no-cpu-baseline --no-drop-in "$@" --steps 5 --warmup 2 > /dev/null 2> $OUT/${TAG}_write.err
cd $ROOT
python tools/rocpd_summary.py $(find $OUT/${TAG}_trace -name '*.db') > $OUT/${TAG}_kernel_stats.txt
python tools/rocpd_summary.py $(find $OUT/${TAG}_fetch -name '*.db') $(find $OUT/${TAG}_write -name '*.db') > $OUT/${TAG}_pmc.txt
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_fetch $OUT/${TAG}_write
tail -2 $OUT/${TAG}_trace.err
