#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash tools/trace_gpu.sh <tag> [bench args...]
# rocprofv3 kernel trace of a short bench run -> gpurun_out/<tag>_kernel_stats.txt (per-kernel avg durations)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o t -- python $ROOT/bench.py --no-cpu-baseline --no-drop-in "$@" > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.err
cd $ROOT
python tools/rocpd_summary.py $(find $OUT/${TAG}_trace -name '*.db') > $OUT/${TAG}_kernel_stats.txt
rm -rf $OUT/${TAG}_trace
