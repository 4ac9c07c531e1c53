#!/bin/bash
# Run ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE calibration (tools/fetch_calib.hip) -> gpurun_out/<tag>_fetch_calib.txt
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $ROOT/tools/fetch_calib.hip 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_cf -o f -- /tmp/fetch_calib > $OUT/${TAG}_calib_expected.txt 2> $OUT/${TAG}_cf.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_cw -o w -- /tmp/fetch_calib > /dev/null 2> $OUT/${TAG}_cw.err
cd $ROOT
python tools/rocpd_summary.py $(find $OUT/${TAG}_cf -name '*.db') $(find $OUT/${TAG}_cw -name '*.db') > $OUT/${TAG}_fetch_calib.txt
cat $OUT/${TAG}_calib_expected.txt >> $OUT/${TAG}_fetch_calib.txt
rm -rf $OUT/${TAG}_cf $OUT/${TAG}_cw
