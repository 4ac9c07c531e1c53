#!/bin/bash
# Build torch-pme_amd/libmipme_<name>.so with extra compiler flags (kernel experiments; run with MIPME_LIB=<path>).
#   bash tools/build_variant.sh lanes32 -DMIPME_ROW_LANES=32
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/torch-pme_amd/csrc
OBJ=/tmp/mipme_variant_$NAME
mkdir -p $OBJ
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function $*"
pids=()
for f in api mesh kfilter rspace topology bricks frames live neighbors ewald jets; do
  /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f.hip -o $OBJ/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 $OBJ/*.o -shared -L/opt/rocm/lib -lhipfft -o $ROOT/torch-pme_amd/libmipme_$NAME.so
echo built $ROOT/torch-pme_amd/libmipme_$NAME.so
