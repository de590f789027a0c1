#!/bin/bash
# A -DLAMP_TUNING build of the CURRENT sources with extra flags on chosen translation units, for kernel micro-benchmarks:
#     EXTRA="-DCHAIN_ABL=1" bash tools/build_tuning_variant.sh abl1 chain.hip     -> lamp_amd/build/liblamp_tuning_abl1.so
# (the other units are taken from the regular tuning build's objects: run python -m lamp_amd.build first)
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=$ROOT/lamp_amd/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -Wno-unused-result -DLAMP_TUNING ${EXTRA:-}"
OBJS=""
for u in gemm gemm_gen attention attention_tile attention_small attention_general pointwise backward chain slab api; do
  o=$B/$u.tuning.o; [ -f $o ] || o=$B/$u.o
  for v in "$@"; do
    if [ "$v" = "$u.hip" ]; then o=$B/$u.tuning.$NAME.o; /opt/rocm/bin/hipcc $FLAGS -c $ROOT/lamp_amd/csrc/$u.hip -o $o & fi
  done
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/liblamp_tuning_$NAME.so $OBJS
echo $B/liblamp_tuning_$NAME.so
