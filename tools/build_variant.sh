#!/bin/bash
# Build the product library from the kernel sources of another git revision (or of a directory) into
# lamp_amd/build/liblamp_<name>.so, for same-box A/B runs:
#     bash tools/build_variant.sh HEAD~1 prev        # csrc/ and include/ as of HEAD~1
#     LAMP_HIP_LIBRARY=$PWD/lamp_amd/build/liblamp_prev.so python bench.py ...
# (bench.py / tools/bench_kernels.py lib_ab alternate between such libraries in one gpurun call: box-to-box variance
# is 3-4 %, larger than most kernel changes.)
set -eu
REV=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p "$TMP/lamp_amd/csrc" "$TMP/include" "$ROOT/lamp_amd/build"
if [ -d "$REV" ]; then cp "$REV"/lamp_amd/csrc/* "$TMP/lamp_amd/csrc/"; cp "$REV"/include/* "$TMP/include/";
else
  for f in $(git -C "$ROOT" ls-tree --name-only "$REV" lamp_amd/csrc/ include/); do git -C "$ROOT" show "$REV:$f" > "$TMP/$f"; done
fi
# EXTRA: additional compiler flags of the variant, e.g. EXTRA="-DLAMP_SETPRIO=1" bash tools/build_variant.sh . setprio
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -Wno-unused-result ${EXTRA:-}"
OBJS=""
for src in "$TMP"/lamp_amd/csrc/*.hip; do
  o="$TMP/$(basename "$src" .hip).o"; /opt/rocm/bin/hipcc $FLAGS -c "$src" -o "$o" & OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/lamp_amd/build/liblamp_$NAME.so" $OBJS
rm -rf "$TMP"
echo "$ROOT/lamp_amd/build/liblamp_$NAME.so"
