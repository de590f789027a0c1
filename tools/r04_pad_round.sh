#!/bin/bash
# LDS row padding 4 -> 8 floats for the 16x16x4 fragment layouts (GEMM, small-shape attention): same-box A/B of two builds.
set -u
TAG=${1:-r04q}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
B=$PWD/lamp_amd/build
{ echo "# GEMM shapes, heuristic tile: LDS row padding 4 floats vs 8 floats"; LAMP_HIP_LIBRARY=$B/liblamp_pad4.so timeout 600 python tools/bench_kernels.py lib_ab $B/liblamp_pad4.so $B/liblamp_pad8.so 2>&1 | grep -v amdgpu.ids
  echo "# attention shapes, heuristic variant: padding 4 vs 8"; LAMP_HIP_LIBRARY=$B/liblamp_pad4.so timeout 600 python tools/bench_kernels.py attn_lib_ab $B/liblamp_pad4.so $B/liblamp_pad8.so 2>&1 | grep -v amdgpu.ids; } > "$OUT/lds_pad.txt"
cat "$OUT/lds_pad.txt"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "linear or sdpa or mha or model_golden or chain" -p no:cacheprovider 2>&1 | tail -3
AB_STEPS=300 bash tools/ab_bench.sh $TAG "" pad4 pad8
AB_STEPS=200 bash tools/ab_bench.sh ${TAG}_ragged "--ragged" pad4 pad8
