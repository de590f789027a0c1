#!/bin/bash
# s_setprio(1) around the MFMA runs of the GEMM and attention kernels: same-box A/B of two builds of the same sources
# (tools/build_variant.sh . base ; EXTRA=-DLAMP_SETPRIO=1 tools/build_variant.sh . setprio).  Output -> gpurun_out/<tag>/
set -u
TAG=${1:-r04c}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
B=$PWD/lamp_amd/build
{ echo "# GEMM shapes, heuristic tile: base vs -DLAMP_SETPRIO"; LAMP_HIP_LIBRARY=$B/liblamp_base.so timeout 600 python tools/bench_kernels.py lib_ab $B/liblamp_base.so $B/liblamp_setprio.so 2>&1 | grep -v amdgpu.ids
  echo "# attention shapes, heuristic variant: base vs -DLAMP_SETPRIO"; LAMP_HIP_LIBRARY=$B/liblamp_base.so timeout 600 python tools/bench_kernels.py attn_lib_ab $B/liblamp_base.so $B/liblamp_setprio.so 2>&1 | grep -v amdgpu.ids; } > "$OUT/setprio.txt"
cat "$OUT/setprio.txt"
AB_STEPS=200 bash tools/ab_bench.sh $TAG "" base setprio
