#!/bin/bash
# Round-4 GEMM session on one GPU box: direct-to-LDS staging variants (correctness, then same-process A/B), the MFMA issue
# micro-benchmark with in-kernel clock measurement, the clock the product GEMMs run at.   Output -> gpurun_out/<tag>/
set -u
TAG=${1:-r04a}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "every_tile_config or direct_to_lds or linear_vs_torch" -p no:cacheprovider > "$OUT/pytest_gemm.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest_gemm.log"; tail -5 "$OUT/pytest_gemm.log"
timeout 300 tools/probes/mfma_issue 20 > "$OUT/mfma_issue.txt" 2>&1; cat "$OUT/mfma_issue.txt"
timeout 900 python tools/bench_kernels.py gemm_ab ${2:-0,9,20,40,28,48,11,21,41,12,22,42,18,23,43,24,44,15,25,45} 2>&1 | grep -v amdgpu.ids > "$OUT/gemm_tiles_dma.txt"; cat "$OUT/gemm_tiles_dma.txt"
timeout 600 python tools/bench_kernels.py gemm_clock ${3:-0,23,43} 2>&1 | grep -v amdgpu.ids > "$OUT/gemm_clock.txt"; cat "$OUT/gemm_clock.txt"
