#!/bin/bash
# One GPU-box session: tests, a short bench, kernel micro-benchmarks, the GEMM timeline.  Output -> gpurun_out/<tag>/
#     gpurun --timeout 2400 -- 'bash tools/gpu_round.sh r02a [tests|notests]'
set -u
TAG=${1:-rXX}
WHAT=${2:-tests}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "$WHAT" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 --durations=12 -p no:cacheprovider > "$OUT/pytest.log" 2>&1
  echo "pytest rc=$?" >> "$OUT/pytest.log"
  tail -40 "$OUT/pytest.log"
fi
timeout 600 python bench.py --steps 100 --warmup 10 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
tail -c 1500 "$OUT/bench.err"
timeout 300 python tools/bench_kernels.py attn 2>&1 | grep -v amdgpu.ids > "$OUT/attn_variants.txt"
timeout 300 python tools/bench_kernels.py gemm_trace 2>&1 | grep -v amdgpu.ids > "$OUT/gemm_trace.txt"
timeout 300 python tools/bench_kernels.py gemm_ab 2>&1 | grep -v amdgpu.ids > "$OUT/gemm_tiles.txt"
head -c 3000 "$OUT/bench.json"; echo
cat "$OUT/attn_variants.txt" | head -70
cat "$OUT/gemm_trace.txt"
