#!/bin/bash
# Round-4 decoder-chain session: bit-identity tests, then same-box whole-forward A/B (chain on = cur, off = a build with
# the row-count rule disabled), then the per-phase timeline.   Output -> gpurun_out/<tag>/
set -u
TAG=${1:-r04d}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "decoder_chain or layernorm or model_golden" -p no:cacheprovider > "$OUT/pytest_chain.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest_chain.log"; tail -8 "$OUT/pytest_chain.log"
AB_STEPS=300 bash tools/ab_bench.sh $TAG "" ${2:-cur nochain}
AB_STEPS=300 bash tools/ab_bench.sh ${TAG}_ragged "--ragged" ${2:-cur nochain}
timeout 300 python tools/bench_kernels.py chain 2>&1 | grep -v amdgpu.ids > "$OUT/chain.txt"; cat "$OUT/chain.txt"
