#!/bin/bash
# Merged plan + gather launch: full GPU suite, then same-box whole-forward A/B against the previous commit's kernels.
set -u
TAG=${1:-r04o}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --maxfail=5 -p no:cacheprovider > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"; tail -6 "$OUT/pytest.log"
AB_STEPS=300 bash tools/ab_bench.sh $TAG "" cur prev
AB_STEPS=300 bash tools/ab_bench.sh ${TAG}_ragged "--ragged" cur prev
bash tools/ab_kernel_stats.sh $TAG "" cur
