set -u
OUT=$PWD/gpurun_out/r03h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|^FAILED" $OUT/pytest.log | head -30
bash tools/ab_bench.sh r03h "" prev cur
