set -u
OUT=$PWD/gpurun_out/r03f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log | head -3
bash tools/ab_bench.sh r03f "" r02 prev cur
