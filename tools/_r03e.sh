set -u
OUT=$PWD/gpurun_out/r03e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
bash tools/ab_bench.sh r03e "" r02 norot cur
bash tools/ab_bench.sh r03e "--ragged" r02 norot cur
