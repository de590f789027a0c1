set -u
OUT=$PWD/gpurun_out/r03g; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log
bash tools/ab_bench.sh r03g "" r02 prev cur
bash tools/ab_bench.sh r03g "--ragged" prev cur
python tools/bench_eval_epoch.py 2>/dev/null | tee $OUT/eval_epoch.json
bash tools/pmc_walk.sh r03g_walk > $OUT/walk.log 2>&1; cat gpurun_out/r03g_walk/walk_traffic.txt
