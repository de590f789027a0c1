#!/bin/bash
# Deferred / grouped weight gradients: tests, same-box A/B of the training step, kernel stats.
set -u
TAG=${1:-r05t}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_training.py -m gpu -q --maxfail=20 -p no:cacheprovider > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"; tail -15 "$OUT/pytest.log"
for rep in 1 2; do
  python tools/bench_train.py --no-defer --per-launch > "$OUT/train_nodefer_perlaunch_$rep.json" 2>/dev/null
  python tools/bench_train.py --per-launch > "$OUT/train_defer_perlaunch_$rep.json" 2>/dev/null
  python tools/bench_train.py > "$OUT/train_defer_$rep.json" 2>/dev/null
done
python tools/bench_train.py --host-profile "$OUT/host_profile.txt" > /dev/null 2>&1
python tools/bench_train.py --fused-adam > "$OUT/train_defer_fused_adam.json" 2>/dev/null
python tools/bench_train.py --workload delicious --steps 5 --warmup 2 > "$OUT/train_delicious.json" 2>/dev/null
python tools/bench_train.py --workload delicious --steps 5 --warmup 2 --no-defer > "$OUT/train_delicious_nodefer.json" 2>/dev/null
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/train_stats" -o p -f csv -- python $REPO/tools/bench_train.py --steps 20 > /dev/null 2>&1 )
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + '/train_*.json')):
    try:
        d = json.load(open(f)); print(os.path.basename(f), round(d['ms_per_step'], 3), 'ms', round(d['value']), d['synchronised_split_ms'])
    except Exception as e:
        print(f, 'unreadable', e)
PY
find "$OUT/train_stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/train_kernel_stats.csv"
head -12 "$OUT/train_kernel_stats.csv" | cut -c1-180
