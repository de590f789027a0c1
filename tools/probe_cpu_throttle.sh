#!/bin/bash
# Evaluation epoch (tools/bench_eval_epoch.py) three times with the intra-op pool fitted to the cgroup quota (lamp_amd/hostcpu.py),
# once with torch's default pool (LAMP_EVAL_NO_FIT=1), then bench.py; the cgroup's throttle counters after each -> gpurun_out/r06thr2/
mkdir -p gpurun_out/r06thr2
thr() { grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; }
{
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max)  nproc: $(nproc)"
echo "before: $(thr)"
for i in 1 2 3; do
  timeout 300 python tools/bench_eval_epoch.py > gpurun_out/r06thr2/fitted_$i.json 2>/dev/null
  echo "after fitted run $i: $(thr)"
done
LAMP_EVAL_NO_FIT=1 timeout 300 python tools/bench_eval_epoch.py > gpurun_out/r06thr2/unfitted.json 2>/dev/null
echo "after unfitted run: $(thr)"
python bench.py > gpurun_out/r06thr2/bench.json 2>gpurun_out/r06thr2/bench.err
echo "after bench.py: $(thr)"
} > gpurun_out/r06thr2/throttle.txt 2>&1
