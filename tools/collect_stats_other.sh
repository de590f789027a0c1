#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries of the three non-headline BASELINE configurations (kernel-only durations
# behind the `workloads` entries of the bench line).   gpurun --timeout 900 -- 'bash tools/collect_stats_other.sh r02'
set -u
TAG=${1:-rXX}
OUT=$PWD/gpurun_out/${TAG}_stats_other
mkdir -p "$OUT"
export TMPDIR=/tmp
for spec in "bibtex 50 5" "delicious 10 3" "synthetic4096 3 1"; do
  set -- $spec
  ( cd /tmp && timeout 250 rocprofv3 --kernel-trace --stats -d "$OUT/$1" -o p -f csv -- python $GRAFT_REPO_ROOT/bench.py --workload $1 \
        --steps $2 --warmup $3 --no-cpu-baseline --no-pipelined --no-extra-workloads > "$OUT/bench_$1.json" 2>/dev/null ); echo "$1 rc=$?"
  cp "$OUT/$1/p_kernel_stats.csv" "$OUT/kernel_stats_$1.csv" 2>/dev/null
done
ls "$OUT"
