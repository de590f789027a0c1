#!/bin/bash
# Same-box per-kernel A/B: rocprofv3 --kernel-trace --stats of the bench command under each library build.
#     gpurun -- 'bash tools/ab_kernel_stats.sh <tag> "<bench flags>" name1 name2 ...'     (name "cur" = lamp_amd/liblamp_hip.so)
set -u
TAG=$1; FLAGS=$2; shift 2
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for name in "$@"; do
  lib=$PWD/lamp_amd/build/liblamp_$name.so; [ "$name" = cur ] && lib=$PWD/lamp_amd/liblamp_hip.so
  ( cd /tmp && LAMP_HIP_LIBRARY=$lib rocprofv3 --kernel-trace --stats -d "$OUT/stats_$name" -o p -f csv -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extra-workloads --no-pipelined --no-pmc $FLAGS > /dev/null 2>&1 )
  python - "$OUT/stats_$name/p_kernel_stats.csv" "$name" <<'PY' | tee -a $OUT/ab_kernel_stats.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'lamp::' in r['Name']]
fw = sum(int(r['Calls']) for r in rows if 'seq_plan' in r['Name'] or 'embed_plan' in r['Name']) or sum(int(r['Calls']) for r in rows if 'embed_kernel' in r['Name'])
tot = 0.0
print('== %s (%d forwards)' % (sys.argv[2], fw))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    per = float(r['TotalDurationNs']) / fw / 1e3
    tot += per
    print('  %-58s x%5.2f  avg %8.2f us   %8.2f us/forward' % (r['Name'].split('(')[0].replace('void lamp::', '').replace('lamp::', '')[:58], int(r['Calls']) / fw, float(r['AverageNs']) / 1e3, per))
print('  total %.1f us/forward' % tot)
PY
done
