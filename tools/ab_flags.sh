#!/bin/bash
# Same-box A/B of whole-forward throughput between bench.py flag sets (one library), three round-robin rounds:
#     gpurun -- 'bash tools/ab_flags.sh <tag> "<common flags>" "<flags A>" "<flags B>" ...'      ("" = no extra flag)
set -u
TAG=$1; COMMON=$2; shift 2
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for round in 1 2 3; do
  for flags in "$@"; do
    python bench.py --steps ${AB_STEPS:-300} --warmup ${AB_WARMUP:-20} --no-cpu-baseline --no-extra-workloads --no-pipelined --no-kernel-trace --no-pmc $COMMON $flags 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-22s' % (sys.argv[1] or '(default)'), '%-18s' % sys.argv[2], round(d['value']), {k: round(v['us_per_step'],1) for k,v in d['kernels'].items()})" "$flags" "$COMMON" | tee -a $OUT/ab_flags.txt
  done
done
