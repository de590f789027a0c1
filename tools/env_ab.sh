#!/bin/bash
# Same-box A/B of runtime environment knobs on the headline bench:   gpurun -- 'bash tools/env_ab.sh <tag> "VAR=a" "VAR=b" ...'   ("-" = nothing set)
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for round in 1 2 3; do
  for kv in "$@"; do
    ( [ "$kv" != "-" ] && export "$kv"; python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extra-workloads --no-pipelined --no-kernel-trace --no-pmc ${AB_FLAGS:-} 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-28s' % sys.argv[1], round(d['value']), d['step_ms_synced']['median'], {k: round(v['us_per_step'],1) for k,v in d['kernels'].items()})" "$kv" ) | tee -a $OUT/env_ab.txt
  done
done
