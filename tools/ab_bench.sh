#!/bin/bash
# Same-box A/B of whole-forward throughput between library builds (tools/build_variant.sh):
#     gpurun -- 'bash tools/ab_bench.sh <tag> "<bench flags>" name1 name2 ...'      (name "cur" = lamp_amd/liblamp_hip.so)
set -u
TAG=$1; FLAGS=$2; shift 2
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for round in 1 2 3; do
  for name in "$@"; do
    base=${name%+single}; single=0; [ "$base" != "$name" ] && single=1     # "<name>+single": LAMP_SINGLE_STREAM=1 (no side lane)
    lib=$PWD/lamp_amd/build/liblamp_$base.so; [ "$base" = cur ] && lib=$PWD/lamp_amd/liblamp_hip.so
    LAMP_SINGLE_STREAM=$single LAMP_HIP_LIBRARY=$lib python bench.py --steps ${AB_STEPS:-300} --warmup ${AB_WARMUP:-20} --no-cpu-baseline --no-extra-workloads --no-pipelined --no-kernel-trace --no-pmc $FLAGS 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s' % sys.argv[1], '%-10s' % sys.argv[2], round(d['value']), {k: round(v['us_per_step'],1) for k,v in d['kernels'].items()})" "$name" "$FLAGS" | tee -a $OUT/ab_bench.txt
  done
done
