#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05c; mkdir -p $OUT; export TMPDIR=/tmp
for a in "$@"; do
  echo "== $a" | tee -a $OUT/chain_variants.txt
  for tg in 7 8; do
  LAMP_HIP_LIBRARY=$PWD/lamp_amd/build/liblamp_tuning_$a.so timeout 300 python tools/bench_kernels.py chain 2880 $tg 0,7,8,9,10 2>&1 | grep -v amdgpu.ids | grep "bitwise\|chain launch\|phase\|clock\|timeline" | tee -a $OUT/chain_variants.txt
  done
done
