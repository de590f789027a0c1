#!/bin/bash
set -u
OUT=$PWD/gpurun_out/${1:-r05b}; shift; mkdir -p $OUT; export TMPDIR=/tmp
G=${GEOMS:-12}
for a in "$@"; do
  echo "== $a (bits: 1 = no W traffic, 2 = no MFMAs, 8 = no A-fragment reads, 16 = no W loads issued)" | tee -a $OUT/chain_abl.txt
  LAMP_HIP_LIBRARY=$PWD/lamp_amd/build/liblamp_tuning_$a.so timeout 300 python tools/bench_kernels.py chain 2880 $G $G 2>&1 | grep -v amdgpu.ids | grep "chain launch\|phase\|clock\|wave 0\|^fc \|^W1 \|^W2 " | tee -a $OUT/chain_abl.txt
done
