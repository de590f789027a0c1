#!/bin/bash
set -u
OUT=$PWD/gpurun_out/${1:-r05e}; mkdir -p $OUT; export TMPDIR=/tmp
python tools/dbg_chain.py 2>&1 | grep -v amdgpu.ids | grep -v "equal True repeat True"
timeout 300 python tools/bench_kernels.py chain 2880 17 0,12,17 2>&1 | grep -v amdgpu.ids | tee -a $OUT/chain_stamps.txt
for m in 720 1440 2048 2400 3072; do
timeout 300 python tools/bench_kernels.py chain $m 0 12,15,16,17 2>&1 | grep -v amdgpu.ids | grep "five\|chain launch\|error" | tr "\n" ";" | tee -a $OUT/chain_stamps.txt; echo " M=$m" | tee -a $OUT/chain_stamps.txt
done
