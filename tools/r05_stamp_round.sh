#!/bin/bash
set -u
OUT=$PWD/gpurun_out/${1:-r05e}; mkdir -p $OUT; export TMPDIR=/tmp
python tools/dbg_chain.py 2>&1 | grep -v amdgpu.ids | grep -v "equal True repeat True"
for m in 3600 4320 5120 5760 6144; do
timeout 300 python tools/bench_kernels.py chain $m 0 12,18,19,20 2>&1 | grep -v amdgpu.ids | grep "five\|chain launch\|error\|Error" | tr "\n" ";" | tee -a $OUT/chain_stamps.txt; echo " M=$m" | tee -a $OUT/chain_stamps.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "chain or slab or pack" -p no:cacheprovider 2>&1 | tail -4
