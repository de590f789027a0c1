#!/bin/bash
# W fragments straight from global memory (gemm.hip DMA = 3, tuning configurations 30-33) against the register-staged tiles:
# bit-identity, main-loop clock, round-robin medians.   gpurun --timeout 1200 -- 'bash tools/r04_wdir_round.sh'
export TMPDIR=/tmp
OUT=gpurun_out/r04wdir; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -q -x -k "every_tile_config or direct_to_lds" 2>&1 | tail -3
python tools/bench_kernels.py gemm_clock 0,11,30,12,31,9,32,18,33 2>&1 | grep -v amdgpu.ids | tee $OUT/gemm_clock.txt
python tools/bench_kernels.py gemm_ab 0,11,30,12,31,9,32,18,33 2>&1 | grep -v amdgpu.ids | tee $OUT/gemm_ab.txt
