#!/bin/bash
# round 5, first look at the W-packed chain geometries: bit identity + stand-alone time, chain tests, whole-forward A/B
set -u
OUT=$PWD/gpurun_out/r05a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/bench_kernels.py chain 2880 8 0,7,8,9,10 2>&1 | grep -v amdgpu.ids | tee $OUT/chain_2880.txt
timeout 300 python tools/bench_kernels.py chain 4096 8 0,7,8,9,10 2>&1 | grep -v amdgpu.ids | grep "launch\|five" | tee $OUT/chain_4096.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "chain or model_golden or baseline_sizes" -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/pytest_chain.txt
for r in 1 2 3; do
  for f in "" "--no-chain-packs"; do
    python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extra-workloads --no-pipelined --no-kernel-trace --no-pmc $f 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-18s' % sys.argv[1], round(d['value']), {k: round(v['us_per_step'],1) for k,v in d['kernels'].items()})" "packs${f}" | tee -a $OUT/ab_bench.txt
  done
done
