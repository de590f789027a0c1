set -u
OUT=$PWD/gpurun_out/r03b; mkdir -p $OUT; export TMPDIR=/tmp
python tools/bench_kernels.py lib_ab lamp_amd/build/liblamp_oldgemm.so lamp_amd/liblamp_hip.so 2>&1 | grep -v amdgpu.ids > $OUT/lib_ab.txt; cat $OUT/lib_ab.txt
B="python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extra-workloads --no-pipelined"
for i in 1 2 3; do
  LAMP_HIP_LIBRARY=$PWD/lamp_amd/build/liblamp_oldgemm.so $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old', round(d['value']), {k: round(v['us_per_step'],1) for k,v in d['kernels'].items()})" | tee -a $OUT/bench_ab.txt
  $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', round(d['value']), {k: round(v['us_per_step'],1) for k,v in d['kernels'].items()})" | tee -a $OUT/bench_ab.txt
done
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_parity.py -m gpu -q -x -k "linear or multi or bench or rccl or ffn or model_golden" -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
