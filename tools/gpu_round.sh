#!/bin/bash
# One GPU-box session: tests, a short bench, kernel micro-benchmarks.  Output -> gpurun_out/<tag>/
#     gpurun --timeout 2400 -- 'bash tools/gpu_round.sh r02a "tests bench attn trace tiles"'
set -u
TAG=${1:-rXX}
WHAT=${2:-"tests bench attn"}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for w in $WHAT; do
  case $w in
    tests)
      timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 --durations=8 -p no:cacheprovider > "$OUT/pytest.log" 2>&1
      echo "pytest rc=$?" >> "$OUT/pytest.log"; tail -25 "$OUT/pytest.log" ;;
    attntests)
      timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=40 -k "sdpa or mha or model_golden or baseline or bitwise or maps" -p no:cacheprovider > "$OUT/pytest_attn.log" 2>&1
      echo "pytest rc=$?" >> "$OUT/pytest_attn.log"; tail -25 "$OUT/pytest_attn.log" ;;
    bench)
      timeout 600 python bench.py --steps 100 --warmup 10 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
      python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('value %.0f samples/s  ms/step %.4f  gemm frac %.3f  fwd frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['forward']['frac_of_fp32_mfma_peak']))
for k, v in d['kernels'].items():
    print('  %-12s %5.1f launches %8.1f us/step  %s TF' % (k, v['launches_per_step'], v['us_per_step'], v['tflops'] and round(v['tflops'], 1)))
print('  pipelined', d['pipelined_batches_in_flight'] and round(d['pipelined_batches_in_flight']['value']))
for k, v in d.get('workloads', {}).items():
    if 'roofline' in v:
        print('  %-14s %9.1f samples/s  %8.3f ms  gemm %.3f  attn %.1f TF  fwd %.3f' % (k, v['value'], v['ms_per_step'], v['roofline']['frac'], v['attention_tflops'], v['forward_frac_of_fp32_mfma_peak']))
    else:
        print('  %-14s %9.1f samples/s  %8.3f ms  fwd %.3f' % (k, v['value'], v['ms_per_step'], v['forward_frac_of_fp32_mfma_peak']))
print('  cpu', d.get('cpu_baseline', {}).get('value'))
PY
      ;;
    ragged)
      timeout 300 python bench.py --ragged --steps 200 --warmup 20 --no-cpu-baseline --no-extra-workloads > "$OUT/bench_ragged.json" 2> "$OUT/bench_ragged.err"; echo "ragged rc=$?"
      python -c "import json,sys; d=json.load(open(sys.argv[1])); print('ragged value %.0f samples/s  ms/step %.4f  pipelined %s' % (d['value'], d['ms_per_step'], d['pipelined_batches_in_flight'] and round(d['pipelined_batches_in_flight']['value'])))" "$OUT/bench_ragged.json" ;;
    attn) timeout 300 python tools/bench_kernels.py attn 2>&1 | grep -v amdgpu.ids > "$OUT/attn_variants.txt"; grep -E "mode=0" "$OUT/attn_variants.txt" ;;
    trace) timeout 300 python tools/bench_kernels.py gemm_trace 2>&1 | grep -v amdgpu.ids > "$OUT/gemm_trace.txt"; cat "$OUT/gemm_trace.txt" ;;
    tiles) timeout 300 python tools/bench_kernels.py gemm_ab 2>&1 | grep -v amdgpu.ids > "$OUT/gemm_tiles.txt"; cat "$OUT/gemm_tiles.txt" ;;
    *) timeout 600 python tools/bench_kernels.py $w 2>&1 | grep -v amdgpu.ids > "$OUT/$w.txt"; cat "$OUT/$w.txt" ;;
  esac
done
