#!/bin/bash
set -u
OUT=$PWD/gpurun_out/${1:-r05n}; mkdir -p $OUT; export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d['roofline']
print('value %.0f  ms %.4f  frac %.3f  kernel_only %.3f  traffic %s (%s)' % (d['value'], d['ms_per_step'], r['frac'], r.get('frac_kernel_only') or 0, r.get('traffic'), r.get('traffic_source')))
print('attention', json.dumps(r.get('attention'))[:600])
print('pmc', (d.get('pmc') or {}).get('skipped'), (d.get('pmc') or {}).get('command'))
print({k: (round(v['value']), round(v['ms_per_step'], 3)) for k, v in d.get('workloads', {}).items()})
PY
python tools/batch_sweep.py > $OUT/batch_sweep.txt 2> $OUT/batch_sweep.err; cat $OUT/batch_sweep.txt; tail -3 $OUT/batch_sweep.err
timeout 900 python -m pytest tests/test_gpu_multi.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
