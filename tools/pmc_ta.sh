#!/bin/bash
# Texture-addresser / vector-L1 load of ONE attention shape (own PMC passes, kernel-trace only, each under a timeout).
#     gpurun --timeout 600 -- 'bash tools/pmc_ta.sh "synthetic self" r02ta'
set -u
CASE=${1:-synthetic self}
TAG=${2:-rXXta}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
RUN="python $PWD/tools/bench_kernels.py attn_one"
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE \
      -d "$OUT/pmc_ta1" -o p -f csv -- $RUN "$CASE" > "$OUT/run1.log" 2>&1 ); echo "pass1 rc=$?"
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum \
      -d "$OUT/pmc_ta2" -o p -f csv -- $RUN "$CASE" > "$OUT/run2.log" 2>&1 ); echo "pass2 rc=$?"
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum \
      -d "$OUT/pmc_ta3" -o p -f csv -- $RUN "$CASE" > "$OUT/run3.log" 2>&1 ); echo "pass3 rc=$?"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in ('pmc_ta1', 'pmc_ta2', 'pmc_ta3'):
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (out, d), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][-40:]
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, cs in acc.items():
            if 'attn' in k:
                print(d, k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in cs.items()})
PY
tail -2 "$OUT"/run1.log
