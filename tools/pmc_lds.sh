#!/bin/bash
# LDS counters of ONE attention shape (own PMC pass, kernel-trace only, under a timeout).
#     gpurun --timeout 600 -- 'bash tools/pmc_lds.sh "reuters enc-attn" r02lds'
set -u
CASE=${1:-reuters enc-attn}
TAG=${2:-rXXlds}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
RUN="python $PWD/tools/bench_kernels.py attn_one"
( cd /tmp && rocprofv3 -L 2>/dev/null | grep -o -E "\bSQ_[A-Z_a-z0-9]*LDS[A-Z_a-z0-9]*" | sort -u | tr "\n" " " ) > "$OUT/lds_counters.txt"
cat "$OUT/lds_counters.txt"; echo
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
      -d "$OUT/pmc1" -o p -f csv -- $RUN "$CASE" > "$OUT/run1.log" 2>&1 ); echo "pass1 rc=$?"
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY \
      -d "$OUT/pmc2" -o p -f csv -- $RUN "$CASE" > "$OUT/run2.log" 2>&1 ); echo "pass2 rc=$?"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in ('pmc1', 'pmc2'):
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (out, d), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][-40:]
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, cs in acc.items():
            if 'attn' in k:
                print(d, k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()})
PY
tail -2 "$OUT"/run2.log
