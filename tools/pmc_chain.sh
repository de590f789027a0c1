#!/bin/bash
# SQ counters of the decoder chain launch (chain.hip) and of the five launches it replaces: LDS conflicts, where the waves
# wait, how busy the matrix pipe is (own PMC passes, kernel-trace only).   gpurun --timeout 600 -- 'bash tools/pmc_chain.sh r04k'
set -u
TAG=${1:-rXXchain}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
RUN="python $PWD/tools/bench_kernels.py chain"
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES \
      -d "$OUT/pmc1" -o p -f csv -- $RUN > "$OUT/run1.log" 2>&1 ); echo "pass1 rc=$?"
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 \
      -d "$OUT/pmc2" -o p -f csv -- $RUN > "$OUT/run2.log" 2>&1 ); echo "pass2 rc=$?"
python - "$OUT" <<'PY' | tee "$OUT/chain_pmc.txt"
import csv, glob, sys, collections
out = sys.argv[1]
for d in ('pmc1', 'pmc2'):
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (out, d), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][-60:]
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, cs in sorted(acc.items()):
            if 'chain' in k or 'gemm_nt' in k or 'layernorm' in k:
                print(d, k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, 'n=%d' % len(next(iter(cs.values()))))
PY
tail -3 "$OUT"/run2.log
