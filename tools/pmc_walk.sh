#!/bin/bash
# HBM-side traffic of the GEMM tile walks (tools/bench_kernels.py walk_pmc): FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 passes (kernel-trace only), read back by dispatch order.   gpurun --timeout 900 -- 'bash tools/pmc_walk.sh r03walk'
set -u
TAG=${1:-rXXwalk}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python tools/bench_kernels.py walk 2>&1 | grep -v amdgpu.ids > "$OUT/walk_times.txt"; cat "$OUT/walk_times.txt"
RUN="python $PWD/tools/bench_kernels.py walk_pmc"
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_$c" -o p -f csv -- $RUN > "$OUT/run_$c.log" 2>&1 ); echo "$c rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, sys
sys.path.insert(0, 'tools')
out = sys.argv[1]
import importlib.util
spec = importlib.util.spec_from_file_location('bk', 'tools/bench_kernels.py')
shapes = [('delic ffn1', 31456, 2048, 1024, 1, 0), ('delic ffn2', 31456, 1024, 2048, 1, 1), ('delic fc +R', 31456, 1024, 1024, 1, 1),
          ('delic qkv', 31456, 1024, 1024, 3, 0), ('reuters kv', 9664, 512, 512, 4, 0), ('syn ffn1', 65536, 2048, 1024, 1, 0)]
walks = [0, 2, 4, 8, 16]
vals = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('%s/pmc_%s/**/*counter_collection.csv' % (out, c), recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if 'gemm_nt_kernel' in r['Kernel_Name'] and r['Counter_Name'] == c]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    vals[c] = [float(r['Counter_Value']) for r in rows]
n = 3
L = ['# fabric-side bytes per launch (rocprofv3 --pmc, KiB counters; FETCH_SIZE x2 per the gfx950 calibration), median of 3 launches',
     '%-14s %10s' % ('shape', 'operands') + ''.join('%22s' % ('walk %d fetch/write MB' % g) for g in walks)]
i = 0
for name, M, N, K, nseg, res in shapes:
    alg = 4.0 * (M * K + nseg * N * K + (M * N * nseg if res else 0)) / 1e6
    row = '%-14s %8.0fMB' % (name, alg)
    for g in walks:
        fe = sorted(vals['FETCH_SIZE'][i:i + n])[n // 2] * 1024 * 2 / 1e6
        wr = sorted(vals['WRITE_SIZE'][i:i + n])[n // 2] * 1024 / 1e6
        row += '%14.0f/%7.0f' % (fe, wr)
        i += n
    L.append(row)
open(out + '/walk_traffic.txt', 'w').write('\n'.join(L) + '\n')
print('\n'.join(L))
PY
