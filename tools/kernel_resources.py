#!/usr/bin/env python3
"""Compact register / occupancy table of the kernels in one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py lamp_amd/csrc/attention_small.hip [substring filter] [-DLAMP_TUNING]
"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = [a for a in sys.argv[2:] if not a.startswith('-')]
extra = [a for a in sys.argv[2:] if a.startswith('-')]
r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-c', src, '-o', '/dev/null',
                    '-Rpass-analysis=kernel-resource-usage'] + extra, capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r'Function Name: (\S+)', line) or re.search(r' Name: (\S+)', line)
    if m:
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        name = name.replace('(anonymous namespace)::', '')
        cur = {'name': name.split('(')[0].replace('void lamp::', '').replace('lamp::', '')}
        rows.append(cur)
        continue
    for key, pat in (('sgpr', r'TotalSGPRs: (\d+)'), ('vgpr', r' VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'),
                     ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'), ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'),
                     ('lds', r'LDS Size \[bytes/block\]: (\d+)')):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print('%-64s %5s %5s %5s %7s %4s' % ('kernel', 'sgpr', 'vgpr', 'agpr', 'scratch', 'occ'))
for c in rows:
    if all(f in c['name'] for f in flt):
        print('%-64s %5s %5s %5s %7s %4s' % (c['name'][:64], c.get('sgpr'), c.get('vgpr'), c.get('agpr'), c.get('scratch'), c.get('occ')))
