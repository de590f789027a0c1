#!/usr/bin/env python3
"""Vector instructions inside the MFMA loops of a kernel (device assembly from hipcc -S).

On gfx950 a wave's vector instructions are not hidden under fp32 MFMAs -- each costs ~4.6 cycles of matrix-pipe time (8.75 for a
transcendental), its SIMD partner's included (tools/probes/mfma_chain.hip, profiles/r05_mfma_chain.txt) -- so the COUNT per loop
trip is the number to minimise, and tests/test_kernel_resources.py pins it for the hot loops.

    python tools/count_loop_valu.py file.{s,hip} kernel-name-substring [-DFLAG ...]
"""
import re
import subprocess
import sys
import tempfile
from collections import Counter

TRANS = ('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_sin', 'v_cos')


def device_asm(path, extra=()):
    if path.endswith('.s'):
        return open(path).read()
    import os
    inc = [os.path.dirname(os.path.abspath(path)), os.path.join(os.path.dirname(os.path.abspath(path)), '..', '..', 'include')]
    with tempfile.NamedTemporaryFile(suffix='.s') as f:
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', f.name, path,
                        *['-I' + i for i in inc], *extra], check=True, capture_output=True)
        return open(f.name).read()


def kernels(asm):
    """-> {demangled name: [lines]} for every kernel (function symbol up to .Lfunc_end)."""
    txt = asm.split('\n')
    out, cur, name = {}, None, None
    for l in txt:
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if l.startswith('.Lfunc_end'):
                out[name] = cur
                cur = None
            else:
                cur.append(l)
    names = list(out)
    dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
    return {d.strip(): out[n] for n, d in zip(names, dem)}


def classify(lines):
    c = Counter()
    for l in lines:
        t = l.strip()
        if not t or t.startswith((';', '.')) or t.endswith(':'):
            continue
        op = t.split()[0]
        if op.startswith('v_mfma'):
            c['mfma'] += 1
        elif op.startswith(TRANS):
            c['trans'] += 1
            c[op] += 1
        elif op.startswith('v_accvgpr'):
            c['accvgpr'] += 1
        elif op.startswith('v_'):
            c['valu'] += 1
            c[op] += 1
    return c


def mfma_loops(lines):
    """Every backward branch whose body holds MFMAs: [(first line, last line, Counter)], innermost first (by length)."""
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    out = []
    for i, l in enumerate(lines):
        m = re.match(r'\s*s_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            c = classify(lines[labels[m.group(1)]:i + 1])
            if c['mfma']:
                out.append((labels[m.group(1)], i, c))
    return sorted(out, key=lambda x: x[1] - x[0])


def barrier_segments(lines):
    """Straight stretches between consecutive s_barrier instructions that hold MFMAs: [Counter] (a tile step of the attention
    kernels; both sides of a rarely taken branch are counted, e.g. the 32 v_pk_mul of the online-softmax rescale)."""
    bars = [i for i, l in enumerate(lines) if l.strip().startswith('s_barrier')]
    return [c for a, b in zip(bars, bars[1:]) for c in [classify(lines[a:b])] if c['mfma']]


def main():
    asm = device_asm(sys.argv[1], [a for a in sys.argv[3:]])
    for name, lines in kernels(asm).items():
        if sys.argv[2] not in name:
            continue
        print(name[:150])
        for a, b, c in mfma_loops(lines):
            ops = ', '.join('%s %d' % (k, v) for k, v in c.most_common() if k.startswith('v_'))
            print('  loop of %4d lines: %3d MFMAs, %3d vector + %2d transcendental (+%d v_accvgpr)   %s' %
                  (b - a, c['mfma'], c['valu'], c['trans'], c['accvgpr'], ops[:160]))


if __name__ == '__main__':
    main()
