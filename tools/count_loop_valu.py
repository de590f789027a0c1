#!/usr/bin/env python3
"""Vector instructions per MFMA loop of a kernel (hipcc -S output): on gfx950 every fp32 vector instruction costs matrix-pipe
time (profiles/r05_mfma_chain.txt), so the count per tile step is the number to minimise.
    python tools/count_loop_valu.py file.s kernel-substring
Prints, for the innermost loop that holds MFMAs (label .. backward branch), the instruction mix."""
import re
import sys
from collections import Counter


def kernel_lines(path, want):
    txt = open(path).read().split('\n')
    out, on = [], False
    for l in txt:
        m = re.match(r'^(_Z\w+):', l)
        if m:
            on = want in m.group(1)
            continue
        if on:
            out.append(l)
            if l.strip().startswith('.Lfunc_end'):
                on = False
    return out


def main():
    lines = kernel_lines(sys.argv[1], sys.argv[2])
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    best = None
    for i, l in enumerate(lines):
        m = re.match(r'\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.match(r'\s*s_branch\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = lines[labels[m.group(1)]:i + 1]
            n = sum('v_mfma' in x for x in body)
            if n and (best is None or len(body) > len(best[0])):
                best = (body, n)
    if best is None:
        print('no MFMA loop found')
        return
    body, n = best
    c = Counter()
    for l in body:
        t = l.strip()
        if not t or t.startswith((';', '.')) or t.endswith(':'):
            continue
        op = t.split()[0]
        kind = 'mfma' if op.startswith('v_mfma') else 'trans' if op.startswith(('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_sin', 'v_cos')) \
            else 'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('buffer_', 'global_')) \
            else 'nop' if op == 's_nop' else 'wait' if op == 's_waitcnt' else 'salu'
        c[kind] += 1
        if kind in ('valu', 'trans'):
            c['  ' + op] += 1
    print('loop of %d lines, %d MFMAs' % (len(body), n))
    for k in ('mfma', 'valu', 'trans', 'lds', 'vmem', 'salu', 'wait', 'nop'):
        print('%-6s %4d' % (k, c[k]))
    print('vector instructions by opcode:', ', '.join('%s %d' % (k.strip(), v) for k, v in sorted(c.items(), key=lambda kv: -kv[1]) if k.startswith('  ')))
    print('=> per 128 MFMAs: %.0f vector + %.0f transcendental = ~%.0f cycles on 8192' %
          (c['valu'] * 128.0 / n, c['trans'] * 128.0 / n, (c['valu'] * 4.6 + c['trans'] * 8.75) * 128.0 / n))


if __name__ == '__main__':
    main()
