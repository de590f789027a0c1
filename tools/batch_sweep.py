#!/usr/bin/env python3
"""Batch sweep of the headline model (reuters d512 2+2L 4h), fixed and ragged lengths, B = 8 .. 512, with the decoder chain
on packed weights (default), on the native weight layouts (round 4) and off (five separate launches; the -DLAMP_NO_CHAIN build).

    python tools/batch_sweep.py [nochain-library.so]       -> one table on stdout (profiles/r05_batch_sweep.txt)
Each configuration: 0.5 s device warm-up, then samples/s over >= 0.3 s of forwards; the three variants of a batch size are
timed round-robin three times and the median is reported (box-to-box variance is larger than some of the differences).
"""
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NOCHAIN = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'lamp_amd', 'build', 'liblamp_nochain.so')

CHILD = r'''
import sys, time, json, torch
sys.path.insert(0, %r)
import bench
from lamp_amd.Models import LAMP
packs = sys.argv[1] == '1'
LAMP.use_chain_packs = packs
dev = torch.device('cuda:0')
out = {}
for ragged in (False, True):
    for B in (8, 16, 24, 32, 40, 45, 48, 64, 96, 128, 256, 512):
        w = dict(bench.WORKLOADS['reuters'])
        lengths = None
        if ragged:
            g = torch.Generator().manual_seed(1000)
            lengths = torch.randint(20, 303, (B,), generator=g).tolist()
            w['T'] = max(lengths)
        model, sd, adj, seq, pos = bench.build(w, B, dev, seed=0, lengths=lengths, n_max=302)
        src = (seq.to(dev), pos.to(dev))
        step = lambda: model(src, None, None, None)
        bench.warm_device(step, 0.3)
        vals = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
            while time.perf_counter() - t0 < 0.25:
                for _ in range(8): step()
                n += 8
                torch.cuda.synchronize()
            vals.append(B * n / (time.perf_counter() - t0))
        out['%%s %%d' %% ('ragged' if ragged else 'fixed', B)] = sorted(vals)[1]
        del model
print(json.dumps(out))
''' % ROOT


def run(lib, packs):
    env = dict(os.environ)
    if lib:
        env['LAMP_HIP_LIBRARY'] = lib
    r = subprocess.run([sys.executable, '-c', CHILD, '1' if packs else '0'], capture_output=True, text=True, env=env, timeout=1500)
    if r.returncode != 0:
        raise SystemExit(r.stderr[-3000:])
    import json
    return json.loads(r.stdout.strip().splitlines()[-1])


def main():
    variants = [('chain, packed weights', None, True), ('chain, native layouts', None, False)]
    if os.path.exists(NOCHAIN):
        variants.append(('five launches', NOCHAIN, False))
    rounds = [[run(lib, packs) for _, lib, packs in variants] for _ in range(2)]
    keys = list(rounds[0][0])
    print('# reuters d512 2+2L 4h, one MI355X, samples/s (median of round-robin runs); rows of the decoder = 90 x B')
    print('%-12s %8s' % ('batch', 'dec rows') + ''.join('%24s' % n for n, _, _ in variants) + '   packed vs five launches')
    for k in keys:
        vals = [statistics.median(r[i][k] for r in rounds) for i in range(len(variants))]
        B = int(k.split()[1])
        gain = ('%+6.1f %%' % ((vals[0] / vals[-1] - 1) * 100)) if len(variants) == 3 else ''
        print('%-12s %8d' % (k, 90 * B) + ''.join('%24.0f' % v for v in vals) + '   ' + gain)


if __name__ == '__main__':
    main()
