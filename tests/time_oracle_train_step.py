#!/usr/bin/env python3
"""CPU counterpart of tools/bench_train.py: forward + backward of the oracle (torch autograd, no optimizer, dropout off)
on one synthetic batch, on the host cores -- the number quoted beside the HIP training-step rate.  Lives under tests/
because only test code may import the oracle.

    python tests/time_oracle_train_step.py [--workload reuters] [--batch 32] [--threads 16]
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import lamp_ref as R  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='reuters', choices=sorted(bench.WORKLOADS))
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--threads', type=int, default=16)
    a = ap.parse_args()
    from lamp_amd import synthetic as S
    w = bench.WORKLOADS[a.workload]
    sd = S.make_state_dict(w['V'], w['L'], w['T'], w['d'], w['dff'], w['h'], 2, 2, pos_emb=w['pos'], seed=0)
    adj = S.make_adjacency(w['L'], w['p'], 0) if w['mask'] == 'prior' else None
    seq, pos = S.make_batch(a.batch, w['V'], w['T'], seed=0)
    blocked = R.label_block_mask(adj, w['mask'], w['L'])
    sdg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    tgt = (torch.rand(a.batch, w['L']) < 0.05).float()
    torch.set_num_threads(a.threads)

    def step():
        lg, _, _ = R.forward(sdg, seq, pos, w['h'], blocked)
        F.binary_cross_entropy_with_logits(lg, tgt).backward()
    step()
    ts = []
    while len(ts) < 5:
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    print(json.dumps({'metric': 'oracle autograd forward + backward, %s' % a.workload, 'value': a.batch / sorted(ts)[2],
                      'unit': 'samples/s', 'threads': a.threads, 'sample': '5 steps (median), dropout off, no optimizer'}))


if __name__ == '__main__':
    main()
