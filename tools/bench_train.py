#!/usr/bin/env python3
"""Training-step throughput of the HIP path: the reference's train loop body (train.py:34-48) --
forward in train() mode with dropout, BCE-with-logits, loss.backward(), Adam step -- on one synthetic batch.

    python tools/bench_train.py [--workload reuters] [--batch 32] [--steps 30] [--dropout 0.1]

Prints one JSON line: samples/s of the whole step, the split into forward / backward / optimizer wall time, the
summed HIP-event kernel time per class (so host overhead = wall - kernels is visible).  The same step on the CPU
oracle's autograd is timed by tests/time_oracle_train_step.py (the oracle is test infrastructure).
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='reuters', choices=sorted(bench.WORKLOADS))
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--host-profile', default=None, help='write a cProfile listing of 20 steps to this file')
    ap.add_argument('--per-launch', action='store_true', help='one Python round trip per launch instead of one library call per sub-layer')
    ap.add_argument('--single-thread-autograd', action='store_true', help='torch.autograd.set_multithreading_enabled(False): backward on the calling thread')
    ap.add_argument('--no-defer', action='store_true', help='weight gradients through autograd, one launch (+ split-K reduce) each')
    ap.add_argument('--fused-adam', action='store_true', help="torch.optim.Adam(fused=True) instead of the reference's call (main.py:99)")
    a = ap.parse_args()
    from lamp_amd import _native as N
    from lamp_amd import hostcpu
    hostcpu.fit_intra_op_threads()   # a 128-thread OpenMP pool under the boxes' 16-core quota gets the issuing thread throttled
    dev = torch.device('cuda:0')
    w = bench.WORKLOADS[a.workload]
    model, sd, adj, seq, pos = bench.build(w, a.batch, dev)
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = a.dropout
    seq, pos = seq.to(dev), pos.to(dev)
    tgt = (torch.rand(a.batch, w['L'], device=dev) < 0.05).float()
    from lamp_amd import training
    training.DEFER_WEIGHT_GRADS = not a.no_defer
    training.COMPOSITE_CALLS = not a.per_launch
    opt = torch.optim.Adam(model.get_trainable_parameters(), lr=2e-4, betas=(0.9, 0.98), eps=1e-9,
                           **({'fused': True} if a.fused_adam else {}))
    model.train()
    if a.single_thread_autograd:
        torch.autograd.set_multithreading_enabled(False)

    def step(timers=None):
        t0 = time.perf_counter()
        opt.zero_grad()
        pred, enc, *_ = model((seq, pos), None, None, tgt)
        loss = F.binary_cross_entropy_with_logits(pred, tgt, reduction='mean')
        if timers is not None:
            torch.cuda.synchronize(); t1 = time.perf_counter()
        loss.backward()
        if timers is not None:
            torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.step()
        if timers is not None:
            torch.cuda.synchronize(); t3 = time.perf_counter()
            timers.append((t1 - t0, t2 - t1, t3 - t2))
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    host = (time.perf_counter() - t0) / a.steps   # the issuing thread's share: what the step costs when the device never blocks it
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    if a.host_profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(20):
            step()
        pr.disable()
        torch.cuda.synchronize()
        with open(a.host_profile, 'w') as f:
            pstats.Stats(pr, stream=f).sort_stats('tottime').print_stats(45)
            pstats.Stats(pr, stream=f).sort_stats('cumulative').print_stats(60)
    timers = []
    for _ in range(5):
        step(timers)
    fwd, bwd, optim = (sorted(t[i] for t in timers)[2] for i in range(3))
    N.prof_reset(); N.prof_enable(True)
    step()
    torch.cuda.synchronize()
    N.prof_enable(False)
    prof = N.prof_read()
    out = {'metric': 'training samples/sec (forward + backward + Adam), %s' % a.workload, 'value': a.batch / dt,
           'unit': 'samples/s', 'ms_per_step': dt * 1e3, 'host_issue_ms_per_step': host * 1e3, 'batch': a.batch, 'dropout': a.dropout, 'steps': a.steps,
           'final_loss': float(loss.detach()), 'dtype': 'f32', 'data': 'synthetic',
           'deferred_weight_gradients': not a.no_defer, 'composite_calls': not a.per_launch, 'autograd_multithreading': not a.single_thread_autograd, 'optimizer': 'torch.optim.Adam' + ('(fused=True)' if a.fused_adam else ''),
           'synchronised_split_ms': {'forward': fwd * 1e3, 'backward': bwd * 1e3, 'optimizer': optim * 1e3},
           'hip_kernels_one_step': {k: {'launches': v['launches'], 'ms': round(v['ms'], 3),
                                        'tflops': round(v['flops'] / v['ms'] / 1e9, 1) if v['ms'] > 0 and v['flops'] else 0.0}
                                    for k, v in prof.items()},
           'note': 'only launches bracketed by the library profiler are listed (pointwise backward kernels and torch '
                   'optimizer kernels are not); wall - kernels = host-side autograd / launch overhead'}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
