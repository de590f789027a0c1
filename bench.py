#!/usr/bin/env python3
"""Benchmark of the LaMP label-graph forward path on MI355X (the BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one forward pass of the hot path (LAMP.forward, eval mode) over one synthetic batch of
the configuration the metric is quoted on: reuters-shaped, batch 32 per GPU, T = 302 fixed,
L = 90 labels, d_model 512, 2 + 2 graph layers, 4 heads, label_mask = prior.
Inputs and weights are resident in HBM before the timed region.  Samples shard over GPUs with no
collective on the data path (weak scaling: every rank runs its own batch of 32); torch.distributed
is used only for the barrier and the max-over-ranks of the elapsed time.

Rank 0 prints ONE JSON line with, besides the contract fields,
  roofline      the dominant kernel class (fp32-MFMA GEMM): algorithmic FLOPs of its launches divided
                by their HIP-event durations (events recorded by liblamp_hip.so on the launch stream,
                in an instrumented replay of the same K steps right after the timed region),
  forward       whole-forward achieved fraction of the fp32 MFMA roof with F_live of SURVEY.md 8d,
  cpu_baseline  the oracle (a port of the reference's op sequence, `as_written`, autograd graph
                built as the reference's test loop does) timed on this host's cores on a bounded
                sample (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0

RAGGED = {'reuters': (20, 302), 'bibtex': (10, 150), 'delicious': (5, 60)}  # SURVEY.md 8d length variant (ii)
WORKLOADS = {
    # name: V, L, T, d, d_ff, heads, label_mask, pos_emb, prior p
    'reuters': dict(V=23666, L=90, T=302, d=512, dff=512, h=4, mask='prior', pos=True, p=0.10),
    'bibtex': dict(V=1840, L=159, T=100, d=512, dff=1024, h=4, mask='prior', pos=False, p=0.05),
    'delicious': dict(V=504, L=983, T=40, d=1024, dff=2048, h=8, mask='none', pos=False, p=0.0),
    'synthetic4096': dict(V=32004, L=4096, T=512, d=1024, dff=2048, h=8, mask='prior', pos=True, p=0.05),
}


def f_live(w, n_enc=2, n_dec=2):
    """Algorithmic GEMM FLOPs per sample: dead encoder attention excluded."""
    T, L, d, dff = w['T'], w['L'], w['d'], w['dff']
    return (n_enc * 4 * T * d * dff + n_dec * (4 * L * d * d + 4 * T * d * d + 4 * L * T * d) +
            n_dec * (8 * L * d * d + 4 * L * L * d) + n_dec * 8 * L * d * dff + 2 * L * d)


def build(w, batch, device, seed=0, lengths=None):
    from lamp_amd import synthetic as R
    from lamp_amd.Models import LAMP
    n_max = max([w['T']] + list(lengths or []))
    sd = R.make_state_dict(w['V'], w['L'], n_max, w['d'], w['dff'], w['h'], 2, 2, pos_emb=w['pos'], seed=seed)
    adj = R.make_adjacency(w['L'], w['p'], seed) if w['mask'] == 'prior' else None
    seq, pos = R.make_batch(batch, w['V'], w['T'], lengths=lengths, seed=seed)
    h, d = w['h'], w['d']
    model = LAMP(w['V'], w['L'], n_max, w['L'], n_layers_enc=2, n_layers_dec=2, n_head=h, n_head2=h,
                 d_word_vec=d, d_model=d, d_inner_hid=w['dff'], d_k=d // h, d_v=d // h, encoder='graph',
                 decoder='graph', no_enc_pos_embedding=not w['pos'],
                 label_adj_matrix=adj.clone() if adj is not None else None, label_mask=w['mask'],
                 dec_dropout2=False)
    model.load_state_dict(sd)
    model = model.to(device).eval()
    return model, sd, adj, seq, pos


def cpu_baseline(w, sd, adj, seq, pos, budget_s=20.0):
    """Reference-as-written op sequence on the host CPU (oracle port), bounded sample.  The ONLY place
    bench.py touches oracle/."""
    from oracle import lamp_ref as R
    h = w['h']
    blocked = R.label_block_mask(adj, w['mask'], w['L'])
    sd_g = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}

    def timed(fn, max_iters, budget):
        fn()  # warm-up
        ts = []
        t_end = time.perf_counter() + budget
        while len(ts) < max_iters and (len(ts) < 2 or time.perf_counter() < t_end):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2], len(ts)

    B = seq.size(0)
    # torch's default thread count (all physical cores of a big host) is usually NOT the fastest for these
    # small ops: probe a few counts on the as-written / autograd-on path and report the best one.
    default_threads = torch.get_num_threads()
    candidates = sorted({t for t in (default_threads, 64, 32, 16, 8) if t <= max(default_threads, 1)}, reverse=True)
    probe = {}
    for t in candidates:
        torch.set_num_threads(t)
        dt, n = timed(lambda: R.forward(sd_g, seq, pos, h, blocked, as_written=True), 8,
                      budget_s * 0.4 / len(candidates))
        probe[t] = B / dt
    best = max(probe, key=probe.get)
    torch.set_num_threads(best)
    t_aw, n_aw = timed(lambda: R.forward(sd_g, seq, pos, h, blocked, as_written=True), 30, budget_s * 0.3)
    with torch.no_grad():
        t_ng, n_ng = timed(lambda: R.forward(sd, seq, pos, h, blocked, as_written=True), 15, budget_s * 0.15)
        t_dce, n_dce = timed(lambda: R.forward(sd, seq, pos, h, blocked, as_written=False), 15, budget_s * 0.15)
    torch.set_num_threads(default_threads)
    return {
        'value': B / t_aw, 'unit': 'samples/s', 'cores': best, 'kind': 'port',
        'sample': '%d timed forwards of one batch of %d (median), eval() with autograd graph built as '
                  'reference test.py:17,41; oracle as_written=True; best of thread counts %s' %
                  (n_aw, B, sorted(probe)),
        'no_grad_value': B / t_ng, 'dead_code_eliminated_no_grad_value': B / t_dce,
        'threads_probe_samples_per_s': {str(k): v for k, v in sorted(probe.items())},
        'default_torch_threads': default_threads, 'host_cpu_count': os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='reuters', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=32, help='samples per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=20.0)
    ap.add_argument('--graph', action='store_true', help='replay the step from a captured HIP graph')
    ap.add_argument('--no-pipelined', action='store_true',
                    help='skip the extra two-batches-in-flight measurement (use under rocprofv3: overlapping '
                         'kernels inflate per-kernel durations)')
    ap.add_argument('--fuse-ln', type=int, default=None, choices=[0, 1],
                    help='override LAMP.fuse_layernorm (deferred LayerNorm: fewer launches, logits equal to rounding)')
    ap.add_argument('--ragged', action='store_true',
                    help='sequence lengths U{lo..hi} padded to the batch maximum (SURVEY.md 8d variant ii) instead of fixed T')
    ap.add_argument('--mask', default=None, choices=['prior', 'none', 'inveye'], help='override the workload label mask')
    ap.add_argument('--streams', type=int, default=1, choices=[1, 2],
                    help='HIP streams one forward spreads its batch over (lamp_set_forward_streams)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: no HIP device is visible (there is no CPU path)')
    dev_index = local_rank % torch.cuda.device_count()  # > 1 rank per GPU only happens in the 1-GPU smoke test
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    dist = None
    # "nccl" is RCCL on ROCm.  LAMP_BENCH_BACKEND=gloo lets two ranks share one GPU to smoke-test this path.
    backend = os.environ.get('LAMP_BENCH_BACKEND', 'nccl')
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    n_gpus = world if world > 1 else 1
    if world > 1 and args.gpus != world and rank == 0:
        print('note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE' % (args.gpus, world), file=sys.stderr)

    from lamp_amd import _native as N
    N.lib()
    N.set_forward_streams(args.streams)
    w = dict(WORKLOADS[args.workload])
    if args.mask:
        w['mask'] = args.mask
    lengths = None
    if args.ragged:
        lo, hi = RAGGED[args.workload]
        g = torch.Generator().manual_seed(1000 + rank)
        lengths = torch.randint(lo, hi + 1, (args.batch,), generator=g).tolist()
        w['T'] = max(lengths)  # padded length of this batch: what the kernels process and what F_live counts
    model, sd, adj, seq, pos = build(w, args.batch, device, seed=rank, lengths=lengths)
    if args.fuse_ln is not None:
        model.fuse_layernorm = bool(args.fuse_ln)
    src = (seq.to(device), pos.to(device))

    def step():
        return model(src, None, None, None)

    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()

    graph = None
    if args.graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = step()
        run = graph.replay
    else:
        run = step
    run()
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # per-step latency with a device sync after every step (median and min), outside the timed region
    lat = []
    for _ in range(min(args.steps, 50)):
        torch.cuda.synchronize()
        ts = time.perf_counter()
        run()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - ts) * 1e3)
    lat.sort()

    # sanity: the timed work produced finite logits
    logits = out[0]
    assert torch.isfinite(logits).all(), 'non-finite logits'

    # ---- throughput mode (reported beside `value`, never as `value`): successive batches issued round-robin
    # on two HIP streams, so that one forward's launch gaps / ramp / tail are filled by the other's kernels ----
    pipelined = None
    if not args.graph and not args.no_pipelined:
        pipelined = {'unit': 'samples/s per GPU',
                     'note': 'independent batches in flight on round-robin HIP streams (what evaluate.test_epoch(streams=n) '
                             'does); latency per batch is NOT reduced'}
        for depth in (2, 4):
            streams = [torch.cuda.Stream(device=device) for _ in range(depth)]
            for st in streams:
                with torch.cuda.stream(st):
                    step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                with torch.cuda.stream(streams[i % depth]):
                    step()
            torch.cuda.synchronize()
            e2 = time.perf_counter() - t1
            pipelined['depth_%d' % depth] = {'value': args.batch * args.steps / e2,
                                              'ms_per_step_amortised': e2 / args.steps * 1e3}
        pipelined['value'] = pipelined['depth_2']['value']
        pipelined['streams'] = 2

    # ---- instrumented replay: per-kernel HIP-event durations on the launch stream ----
    prof_steps = min(args.steps, 20)
    N.prof_reset()
    N.prof_enable(True)
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    N.prof_enable(False)
    prof = N.prof_read()
    N.prof_reset()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    samples = args.batch * n_gpus * args.steps
    value = samples / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    fl = f_live(w)
    gemm = prof['gemm']
    gemm_tflops = gemm['flops'] / (gemm['ms'] * 1e-3) / 1e12 if gemm['ms'] > 0 else 0.0
    kernels = {}
    for name, r in prof.items():
        if r['launches']:
            kernels[name] = {
                'launches_per_step': r['launches'] / prof_steps,
                'us_per_step': r['ms'] * 1e3 / prof_steps,
                'avg_us_per_launch': r['ms'] * 1e3 / r['launches'],
                'tflops': r['flops'] / (r['ms'] * 1e-3) / 1e12 if r['ms'] > 0 else None,
                'algorithmic_gbs': r['bytes'] / (r['ms'] * 1e-3) / 1e9 if r['ms'] > 0 else None,
            }
    result = {
        'metric': 'forward samples/sec, reuters d512 2+2L 4h' if args.workload == 'reuters' else
                  'forward samples/sec, %s d%d 2+2L %dh' % (args.workload, w['d'], w['h']),
        'value': value, 'unit': 'samples/s', 'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'step_ms_synced': {'median': lat[len(lat) // 2], 'min': lat[0], 'n': len(lat)},
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s: batch %d/GPU, T=%d %s, L=%d, d_model=%d, d_ff=%d, 2 enc + 2 dec graph layers, '
                               '%d heads, label_mask=%s, fp32' %
                               (args.workload, args.batch, w['T'],
                                '(lengths U{%d..%d} padded to the batch maximum)' % RAGGED[args.workload] if args.ragged
                                else 'fixed', w['L'], w['d'], w['dff'], w['h'], w['mask']),
                   'batch_per_gpu': args.batch, 'parallelism': 'batch-sharded x%d, no collectives' % n_gpus,
                   'launch': 'hip-graph replay' if args.graph else 'eager (one lamp_forward call per step)',
                   'streams_per_forward': args.streams, 'deferred_layernorm': bool(model.fuse_layernorm)},
        'roofline': {
            'bound': 'mfma', 'kernel': 'gemm_nt_kernel (fp32 MFMA 16x16x4), all launches of a forward',
            'achieved': gemm_tflops, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': gemm_tflops / PEAK_FP32_MFMA_TFLOPS, 'traffic': None,
            'launches_per_step': gemm['launches'] / prof_steps if prof_steps else None,
            'avg_launch_us': gemm['ms'] * 1e3 / gemm['launches'] if gemm['launches'] else None,
            'algorithmic_gflop_per_step': gemm['flops'] / prof_steps / 1e9 if prof_steps else None,
        },
        'forward': {
            'f_live_gflop_per_sample': fl / 1e9,
            'achieved_tflops_per_gpu': value / n_gpus * fl / 1e12,
            'frac_of_fp32_mfma_peak': value / n_gpus * fl / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            'kernel_time_us_per_step': sum(k['us_per_step'] for k in kernels.values()),
        },
        'kernels': kernels,
        'pipelined_batches_in_flight': pipelined,
    }
    if n_gpus == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(w, sd, adj, seq, pos, args.cpu_budget)
        result['cpu_baseline'] = cb
        result['speedup_vs_cpu_as_written'] = value / cb['value']
    print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
