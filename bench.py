#!/usr/bin/env python3
"""Benchmark of the LaMP label-graph forward path on MI355X (the BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one forward pass of the hot path (LAMP.forward, eval mode) over one synthetic batch of
the configuration the metric is quoted on: reuters-shaped, batch 32 per GPU, T = 302 fixed,
L = 90 labels, d_model 512, 2 + 2 graph layers, 4 heads, label_mask = prior.
Inputs and weights are resident in HBM before the timed region.  Samples shard over GPUs with no
collective on the data path (weak scaling: every rank runs its own batch of 32); torch.distributed
is used only for the barrier, the max-over-ranks of the elapsed time and the per-rank report.

Launching N > 1 ranks: either under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or plainly as `python bench.py --gpus N`:
without WORLD_SIZE in the environment the script starts the N ranks itself (one process per visible
device, rendezvous on 127.0.0.1) and exits non-zero unless exactly N ranks report.  Replaces the
reference's single-process nn.DataParallel scatter (main.py:106-108).

BASELINE.json configs[4] (4096 labels x 512 tokens, d_model 1024, batch 8192 over 8 GPUs) is
    python bench.py --gpus 8 --workload synthetic4096 --batch 1024 --steps 2 --warmup 1

Rank 0 prints ONE JSON line with, besides the contract fields,
  ranks_seen / per_rank   which ranks reported and each one's own samples/s,
  cross_rank_check        SURVEY.md 8e's contract, checked across real devices: every rank holds the SAME weights and
                          its OWN batch; after the timed region rank 0 re-runs the first samples of every other rank's
                          batch on its own device and compares the logits that rank computed bit for bit (exit 4 on a
                          mismatch).  Under "nccl" rank 0 also requires one physical device per rank (exit 5),
  roofline      the dominant kernel class (fp32-MFMA GEMM): algorithmic FLOPs of its launches divided
                by their HIP-event durations (events recorded by liblamp_hip.so on the launch stream,
                in an instrumented replay of the same K steps right after the timed region);
                `frac_kernel_only` = the same FLOPs over kernel-only durations from a `rocprofv3 --kernel-trace
                --stats` sub-run of this invocation (N = 1; `kernel_trace` holds its per-kernel table;
                --no-kernel-trace or a missing profiler falls back to the committed profile, and says so);
                `traffic` = HBM-side bytes per GEMM launch from the committed rocprofv3 PMC passes
                (profiles/hbm_traffic.json, written by tools/summarize_profiles.py),
  forward       whole-forward achieved fraction of the fp32 MFMA roof with F_live of SURVEY.md 8d,
  workloads     (N = 1) the other GPU configurations of BASELINE.json -- bibtex, delicious, synthetic4096 --
                each for a bounded number of steps: value, ms_per_step, GEMM roofline, attention TFLOP/s; and
                reuters_ragged, the headline model on lengths U{20..302} (SURVEY.md 8d variant ii),
  cpu_baseline  the oracle (a port of the reference's op sequence, `as_written`, autograd graph
                built as the reference's test loop does) timed on this host's cores on a bounded
                sample (N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from lamp_amd import hostcpu  # noqa: E402

TORCH_DEFAULT_THREADS = torch.get_num_threads()   # before main() fits the intra-op pool to the container's CPU quota

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0
DEVICE_WARMUP_S = 0.5           # fixed device warm-up before every timed region (clock ramp), stated in `config`
PIPELINED_WINDOW_S = 0.4        # the batches-in-flight rates are measured over at least this much device time

RAGGED = {'reuters': (20, 302), 'bibtex': (10, 150), 'delicious': (5, 60)}  # SURVEY.md 8d length variant (ii)
WORKLOADS = {
    # name: V, L, T, d, d_ff, heads, label_mask, pos_emb, prior p
    'reuters': dict(V=23666, L=90, T=302, d=512, dff=512, h=4, mask='prior', pos=True, p=0.10),
    'bibtex': dict(V=1840, L=159, T=100, d=512, dff=1024, h=4, mask='prior', pos=False, p=0.05),
    'delicious': dict(V=504, L=983, T=40, d=1024, dff=2048, h=8, mask='none', pos=False, p=0.0),
    'synthetic4096': dict(V=32004, L=4096, T=512, d=1024, dff=2048, h=8, mask='prior', pos=True, p=0.05),
}
# bounded (batch, steps, warmup) of the secondary workloads reported beside the headline at N = 1.  synthetic4096 runs the
# PER-GPU SHARE of BASELINE.json configs[4] (batch 8192 over 8 GPUs = 1024 samples per GPU, micro-batched inside
# lamp_forward): ~3 s per step, so one warm-up step, two timed, one instrumented.
EXTRA_WORKLOADS = (('bibtex', 32, 100, 10), ('delicious', 32, 20, 3), ('synthetic4096', 1024, 2, 1))


KERNEL_TRACE_STEPS = 60         # forwards of the rocprofv3 --kernel-trace --stats sub-run behind roofline.frac_kernel_only
KERNEL_TRACE_TIMEOUT_S = 150


def under_profiler():
    """True when this process was itself started by rocprofv3 / rocprof (no nested profiler then)."""
    return any(k.startswith(('ROCPROF', 'ROCP_TOOL', 'ROCPROFILER')) for k in os.environ) or \
        'rocprofiler' in os.environ.get('LD_PRELOAD', '')


def is_chain_kernel(name):
    """The decoder chain launch in any of its forms (chain.hip: chain_kernel, chain_packed_kernel, chain_rows4_kernel)."""
    return 'chain_' in name and 'kernel' in name and 'pack_weight' not in name


def live_kernel_trace(args):
    """Kernel-only durations measured IN THIS INVOCATION: `rocprofv3 --kernel-trace --stats` around a short run of this
    same script on the same workload (no counters: tracing only, the mode the MI355X guide's profiling recipe starts
    with), read back from its p_kernel_stats.csv.  -> dict, or a dict with only 'skipped' when the profiler is absent, this
    process already runs under it, or the sub-run fails / times out (the bench line then falls back to the committed
    profile's durations, marked as such).  The GPU is idle in this process while the sub-run holds it."""
    import csv
    import shutil
    import tempfile
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return {'skipped': 'rocprofv3 not found'}
    if under_profiler():
        return {'skipped': 'this process already runs under the profiler'}
    try:
        out = tempfile.mkdtemp(prefix='lamp_bench_trace_', dir='/tmp' if os.path.isdir('/tmp') else None)
    except OSError as e:
        return {'skipped': 'no scratch directory for the profiler output: %s' % e}
    cmd = [exe, '--kernel-trace', '--stats', '-d', out, '-o', 'p', '-f', 'csv', '--', sys.executable,
           os.path.join(ROOT, 'bench.py'), '--workload', args.workload, '--batch', str(args.batch),
           '--steps', str(KERNEL_TRACE_STEPS), '--warmup', '10', '--no-cpu-baseline', '--no-extra-workloads', '--no-pipelined',
           '--no-kernel-trace', '--no-pmc']
    env = dict(os.environ, TMPDIR='/tmp', LAMP_BENCH_INNER='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'LAMP_BENCH_SPAWNED'):
        env.pop(k, None)
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                           timeout=KERNEL_TRACE_TIMEOUT_S)
        path = os.path.join(out, 'p_kernel_stats.csv')
        if r.returncode != 0 or not os.path.exists(path):
            return {'skipped': 'rocprofv3 sub-run failed (exit %d)' % r.returncode}
        with open(path) as f:
            rows = list(csv.DictReader(f))
    except subprocess.TimeoutExpired:
        return {'skipped': 'rocprofv3 sub-run exceeded %d s' % KERNEL_TRACE_TIMEOUT_S}
    except (OSError, ValueError, KeyError) as e:
        return {'skipped': 'rocprofv3 sub-run: %s' % e}
    finally:
        shutil.rmtree(out, ignore_errors=True)
    # one embed launch per forward (per micro-batch for the large batches: then per-forward figures are per micro-batch)
    fwd = sum(int(r['Calls']) for r in rows if 'embed_plan_kernel' in r['Name'] or 'embed_packed_kernel' in r['Name']
              or 'embed_kernel' in r['Name'])
    if not fwd:
        return {'skipped': 'no lamp_forward launches in the trace'}
    lamp = [r for r in rows if 'lamp::' in r['Name']]
    short = lambda n: n.split('(')[0].replace('void lamp::', '').replace('lamp::', '')
    is_gemm = lambda n: 'gemm_nt_kernel' in n or is_chain_kernel(n)
    by_kernel = {}
    for r in sorted(lamp, key=lambda r: -float(r['TotalDurationNs'])):
        by_kernel[short(r['Name'])] = {'launches_per_forward': int(r['Calls']) / fwd,
                                       'avg_us': float(r['AverageNs']) / 1e3,
                                       'us_per_forward': float(r['TotalDurationNs']) / fwd / 1e3}
    return {
        'command': 'rocprofv3 --kernel-trace --stats -- python bench.py --workload %s --batch %d --steps %d ... (this '
                   'invocation, %.0f s)' % (args.workload, args.batch, KERNEL_TRACE_STEPS, time.perf_counter() - t0),
        'forwards_traced': fwd,
        'gemm_class_us_per_forward': sum(float(r['TotalDurationNs']) for r in lamp if is_gemm(r['Name'])) / fwd / 1e3,
        'all_kernels_us_per_forward': sum(float(r['TotalDurationNs']) for r in lamp) / fwd / 1e3,
        'by_kernel': by_kernel,
    }

PMC_PASSES = (('FETCH_SIZE',), ('WRITE_SIZE',), ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES'))
PMC_TIMEOUT_S = 120


def live_pmc(args):
    """Hardware counters measured IN THIS INVOCATION (N = 1): three short `rocprofv3 --kernel-trace --pmc ...` passes around this
    same script -- FETCH_SIZE, WRITE_SIZE (the L2s' memory-side traffic; separate passes, FETCH_SIZE doubled per the gfx950
    calibration of MI355X_MICROARCH.md) and the matrix-pipe busy cycles -- read back per dispatch and averaged per launch of each
    kernel.  Counters are never combined with any trace domain but the kernel trace.  -> dict {'by_kernel': {name: {launches,
    avg_us, fetch_bytes, write_bytes, hbm_gbs, mfma_busy}}, ...} or {'skipped': reason}."""
    import collections
    import csv
    import shutil
    import tempfile
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return {'skipped': 'rocprofv3 not found'}
    if under_profiler():
        return {'skipped': 'this process already runs under the profiler'}
    env = dict(os.environ, TMPDIR='/tmp', LAMP_BENCH_INNER='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'LAMP_BENCH_SPAWNED'):
        env.pop(k, None)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter / 'us' -> per-dispatch values
    t0 = time.perf_counter()
    for counters in PMC_PASSES:
        try:
            out = tempfile.mkdtemp(prefix='lamp_bench_pmc_', dir='/tmp' if os.path.isdir('/tmp') else None)
        except OSError as e:
            return {'skipped': 'no scratch directory for the profiler output: %s' % e}
        cmd = [exe, '--kernel-trace', '--pmc'] + list(counters) + ['-d', out, '-o', 'p', '-f', 'csv', '--', sys.executable,
               os.path.join(ROOT, 'bench.py'), '--workload', args.workload, '--batch', str(args.batch), '--steps', '4', '--warmup', '2',
               '--no-cpu-baseline', '--no-extra-workloads', '--no-pipelined', '--no-kernel-trace', '--no-pmc']
        try:
            r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=PMC_TIMEOUT_S)
            cc, kt = os.path.join(out, 'p_counter_collection.csv'), os.path.join(out, 'p_kernel_trace.csv')
            if r.returncode != 0 or not os.path.exists(cc) or not os.path.exists(kt):
                return {'skipped': 'rocprofv3 --pmc %s sub-run failed (exit %d)' % (' '.join(counters), r.returncode)}
            with open(kt) as f:
                dur = {row['Dispatch_Id']: (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1000.0 for row in csv.DictReader(f)}
            seen = set()
            with open(cc) as f:
                for row in csv.DictReader(f):
                    name = row['Kernel_Name']
                    if 'lamp::' not in name:
                        continue
                    short = name.split('(')[0].replace('void lamp::', '').replace('lamp::', '')
                    acc[short][row['Counter_Name']].append(float(row['Counter_Value']))
                    if counters[0] == 'FETCH_SIZE' and row['Dispatch_Id'] not in seen:
                        seen.add(row['Dispatch_Id'])
                        acc[short]['us'].append(dur.get(row['Dispatch_Id'], 0.0))
                    if counters[0].startswith('SQ_') and row['Counter_Name'] == 'SQ_BUSY_CYCLES':
                        acc[short]['us_sq'].append(dur.get(row['Dispatch_Id'], 0.0))
        except subprocess.TimeoutExpired:
            return {'skipped': 'rocprofv3 --pmc sub-run exceeded %d s' % PMC_TIMEOUT_S}
        except (OSError, ValueError, KeyError) as e:
            return {'skipped': 'rocprofv3 --pmc sub-run: %s' % e}
        finally:
            shutil.rmtree(out, ignore_errors=True)
    mean = lambda v: sum(v) / len(v) if v else None   # noqa: E731
    by_kernel = {}
    for k, c in acc.items():
        if 'pack_weight' in k or not c.get('us'):
            continue
        us = mean(c['us'])
        fetch = mean(c.get('FETCH_SIZE', [])) or 0.0
        write = mean(c.get('WRITE_SIZE', [])) or 0.0
        e = {'launches': len(c['us']), 'avg_us': us, 'fetch_bytes': fetch * 1024 * 2, 'write_bytes': write * 1024}
        e['hbm_gbs'] = (e['fetch_bytes'] + e['write_bytes']) / (us * 1e-6) / 1e9 if us else None
        busy, mf, us_sq = mean(c.get('SQ_BUSY_CYCLES', [])), mean(c.get('SQ_VALU_MFMA_BUSY_CYCLES', [])), mean(c.get('us_sq', []))
        if busy and us_sq:
            clk_ghz = busy / 32 / us_sq / 1000          # SQ_BUSY_CYCLES sums the 32 shader engines
            e['clock_ghz_under_profiler'] = clk_ghz
            e['mfma_busy'] = (mf or 0.0) / (1024 * us_sq * 1000 * clk_ghz)   # of the 1024 SIMDs' cycles
        by_kernel[k] = e
    return {'command': 'rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES> -- python bench.py '
                       '--workload %s --batch %d --steps 4 ... (three passes of this invocation, %.0f s)' %
                       (args.workload, args.batch, time.perf_counter() - t0),
            'units': 'bytes per launch (FETCH_SIZE x 1 KiB x 2: gfx950 calibration; WRITE_SIZE x 1 KiB), HBM-side incl. Infinity-Cache hits',
            'by_kernel': by_kernel}


def f_live(w, n_enc=2, n_dec=2):
    """Algorithmic GEMM FLOPs per sample: dead encoder attention excluded."""
    T, L, d, dff = w['T'], w['L'], w['d'], w['dff']
    return (n_enc * 4 * T * d * dff + n_dec * (4 * L * d * d + 4 * T * d * d + 4 * L * T * d) +
            n_dec * (8 * L * d * d + 4 * L * L * d) + n_dec * 8 * L * d * dff + 2 * L * d)


def executed_flops(w, model, n_enc=2, n_dec=2):
    """GEMM-class + attention FLOPs per sample the kernels really EXECUTE: F_live minus what the weights-only
    precomputations remove (decoder layer 0's hoisted query, encoder layer 0's folded W1) and, when the label self-attention runs
    the pair kernel (attention_sparse.hip), with its 4 L^2 d replaced by 4 nnz d.  -> (flops, dict of what was subtracted)."""
    from lamp_amd.Models import LAMP
    T, L, d, dff = w['T'], w['L'], w['d'], w['dff']
    fl = float(f_live(w, n_enc, n_dec))
    sub = {}
    if LAMP.cache_layer0_query:
        sub['hoisted_dec0_query'] = 2.0 * L * d * d
    if LAMP.fold_embedding:
        sub['folded_enc0_w1'] = 2.0 * T * d * dff
    dec = model.decoder
    dk = d // w['h']
    if (LAMP.use_sparse_label_attention and getattr(dec, 'label_rows_sparse', False) and L >= 1024 and dk == 128):
        sub['blocked_label_pairs_skipped'] = n_dec * 4.0 * d * (float(L) * L - dec.label_allowed_pairs)
    return fl - sum(sub.values()), sub


def build(w, batch, device, seed=0, lengths=None, n_max=None):
    """Model + one batch.  The weights and the label graph are the same on every rank (seed 0: what replicating a
    checkpoint gives); `seed` only draws the batch, so that ranks work on different samples."""
    from lamp_amd import synthetic as R
    from lamp_amd.Models import LAMP
    n_max = n_max or max([w['T']] + list(lengths or []))
    sd = R.make_state_dict(w['V'], w['L'], n_max, w['d'], w['dff'], w['h'], 2, 2, pos_emb=w['pos'], seed=0)
    adj = R.make_adjacency(w['L'], w['p'], 0) if w['mask'] == 'prior' else None
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
    default_threads = torch.get_num_threads()   # (fitted to the CPU quota by main(): the probe below goes up to torch's own default)
    usable = hostcpu.usable_cores()      # min(affinity, cgroup quota): more threads than this only burn the quota
    candidates = sorted({t for t in (TORCH_DEFAULT_THREADS, usable, 64, 32, 16, 8) if t <= max(TORCH_DEFAULT_THREADS, 1)}, reverse=True)
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
        'default_torch_threads': TORCH_DEFAULT_THREADS, 'host_cpu_count': os.cpu_count(), 'usable_cores_under_cgroup_quota': usable,
    }


def training_step_line(timeout=120):
    """SURVEY.md 8f n4 beside the forward metric: the reference's train-loop body (train.py:34-48: forward with dropout, BCE,
    loss.backward(), torch.optim.Adam step) on the headline model, batch 32 -- tools/bench_train.py in a child process (its own
    import and build of the model; ~10 s), bounded, never part of `value`.  -> the child's line, trimmed, or a 'skipped' note."""
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'bench_train.py'), '--steps', '30', '--warmup', '5']
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout, check=True,
                             cwd=ROOT).stdout.decode()
        d = json.loads(out.strip().splitlines()[-1])
        keep = ('metric', 'value', 'unit', 'ms_per_step', 'host_issue_ms_per_step', 'batch', 'dropout', 'steps', 'dtype', 'data',
                'optimizer', 'deferred_weight_gradients', 'composite_calls', 'synchronised_split_ms')
        line = {k: d[k] for k in keep if k in d}
        line['command'] = 'python tools/bench_train.py --steps 30 --warmup 5 (child process of this invocation)'
        return line
    except Exception as e:   # the forward line must not depend on this leg
        return {'skipped': '%s: %s' % (type(e).__name__, str(e)[:200])}


def warm_device(step, seconds=DEVICE_WARMUP_S):
    """Keep the device busy for a fixed wall time so the timed region does not measure the clock ramp.  Steps longer than
    an eighth of that time (the 1024-sample share of configs[4]: seconds each) are issued one at a time."""
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    n, burst = 1, (1 if time.perf_counter() - t0 > seconds / 8 else 8)
    while time.perf_counter() - t0 < seconds:
        for _ in range(burst):
            step()
        torch.cuda.synchronize()
        n += burst
    return n


def csrc_fingerprint():
    """sha256 over the kernel sources (names and contents), first 16 hex digits: stamps a committed profile with the
    kernels it was measured on (there is no .git on the GPU box)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'lamp_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(f.encode())
            with open(os.path.join(d, f), 'rb') as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def _built_toolchain():
    """The compiler that built liblamp_hip.so and passed the ISA guard of the hand-scheduled kernels (lamp_amd/build.py)."""
    from lamp_amd import build as B
    return B.built_toolchain()


def load_traffic(workload):
    """HBM-side bytes per GEMM launch measured by the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
    separate runs, FETCH_SIZE doubled per the MI355X guide's gfx950 calibration) -- or None when no profile of this
    workload is committed.  Written by tools/summarize_profiles.py together with the fingerprint of the kernel sources
    the passes ran on: the bench line marks the figure stale when the sources have changed since."""
    path = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    try:
        with open(path) as f:
            return json.load(f).get(workload)
    except (OSError, ValueError):
        return None


def profile_steps(N, step, n_steps):
    """Instrumented replay: per-kernel-class HIP-event durations on the launch stream."""
    N.prof_reset()
    N.prof_enable(True)
    for _ in range(n_steps):
        step()
    torch.cuda.synchronize()
    N.prof_enable(False)
    prof = N.prof_read()
    N.prof_reset()
    kernels = {}
    for name, r in prof.items():
        if r['launches']:
            kernels[name] = {
                'launches_per_step': r['launches'] / n_steps,
                'us_per_step': r['ms'] * 1e3 / n_steps,
                'avg_us_per_launch': r['ms'] * 1e3 / r['launches'],
                'tflops': r['flops'] / (r['ms'] * 1e-3) / 1e12 if r['ms'] > 0 else None,
                'algorithmic_gbs': r['bytes'] / (r['ms'] * 1e-3) / 1e9 if r['ms'] > 0 else None,
            }
    return prof, kernels


def roofline_of(prof, n_steps, workload, live=None, chain_gflop=None, pmc=None, attn_prof=None):
    gemm = prof['gemm']
    tf = gemm['flops'] / (gemm['ms'] * 1e-3) / 1e12 if gemm['ms'] > 0 else 0.0
    tr = load_traffic(workload)
    launches = gemm['launches'] / n_steps if n_steps else 0
    out = {
        'bound': 'mfma', 'kernel': 'gemm_nt_kernel (fp32 MFMA 16x16x4), all launches of a forward',
        'achieved': tf, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / PEAK_FP32_MFMA_TFLOPS,
        'traffic': None,
        'launches_per_step': launches,
        'avg_launch_us': gemm['ms'] * 1e3 / gemm['launches'] if gemm['launches'] else None,
        'algorithmic_gflop_per_step': gemm['flops'] / n_steps / 1e9 if n_steps else None,
        'algorithmic_bytes_per_launch': gemm['bytes'] / gemm['launches'] if gemm['launches'] else None,
    }
    out['timing'] = ('HIP events recorded by the library around every GEMM launch in an INSTRUMENTED replay of the timed '
                     'steps (the event pairs perturb the stream: conservative); frac_kernel_only = the same FLOPs over '
                     'rocprofv3 --kernel-trace durations -- of a sub-run of this invocation when kernel_only_source says '
                     '"live", else of the committed profile')
    fresh = bool(tr) and tr.get('csrc_fingerprint') == csrc_fingerprint()
    if live and live.get('gemm_class_us_per_forward') and n_steps:
        # kernel-only: this run's FLOPs over the kernel durations of the rocprofv3 --kernel-trace sub-run of THIS invocation
        ko = gemm['flops'] / n_steps / (live['gemm_class_us_per_forward'] * 1e-6) / 1e12
        out['achieved_kernel_only'] = ko
        out['frac_kernel_only'] = ko / PEAK_FP32_MFMA_TFLOPS
        out['kernel_only_source'] = 'live: ' + live['command']
        chain_us = sum(k['us_per_forward'] for n, k in live['by_kernel'].items() if is_chain_kernel(n))
        if chain_us and chain_gflop:
            # the decoder chain launch holds three GEMMs AND their two LayerNorms / residual adds: split the class so that
            # neither hides behind the other (the class figure above stays the conservative sum)
            nt_us = live['gemm_class_us_per_forward'] - chain_us
            nt_gf = gemm['flops'] / n_steps / 1e9 - chain_gflop
            out['kernel_only_split'] = {
                'gemm_nt_kernel': {'us_per_forward': nt_us, 'gflop': nt_gf, 'tflops': nt_gf / nt_us * 1e3,
                                   'frac': nt_gf / nt_us * 1e3 / PEAK_FP32_MFMA_TFLOPS},
                'chain_kernel': {'us_per_forward': chain_us, 'gflop': chain_gflop, 'tflops': chain_gflop / chain_us * 1e3,
                                 'frac': chain_gflop / chain_us * 1e3 / PEAK_FP32_MFMA_TFLOPS,
                                 'note': 'output projection + residual, LayerNorm, FFN (two GEMMs), LayerNorm in one launch: '
                                         'the time includes the LayerNorms, the FLOPs do not'}}
    elif tr and tr.get('gemm_kernel_only_us_per_step') and n_steps:
        # this run's FLOPs over the committed profile's kernel durations: only meaningful while the kernels are the ones
        # that profile ran on
        if fresh:
            ko = gemm['flops'] / n_steps / (tr['gemm_kernel_only_us_per_step'] * 1e-6) / 1e12
            out['achieved_kernel_only'] = ko
            out['frac_kernel_only'] = ko / PEAK_FP32_MFMA_TFLOPS
            out['kernel_only_source'] = 'committed profile (profiles/hbm_traffic.json), same kernel sources'
        else:
            out['achieved_kernel_only'] = out['frac_kernel_only'] = None
            out['kernel_only_stale'] = True
    if live and live.get('skipped'):
        out['kernel_trace_skipped'] = live['skipped']
    if pmc and pmc.get('by_kernel'):
        # measured in this invocation: GEMM-class launches weighted by their launch counts
        g = [e for k, e in pmc['by_kernel'].items() if 'gemm_nt_kernel' in k or is_chain_kernel(k)]
        n = sum(e['launches'] for e in g)
        if n:
            fetch = sum(e['fetch_bytes'] * e['launches'] for e in g) / n
            write = sum(e['write_bytes'] * e['launches'] for e in g) / n
            out['traffic'] = fetch + write
            out['traffic_source'] = 'live'
            out['traffic_stale'] = False
            out['traffic_detail'] = {'unit': 'bytes per GEMM-class launch (HBM-side: L2 fabric requests incl. Infinity-Cache hits)',
                                     'fetch': fetch, 'write': write, 'launches_measured': n, 'measured': pmc['command']}
        # the masked-softmax attention kernels: HBM-side GB/s against the 8 TB/s peak and the matrix pipes' busy fraction, as
        # BASELINE.json's north_star words it ("evidenced by rocprof HBM-GB/s and MFMA-busy")
        a = {k: e for k, e in pmc['by_kernel'].items() if 'attn' in k}
        n = sum(e['launches'] for e in a.values())
        if n:
            us = sum(e['avg_us'] * e['launches'] for e in a.values()) / n
            by = sum((e['fetch_bytes'] + e['write_bytes']) * e['launches'] for e in a.values()) / n
            busy = [e['mfma_busy'] * e['launches'] for e in a.values() if e.get('mfma_busy') is not None]
            out['attention'] = {
                'kernel': 'attn16_kernel / attn_tile_kernel / attn_kernel (masked softmax attention: scores never leave the CU)',
                'bound': 'mfma', 'hbm_gbs': by / (us * 1e-6) / 1e9, 'hbm_peak_gbs': PEAK_HBM_GBS, 'hbm_frac': by / (us * 1e-6) / 1e9 / PEAK_HBM_GBS,
                'traffic_bytes_per_launch': by, 'avg_launch_us_under_profiler': us, 'mfma_busy': sum(busy) / n if busy else None,
                'algorithmic_gbs': attn_prof.get('algorithmic_gbs') if attn_prof else None,
                'achieved_tflops': attn_prof.get('tflops') if attn_prof else None,
                'frac_of_fp32_mfma_peak': attn_prof['tflops'] / PEAK_FP32_MFMA_TFLOPS if attn_prof and attn_prof.get('tflops') else None,
                'by_kernel': {k: {'avg_us': e['avg_us'], 'hbm_gbs': e['hbm_gbs'], 'mfma_busy': e.get('mfma_busy')} for k, e in a.items()},
                'source': 'live: ' + pmc['command']}
    if pmc and pmc.get('skipped'):
        out['pmc_skipped'] = pmc['skipped']
    if out.get('traffic_source') == 'live':
        pass
    elif tr and tr.get('gemm_launches'):
        out['traffic_source'] = 'committed'
        out['traffic'] = (tr['gemm_fetch_bytes'] + tr['gemm_write_bytes']) / tr['gemm_launches']
        out['traffic_stale'] = not fresh
        out['traffic_detail'] = {
            'unit': 'bytes per GEMM launch (HBM-side: L2 fabric requests incl. Infinity-Cache hits)',
            'fetch': tr['gemm_fetch_bytes'] / tr['gemm_launches'], 'write': tr['gemm_write_bytes'] / tr['gemm_launches'],
            'source': tr.get('source'), 'batch': tr.get('batch'),
            'measured': 'NOT in this run: read from the committed PMC profile; traffic_stale says whether the kernel '
                        'sources have changed since (fingerprint %s then)' % tr.get('csrc_fingerprint'),
        }
    else:
        out['traffic_detail'] = 'no committed PMC profile of this workload (profiles/hbm_traffic.json)'
    return out


def measure(N, name, w, batch, steps, warmup, device, rank, sync, lengths=None, graph=False, n_max=None):
    """Build one workload on `device`, warm up, time exactly `steps` steps between barriers.  -> dict."""
    model, sd, adj, seq, pos = build(w, batch, device, seed=rank, lengths=lengths, n_max=n_max)
    src = (seq.to(device), pos.to(device))

    def step():
        return model(src, None, None, None)

    for _ in range(max(warmup, 1)):
        out = step()
    torch.cuda.synchronize()
    run = step
    g = None
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = step()
        run = g.replay
    warm_n = warm_device(run)

    sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sync()
    logits = out[0]
    assert torch.isfinite(logits).all(), 'non-finite logits'
    return dict(model=model, sd=sd, adj=adj, seq=seq, pos=pos, step=step, run=run, elapsed=elapsed,
                device_warmup_steps=warm_n, graph=g)


def device_identity(index):
    """A string that is the same for two ranks iff they sit on the same physical GPU, whatever HIP_VISIBLE_DEVICES says.
    -> (identity, strong): strong = the PCI / uuid attributes were there; without them the string falls back to the host
    name, the rank's own device-visibility variables and the index (distinct per rank under one-device-per-rank launchers,
    but not proof of distinct hardware -- rank 0 then reports the device count as a warning instead of failing the run)."""
    p = torch.cuda.get_device_properties(index)
    parts = [str(getattr(p, a, '') or '') for a in ('pci_domain_id', 'pci_bus_id', 'pci_device_id', 'uuid')]
    strong = any(x not in ('', '0', 'None') for x in parts)
    if strong:
        return ':'.join(parts), True
    vis = ','.join('%s=%s' % (k, os.environ[k]) for k in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES')
                   if k in os.environ)
    return '%s|%s|index-%d' % (socket.gethostname(), vis, index), False


def batch_of_rank(args, w_base, rank):
    """The (workload dict, lengths) of rank `rank`'s batch -- every rank can rebuild every other rank's batch."""
    w = dict(w_base)
    lengths = None
    if args.ragged:
        lo, hi = RAGGED[args.workload]
        g = torch.Generator().manual_seed(1000 + rank)
        lengths = torch.randint(lo, hi + 1, (args.batch,), generator=g).tolist()
        w['T'] = max(lengths)  # padded length of this batch: what F_live counts (the kernels skip the PAD positions)
    return w, lengths


def gemm_flops_per_step(w, batch, n_tok, n_enc=2, n_dec=2):
    """Algorithmic FLOPs of the GEMM launches of one forward: what liblamp_hip.so's launchers count for a fixed-length
    batch (n_tok = batch * T), evaluated for the real token count of a ragged one (the packed encoder and the K / V
    projections run on n_tok rows; the launchers only know the padded upper bound).  Layer 0's hoisted query
    projection is not counted, as in the library."""
    L, d, dff = w['L'], w['d'], w['dff']
    enc = n_enc * 4 * n_tok * d * dff + n_dec * 4 * n_tok * d * d
    dec = batch * L * ((n_dec - 1) * 2 * d * d + n_dec * 2 * d * d + n_dec * 8 * d * d + n_dec * 8 * d * dff)
    return float(enc + dec)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per visible device."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0',
                   LAMP_BENCH_SPAWNED='1')
        env.setdefault('OMP_NUM_THREADS', str(max(1, hostcpu.usable_cores() // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0:
                rc = rc or code
                for q in alive:      # a dead rank leaves the others in a barrier: stop exactly the ones we started
                    q.terminate()
    return rc


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
    ap.add_argument('--no-extra-workloads', action='store_true',
                    help='skip the bounded bibtex / delicious / synthetic4096 runs reported beside the headline (N = 1)')
    ap.add_argument('--ragged', action='store_true',
                    help='sequence lengths U{lo..hi} padded to the batch maximum (SURVEY.md 8d variant ii) instead of fixed T')
    ap.add_argument('--no-kernel-trace', action='store_true',
                    help='skip the rocprofv3 --kernel-trace --stats sub-run behind roofline.frac_kernel_only (N = 1)')
    ap.add_argument('--no-pmc', action='store_true',
                    help='skip the three rocprofv3 --pmc sub-runs behind roofline.traffic / roofline.attention (N = 1)')
    ap.add_argument('--no-chain-packs', action='store_true',
                    help='A/B switch: run the decoder chain launch from the native weight layouts (no weights-only repack)')
    ap.add_argument('--no-embed-fold', action='store_true',
                    help="A/B switch: launch encoder layer 0's first FFN GEMM instead of gathering its hidden rows from the "
                         'weights-only folded tables (LAMP.fold_embedding)')
    ap.add_argument('--no-sparse-label-attention', action='store_true',
                    help='A/B switch: the dense tile kernels for the label self-attention even where the decoder flags the label '
                         'graph as sparse and unstructured (LAMP.use_sparse_label_attention; synthetic4096)')
    ap.add_argument('--mask', default=None, choices=['prior', 'none', 'inveye'], help='override the workload label mask')
    args = ap.parse_args()
    hostcpu.fit_intra_op_threads(int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1'))))   # lamp_amd/hostcpu.py

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    # ONE JSON line on stdout, whatever the libraries underneath print (gloo and RCCL write banners to fd 1): everything
    # but the result line is sent to stderr from here on
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: no HIP device is visible (there is no CPU path)')
    dev_index = local_rank % torch.cuda.device_count()  # > 1 rank per GPU only happens in the 1-GPU smoke test
    torch.cuda.set_device(dev_index)
    affinity = None
    if world > 1:
        from lamp_amd.sharding import pin_rank_to_device_cpus
        affinity = pin_rank_to_device_cpus(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    device = torch.device('cuda', dev_index)
    # "nccl" is RCCL on ROCm.  LAMP_BENCH_BACKEND=gloo lets two ranks share one GPU to smoke-test this path.
    from lamp_amd.sharding import ControlPlane
    cp = ControlPlane(rank, world, device, os.environ.get('LAMP_BENCH_BACKEND', 'nccl'))
    if world != args.gpus and rank == 0:
        print('error: --gpus %d but %d rank(s) were launched' % (args.gpus, world), file=sys.stderr)

    from lamp_amd import _native as N
    N.lib()
    if args.no_chain_packs:
        from lamp_amd.Models import LAMP
        LAMP.use_chain_packs = False
    if args.no_embed_fold:
        from lamp_amd.Models import LAMP
        LAMP.fold_embedding = False
    if args.no_sparse_label_attention:
        from lamp_amd.Models import LAMP
        LAMP.use_sparse_label_attention = False
    w_base = dict(WORKLOADS[args.workload])
    if args.mask:
        w_base['mask'] = args.mask
    w, lengths = batch_of_rank(args, w_base, rank)
    n_max = RAGGED[args.workload][1] if args.ragged else None   # one position table for every rank's batch

    m = measure(N, args.workload, w, args.batch, args.steps, args.warmup, device, rank, cp.barrier, lengths=lengths,
                graph=args.graph, n_max=n_max)
    model, step, run, my_elapsed = m['model'], m['step'], m['run'], m['elapsed']

    # host issue time per forward: how long the Python / ctypes side takes to ENQUEUE one forward (all its launches) while the
    # device still has work queued -- what eight issue loops on one host compete with (outside the timed region)
    torch.cuda.synchronize()
    n_issue = min(args.steps, 50)
    t_i = time.perf_counter()
    for _ in range(n_issue):
        run()
    host_issue_us = (time.perf_counter() - t_i) / n_issue * 1e6
    torch.cuda.synchronize()
    cp.barrier()

    # every rank reports (rank, device, elapsed, samples, tokens); rank 0 aggregates: total samples / max elapsed
    n_tok = float(sum(lengths)) if lengths else float(args.batch * w['T'])
    rows = [r.tolist() for r in cp.gather(torch.tensor(
        [float(rank), float(dev_index), my_elapsed, float(args.batch * args.steps), n_tok, float(w['T']), host_issue_us],
        dtype=torch.float64))]
    affinities = cp.gather_objects(affinity)
    ident_rows = cp.gather_objects(device_identity(dev_index))
    identities = [i for i, _ in ident_rows]
    identity_strong = all(st for _, st in ident_rows)
    elapsed = max(r[2] for r in rows)

    # ---- SURVEY.md 8e across ranks: same weights, different batches; rank 0 recomputes the first samples of every other
    # rank's batch on ITS device and compares with what that rank computed, bit for bit ----
    n_chk = min(args.batch, 64)
    own = step()[0][:n_chk].detach().clone()
    torch.cuda.synchronize()
    theirs = cp.gather(own)
    cross = None
    if rank == 0 and world > 1:
        from lamp_amd import synthetic as S
        bad, worst = [], 0.0
        for r in range(1, world):
            w_r, len_r = batch_of_rank(args, w_base, r)
            seq_r, pos_r = S.make_batch(args.batch, w_r['V'], w_r['T'], lengths=len_r, seed=r)
            mine = model((seq_r[:n_chk].to(device), pos_r[:n_chk].to(device)), None, None, None)[0].cpu()
            if not torch.equal(mine, theirs[r]):
                bad.append(r)
                worst = max(worst, float((mine.double() - theirs[r].double()).abs().max()))
        cross = {'weights': 'identical on every rank (seed 0)', 'batches': 'rank r draws its batch with seed r',
                 'recomputed_on': 'rank 0', 'ranks_checked': list(range(1, world)), 'samples_per_rank': n_chk,
                 'bitwise_equal': not bad, 'mismatching_ranks': bad, 'max_abs_diff': worst}

    # per-step latency with a device sync after every step (median and min), outside the timed region
    lat = []
    for _ in range(min(args.steps, 50)):
        torch.cuda.synchronize()
        ts = time.perf_counter()
        run()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - ts) * 1e3)
    lat.sort()

    # ---- throughput mode (reported beside `value`, never as `value`): successive batches issued round-robin
    # on two HIP streams, so that one forward's launch gaps / ramp / tail are filled by the other's kernels.  Measured
    # over a FIXED window (>= PIPELINED_WINDOW_S of device time, at least --steps steps), not over --steps iterations:
    # a 20-step run would otherwise time 16 ms, most of it the two streams' ramp ----
    pipelined = None
    if not args.graph and not args.no_pipelined:
        n_pipe = max(args.steps, int(PIPELINED_WINDOW_S / max(my_elapsed / args.steps, 1e-6)) + 1)
        pipelined = {'unit': 'samples/s per GPU', 'steps': n_pipe, 'window_s': PIPELINED_WINDOW_S,
                     'note': 'independent batches in flight on round-robin HIP streams (what evaluate.test_epoch(streams=n) '
                             'does); latency per batch is NOT reduced'}
        for depth in (2, 4):
            streams = [torch.cuda.Stream(device=device) for _ in range(depth)]
            for _ in range(2):
                for st in streams:
                    with torch.cuda.stream(st):
                        step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(n_pipe):
                with torch.cuda.stream(streams[i % depth]):
                    step()
            torch.cuda.synchronize()
            e2 = time.perf_counter() - t1
            pipelined['depth_%d' % depth] = {'value': args.batch * n_pipe / e2,
                                              'ms_per_step_amortised': e2 / n_pipe * 1e3}
        pipelined['value'] = pipelined['depth_2']['value']
        pipelined['streams'] = 2

    prof_steps = min(args.steps, 20)
    prof, kernels = profile_steps(N, step, prof_steps)

    if rank != 0:
        cp.close()
        return

    ranks_seen = sorted(int(r[0]) for r in rows)
    n_gpus = len(ranks_seen)
    # distinct devices: by PCI / uuid identity, and never fewer than the distinct device indices the ranks opened (identity
    # strings can coincide under virtualisation; indices cannot, unless every rank was given its own HIP_VISIBLE_DEVICES)
    physical = max(len(set(identities)), len({int(r[1]) for r in rows}))
    samples = sum(r[3] for r in rows)
    value = samples / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    fl = f_live(w)
    plain = not (args.ragged or args.mask)
    live = None
    if n_gpus == 1 and world == 1 and plain and not args.graph and not args.no_kernel_trace and args.batch <= 64:
        live = live_kernel_trace(args)
    # FLOPs of the decoder chain launches of one forward (lamp_amd/csrc/chain.hip; two per decoder layer, each the
    # d x d output projection and the d -> d_ff -> d FFN over the B*L label rows)
    # one chain per attention block of the decoder: n_layers_dec x (enc-dec attention + label self-attention unless
    # no_dec_self_att); each = the (h d_v) -> d output projection and the d -> d_ff -> d FFN over the B * L label rows
    rows_dec = args.batch * w['L']
    dec = model.decoder.layer_stack
    n_chains = sum(2 if hasattr(l, 'slf_attn') else 1 for l in dec)
    hdv = dec[0].enc_attn.n_head * dec[0].enc_attn.d_v
    chain_gflop = n_chains * (2.0 * rows_dec * hdv * w['d'] + 4.0 * rows_dec * w['d'] * w['dff']) / 1e9
    pmc = None
    if n_gpus == 1 and world == 1 and plain and not args.graph and not args.no_pmc and args.batch <= 64:
        pmc = live_pmc(args)
    roof = roofline_of(prof, prof_steps, args.workload if plain else None, live, chain_gflop, pmc, kernels.get('attention'))
    if args.ragged and prof['gemm']['ms'] > 0:
        # the launchers count the padded upper bound of the packed encoder's rows: use the real token count
        gf = gemm_flops_per_step(w, args.batch, n_tok)
        tf = gf * prof_steps / (prof['gemm']['ms'] * 1e-3) / 1e12
        roof.update(achieved=tf, frac=tf / PEAK_FP32_MFMA_TFLOPS, algorithmic_gflop_per_step=gf / 1e9,
                    flops_source='analytic, real token count %d of %d padded positions' % (n_tok, args.batch * w['T']))
        kernels['gemm']['tflops'] = tf
    # what the model object computed ONCE per weight version, outside every forward (all of it a function of the weights alone)
    from lamp_amd.Models import LAMP as _LAMP
    fx_head, fx_head_sub = executed_flops(w, model)
    weights_only = {
        'dec0_query': bool(_LAMP.cache_layer0_query), 'chain_packs': bool(_LAMP.use_chain_packs),
        'embed_fold': bool(_LAMP.fold_embedding), 'sparse_label_attention': 'blocked_label_pairs_skipped' in fx_head_sub,
        'note': 'decoder layer 0 query = label table x W_q (SURVEY.md G11); encoder layer 0 hidden = relu((Emb W1^T)[tok] + '
                '(Pos W1^T + b1)[pos]) -- the gather is a one-hot product, so W1 folds into the tables; both re-associations '
                'of the same linear maps, built once per weight version with lamp_linear_fwd.  F_live (SURVEY.md 8d) is NOT '
                'reduced for either; forward.executed_gflop_per_sample is.  --no-embed-fold runs the unfolded route.'}
    result = {
        'metric': 'forward samples/sec, reuters d512 2+2L 4h' if args.workload == 'reuters' else
                  'forward samples/sec, %s d%d 2+2L %dh' % (args.workload, w['d'], w['h']),
        'value': value, 'unit': 'samples/s', 'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'step_ms_synced': {'median': lat[len(lat) // 2], 'min': lat[0], 'n': len(lat)},
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'ranks_seen': ranks_seen,
        'per_rank': [{'rank': int(r[0]), 'device': int(r[1]), 'device_identity': identities[int(r[0])],
                      'value': r[3] / r[2], 'ms_per_step': r[2] / args.steps * 1e3, 'tokens_per_batch': int(r[4]),
                      'padded_length': int(r[5]), 'host_issue_us_per_forward': r[6],
                      'host_issue_frac_of_step': r[6] / (r[2] / args.steps * 1e6),
                      'cpu_affinity': affinities[int(r[0])]} for r in sorted(rows)],
        'host_issue_us_per_forward': max(r[6] for r in rows),
        'physical_devices': physical, 'physical_devices_from': 'pci/uuid' if identity_strong else 'host/visibility/index (weak)',
        'backend': cp.backend, 'control_plane_ranks': cp.ranks_in_group(),
        'cross_rank_check': cross,
        'config': {'workload': '%s: batch %d/GPU, T=%d %s, L=%d, d_model=%d, d_ff=%d, 2 enc + 2 dec graph layers, '
                               '%d heads, label_mask=%s, fp32' %
                               (args.workload, args.batch, w['T'],
                                '(lengths U{%d..%d} padded to the batch maximum; %d real tokens)' %
                                (RAGGED[args.workload] + (int(n_tok),)) if args.ragged
                                else 'fixed', w['L'], w['d'], w['dff'], w['h'], w['mask']),
                   'batch_per_gpu': args.batch, 'parallelism': 'batch-sharded x%d, no collectives' % n_gpus,
                   'launch': 'hip-graph replay' if args.graph else 'eager (one lamp_forward call per step)',
                   'launcher': 'self-spawned ranks' if os.environ.get('LAMP_BENCH_SPAWNED') else
                               ('torch.distributed.run' if world > 1 else 'single process'),
                   'backend': cp.backend, 'backend_note': cp.note,
                   'device_warmup_s': DEVICE_WARMUP_S, 'device_warmup_steps': m['device_warmup_steps'],
                   'weights_only_precomputation': weights_only,
                   'csrc_fingerprint': csrc_fingerprint(), 'hipcc': _built_toolchain(),
                   'host_threads': {'torch_default': TORCH_DEFAULT_THREADS, 'intra_op_fitted_to_cpu_quota': torch.get_num_threads(),
                                    'usable_cores': hostcpu.usable_cores()},
                   },
        'roofline': roof,
        'forward': {
            'f_live_gflop_per_sample': fl / 1e9,
            'achieved_tflops_per_gpu': value / n_gpus * fl / 1e12,
            'frac_of_fp32_mfma_peak': value / n_gpus * fl / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            'executed_gflop_per_sample': fx_head / 1e9,
            'executed_frac_of_fp32_mfma_peak': value / n_gpus * fx_head / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            'not_executed_gflop_per_sample': {k: x / 1e9 for k, x in fx_head_sub.items()},
            'kernel_time_us_per_step': sum(k['us_per_step'] for k in kernels.values()),
            'kernel_time_note': 'HIP-event durations of an INSTRUMENTED replay (an event pair around every launch adds '
                                '~2-3 us per kernel): the sum may exceed ms_per_step; rocprofv3 kernel-only figures are '
                                'under profiles/',
        },
        'kernels': kernels,
        'kernel_trace': live,
        'pmc': pmc,
        'pipelined_batches_in_flight': pipelined,
    }

    if n_gpus == 1 and not args.no_extra_workloads and args.workload == 'reuters' and plain:
        # the other GPU configurations of BASELINE.json, bounded; the headline model is dropped first
        sd_h, adj_h, seq_h, pos_h = m['sd'], m['adj'], m['seq'], m['pos']
        del model, step, run
        m.clear()
        torch.cuda.empty_cache()
        extra = {}
        for name, eb, steps, warm in EXTRA_WORKLOADS:
            we = dict(WORKLOADS[name])
            me = measure(N, name, we, eb, steps, warm, device, 0, lambda: None)
            psteps = 1 if eb > 32 else min(steps, 10)
            pe, ke = profile_steps(N, me['step'], psteps)
            v = eb * steps / me['elapsed']
            fe = f_live(we)
            fx, fx_sub = executed_flops(we, me['model'])
            rf = roofline_of(pe, psteps, name if eb == 32 else None)
            extra[name] = {
                'value': v, 'unit': 'samples/s', 'batch': eb, 'steps': steps, 'warmup': warm,
                'ms_per_step': me['elapsed'] / steps * 1e3,
                'config': 'T=%d fixed, L=%d, d_model=%d, d_ff=%d, %d heads, label_mask=%s' %
                          (we['T'], we['L'], we['d'], we['dff'], we['h'], we['mask']),
                'roofline': {'achieved': rf['achieved'], 'frac': rf['frac'], 'unit': 'TFLOP/s', 'traffic': rf['traffic'],
                             'traffic_stale': rf.get('traffic_stale')},
                'attention_tflops': ke.get('attention', {}).get('tflops'),
                # utilisation on EXECUTED FLOPs (never above 1); the F_live figure is a dense-equivalent rate: where the pair
                # kernel skips blocked label pairs it says how fast a dense implementation would have to be, not how busy the chip is
                'forward_frac_of_fp32_mfma_peak': v * fx / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                'executed_gflop_per_sample': fx / 1e9, 'f_live_gflop_per_sample': fe / 1e9,
                'not_executed_gflop_per_sample': {k: x / 1e9 for k, x in fx_sub.items()},
                'f_live_equivalent_tflops': v * fe / 1e12,
            }
            me.clear()
            torch.cuda.empty_cache()
        # SURVEY.md 8d variant (ii) of the headline configuration: the same model, sequence lengths U{20..302} padded to the batch
        # maximum (the real-data shape; `python bench.py --ragged` is the full line) -- bounded like the others
        import argparse as _ap
        ra = _ap.Namespace(ragged=True, workload='reuters', batch=32)
        wr, len_r = batch_of_rank(ra, WORKLOADS['reuters'], 0)
        mr = measure(N, 'reuters', wr, 32, 200, 20, device, 0, lambda: None, lengths=len_r, n_max=RAGGED['reuters'][1])
        vr = 32 * 200 / mr['elapsed']
        extra['reuters_ragged'] = {
            'value': vr, 'unit': 'samples/s', 'batch': 32, 'steps': 200, 'warmup': 20, 'ms_per_step': mr['elapsed'] / 200 * 1e3,
            'config': 'lengths U{%d..%d} padded to the batch maximum %d (%d real tokens), otherwise the headline model' %
                      (RAGGED['reuters'] + (wr['T'], int(sum(len_r)))),
            'forward_frac_of_fp32_mfma_peak': vr * f_live(wr) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            'note': 'F_live counts the padded length; the kernels do not compute on the padding (DESIGN.md 3b)',
        }
        mr.clear()
        torch.cuda.empty_cache()
        result['workloads'] = extra
        result['training_step'] = training_step_line()
        m.update(sd=sd_h, adj=adj_h, seq=seq_h, pos=pos_h)

    if n_gpus == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(w, m['sd'], m['adj'], m['seq'], m['pos'], args.cpu_budget)
        result['cpu_baseline'] = cb
        result['speedup_vs_cpu_as_written'] = value / cb['value']
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(result) + '\n').encode())
    rc = 0
    if n_gpus != args.gpus:
        rc = 3
    elif cross is not None and not cross['bitwise_equal']:
        print('error: cross-rank bitwise check failed on ranks %s' % cross['mismatching_ranks'], file=sys.stderr)
        rc = 4
    elif cp.nccl and physical != n_gpus:
        print('%s: %d ranks on %d physical device(s) under nccl' % ('error' if identity_strong else 'warning (no PCI / uuid '
              'attributes to tell devices apart)', n_gpus, physical), file=sys.stderr)
        if identity_strong:
            rc = 5
    cp.close()
    if rc:
        sys.exit(rc)


if __name__ == '__main__':
    main()
