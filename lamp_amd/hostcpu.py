"""How many host cores this process may really use, and the intra-op thread count that fits them.

A GPU box hands a container a CPU *quota* (cgroup cpu.max, e.g. 16 cores' worth of time per 100 ms) on a host whose
`nproc` says 256.  torch sizes its OpenMP pool from nproc: the first CPU op above the parallel grain (the (n, L) prediction
matrix of an evaluation epoch, 270 k floats) wakes 256 threads, each of which spins for a while after the region ends;
a few such regions use up the whole quota, and the kernel then stops EVERY thread of the process -- the one that issues
kernel launches included -- until the next 100 ms period.  Measured on the evaluation epoch (profiles/r06_eval_epoch_threads.txt):
epochs of 60 / 62 / 71 / 90 / 83 ms with the pool at 256, 63 / 62 / 62 / 62 / 63 ms with it fitted; cpu.stat nr_throttled 20 vs 0.

The reference has no counterpart (it leaves torch's defaults alone); this is host plumbing of the evaluation / training
drivers (lamp_amd/run_eval.py, lamp_amd/evaluate.test_epoch, bench.py), not part of the operator surface.
"""
import os

import torch

_CGROUP_V2 = '/sys/fs/cgroup/cpu.max'
_CGROUP_V1 = ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us')


def parse_cpu_max(text):
    """cgroup v2 cpu.max ('<quota> <period>' or 'max <period>') -> cores (float) or None for no limit / unreadable."""
    parts = text.split()
    if len(parts) != 2 or parts[0] == 'max':
        return None
    try:
        quota, period = int(parts[0]), int(parts[1])
    except ValueError:
        return None
    return quota / period if quota > 0 and period > 0 else None


def cpu_quota():
    """-> cores' worth of CPU time this cgroup may use per period (float), or None when there is no limit."""
    try:
        with open(_CGROUP_V2) as f:
            return parse_cpu_max(f.read())
    except OSError:
        pass
    try:
        with open(_CGROUP_V1[0]) as q, open(_CGROUP_V1[1]) as p:
            return parse_cpu_max('%s %s' % (q.read().strip(), p.read().strip()))
    except OSError:
        return None


def usable_cores():
    """min(cores in the affinity mask, cgroup quota), at least 1."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = cpu_quota()
    if quota is not None:
        n = min(n, int(quota))
    return max(n, 1)


def fitted_threads(usable, current, share=1):
    """Half of this process's share of the usable cores (the other half stays free for the issuing thread, the producer
    thread and the HIP runtime's own threads), never more than torch already uses, at least 1.  `share` = processes of the
    job inside the same cgroup (one rank per GPU on one node: the world size)."""
    return max(1, min(current, usable // (2 * max(int(share), 1))))


_fitted = None


def fit_intra_op_threads(share=1):
    """Lower torch's intra-op thread count to what the quota allows (never raises it; once per process).  -> the count."""
    global _fitted
    if _fitted is None and os.environ.get('LAMP_EVAL_NO_FIT'):     # (measurement hook: tools/bench_eval_epoch.py's A/B)
        _fitted = torch.get_num_threads()
    if _fitted is None:
        want = fitted_threads(usable_cores(), torch.get_num_threads(), share)
        if want < torch.get_num_threads():
            torch.set_num_threads(want)
        _fitted = torch.get_num_threads()
    return _fitted
