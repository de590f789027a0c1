"""Batch sharding of the forward path across GPUs.

Samples never interact, so the multi-GPU scheme is static partitioning of the batch -- what
``nn.DataParallel.scatter`` does in the reference (main.py:106-108) minus the per-forward parameter
broadcast and the threads: one process per GPU, weights replicated once, contiguous chunks of the
batch per rank, NO collective on the data path.  ``torch.distributed`` is only needed to bring the
per-rank logits together when a caller wants them on one rank (evaluation metrics), or for timing
barriers.  Works with any backend ("nccl" = RCCL on GPUs, "gloo" on CPU for tests).
"""
import os

import torch


def shard_bounds(n_samples, world_size, rank):
    """[lo, hi) of the contiguous chunk `rank` owns; chunk sizes differ by at most one sample."""
    if not (0 <= rank < world_size):
        raise ValueError('rank %d outside world of %d' % (rank, world_size))
    base, extra = divmod(n_samples, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(tensors, world_size, rank):
    """Slice every tensor of a batch tuple along dim 0 to this rank's chunk."""
    n = tensors[0].size(0)
    lo, hi = shard_bounds(n, world_size, rank)
    return tuple(t[lo:hi] for t in tensors)


def gather_logits(local_logits, n_samples, group=None):
    """All-gather the per-rank (b_r, L) logits into the full (n_samples, L) tensor on every rank.

    The only collective in the package, and it is OFF the hot path: forward throughput is measured on
    the sharded logits.  Ragged chunks are padded to the largest chunk for the all_gather.
    """
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [shard_bounds(n_samples, world, r) for r in range(world)]
    max_b = max(hi - lo for lo, hi in sizes)
    L = local_logits.size(1)
    padded = local_logits.new_zeros((max_b, L))
    padded[:local_logits.size(0)] = local_logits
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    return torch.cat([bufs[r][:hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


def sharded_forward(forward_fn, src_seq, src_pos, world_size, rank):
    """Run `forward_fn(src_seq_chunk, src_pos_chunk) -> logits` on this rank's chunk."""
    seq, pos = shard_batch((src_seq, src_pos), world_size, rank)
    if seq.size(0) == 0:
        return None
    return forward_fn(seq, pos)


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/.../local_cpulist)."""
    out = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def device_local_cpus(index):
    """The CPUs of the NUMA node the GPU `index` hangs off (its PCI function's local_cpulist), or None when sysfs does not
    say (virtualised PCI, no such attribute)."""
    try:
        p = torch.cuda.get_device_properties(index)
        bdf = '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        with open('/sys/bus/pci/devices/%s/local_cpulist' % bdf) as f:
            cpus = parse_cpulist(f.read())
        return cpus or None
    except (OSError, AttributeError, ValueError, RuntimeError):
        return None


def rank_cpu_set(allowed, local_lists, local_rank):
    """Host logic of the pinning (no GPU needed): `allowed` = the CPUs this process may use, `local_lists[r]` = the CPUs local to
    the device of local rank r (or None).  Rank r gets its device's local CPUs -- split evenly among the ranks whose devices
    share that list, in rank order -- or, when the locality is unknown, an even contiguous share of `allowed`.
    -> (sorted CPU list, 'pci-locality' | 'even-split')."""
    allowed = sorted(allowed)
    n = len(local_lists)
    mine = local_lists[local_rank]
    if mine:
        mine = [c for c in mine if c in set(allowed)]
    if mine:
        sharers = [r for r in range(n) if local_lists[r] == local_lists[local_rank]]
        k, i = len(sharers), sharers.index(local_rank)
        lo, hi = shard_bounds(len(mine), k, i)
        if hi > lo:
            return mine[lo:hi], 'pci-locality'
    lo, hi = shard_bounds(len(allowed), n, local_rank)
    return (allowed[lo:hi] or allowed), 'even-split'


def pin_rank_to_device_cpus(local_rank, local_world, device_of_rank=None):
    """One process per GPU on a many-core host (the MI355X nodes have 256 CPUs on two sockets): keep this rank's Python issue
    loop -- ~20 launches per 0.75 ms forward -- and torch's host threads on the cores next to ITS device instead of letting eight
    ranks migrate over both sockets.  Replaces nothing in the reference (nn.DataParallel runs its replicas as threads of one
    process, main.py:106-108).  LAMP_NO_PIN=1 leaves the affinity alone.  -> dict for the report."""
    if os.environ.get('LAMP_NO_PIN') == '1' or not hasattr(os, 'sched_setaffinity'):
        return {'pinned': False, 'reason': 'disabled' if os.environ.get('LAMP_NO_PIN') == '1' else 'no sched_setaffinity'}
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    device_of_rank = device_of_rank or (lambda r: r % max(n_dev, 1))
    lists = [device_local_cpus(device_of_rank(r)) if n_dev else None for r in range(local_world)]
    allowed = os.sched_getaffinity(0)
    cpus, how = rank_cpu_set(allowed, lists, local_rank)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError as e:
        return {'pinned': False, 'reason': str(e)}
    torch.set_num_threads(max(1, min(len(cpus), 16)))
    return {'pinned': True, 'cpus': len(cpus), 'first_cpu': cpus[0], 'last_cpu': cpus[-1], 'source': how}


class ControlPlane:
    """torch.distributed as a CONTROL plane only (the forward has no collective): bench.py's barrier around the timed
    region, its gathers of per-rank reports and of a few logits for the cross-rank bitwise check, and run_eval's final
    combination of the per-rank prediction rows.

    The rendezvous and the default group are gloo (CPU, cannot fail on GPU topology); the barrier and the gathers run on a
    "nccl" (= RCCL over xGMI on ROCm) group when it works -- first use is probed with one all_reduce, and every rank
    agrees on the outcome through gloo -- so a node where RCCL cannot initialise still produces a line, with
    `backend: "gloo"` and the reason in `config.backend_note`.  want='gloo' (bench.py: LAMP_BENCH_BACKEND=gloo, run_eval: LAMP_EVAL_BACKEND=gloo) skips RCCL (two ranks sharing
    the one GPU of a test box)."""

    def __init__(self, rank, world, device, want):
        self.rank, self.world, self.device = rank, world, device
        self.backend, self.note, self.group, self.dist = None, None, None, None
        if world == 1 and not os.environ.get('LAMP_FORCE_DIST'):   # the variable: tools/check_rccl_control_plane.py
            return
        import datetime
        import torch.distributed as dist
        self.dist = dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
        self.backend = 'gloo'
        if want != 'nccl':
            return
        ok, note = 1, None
        try:
            self.group = dist.new_group(backend='nccl', timeout=datetime.timedelta(seconds=300))
            t = torch.ones(1, device=device)
            dist.all_reduce(t, group=self.group)
            torch.cuda.synchronize()
            ok = int(t.item() == world)
            if not ok:
                note = 'RCCL all_reduce over %d ranks returned %r' % (world, t.item())
        except Exception as e:  # noqa: BLE001 -- whatever RCCL raises here, the bench falls back to gloo and says so
            ok, note = 0, '%s: %s' % (type(e).__name__, e)
        flag = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        notes = [None] * world
        dist.all_gather_object(notes, note)
        if int(flag.item()) == 1:
            self.backend = 'nccl'
        else:
            self.group = None
            self.note = 'nccl (RCCL) control plane unavailable, gloo used: ' + '; '.join(
                'rank %d: %s' % (r, n) for r, n in enumerate(notes) if n)

    @property
    def nccl(self):
        return self.backend == 'nccl'

    def barrier(self):
        if self.dist is None:
            return
        if self.nccl:
            self.dist.barrier(group=self.group, device_ids=[self.device.index])
        else:
            self.dist.barrier()

    def ranks_in_group(self):
        if self.dist is None:
            return 1
        return self.dist.get_world_size(group=self.group) if self.nccl else self.dist.get_world_size()

    def gather(self, t):
        """All ranks' copies of tensor `t` (same shape and dtype everywhere), as CPU tensors, rank order."""
        if self.dist is None:
            return [t.detach().cpu()]
        src = t.detach().to(self.device if self.nccl else 'cpu').contiguous()
        out = [torch.empty_like(src) for _ in range(self.world)]
        self.dist.all_gather(out, src, group=self.group if self.nccl else None)
        return [o.cpu() for o in out]

    def gather_objects(self, obj):
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist is not None:
            self.barrier()
            self.dist.destroy_process_group()
