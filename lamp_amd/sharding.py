"""Batch sharding of the forward path across GPUs.

Samples never interact, so the multi-GPU scheme is static partitioning of the batch -- what
``nn.DataParallel.scatter`` does in the reference (main.py:106-108) minus the per-forward parameter
broadcast and the threads: one process per GPU, weights replicated once, contiguous chunks of the
batch per rank, NO collective on the data path.  ``torch.distributed`` is only needed to bring the
per-rank logits together when a caller wants them on one rank (evaluation metrics), or for timing
barriers.  Works with any backend ("nccl" = RCCL on GPUs, "gloo" on CPU for tests).
"""
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
