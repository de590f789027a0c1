"""N > 1 path on CPU: two gloo processes shard a batch, each runs the forward on its chunk (the CPU
oracle stands in for the HIP path here -- tests may use it), and the gathered logits must equal the
single-process result bit for bit.  Also checks the shard arithmetic."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from lamp_amd import sharding
from oracle import lamp_ref as R


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 32, 33, 8192):
        for w in (1, 2, 3, 8):
            chunks = [sharding.shard_bounds(n, w, r) for r in range(w)]
            assert chunks[0][0] == 0 and chunks[-1][1] == n
            assert all(chunks[i][1] == chunks[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in chunks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        V, L, T, d, dff, h, B = 60, 10, 9, 32, 64, 4, 7   # B = 7 -> ragged chunks 4 + 3
        sd = R.make_state_dict(V, L, T, d, dff, h, 1, 1, seed=3)
        blocked = R.label_block_mask(R.make_adjacency(L, 0.3, 3), 'prior', L)
        seq, pos = R.make_batch(B, V, T, lengths=[9, 2, 5, 9, 1, 7, 3], seed=3)

        def fwd(s, p):
            with torch.no_grad():
                return R.forward(sd, s, p, h, blocked)[0]

        local = sharding.sharded_forward(fwd, seq, pos, world, rank)
        lo, hi = sharding.shard_bounds(B, world, rank)
        assert local.shape == (hi - lo, L)
        full = sharding.gather_logits(local, B)
        if rank == 0:
            torch.save({'sharded': full, 'single': fwd(seq, pos)}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shards_match_single_process(tmp_path):
    out = str(tmp_path / 'res.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    assert res['sharded'].shape == res['single'].shape == (7, 10)
    # per-sample arithmetic is independent of the shard: identical padded length T in both runs
    assert torch.equal(res['sharded'], res['single'])
