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


def _control_plane_worker(rank, world, port, want, out_dir):
    """sharding.ControlPlane (the only use bench.py makes of torch.distributed) between two CPU processes."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    cp = sharding.ControlPlane(rank, world, torch.device('cpu'), want)
    try:
        cp.barrier()
        rows = cp.gather(torch.tensor([float(rank), 10.0 + rank], dtype=torch.float64))
        logits = cp.gather(torch.full((3, 5), float(rank)))
        names = cp.gather_objects('dev-%d' % rank)
        torch.save({'backend': cp.backend, 'note': cp.note, 'rows': rows, 'logits': logits, 'names': names,
                    'ranks': cp.ranks_in_group()}, os.path.join(out_dir, 'cp%d.pt' % rank))
    finally:
        cp.close()


@pytest.mark.parametrize('want', ['gloo', 'nccl'])
def test_bench_control_plane_two_ranks(tmp_path, want):
    """want='nccl' on this GPU-less box exercises the fallback: RCCL cannot start, every rank agrees on gloo and the
    reason is kept for the result line."""
    mp.spawn(_control_plane_worker, args=(2, _free_port(), want, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(tmp_path / ('cp%d.pt' % r))
        assert res['backend'] == 'gloo' and res['ranks'] == 2
        assert (res['note'] is None) == (want == 'gloo')
        assert [t.tolist() for t in res['rows']] == [[0.0, 10.0], [1.0, 11.0]]
        assert [float(t[0, 0]) for t in res['logits']] == [0.0, 1.0] and res['names'] == ['dev-0', 'dev-1']


def test_bench_batches_are_rebuildable_by_every_rank():
    """The cross-rank bitwise check relies on rank 0 rebuilding rank r's batch exactly."""
    import argparse
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    import bench
    from lamp_amd import synthetic as S
    args = argparse.Namespace(ragged=True, workload='reuters', batch=8)
    w1, len1 = bench.batch_of_rank(args, bench.WORKLOADS['reuters'], 1)
    w1b, len1b = bench.batch_of_rank(args, bench.WORKLOADS['reuters'], 1)
    w0, len0 = bench.batch_of_rank(args, bench.WORKLOADS['reuters'], 0)
    assert len1 == len1b and w1 == w1b and len1 != len0 and w1['T'] == max(len1) and 20 <= min(len1)
    a = S.make_batch(8, w1['V'], w1['T'], lengths=len1, seed=1)
    b = S.make_batch(8, w1['V'], w1['T'], lengths=len1, seed=1)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # analytic GEMM FLOPs of a fixed-length reuters step = what the library's launchers count (69.2 GFLOP)
    gf = bench.gemm_flops_per_step(bench.WORKLOADS['reuters'], 32, 32 * 302)
    assert abs(gf / 1e9 - 69.22) < 0.05


def test_rank_cpu_sets_follow_device_locality():
    """sharding.rank_cpu_set (host logic of the per-rank CPU pinning of bench.py --gpus N / run_eval -gpus N): ranks whose devices
    share a NUMA node split its CPUs in rank order; unknown locality falls back to an even share of the allowed CPUs; the sets
    of one node never overlap."""
    from lamp_amd import sharding as S
    assert S.parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11] and S.parse_cpulist('') == []
    node0, node1 = list(range(0, 64)) + list(range(128, 192)), list(range(64, 128)) + list(range(192, 256))
    lists = [node0] * 4 + [node1] * 4
    allowed = set(range(256))
    sets = [S.rank_cpu_set(allowed, lists, r) for r in range(8)]
    assert all(how == 'pci-locality' and len(c) == 32 for c, how in sets)
    assert set(sets[0][0]) <= set(node0) and set(sets[5][0]) <= set(node1)
    flat = [c for cs, _ in sets for c in cs]
    assert len(flat) == len(set(flat)) == 256
    # no sysfs locality (virtualised PCI): an even contiguous share
    sets = [S.rank_cpu_set(set(range(16)), [None] * 8, r) for r in range(8)]
    assert [c for c, _ in sets] == [[2 * r, 2 * r + 1] for r in range(8)] and all(h == 'even-split' for _, h in sets)
    # eight ranks sharing one device (the one-GPU rehearsal): its CPUs split eight ways; a restricted cpuset is honoured
    cs, how = S.rank_cpu_set({0, 1, 2, 3}, [[0, 1, 2, 3, 4, 5, 6, 7]] * 2, 1)
    assert cs == [2, 3] and how == 'pci-locality'
    # more ranks than local CPUs: nobody is left with an empty set
    assert S.rank_cpu_set({0, 1}, [[0, 1]] * 4, 3)[0]
