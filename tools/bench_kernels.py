#!/usr/bin/env python3
"""Micro-benchmarks of the individual HIP kernels at the shapes one forward issues (GPU box only).

    python tools/bench_kernels.py gemm      # every tile configuration x every GEMM shape
    python tools/bench_kernels.py attn
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the -DLAMP_TUNING build of the library: the only one that exports the lamp_debug_* hooks used below
os.environ.setdefault('LAMP_HIP_LIBRARY', os.path.join(ROOT, 'lamp_amd', 'liblamp_hip_tuning.so'))
from lamp_amd import _native as N  # noqa: E402

TILES = {0: 'heuristic', 1: '128x128x32', 2: '64x64x32', 3: '128x64x32', 4: '64x128x32', 5: '128x128x16',
         6: '64x64x16', 7: '128x64x16', 8: '256x128x16(8w)', 9: 'm16:32x64x32',
         10: 'm16:64x32x32', 11: 'm16:64x64x16', 12: 'm16:64x64x32', 13: 'm16:32x128x32',
         14: 'm16:128x128x32', 15: 'm16:128x128x16', 16: 'm16:128x64x32', 17: 'm16:64x128x32', 18: 'm16:128x64x16'}

def time_fn(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def gemm():
    lib = N.lib()
    force = lib.lamp_debug_force_gemm_tile
    force.argtypes = [ctypes.c_int]
    force.restype = None
    dev = torch.device('cuda:0')
    shapes = [('encFFN 9664x512x512', 9664, 512, 512), ('encKV 9664x1024x512', 9664, 1024, 512),
              ('encKVx2 9664x2048x512', 9664, 2048, 512), ('dec 2880x512x512', 2880, 512, 512),
              ('decQKV 2880x1536x512', 2880, 1536, 512), ('Q0 90x512x512', 90, 512, 512),
              ('bibtex ffn 5088x1024x512', 5088, 1024, 512), ('delic ffn1 31456x2048x1024', 31456, 2048, 1024),
              ('delic ffn2 31456x1024x2048', 31456, 1024, 2048), ('sq 4096^3', 4096, 4096, 4096)]
    print('%-28s' % 'shape' + ''.join('%16s' % TILES[t] for t in sorted(TILES)))
    for name, M, Nn, K in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(Nn, K, device=dev) / K ** 0.5
        b = torch.randn(Nn, device=dev)
        r = torch.randn(M, Nn, device=dev)
        out = torch.empty(M, Nn, device=dev)
        row = '%-28s' % name
        for t in sorted(TILES):
            force(t)

            def fn():
                N.check(lib.lamp_linear_fwd(x.data_ptr(), M, K, K, w.data_ptr(), Nn, K, b.data_ptr(),
                                            r.data_ptr(), Nn, 1, out.data_ptr(), Nn, N.stream()), 'linear')
            us = time_fn(fn, iters=10 if M * Nn * K > 1e11 else 30)
            row += '%9.1f/%5.1fT' % (us, 2.0 * M * Nn * K / us / 1e6)
        force(0)
        print(row)


def gemm_ab(cfgs=None, rounds=7):
    """A/B timing robust against clock / thermal drift: the configurations are timed round-robin, `rounds` times each,
    and the median is reported (single back-to-back sweeps differ by 3-4 % between columns for the same config)."""
    import statistics
    lib = N.lib()
    force = lib.lamp_debug_force_gemm_tile
    force.argtypes = [ctypes.c_int]
    force.restype = None
    dev = torch.device('cuda:0')
    if len(sys.argv) > 2 and cfgs is None:
        cfgs = [int(c) for c in sys.argv[2].split(',')]
    cfgs = cfgs or [0, 1, 6, 9, 11, 12, 13, 15, 18]
    shapes = [('encFFN 9664x512x512', 9664, 512, 512), ('encKV 9664x1024x512', 9664, 1024, 512),
              ('encKVx2 9664x2048x512', 9664, 2048, 512), ('dec 2880x512x512', 2880, 512, 512),
              ('decQKV 2880x1536x512', 2880, 1536, 512), ('bibtex dec 5088x512x512', 5088, 512, 512),
              ('bibtex ffn 5088x1024x512', 5088, 1024, 512), ('delic ffn1 31456x2048x1024', 31456, 2048, 1024),
              ('delic ffn2 31456x1024x2048', 31456, 1024, 2048), ('delic enc 6400x2048x1024', 6400, 2048, 1024),
              ('syn dec 131072x256x256', 131072, 256, 256), ('sq 4096^3', 4096, 4096, 4096)]
    print('%-28s' % 'shape (median us)' + ''.join('%16s' % TILES[c] for c in cfgs))
    for name, M, Nn, K in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(Nn, K, device=dev) / K ** 0.5
        b = torch.randn(Nn, device=dev)
        r = torch.randn(M, Nn, device=dev)
        out = torch.empty(M, Nn, device=dev)

        def fn():
            N.check(lib.lamp_linear_fwd(x.data_ptr(), M, K, K, w.data_ptr(), Nn, K, b.data_ptr(),
                                        r.data_ptr(), Nn, 1, out.data_ptr(), Nn, N.stream()), 'linear')
        samples = {c: [] for c in cfgs}
        for _ in range(rounds):
            for c in cfgs:
                force(c)
                samples[c].append(time_fn(fn, iters=5 if M * Nn * K > 1e11 else 20, warm=2))
        force(0)
        med = {c: statistics.median(samples[c]) for c in cfgs}
        best = min(med.values())
        print('%-28s' % name + ''.join('%9.1f/%5.1fT%s' % (med[c], 2.0 * M * Nn * K / med[c] / 1e6,
                                                           '*' if med[c] <= best * 1.01 else ' ') for c in cfgs))


def lib_ab(rounds=9):
    """A/B of two BUILDS of the library (argv[2], argv[3]: paths) on the GEMM shapes of the forwards, heuristic tile,
    round-robin medians in one process -- how a kernel change is judged when box-to-box variance (3-4 %) is larger than
    the effect."""
    import statistics
    libs = [(os.path.basename(p), N.load_library(p)) for p in sys.argv[2:4]]
    dev = torch.device('cuda:0')
    shapes = [('encFFN 9664x512x512 +b relu', 9664, 512, 512, 1, 0), ('encFFN 9664x512x512 +b +R', 9664, 512, 512, 0, 1),
              ('encKVx4 9664x512x512', 9664, 512, 512, 0, 0), ('dec 2880x512x512 +R', 2880, 512, 512, 0, 1),
              ('dec 2880x512x512 relu', 2880, 512, 512, 1, 0), ('bibtex ffn 5088x1024x512', 5088, 1024, 512, 1, 0),
              ('delic ffn1 31456x2048x1024', 31456, 2048, 1024, 1, 0), ('delic ffn2 31456x1024x2048', 31456, 1024, 2048, 0, 1),
              ('delic fc 31456x1024x1024 +R', 31456, 1024, 1024, 0, 1)]
    print('%-30s' % 'shape (median us)' + ''.join('%26s' % n for n, _ in libs))
    for name, M, Nn, K, relu, res in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(Nn, K, device=dev) / K ** 0.5
        b = torch.randn(Nn, device=dev)
        r = torch.randn(M, Nn, device=dev)
        out = torch.empty(M, Nn, device=dev)
        samples = [[] for _ in libs]
        for _ in range(rounds):
            for i, (_, lib) in enumerate(libs):
                def fn():
                    N.check(lib.lamp_linear_fwd(x.data_ptr(), M, K, K, w.data_ptr(), Nn, K, b.data_ptr(),
                                                r.data_ptr() if res else None, Nn, relu, out.data_ptr(), Nn, N.stream()), 'linear')
                samples[i].append(time_fn(fn, iters=5 if M * Nn * K > 1e11 else 20, warm=2))
        med = [statistics.median(v) for v in samples]
        print('%-30s' % name + ''.join('%17.1f/%6.1fT ' % (m, 2.0 * M * Nn * K / m / 1e6) for m in med))


WALK_SHAPES = [('delic ffn1 31456x2048x1024', 31456, 2048, 1024, 1, 1, 0), ('delic ffn2 31456x1024x2048', 31456, 1024, 2048, 1, 0, 1),
               ('delic fc 31456x1024x1024 +R', 31456, 1024, 1024, 1, 0, 1), ('delic qkv 31456x(3x1024)x1024', 31456, 1024, 1024, 3, 0, 0),
               ('reuters kv 9664x(4x512)x512', 9664, 512, 512, 4, 0, 0), ('syn ffn1 65536x2048x1024', 65536, 2048, 1024, 1, 1, 0)]
WALKS = [0, 2, 4, 8, 16]


def _walk_case(lib, name, M, Nn, K, nseg, relu, res, dev):
    """-> fn launching the GEMM (nseg > 1: that many weight matrices sharing A, through lamp_mha-style segments is not
    exposed in the ABI, so the segments are emulated by ONE weight of nseg*N rows -- the same tile walk and traffic)."""
    x = torch.randn(M, K, device=dev)
    w = torch.randn(Nn * nseg, K, device=dev) / K ** 0.5
    b = torch.randn(Nn * nseg, device=dev)
    r = torch.randn(M, Nn * nseg, device=dev) if res else None
    out = torch.empty(M, Nn * nseg, device=dev)

    def fn():
        N.check(lib.lamp_linear_fwd(x.data_ptr(), M, K, K, w.data_ptr(), Nn * nseg, K, b.data_ptr(),
                                    r.data_ptr() if res else None, Nn * nseg, relu, out.data_ptr(), Nn * nseg, N.stream()), 'linear')
    return fn, (x, w, b, r, out)


def walk(rounds=5):
    """Tile-walk A/B on the shapes whose weight matrix crowds the L2: row-panel groups (0) against the W-resident walk
    with 2 / 4 / 8 / 16 column panels per group, round-robin medians (time only; traffic: tools/pmc_walk.sh)."""
    import statistics
    lib = N.lib()
    force = lib.lamp_debug_force_gemm_walk
    force.argtypes = [ctypes.c_int]
    force.restype = None
    dev = torch.device('cuda:0')
    print('%-34s' % 'shape (median us / TFLOP/s)' + ''.join('%18s' % ('walk %d' % g) for g in WALKS))
    for name, M, Nn, K, nseg, relu, res in WALK_SHAPES:
        fn, keep = _walk_case(lib, name, M, Nn, K, nseg, relu, res, dev)
        samples = {g: [] for g in WALKS}
        for _ in range(rounds):
            for g in WALKS:
                force(g)
                samples[g].append(time_fn(fn, iters=5, warm=2))
        force(-1)
        print('%-34s' % name + ''.join('%10.1f/%6.1fT ' % (statistics.median(samples[g]),
                                                            2.0 * M * Nn * nseg * K / statistics.median(samples[g]) / 1e6) for g in WALKS))
        del keep


def walk_pmc():
    """For a rocprofv3 --pmc pass: every (shape, walk) launched 3 times in a fixed order (tools/pmc_walk.sh reads the
    counters back by dispatch order)."""
    lib = N.lib()
    force = lib.lamp_debug_force_gemm_walk
    force.argtypes = [ctypes.c_int]
    force.restype = None
    dev = torch.device('cuda:0')
    for name, M, Nn, K, nseg, relu, res in WALK_SHAPES:
        fn, keep = _walk_case(lib, name, M, Nn, K, nseg, relu, res, dev)
        for g in WALKS:
            force(g)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
        del keep
    force(-1)


def gemm_gen():
    """lamp_gemm (backward-pass GEMM) on the shapes of a reuters training step, every operand layout, next to the
    tuned forward kernel on the same product where it applies."""
    dev = torch.device('cuda:0')
    print('%-34s %10s %10s' % ('product', 'us', 'TFLOP/s'))
    for M, Nn, K in ((2880, 512, 512), (9664, 512, 512), (31456, 1024, 2048)):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(Nn, K, device=dev) / K ** 0.5
        wt = w.t().contiguous()          # (K, Nn): column-stored B operand
        xt = x.t().contiguous()          # (K, M): column-stored A operand
        dy = torch.randn(M, Nn, device=dev)
        cases = [('fwd kernel  x.w^T', lambda: N.linear(x, w), M, Nn, K),
                 ('gen  row.row     x.w^T', lambda: N.matmul_nt(x, w), M, Nn, K),
                 ('gen  row.col     x.(wt)', lambda: N.matmul_nt(x, wt.t()), M, Nn, K),
                 ('gen  col.row     (xt)^T.w^T', lambda: N.matmul_nt(xt.t(), w), M, Nn, K),
                 ('gen  col.col  wgrad dy^T.x', lambda: N.matmul_nt(dy.t(), x.t()), Nn, K, M)]
        for name, fn, m_, n_, k_ in cases:
            us = time_fn(fn, iters=20)
            print('%-34s %10.1f %10.1f   (%d x %d x %d)' % (name, us, 2.0 * m_ * n_ * k_ / us / 1e6, m_, n_, k_))
    # the weight gradients of one reuters training step (28 products: 8 over the 9664 token rows, 20 over the 2880 label
    # rows, 512 x 512 each): one split-K launch + reduce each, against ONE grouped launch (lamp_gemm_grouped)
    ops = []
    for rows_, count in ((9664, 8), (2880, 20)):
        for _ in range(count):
            ops.append((torch.randn(rows_, 512, device=dev), torch.randn(rows_, 512, device=dev)))
    outs = [torch.empty(512, 512, device=dev) for _ in ops]
    flop = sum(2.0 * 512 * 512 * a.size(0) for a, _ in ops)

    def one_by_one():
        for (dy_, x_), o in zip(ops, outs):
            N.matmul_nt(dy_.t(), x_.t(), out=o)

    def grouped():
        N.matmul_nt_grouped([(dy_.t(), x_.t(), o, False) for (dy_, x_), o in zip(ops, outs)])

    for name, fn in (('28 wgrads, split-K + reduce each', one_by_one), ('28 wgrads, one grouped launch', grouped)):
        us = time_fn(fn, iters=10)
        print('%-34s %10.1f %10.1f   (%.1f GFLOP)' % (name, us, flop / us / 1e6, flop / 1e9))


def steady():
    """K = 512 at growing M: separates per-tile efficiency from launch / tail / quantisation effects."""
    lib = N.lib()
    force = lib.lamp_debug_force_gemm_tile
    force.argtypes = [ctypes.c_int]
    dev = torch.device('cuda:0')
    print('%-22s' % 'M x 512 x 512' + ''.join('%16s' % TILES[t] for t in (1, 2, 6, 7, 9, 12)))
    for M in (2880, 9664, 19328, 38656, 77312, 154624):
        x = torch.randn(M, 512, device=dev)
        w = torch.randn(512, 512, device=dev) / 512 ** 0.5
        out = torch.empty(M, 512, device=dev)
        row = '%-22d' % M
        for t in (1, 2, 6, 7, 9, 12):
            force(t)

            def fn():
                N.check(lib.lamp_linear_fwd(x.data_ptr(), M, 512, 512, w.data_ptr(), 512, 512, None, None, 0, 0,
                                            out.data_ptr(), 512, N.stream()), 'linear')
            us = time_fn(fn, iters=20)
            row += '%9.1f/%5.1fT' % (us, 2.0 * M * 512 * 512 / us / 1e6)
        force(0)
        print(row)


def sparse():
    """Label self-attention at L = 4096 with a clustered (block-diagonal) label graph: dense tile visit vs the
    active-tile list."""
    dev = torch.device('cuda:0')
    B, H, L, dk = 4, 8, 4096, 128
    q = torch.randn(B, L, H * dk, device=dev)
    k = torch.randn(B, L, H * dk, device=dev)
    v = torch.randn(B, L, H * dk, device=dev)
    o = torch.empty_like(q)
    lay = N.AttnLayout(L * H * dk, dk, H * dk, L * H * dk, dk, H * dk, L * H * dk, dk, H * dk, L * H * dk, dk, H * dk)
    for n_clusters in (1, 4, 16, 64):
        blocked = torch.ones(L, L, dtype=torch.uint8)
        edges = torch.linspace(0, L, n_clusters + 1).long().tolist()
        for lo, hi in zip(edges[:-1], edges[1:]):
            blocked[lo:hi, lo:hi] = 0
        mu8 = blocked.to(dev)
        tiles = N.active_tile_list(mu8).to(dev)
        for name, tl in (('dense', None), ('tile-list', tiles)):
            ms = N.Mask(N.LAMP_MASK_U8, 0, mu8.data_ptr(), 0, L, tl.data_ptr() if tl is not None else None,
                        tl.size(1) if tl is not None else 0)

            def fn():
                N.check(N.lib().lamp_sdpa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), None, B, H, L, L,
                                              dk, dk, dk ** -0.5, ctypes.byref(ms), ctypes.byref(lay), N.stream()),
                        'sdpa')
            us = time_fn(fn, iters=5, warm=2)
            print('L=4096, %2d clusters (%5.1f%% of tiles active)  %-9s %9.1f us' %
                  (n_clusters, 100.0 * tiles[:, 0].float().mean().item() / (tiles.size(1) - 1), name, us))


def sparse_rows():
    """Label self-attention at L = 4096 over UNSTRUCTURED graphs (symmetric Bernoulli(p) + identity, BASELINE configs[4]'s
    generator): the dense tile kernel against the pair kernel (attention_sparse.hip, LAMP_MASK_SPARSE_ROWS)."""
    from lamp_amd import synthetic as S
    dev = torch.device('cuda:0')
    B, H, L, dk = 16, 8, 4096, 128
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, L, H * dk, generator=g).to(dev)
    k = torch.randn(B, L, H * dk, generator=g).to(dev)
    v = torch.randn(B, L, H * dk, generator=g).to(dev)
    lay = N.AttnLayout(L * H * dk, dk, H * dk, L * H * dk, dk, H * dk, L * H * dk, dk, H * dk, L * H * dk, dk, H * dk)
    for p in (0.005, 0.01, 0.025, 0.05, 0.075, 0.10, 0.15):   # symmetrised: ~ 2 p of the pairs are allowed
        blocked = (S.make_adjacency(L, p, 0) == 0).to(torch.uint8)
        bits = N.pack_mask_bits(blocked).to(dev)
        allowed = int((blocked == 0).sum())
        outs, times = {}, {}
        lpq = N.lib().lamp_debug_sparse_lpq
        lpq.argtypes = [ctypes.c_int]
        lpq.restype = None
        for name, flags, lanes in (('dense', 0, 0), ('pairs', N.LAMP_MASK_SPARSE_ROWS, 8), ('pairs4', N.LAMP_MASK_SPARSE_ROWS, 4)):
            lpq(lanes)
            o = torch.empty_like(q)
            ms = N.Mask(N.LAMP_MASK_BITS_U32, flags, bits.data_ptr(), 0, bits.size(1), None, 0, allowed if flags else 0)

            def fn():
                N.check(N.lib().lamp_sdpa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), None, B, H, L, L,
                                              dk, dk, dk ** -0.5, ctypes.byref(ms), ctypes.byref(lay), N.stream()), 'sdpa')
            times[name] = time_fn(fn, iters=5, warm=2)
            outs[name] = o
        lpq(0)
        diff = max((outs['dense'] - outs[k]).abs().max().item() for k in ('pairs', 'pairs4'))
        best = min(times['pairs'], times['pairs4'])
        pair_tf = 2.0 * B * H * allowed * 2 * dk / (best * 1e-6) / 1e12
        print('L=4096 B=%d H=%d  p=%.3f (%.2f%% of the pairs allowed)  dense %9.1f us   pairs: 8 lanes/query %9.1f us (x%.2f), 4 lanes/query '
              '%9.1f us (x%.2f)   executed %.1f TFLOP/s   max|dense - pairs| %.2g' %
              (B, H, p, 100.0 * allowed / (L * L), times['dense'], times['pairs'], times['dense'] / times['pairs'], times['pairs4'],
               times['dense'] / times['pairs4'], pair_tf, diff))


def attn():
    dev = torch.device('cuda:0')
    cases = [('reuters enc-attn', 32, 4, 90, 302, 128), ('reuters self', 32, 4, 90, 90, 128),
             ('bibtex enc-attn', 32, 4, 159, 100, 128), ('bibtex self', 32, 4, 159, 159, 128),
             ('delicious self', 32, 8, 983, 983, 128),
             ('synthetic self', 4, 8, 4096, 4096, 128), ('synthetic enc', 4, 8, 4096, 512, 128)]
    for name, B, H, lq, lk, dk in cases:
        q = torch.randn(B, lq, H * dk, device=dev)
        k = torch.randn(B, lk, H * dk, device=dev)
        v = torch.randn(B, lk, H * dk, device=dev)
        o = torch.empty(B, lq, H * dk, device=dev)
        mask = (torch.rand(lq, lk, device=dev) < 0.9).to(torch.uint8)
        mask[:, 0] = 0
        lay = N.AttnLayout(lq * H * dk, dk, H * dk, lk * H * dk, dk, H * dk, lk * H * dk, dk, H * dk,
                           lq * H * dk, dk, H * dk)
        force = N.lib().lamp_debug_force_attn
        force.argtypes = [ctypes.c_int]
        small = lq <= 256
        # 0 = heuristic; bits 0-2 forced key split, bits 4-6 query blocks per workgroup (small-shape kernel)
        for mode in ((0, 0x14, 0x24, 0x34, 0x12, 0x22, 0x32, 0x42, 0x41) if small else (0, 0x100, 1, 2, 4, 0xc1, 0x92, 0x94)):   # 0x100: attn_kernel instead of attention_tile.hip
          bits = N.pack_mask_bits(mask).to(dev)
          toks = (torch.rand(B, lk, device=dev) < 0.9).long()
          masks = (('none', None), ('shared-u8', N.Mask(N.LAMP_MASK_U8, 0, mask.data_ptr(), 0, lk)),
                   ('shared-bits', N.Mask(N.LAMP_MASK_BITS_U32, 0, bits.data_ptr(), 0, bits.size(1))),
                   ('key-tokens', N.Mask(N.LAMP_MASK_KEY_TOKENS_I64, 0, toks.data_ptr(), lk, 0)))
          for mname, ms in (masks if mode == 0 else masks[2:]):
            force(mode)

            def fn():
                N.check(N.lib().lamp_sdpa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), None, B, H,
                                              lq, lk, dk, dk, dk ** -0.5,
                                              ctypes.byref(ms) if ms is not None else None, ctypes.byref(lay),
                                              N.stream()), 'sdpa')
            us = time_fn(fn, iters=10)
            force(0)
            fl = 4.0 * B * H * lq * lk * dk
            print('%-20s mode=%02x mask=%-10s %9.1f us  %6.1f TFLOP/s' % (name, mode, mname, us, fl / us / 1e6))


def chain():
    """The decoder chain launch (chain.hip) on its own against the five launches it replaces, same operands: median us of
    both, and the per-workgroup phase timeline of the chain (tuning build: wall_clock64 stamps after the row load, fc,
    LayerNorm 1, W1, W2, LayerNorm 2).  argv[2]: rows (default 2880 = reuters batch 32)."""
    import statistics
    lib = N.lib()
    dev = torch.device('cuda:0')
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 2880
    d = dff = 512
    g = torch.Generator().manual_seed(0)
    rnd = lambda *sh: torch.randn(*sh, generator=g).to(dev)  # noqa: E731
    A, Y = rnd(M, d), rnd(M, d)
    wfc, w1, w2 = rnd(d, d) / d ** 0.5, rnd(dff, d) / d ** 0.5, rnd(d, dff) / dff ** 0.5
    b1, b2 = rnd(dff), rnd(d)
    g1, be1, g2, be2 = rnd(d), rnd(d), rnd(d), rnd(d)
    out = torch.empty(M, d, device=dev)
    H = torch.empty(M, dff, device=dev)
    tmp = torch.empty(M, d, device=dev)
    fnc = lib.lamp_debug_launch_chain
    fnc.restype = ctypes.c_int
    fnc.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong,
                    ctypes.c_int] + [ctypes.c_void_p] * 9 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_void_p] * 6
    pk = [N.weight_pack(w, f) for f in (0, 1) for w in (wfc, w1, w2)]
    hook = lib.lamp_debug_set_chain_trace
    hook.argtypes = [ctypes.c_void_p]
    hook.restype = None

    def fused():
        N.check(fnc(A.data_ptr(), d, d, Y.data_ptr(), 0, M, d, wfc.data_ptr(), g1.data_ptr(), be1.data_ptr(), w1.data_ptr(),
                    b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), g2.data_ptr(), be2.data_ptr(), dff, out.data_ptr(), N.stream(),
                    *[t.data_ptr() for t in pk]), 'chain')

    def separate():
        st = N.stream()
        N.check(lib.lamp_linear_fwd(A.data_ptr(), M, d, d, wfc.data_ptr(), d, d, None, Y.data_ptr(), d, 0, tmp.data_ptr(), d, st), 'fc')
        N.check(lib.lamp_layernorm_fwd(tmp.data_ptr(), M, d, g1.data_ptr(), be1.data_ptr(), 1e-5, tmp.data_ptr(), st), 'ln1')
        N.check(lib.lamp_linear_fwd(tmp.data_ptr(), M, d, d, w1.data_ptr(), dff, d, b1.data_ptr(), None, 0, 1, H.data_ptr(), dff, st), 'w1')
        N.check(lib.lamp_linear_fwd(H.data_ptr(), M, dff, dff, w2.data_ptr(), d, dff, b2.data_ptr(), tmp.data_ptr(), d, 0, tmp.data_ptr(), d, st), 'w2')
        N.check(lib.lamp_layernorm_fwd(tmp.data_ptr(), M, d, g2.data_ptr(), be2.data_ptr(), 1e-5, tmp.data_ptr(), st), 'ln2')
    geom = lib.lamp_debug_chain_geometry
    geom.argtypes = [ctypes.c_int]
    geom.restype = None
    GEOMS = {0: '16 waves x 32 cols, 2 reg sets, 1 slot', 1: '8 x 32, 4 sets, 2 slots', 2: '8 x 32, 2 sets, 2 slots',
             3: '8 x 32, 4 sets, 1 slot', 4: '8 waves x 64 cols, 2 sets, 1 slot',
             5: 'W direct: 16 x 32, 2 sets', 6: 'W direct: 8 x 64, 4 sets', 7: 'W packed: 16 x 32, 2 sets',
             8: 'W packed: 16 x 32, 4 sets', 9: 'W packed: 8 x 64, 2 sets', 10: 'W packed: 8 x 64, 4 sets',
             11: 'packed kernel: 16 x 32, 2 sets', 12: 'packed kernel: 16 x 32, 4 sets', 13: 'packed kernel: 8 x 64, 2 sets',
             14: 'packed kernel: 8 x 64, 4 sets', 15: '4x4x1 kernel: 4-row panels', 16: '4x4x1 kernel: 8-row panels',
             17: '4x4x1 kernel: 12-row panels', 18: '4x4x1 kernel: 16-row panels', 19: '4x4x1 kernel: 20-row panels',
             20: '4x4x1 kernel: 24-row panels'}
    if len(sys.argv) > 4:
        GEOMS = {int(k): GEOMS[int(k)] for k in sys.argv[4].split(',')}
    separate()
    for gi in sorted(GEOMS):
        geom(gi)
        out.zero_()
        fused()
        torch.cuda.synchronize()
        print('# geometry %d (%s): bitwise equal to the five launches: %s' % (gi, GEOMS[gi], torch.equal(out, tmp)))
    b = [time_fn(separate, iters=20, warm=3) for _ in range(9)]
    print('five launches %.1f us' % statistics.median(b))
    for _ in range(2):
        for gi in sorted(GEOMS):
            geom(gi)
            print('chain launch, geometry %d (%s): %.1f us' % (gi, GEOMS[gi], statistics.median(time_fn(fused, iters=20, warm=3) for _ in range(7))))
    tgeom = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    geom(tgeom)
    print('# timeline of geometry %d' % tgeom)
    n_wg = (M + 15) // 16
    buf = torch.zeros(n_wg * 24, dtype=torch.int64, device=dev)
    for _ in range(5):
        fused()
    hook(buf.data_ptr())
    fused()
    torch.cuda.synchronize()
    hook(None)
    t = buf.cpu().view(n_wg, 24).double()
    t0 = t[:, 0].min()
    ghz = (t[:, 6] / ((t[:, 5] - t[:, 0]) * 10.0)).median().item()
    print('# shader clock during the chain (s_memtime cycles / wall_clock64 time, median over workgroups): %.3f GHz' % ghz)
    names = ['rows in LDS', 'fc', 'LayerNorm 1', 'W1', 'W2', 'LayerNorm 2 / exit']
    q = lambda v, f: sorted(v.tolist())[min(len(v) - 1, int(f * len(v)))]  # noqa: E731
    print('# per-workgroup timeline (us, wall_clock64): time of the stamp since the first workgroup\'s first stamp, and phase length')
    prev = None
    for i, nm in enumerate(names):
        at = (t[:, i] - t0) * 0.01
        line = '%-20s at p50 %6.2f max %6.2f' % (nm, q(at, 0.5), at.max().item())
        if prev is not None:
            ph = (t[:, i] - prev) * 0.01
            line += '   phase p50 %6.2f max %6.2f' % (q(ph, 0.5), ph.max().item())
        print(line)
        prev = t[:, i]
    # wave 0's shader-clock stamps inside each GEMM step: prologue (first stages requested and landed), k loop, epilogue + barrier
    cyc = lambda a: '%7.0f' % q(a, 0.5)  # noqa: E731
    print('# wave 0, shader cycles (median over workgroups): prologue / k loop / epilogue + barrier / whole step')
    for gi, nm in enumerate(('fc', 'W1', 'W2')):
        g = t[:, 8 + 4 * gi: 12 + 4 * gi]
        print('%-4s %s %s %s %s   (%.2f us at %.3f GHz)' % (nm, cyc(g[:, 1] - g[:, 0]), cyc(g[:, 2] - g[:, 1]), cyc(g[:, 3] - g[:, 2]),
                                                           cyc(g[:, 3] - g[:, 0]), q(g[:, 3] - g[:, 0], 0.5) / ghz / 1e3, ghz))


def slab():
    """The slab kernel (slab.hip: one slab of rows per CU, packed weights, 4x4x1 MFMA) against the tile kernel on the token-row GEMMs:
    bit identity and round-robin median times.  argv[2:]: row counts."""
    import statistics
    lib = N.lib()
    dev = torch.device('cuda:0')
    fn = lib.lamp_debug_slab_gemm
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong,
                   ctypes.c_void_p, ctypes.c_void_p]
    rows = [int(a) for a in sys.argv[2:]] or [9665, 9664, 5000, 3200, 1283]
    for M in rows:
        for name, Nn, K, nseg, relu, res in (('W1 +b relu', 512, 512, 1, 1, 0), ('W2 +b +R', 512, 512, 1, 0, 1), ('K,V (2 x 1024)', 1024, 512, 2, 0, 0),
                                             ('ffn 1024', 1024, 512, 1, 1, 0)):
            g = torch.Generator().manual_seed(M + Nn)
            x = torch.randn(M, K, generator=g).to(dev)
            ws = [(torch.randn(Nn, K, generator=g) / K ** 0.5).to(dev) for _ in range(nseg)]
            wq = [N.weight_pack(w, 1) for w in ws]
            b = torch.randn(Nn, generator=g).to(dev) if nseg == 1 else None
            r = torch.randn(M, Nn, generator=g).to(dev) if res else None
            want = [N.linear(x, w, b, residual=r, relu=bool(relu)) for w in ws]
            outs = [torch.zeros(M, Nn, device=dev) for _ in range(nseg)]

            def run_slab():
                N.check(fn(x.data_ptr(), M, K, K, wq[0].data_ptr(), wq[1].data_ptr() if nseg > 1 else None, Nn, N.ptr(b), N.ptr(r), Nn, relu,
                           outs[0].data_ptr(), outs[1].data_ptr() if nseg > 1 else None, Nn, None, N.stream()), 'slab')

            def run_tile():
                for w in ws:
                    N.linear(x, w, b, residual=r, relu=bool(relu))
            run_slab()
            torch.cuda.synchronize()
            same = all(torch.equal(a, w_) for a, w_ in zip(outs, want))
            ts, tt = [], []
            for _ in range(7):
                tt.append(time_fn(run_tile, iters=20, warm=3))
                ts.append(time_fn(run_slab, iters=20, warm=3))
            fl = 2.0 * M * Nn * K * nseg
            ms, mt = statistics.median(ts), statistics.median(tt)
            print('M=%5d %-16s bitwise %s   tile %7.1f us (%5.1f TF)   slab %7.1f us (%5.1f TF)' % (M, name, same, mt, fl / mt / 1e6, ms, fl / ms / 1e6))


def attn_tile(rounds=7):
    """attention_tile.hip against attn_kernel (bit 8 of the tuning hook) on the long-sequence shapes, round-robin medians."""
    import statistics
    dev = torch.device('cuda:0')
    cases = [('delicious self', 32, 8, 983, 983, 128, False), ('delicious self bits', 32, 8, 983, 983, 128, True),
             ('synthetic self', 4, 8, 4096, 4096, 128, True), ('synthetic enc', 4, 8, 4096, 512, 128, False),
             ('synthetic self B=8', 8, 8, 4096, 4096, 128, True)]
    force = N.lib().lamp_debug_force_attn
    force.argtypes = [ctypes.c_int]
    for name, B, H, lq, lk, dk, masked in cases:
        q = torch.randn(B, lq, H * dk, device=dev)
        k = torch.randn(B, lk, H * dk, device=dev)
        v = torch.randn(B, lk, H * dk, device=dev)
        o = torch.empty(B, lq, H * dk, device=dev)
        mask = (torch.rand(lq, lk, device=dev) < 0.9).to(torch.uint8)
        mask[:, 0] = 0
        bits = N.pack_mask_bits(mask).to(dev)
        ms = N.Mask(N.LAMP_MASK_BITS_U32, 0, bits.data_ptr(), 0, bits.size(1)) if masked else None
        lay = N.AttnLayout(lq * H * dk, dk, H * dk, lk * H * dk, dk, H * dk, lk * H * dk, dk, H * dk,
                           lq * H * dk, dk, H * dk)
        modes = (0, 0x100)
        samples = [[] for _ in modes]
        outs = []
        for _ in range(rounds):
            for i, mode in enumerate(modes):
                force(mode)

                def fn():
                    N.check(N.lib().lamp_sdpa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), None, B, H, lq, lk,
                                                  dk, dk, dk ** -0.5, ctypes.byref(ms) if ms is not None else None,
                                                  ctypes.byref(lay), N.stream()), 'sdpa')
                samples[i].append(time_fn(fn, iters=10, warm=2))
                if len(outs) < 2:
                    outs.append(o.clone())
                force(0)
        fl = 4.0 * B * H * lq * lk * dk
        med = [statistics.median(x) for x in samples]
        # the clock the key loop runs at: shader cycles / wall clock of wave 0 of every workgroup of the last of 4 launches
        nwg = B * H * ((lq + 127) // 128)
        buf = torch.zeros(nwg * (8 + 256 + 4), dtype=torch.int64, device=dev)
        hook = N.lib().lamp_debug_set_attn_tile_trace
        hook.argtypes = [ctypes.c_void_p]
        hook.restype = None
        hook(buf.data_ptr())
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        hook(None)
        tiles_t = buf[nwg * 8:nwg * (8 + 256)].cpu().view(nwg, 256).double()
        phases = buf[nwg * (8 + 256):].cpu().view(nwg, 4).double()
        t = buf[:nwg * 8].cpu().view(-1, 8).double()
        if t[:, 3].max().item() == 0:   # a library without -DTILE_TRACE: no stamps
            print('%-22s tile kernel %9.1f us %6.1f TFLOP/s | attn_kernel %9.1f us %6.1f TFLOP/s | same bits: %s' %
                  (name, med[0], fl / med[0] / 1e6, med[1], fl / med[1] / 1e6, torch.equal(outs[0], outs[1])))
            continue
        ghz = statistics.median(((t[:, 1] - t[:, 0]) / ((t[:, 3] - t[:, 2]) * 10.0)).tolist())
        loop_us = statistics.median(((t[:, 3] - t[:, 2]) / 100.0).tolist())
        span_us = (t[:, 3].max() - t[:, 2].min()).item() / 100.0
        pipe = fl / 65536.0 / (med[0] * ghz * 1e3)     # MFMA cycles every SIMD needs / cycles of the launch at that clock
        if len(sys.argv) > 2 and sys.argv[2] == 'timeline':
            t0 = t[:, 2].min()
            st, en = (t[:, 2] - t0) / 100.0, (t[:, 3] - t0) / 100.0
            dur = en - st
            q = lambda x, f: x.sort().values[int(f * (x.numel() - 1))].item()
            print('   loop duration us  p5 %.0f p25 %.0f p50 %.0f p75 %.0f p95 %.0f max %.0f | start us p25 %.0f p50 %.0f p75 %.0f max %.0f | '
                  'end us p50 %.0f p95 %.0f max %.0f' % (q(dur, .05), q(dur, .25), q(dur, .5), q(dur, .75), q(dur, .95), dur.max().item(),
                                                         q(st, .25), q(st, .5), q(st, .75), st.max().item(), q(en, .5), q(en, .95), en.max().item()))
            print('   kernel entry -> first request of the first tile: median %.1f us' % statistics.median(((t[:, 2] - t[:, 7]) / 100.0).tolist()))
            cu = ((t[:, 6].long() & 15) << 8) | ((t[:, 5].long() >> 8) & 0xff)      # XCC, SE / SH / CU
            ids = cu.unique()
            busy = torch.tensor([dur[cu == i].sum().item() for i in ids])
            cnt = torch.tensor([(cu == i).sum().item() for i in ids])
            for i in ids[:3].tolist() + ids[-2:].tolist():
                sel = (cu == i).nonzero().flatten().tolist()
                print('   CU %03x (entry | loop start, end): ' % i + '  '.join('[%4.0f | %4.0f, %4.0f]' % ((t[j, 7] - t0).item() / 100.0, st[j].item(), en[j].item()) for j in sorted(sel, key=lambda j: st[j].item())))
            # tile steps of the workgroups of one CU: us per 16 tiles along each workgroup's life (2 x 8192 MFMA cycles per SIMD and
            # pair of tiles = 6.86 us at 2.39 GHz when two workgroups share the CU, 3.43 us per tile for one alone)
            nt_ = min((lk + 31) // 32, 256)
            for j in sorted((cu == ids[0]).nonzero().flatten().tolist(), key=lambda j: st[j].item()):
                tt = (tiles_t[j, :nt_] - t0) / 100.0
                steps = [(tt[min(a + 16, nt_ - 1)] - tt[a]).item() / (min(a + 16, nt_ - 1) - a) for a in range(0, nt_ - 1, 16)]
                print('   CU %03x workgroup from %4.0f us: us per tile, 16-tile windows: ' % (ids[0], st[j].item()) + ' '.join('%.2f' % x for x in steps))
                print('        wave 0, shader cycles per tile step: QK^T + softmax %.0f | PV + mask %.0f | wait for the DMA %.0f | barrier %.0f' %
                      tuple((phases[j] / max(t[j, 4].item(), 1)).tolist()))
            print('   %d CUs seen; workgroups per CU min %d max %d; sum of loop time per CU (2 slots): min %.0f median %.0f max %.0f us of a %.0f us span' %
                  (ids.numel(), cnt.min().item(), cnt.max().item(), busy.min().item(), busy.median().item(), busy.max().item(), en.max().item()))
        print('%-22s tile kernel %9.1f us %6.1f TFLOP/s | attn_kernel %9.1f us %6.1f TFLOP/s | same bits: %s | loop clock %.3f GHz, '
              'MFMA pipe busy %.3f of the launch; a workgroup\'s key loop %.1f us, first entry to last exit %.1f us' %
              (name, med[0], fl / med[0] / 1e6, med[1], fl / med[1] / 1e6, torch.equal(outs[0], outs[1]), ghz, pipe, loop_us, span_us))


def attn_lib_ab(rounds=9):
    """A/B of BUILDS of the library (argv[2:]: paths) on the attention shapes of the forwards, heuristic variant, the
    masks the forward uses (bit-packed label graph for self-attention, none for enc-dec: the padding mask of a full-length
    batch blocks nothing), round-robin medians in one process."""
    import statistics
    libs = [(os.path.basename(p), N.load_library(p)) for p in sys.argv[2:]]
    dev = torch.device('cuda:0')
    cases = [('reuters enc-attn', 32, 4, 90, 302, 128, False), ('reuters self', 32, 4, 90, 90, 128, True),
             ('bibtex enc-attn', 32, 4, 159, 100, 128, False), ('bibtex self', 32, 4, 159, 159, 128, True),
             ('delicious enc-attn', 32, 8, 983, 40, 128, False), ('delicious self', 32, 8, 983, 983, 128, False),
             ('synthetic self', 4, 8, 4096, 4096, 128, True), ('synthetic enc', 4, 8, 4096, 512, 128, False)]
    print('%-22s' % 'shape (median us)' + ''.join('%28s' % n for n, _ in libs))
    for name, B, H, lq, lk, dk, masked in cases:
        q = torch.randn(B, lq, H * dk, device=dev)
        k = torch.randn(B, lk, H * dk, device=dev)
        v = torch.randn(B, lk, H * dk, device=dev)
        o = torch.empty(B, lq, H * dk, device=dev)
        mask = (torch.rand(lq, lk, device=dev) < 0.9).to(torch.uint8)
        mask[:, 0] = 0
        bits = N.pack_mask_bits(mask).to(dev)
        ms = N.Mask(N.LAMP_MASK_BITS_U32, 0, bits.data_ptr(), 0, bits.size(1)) if masked else None
        lay = N.AttnLayout(lq * H * dk, dk, H * dk, lk * H * dk, dk, H * dk, lk * H * dk, dk, H * dk,
                           lq * H * dk, dk, H * dk)
        samples = [[] for _ in libs]
        for _ in range(rounds):
            for i, (_, lib) in enumerate(libs):
                def fn():
                    N.check(lib.lamp_sdpa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), None, B, H, lq, lk, dk,
                                              dk, dk ** -0.5, ctypes.byref(ms) if ms is not None else None,
                                              ctypes.byref(lay), N.stream()), 'sdpa')
                samples[i].append(time_fn(fn, iters=20, warm=3))
        fl = 4.0 * B * H * lq * lk * dk
        med = [statistics.median(x) for x in samples]
        print('%-22s' % name + ''.join('%19.1f/%6.1fT ' % (m, fl / m / 1e6) for m in med))


def attn_one():
    """One attention shape, heuristic variant, bit-packed shared mask, a handful of launches: small enough for a
    rocprofv3 --pmc pass (tools/pmc_ta.sh).  argv[2] = case name prefix (default 'synthetic self')."""
    dev = torch.device('cuda:0')
    want = sys.argv[2] if len(sys.argv) > 2 else 'synthetic self'
    cases = {'reuters enc-attn': (32, 4, 90, 302, 128), 'reuters self': (32, 4, 90, 90, 128),
             'delicious self': (32, 8, 983, 983, 128), 'synthetic self': (4, 8, 4096, 4096, 128)}
    B, H, lq, lk, dk = cases[want]
    q, k, v = (torch.randn(B, l, H * dk, device=dev) for l in (lq, lk, lk))
    o = torch.empty(B, lq, H * dk, device=dev)
    mask = (torch.rand(lq, lk, device=dev) < 0.9).to(torch.uint8)
    mask[:, 0] = 0
    bits = N.pack_mask_bits(mask).to(dev)
    ms = N.Mask(N.LAMP_MASK_BITS_U32, 0, bits.data_ptr(), 0, bits.size(1))
    lay = N.AttnLayout(lq * H * dk, dk, H * dk, lk * H * dk, dk, H * dk, lk * H * dk, dk, H * dk, lq * H * dk, dk, H * dk)

    def fn():
        N.check(N.lib().lamp_sdpa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), None, B, H, lq, lk, dk, dk,
                                      dk ** -0.5, ctypes.byref(ms), ctypes.byref(lay), N.stream()), 'sdpa')
    us = time_fn(fn, iters=4, warm=2)
    print('%-20s %9.1f us  %6.1f TFLOP/s' % (want, us, 4.0 * B * H * lq * lk * dk / us / 1e6))


def attn_maps():
    """The map-writing variants of the large-shape kernel (PM = 1 exact two-pass / PM = 2 single pass + scores), us."""
    dev = torch.device('cuda:0')
    for name, B, H, lq, lk, dk in (('delicious self', 8, 8, 983, 983, 128), ('synthetic self', 1, 8, 4096, 4096, 128),
                                   ('reuters enc self-attn maps', 32, 4, 302, 302, 128)):
        q, k, v = (torch.randn(B, l, H * dk, device=dev) for l in (lq, lk, lk))
        mask = (torch.rand(lq, lk, device=dev) < 0.9).to(torch.uint8)
        mask[:, 0] = 0
        bits = N.pack_mask_bits(mask).to(dev)
        toks = (torch.rand(B, lk, device=dev) < 0.9).long()
        toks[:, 0] = 1
        for mname, ms in (('shared-bits', N.Mask(N.LAMP_MASK_BITS_U32, 0, bits.data_ptr(), 0, bits.size(1))),
                          ('key-tokens', N.Mask(N.LAMP_MASK_KEY_TOKENS_I64, 0, toks.data_ptr(), lk, 0))):
            t0 = time_fn(lambda: N.sdpa_fused(q, k, v, H, ms, dk ** -0.5, need_attn=False), iters=5, warm=2)
            t1 = time_fn(lambda: N.sdpa_fused(q, k, v, H, ms, dk ** -0.5, need_attn=True), iters=5, warm=2)
            t2 = time_fn(lambda: N.sdpa_fused(q, k, v, H, ms, dk ** -0.5, need_attn=True, fast_maps=True), iters=5, warm=2)
            print('%-28s %-12s no maps %9.1f   exact two-pass maps %9.1f   single-pass maps %9.1f' % (name, mname, t0, t1, t2))


def residency():
    """Forward-GEMM shapes with the workgroups per CU limited by extra LDS (tuning build): does a launch whose tiles no
    longer fit one round -- phases out of lockstep -- beat the all-resident one?"""
    lib = N.lib()
    hook = lib.lamp_debug_set_gemm_extra_lds
    hook.argtypes = [ctypes.c_int]
    hook.restype = None
    dev = torch.device('cuda:0')
    shapes = [('encFFN 9664x512x512', 9664, 512, 512), ('dec 2880x512x512', 2880, 512, 512),
              ('decQKV 2880x1536x512', 2880, 1536, 512), ('encKVx2 9664x2048x512', 9664, 2048, 512)]
    extras = (0, 12, 20, 32, 44, 60)
    print('%-26s' % 'extra LDS KiB ->' + ''.join('%14d' % e for e in extras))
    for name, M, Nn, K in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(Nn, K, device=dev) / K ** 0.5
        b = torch.randn(Nn, device=dev)
        r = torch.randn(M, Nn, device=dev)
        out = torch.empty(M, Nn, device=dev)

        def fn():
            N.check(lib.lamp_linear_fwd(x.data_ptr(), M, K, K, w.data_ptr(), Nn, K, b.data_ptr(), r.data_ptr(), Nn, 1,
                                        out.data_ptr(), Nn, N.stream()), 'linear')
        row = '%-26s' % name
        import statistics
        samples = {e: [] for e in extras}
        for _ in range(5):
            for e in extras:
                hook(e * 1024)
                samples[e].append(time_fn(fn, iters=20, warm=3))
        hook(0)
        print(row + ''.join('%9.1f us   ' % statistics.median(samples[e]) for e in extras))


def ln():
    """Stand-alone LayerNorm launches at the shapes of one forward."""
    dev = torch.device('cuda:0')
    for M, d in ((2880, 512), (9664, 512), (5088, 512), (31456, 1024), (131072, 1024)):
        x = torch.randn(M, d, device=dev)
        g_, b_ = torch.randn(d, device=dev), torch.randn(d, device=dev)
        y = torch.empty_like(x)

        def fn():
            N.check(N.lib().lamp_layernorm_fwd(x.data_ptr(), M, d, g_.data_ptr(), b_.data_ptr(), 1e-5, y.data_ptr(), N.stream()), 'ln')
        us = time_fn(fn, iters=50)
        print('layernorm %6d x %4d  %7.2f us  %7.1f GB/s' % (M, d, us, 8.0 * M * d / us / 1e3))


def attn_trace():
    """Per-workgroup timeline of the small-shape attention kernel (stamps of wave 0 of each workgroup: entry, Q block
    staged, key loop done, exit) for the reuters / bibtex shapes, heuristic variant."""
    dev = torch.device('cuda:0')
    hook = N.lib().lamp_debug_set_attn_trace
    hook.argtypes = [ctypes.c_void_p]
    hook.restype = None
    print('%-18s %6s %8s | %-22s | %-16s %-16s %-16s' % ('shape', 'WGs', 'span', 'start p50 p90 max', 'stage Q p50 max',
                                                      'key loop p50 max', 'merge+store p50 max'))
    for name, B, H, lq, lk, dk in (('reuters enc-attn', 32, 4, 90, 302, 128), ('reuters self', 32, 4, 90, 90, 128),
                                   ('bibtex enc-attn', 32, 4, 159, 100, 128), ('bibtex self', 32, 4, 159, 159, 128)):
        q = torch.randn(B, lq, H * dk, device=dev)
        k = torch.randn(B, lk, H * dk, device=dev)
        v = torch.randn(B, lk, H * dk, device=dev)
        o = torch.empty(B, lq, H * dk, device=dev)
        lay = N.AttnLayout(lq * H * dk, dk, H * dk, lk * H * dk, dk, H * dk, lk * H * dk, dk, H * dk, lq * H * dk, dk, H * dk)
        buf = torch.zeros(8 * 8192, dtype=torch.int64, device=dev)

        def fn():
            N.check(N.lib().lamp_sdpa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), None, B, H, lq, lk, dk, dk,
                                          dk ** -0.5, None, ctypes.byref(lay), N.stream()), 'sdpa')
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        hook(buf.data_ptr())
        fn()
        torch.cuda.synchronize()
        hook(None)
        t = buf.cpu().view(-1, 8)
        rows = t[t[:, 3] != 0]
        t0 = rows[:, 0].min().item()
        us = lambda x: (x - t0) * 0.01  # noqa: E731  (10 ns ticks)
        start = sorted(us(x) for x in rows[:, 0].tolist())
        ph = [sorted(((rows[:, j + 1] - rows[:, j]).double() * 0.01).tolist()) for j in range(3)]
        qf = lambda a, f: a[min(len(a) - 1, int(f * len(a)))]  # noqa: E731
        print('%-18s %6d %8.2f | %6.2f %6.2f %6.2f   | %7.2f %7.2f  %7.2f %7.2f  %7.2f %7.2f' %
              (name, len(start), us(rows[:, 3].max().item()), qf(start, 0.5), qf(start, 0.9), start[-1],
               qf(ph[0], 0.5), ph[0][-1], qf(ph[1], 0.5), ph[1][-1], qf(ph[2], 0.5), ph[2][-1]))


def gemm_trace():
    """Per-workgroup timeline of the GEMM launches of ONE reuters forward (in situ: every launch runs behind its real
    predecessor), from the wall_clock64 stamps the tuning build records at kernel entry, after the prologue (first
    tile staged, two more in flight), after the main loop and at exit.  Prints, per launch: grid, span (first entry ->
    last exit), how the workgroups' start times spread, and the median / max length of the three phases."""
    import statistics
    sys.path.insert(0, ROOT)
    import bench
    lib = N.lib()
    hook = lib.lamp_debug_set_gemm_trace
    hook.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int]
    hook.restype = None
    dev = torch.device('cuda:0')
    wl = sys.argv[2] if len(sys.argv) > 2 else 'reuters'
    w = dict(bench.WORKLOADS[wl])
    model, sd, adj, seq, pos = bench.build(w, 32, dev)
    src = (seq.to(dev), pos.to(dev))
    for _ in range(50):
        model(src, None, None, None)
    torch.cuda.synchronize()
    slab, n_slabs = 8 * 8192, 32
    buf = torch.zeros(slab * n_slabs, dtype=torch.int64, device=dev)
    hook(buf.data_ptr(), slab, n_slabs)
    model(src, None, None, None)
    torch.cuda.synchronize()
    hook(None, 0, 0)
    t = buf.cpu().view(n_slabs, -1, 8)
    tick_ns = 10.0  # wall_clock64: 100 MHz constant clock
    print('# %s, batch 32: GEMM launches of one forward; times in us (wall_clock64, %g ns ticks)' % (wl, tick_ns))
    print('%3s %6s %8s | %-26s | %-17s %-17s %-17s | %s' % ('#', 'WGs', 'span', 'WG start: p50 p90 max', 'prologue p50 max',
                                                          'main loop p50 max', 'epilogue p50 max', 'WGs/CU max, CUs used'))
    for i in range(n_slabs):
        rows = t[i][t[i][:, 3] != 0]
        if rows.numel() == 0:
            continue
        t0 = rows[:, 0].min().item()
        us = lambda x: (x - t0) * tick_ns / 1e3  # noqa: E731
        start = sorted(us(v) for v in rows[:, 0].tolist())
        ph = [sorted(((rows[:, j + 1] - rows[:, j]).double() * tick_ns / 1e3).tolist()) for j in range(3)]
        span = us(rows[:, 3].max().item())
        hw = rows[:, 4].tolist()
        xcc = rows[:, 5].tolist()
        # HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]; XCC_ID[3:0]
        cu = [((x & 0xf), (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 0xf) for h, x in zip(hw, xcc)]
        per_cu = {}
        for c in cu:
            per_cu[c] = per_cu.get(c, 0) + 1
        q = lambda a, f: a[min(len(a) - 1, int(f * len(a)))]  # noqa: E731
        live = rows[rows[:, 2] > rows[:, 1]]
        ghz = sorted((live[:, 7].double() / ((live[:, 2] - live[:, 1]).double() * tick_ns)).tolist()) if live.numel() else [0.0]
        print('%3d %6d %8.2f | %7.2f %7.2f %7.2f    | %7.2f %7.2f   %7.2f %7.2f   %7.2f %7.2f   | %d, %d | clock %.2f GHz' %
              (i, len(start), span, q(start, 0.5), q(start, 0.9), start[-1], q(ph[0], 0.5), ph[0][-1], q(ph[1], 0.5),
               ph[1][-1], q(ph[2], 0.5), ph[2][-1], max(per_cu.values()), len(per_cu), q(ghz, 0.5)))


def gemm_clock():
    """The shader clock the GEMM main loops actually run at (tuning build: s_memtime cycles of the main loop / its
    wall_clock64 duration, median over the workgroups of the LAST of 12 back-to-back launches), beside the TFLOP/s of
    the same shape measured without stamps.  argv[2]: comma-separated tile configurations (default: heuristic)."""
    import statistics
    lib = N.lib()
    hook = lib.lamp_debug_set_gemm_trace
    hook.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int]
    hook.restype = None
    force = lib.lamp_debug_force_gemm_tile
    force.argtypes = [ctypes.c_int]
    force.restype = None
    dev = torch.device('cuda:0')
    cfgs = [int(c) for c in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0]
    shapes = [('encFFN 9664x512x512', 9664, 512, 512), ('encKVx2 9664x2048x512', 9664, 2048, 512),
              ('dec 2880x512x512', 2880, 512, 512), ('delic ffn1 31456x2048x1024', 31456, 2048, 1024),
              ('delic ffn2 31456x1024x2048', 31456, 1024, 2048), ('syn ffn1 65536x2048x1024', 65536, 2048, 1024),
              ('sq 4096^3', 4096, 4096, 4096)]
    slab = 8 * 70000
    buf = torch.zeros(slab, dtype=torch.int64, device=dev)
    print('# main-loop clock = s_memtime cycles / wall_clock64 time, median over workgroups; peak 157.3 TFLOP/s is quoted at 2.4 GHz')
    print('%-30s %-16s %9s %9s %10s %12s' % ('shape', 'tile', 'us', 'TFLOP/s', 'clock GHz', 'TF at 2.4GHz'))
    for name, M, Nn, K in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(Nn, K, device=dev) / K ** 0.5
        b = torch.randn(Nn, device=dev)
        out = torch.empty(M, Nn, device=dev)

        def fn():
            N.check(lib.lamp_linear_fwd(x.data_ptr(), M, K, K, w.data_ptr(), Nn, K, b.data_ptr(), None, Nn, 1,
                                        out.data_ptr(), Nn, N.stream()), 'linear')
        for c in cfgs:
            force(c)
            us = statistics.median(time_fn(fn, iters=8 if M * Nn * K > 1e11 else 30, warm=3) for _ in range(5))
            buf.zero_()
            hook(buf.data_ptr(), slab, 1)
            for _ in range(12):
                fn()
            torch.cuda.synchronize()
            hook(None, 0, 0)
            t = buf.cpu().view(-1, 8)
            live = t[(t[:, 3] != 0) & (t[:, 2] > t[:, 1])]
            ghz = statistics.median((live[:, 7].double() / ((live[:, 2] - live[:, 1]).double() * 10.0)).tolist()) if live.numel() else float('nan')
            tf = 2.0 * M * Nn * K / us / 1e6
            print('%-30s %-16s %9.1f %9.1f %10.3f %12.1f' % (name, TILES.get(c, str(c)), us, tf, ghz, tf * 2.4 / ghz if ghz == ghz else float('nan')))
        force(0)


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'gemm'
    {'gemm': gemm, 'gemm_ab': gemm_ab, 'lib_ab': lib_ab, 'walk': walk, 'walk_pmc': walk_pmc, 'gemm_gen': gemm_gen, 'attn': attn, 'steady': steady, 'sparse': sparse, 'sparse_rows': sparse_rows,
     'gemm_trace': gemm_trace, 'gemm_clock': gemm_clock, 'attn_lib_ab': attn_lib_ab, 'attn_tile': attn_tile, 'chain': chain, 'ln': ln, 'attn_one': attn_one, 'attn_maps': attn_maps, 'attn_trace': attn_trace, 'residency': residency, 'slab': slab}[which]()
