"""Backward-pass building blocks on the MI355X against fp64 torch / torch.autograd of the oracle."""
import pytest
import torch

from conftest import max_abs_diff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return torch.device('cuda:0')


def _rand(g, *shape):
    return torch.randn(*shape, generator=g)


@pytest.mark.parametrize('ta', [False, True])
@pytest.mark.parametrize('tb', [False, True])
@pytest.mark.parametrize('M,N,K', [(64, 64, 16), (300, 200, 512), (67, 130, 72), (1, 1, 4), (90, 90, 90), (2880, 512, 37),
                                   (5, 129, 1030)])
def test_matmul_nt_all_operand_layouts(dev, ta, tb, M, N, K):
    """C = A . B^T with each operand stored either k-contiguous or transposed (m-contiguous), ragged edges, K tails,
    rows that are not 16-byte aligned (scalar-load path)."""
    from lamp_amd import _native as N_
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + ta * 2 + tb)
    a = _rand(g, K, M).t() if ta else _rand(g, M, K)
    b = _rand(g, K, N).t() if tb else _rand(g, N, K)
    ref = a.double() @ b.double().t()
    out = N_.matmul_nt(a.to(dev), b.to(dev))
    assert a.to(dev).stride() == a.stride()
    assert max_abs_diff(out, ref) < 3e-5 * max(1.0, K ** 0.5), (ta, tb, M, N, K)


def test_matmul_nt_split_k_is_deterministic_and_exact(dev):
    """Weight-gradient shape: K = B*L rows deep, 512 x 512 output -> split-K with a fixed summation order."""
    from lamp_amd import _native as N_
    g = torch.Generator().manual_seed(5)
    dy, x = _rand(g, 2880, 512), _rand(g, 2880, 384)
    ref = dy.double().t() @ x.double()
    outs = [N_.matmul_nt(dy.to(dev).t(), x.to(dev).t()) for _ in range(3)]
    assert max_abs_diff(outs[0], ref) < 2e-3
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert N_.lib().lamp_gemm_workspace_bytes(512, 384, 2880, 1) > 0


def _gemm_one_launch_no_split(N_, a, b, accumulate_into=None):
    """lamp_gemm without a workspace: no K split -- the summation order lamp_gemm_grouped promises."""
    import ctypes as C
    A, ash, ars, acs, _ = N_._operand(a)
    Bm, bsh, brs, bcs, _ = N_._operand(b)
    M, K, Nn = ash[2], ash[3], bsh[2]
    out = accumulate_into if accumulate_into is not None else torch.empty(M, Nn, device=a.device)
    d = N_.GemmDesc(A.data_ptr(), Bm.data_ptr(), out.data_ptr(), M, Nn, K, 1, 1, 1 if accumulate_into is not None else 0,
                    ars, acs, 0, 0, brs, bcs, 0, 0, out.stride(0), 0, 0, None, 0, 1.0, 0)
    N_.check(N_.lib().lamp_gemm(C.byref(d), None, 0, N_.stream()), 'lamp_gemm')
    return out


@pytest.mark.parametrize('ta,tb', [(True, True), (False, False), (True, False), (False, True)])
def test_grouped_gemm_equals_the_single_launches_bit_for_bit(dev, ta, tb):
    """lamp_gemm_grouped: many products in one launch (the deferred weight gradients of lamp_amd/training.py) -- every
    result equal to lamp_gemm's unsplit launch of the same product, whatever shares the grid: mixed K depths (whole and
    partial k-tiles, scalar-load rows), ragged tile edges, a lone tile, accumulation into an existing gradient, and
    more problems than one kernel-argument table holds (32)."""
    from lamp_amd import _native as N_
    g = torch.Generator().manual_seed(11 + 2 * ta + tb)
    shapes = [(512, 512, 2880), (512, 512, 9664), (512, 384, 2880), (64, 64, 16), (70, 130, 1000), (5, 129, 1030),
              (1, 1, 4), (512, 512, 37), (90, 200, 333)] + [(128, 64, 160 + 16 * i) for i in range(30)]
    ops = []
    for M, Nn, K in shapes:
        a = (_rand(g, K, M).t() if ta else _rand(g, M, K)).to(dev)
        b = (_rand(g, K, Nn).t() if tb else _rand(g, Nn, K)).to(dev)
        ops.append((a, b))
    want = [_gemm_one_launch_no_split(N_, a, b) for a, b in ops]
    outs = [torch.full((a.size(0), b.size(0)), float('nan'), device=dev) for a, b in ops]
    N_.matmul_nt_grouped([(a, b, o, False) for (a, b), o in zip(ops, outs)])
    for (M, Nn, K), o, w, (a, b) in zip(shapes, outs, want, ops):
        assert torch.equal(o, w), (M, Nn, K)
        assert max_abs_diff(o, a.double() @ b.double().t()) < 3e-5 * max(1.0, K ** 0.5) * 4
    # the whole-k-tile, 16-byte-loadable subset alone takes the kernel without per-tile address arithmetic: same bits
    fast = [i for i, (M, Nn, K) in enumerate(shapes) if K % 16 == 0 and M % 4 == 0 and Nn % 4 == 0]
    outs2 = [torch.empty_like(outs[i]) for i in fast]
    N_.matmul_nt_grouped([(ops[i][0], ops[i][1], o, False) for i, o in zip(fast, outs2)])
    assert len(fast) >= 30 and all(torch.equal(o, want[i]) for i, o in zip(fast, outs2))
    # accumulate
    base = [_rand(g, *o.shape).to(dev) for o in outs[:9]]
    acc = [t.clone() for t in base]
    N_.matmul_nt_grouped([(a, b, o, True) for (a, b), o in zip(ops[:9], acc)])
    for (a, b), t, o in zip(ops[:9], base, acc):
        assert torch.equal(o, _gemm_one_launch_no_split(N_, a, b, accumulate_into=t.clone()))
    with pytest.raises(ValueError):
        N_.matmul_nt_grouped([(ops[0][0], ops[0][1], torch.empty(3, 3, device=dev), False)])


def test_training_composites_validate_their_arguments(dev):
    """lamp_ffn_bwd / lamp_mha_bwd / lamp_reduce_partials_grouped return status codes, never launch on bad input."""
    import ctypes as C
    from lamp_amd import _native as N_
    L = N_.lib()
    M, d, dff = 24, 32, 64
    t = lambda *s: torch.randn(*s, device=dev)   # noqa: E731
    x, h, o, dy, w1, w2, g = t(M, d), t(M, dff), t(M, d), t(M, d), t(dff, d), t(d, dff), t(d)
    with pytest.raises(N_.LampError) as e:
        N_.ffn_bwd(x, h, o, dy, w1, w2, g, 1.5, 0, True, True)
    assert e.value.status == -4                                   # LAMP_E_UNSUPPORTED: dropout probability
    with pytest.raises(N_.LampError) as e:
        N_.ffn_bwd(x, h, o, dy, w1, w2, g, 0.0, 0, True, False)   # without dropout d_o IS dx: dW2 cannot be deferred
    assert e.value.status == -4
    wts = N_.FfnWeights(w1.data_ptr(), None, w2.data_ptr(), None, g.data_ptr(), None)
    out = [t(M, d), t(M, d), t(M, dff), t(dff), t(d), t(d), t(d)]
    rc = L.lamp_ffn_bwd(x.data_ptr(), h.data_ptr(), o.data_ptr(), dy.data_ptr(), M, d, dff, C.byref(wts), 0.1, 7,
                        out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), None, None, out[3].data_ptr(),
                        out[4].data_ptr(), out[5].data_ptr(), out[6].data_ptr(), None, 0, None, 0, None, N_.stream())
    assert rc == -3                                               # LAMP_E_WORKSPACE
    desc = N_.MhaTrainDesc(2, 5, 7, 32, 2, 16, 16, 0.25, 0.0, 0.0, 1, 2)
    assert L.lamp_mha_bwd_workspace_bytes(C.byref(desc)) > 0 and L.lamp_mha_bwd_partials_bytes(C.byref(desc)) > 0
    desc_wide = N_.MhaTrainDesc(2, 5, 7, 32, 1, 256, 256, 0.25, 0.0, 0.0, 1, 2)
    q = t(2, 5, 256)
    wts = N_.MhaWeights(q.data_ptr(), q.data_ptr(), q.data_ptr(), None, g.data_ptr(), g.data_ptr(), 1, 1)
    rc = L.lamp_mha_train_fwd(C.byref(desc_wide), C.byref(wts), *([q.data_ptr()] * 3), None, *([q.data_ptr()] * 9),
                              N_.stream())
    assert rc == -4                                               # wide heads: the per-launch route
    job = N_.ReduceJob(x.data_ptr(), 0, 8, (C.c_void_p * 3)(x.data_ptr(), None, None), 4, 0)
    assert L.lamp_reduce_partials_grouped((N_.ReduceJob * 1)(job), 1, N_.stream()) == -1   # LAMP_E_DIMS
    assert L.lamp_reduce_partials_grouped(None, 0, N_.stream()) == 0


def test_matmul_nt_batched_head_views(dev):
    """Attention-backward products straight on head-split views of [B, l, h*d] buffers and (h*B, lq, lk) maps."""
    from lamp_amd import _native as N_
    g = torch.Generator().manual_seed(9)
    B, H, lq, lk, dk = 3, 4, 90, 300, 32
    q = _rand(g, B, lq, H * dk).to(dev)
    k = _rand(g, B, lk, H * dk).to(dev)
    p = torch.softmax(_rand(g, H, B, lq, lk), -1).to(dev)      # index head*B + b, as the reference's maps
    qh = q.view(B, lq, H, dk).permute(2, 0, 1, 3)               # (H, B, lq, dk) view
    kh = k.view(B, lk, H, dk).permute(2, 0, 1, 3)
    # dS-like product: S = Q K^T
    s = N_.matmul_nt(qh, kh, alpha=0.5)
    assert max_abs_diff(s, 0.5 * qh.double() @ kh.double().transpose(-1, -2)) < 1e-4
    # dK = P^T Q written into a head-split view of a [B, lk, H*dk] buffer
    dk_buf = torch.zeros(B, lk, H * dk, device=dev)
    out_view = dk_buf.view(B, lk, H, dk).permute(2, 0, 1, 3)
    N_.matmul_nt(p.transpose(-1, -2), qh.transpose(-1, -2), out=out_view)
    assert max_abs_diff(out_view, p.double().transpose(-1, -2) @ qh.double()) < 1e-4
    # dQ = P K
    dq = N_.matmul_nt(p, kh.transpose(-1, -2))
    assert max_abs_diff(dq, p.double() @ kh.double()) < 1e-4


def test_matmul_nt_relu_mask_accumulate_alpha(dev):
    from lamp_amd import _native as N_
    g = torch.Generator().manual_seed(11)
    dy, w, h, c0 = _rand(g, 130, 70), _rand(g, 70, 96), _rand(g, 130, 96), _rand(g, 130, 96)
    ref = (dy.double() @ w.double()) * 0.25 * (h.double() > 0) + c0.double()
    out = c0.clone().to(dev)
    N_.matmul_nt(dy.to(dev), w.to(dev).t(), out=out, alpha=0.25, accumulate=True, relu_mask=h.to(dev))
    assert max_abs_diff(out, ref) < 1e-4
    # and through the split-K reduce kernel
    dy, x, c0 = _rand(g, 4096, 64), _rand(g, 4096, 64), _rand(g, 64, 64)
    m = _rand(g, 64, 64)
    ref = (dy.double().t() @ x.double()) * 2.0 * (m.double() > 0) + c0.double()
    out = c0.clone().to(dev)
    N_.matmul_nt(dy.to(dev).t(), x.to(dev).t(), out=out, alpha=2.0, accumulate=True, relu_mask=m.to(dev))
    assert max_abs_diff(out, ref) < 2e-3


def test_matmul_nt_rejects_bad_operands(dev):
    from lamp_amd import _native as N_
    a = torch.randn(8, 8, device=dev)
    with pytest.raises(ValueError):
        N_.matmul_nt(a, torch.randn(8, 12, device=dev))
    with pytest.raises(RuntimeError):
        N_.matmul_nt(a.cpu(), a.cpu())
    with pytest.raises(TypeError):
        N_.matmul_nt(a.double(), a.double())


@pytest.mark.parametrize('M,d,r_rows', [(7, 64, 0), (2880, 512, 0), (2880, 512, 90), (333, 1024, 0), (5, 36, 0), (9664, 512, 0),
                                        (130, 2048, 0), (67, 4096, 0)])
def test_layernorm_residual_fwd_and_bwd_vs_autograd(dev, M, d, r_rows):
    from lamp_amd import _native as N_
    g = torch.Generator().manual_seed(M + d)
    x = _rand(g, M, d)
    res = _rand(g, r_rows or M, d)
    gamma, beta, dy = 1 + 0.3 * _rand(g, d), 0.1 * _rand(g, d), _rand(g, M, d)
    xd, rd = x.double().requires_grad_(), res.double().requires_grad_()
    gd, bd = gamma.double().requires_grad_(), beta.double().requires_grad_()
    z = xd + (rd.repeat(M // r_rows, 1) if r_rows else rd)
    y_ref = torch.nn.functional.layer_norm(z, (d,), gd, bd, 1e-5)
    y_ref.backward(dy.double())
    y = N_.layernorm_residual(x.to(dev), res.to(dev), gamma.to(dev), beta.to(dev))
    assert max_abs_diff(y, y_ref.detach()) < 2e-5
    dz, dx, dgamma, dbeta, dbias = N_.layernorm_bwd(x.to(dev), res.to(dev), gamma.to(dev), dy.to(dev), want_dbias=True)
    assert dx is dz and max_abs_diff(dz, xd.grad) < 3e-5
    assert max_abs_diff(dbias, xd.grad.sum(0)) < 1e-4 * max(1.0, (M / 100) ** 0.5)
    assert max_abs_diff(dgamma, gd.grad) < 1e-4 * max(1.0, (M / 100) ** 0.5)
    assert max_abs_diff(dbeta, bd.grad) < 1e-4 * max(1.0, (M / 100) ** 0.5)
    if r_rows:  # the broadcast residual's gradient is the sum over its repeats: colsum of the (M/r, r*d) view
        dres = N_.colsum(dz.view(M // r_rows, r_rows * d)).view(r_rows, d)
        assert max_abs_diff(dres, rd.grad) < 1e-4
    # deterministic
    dz2, _, dgamma2, _, _ = N_.layernorm_bwd(x.to(dev), res.to(dev), gamma.to(dev), dy.to(dev))
    assert torch.equal(dz, dz2) and torch.equal(dgamma, dgamma2)
    # with dropout on x inside the kernels: against autograd on the restatement that uses the library's keep mask
    p, seed = 0.3, 77
    keep = N_.dropout_keep_mask(M * d, p, seed).view(M, d)
    xd2, rd2 = x.double().requires_grad_(), res.double().requires_grad_()
    gd2, bd2 = gamma.double().requires_grad_(), beta.double().requires_grad_()
    z2 = xd2 * keep / (1 - p) + (rd2.repeat(M // r_rows, 1) if r_rows else rd2)
    y2 = torch.nn.functional.layer_norm(z2, (d,), gd2, bd2, 1e-5)
    y2.backward(dy.double())
    yk = N_.layernorm_residual(x.to(dev), res.to(dev), gamma.to(dev), beta.to(dev), dropout_p=p, seed=seed)
    assert max_abs_diff(yk, y2.detach()) < 3e-5
    dzk, dxk, dgk, dbk, dbiask = N_.layernorm_bwd(x.to(dev), res.to(dev), gamma.to(dev), dy.to(dev), dropout_p=p, seed=seed,
                                                  want_dbias=True)
    assert max_abs_diff(dxk, xd2.grad) < 5e-5
    assert max_abs_diff(dgk, gd2.grad) < 1e-4 * max(1.0, (M / 100) ** 0.5)
    assert max_abs_diff(dbiask, xd2.grad.sum(0)) < 1e-4 * max(1.0, (M / 100) ** 0.5)
    if r_rows:   # broadcast residual: its gradient is dz summed over the repeats
        assert max_abs_diff(N_.colsum(dzk.view(M // r_rows, r_rows * d)).view(r_rows, d), rd2.grad) < 1e-4
    else:
        assert max_abs_diff(dzk, rd2.grad) < 5e-5


@pytest.mark.parametrize('M,N', [(1, 5), (32, 46080), (9664, 512), (2880, 2048), (129, 257)])
def test_colsum(dev, M, N):
    from lamp_amd import _native as N_
    x = torch.randn(M, N, generator=torch.Generator().manual_seed(M + N))
    out = N_.colsum(x.to(dev))
    assert max_abs_diff(out, x.double().sum(0)) < 1e-5 * max(1.0, M ** 0.5) * 4


@pytest.mark.parametrize('p', [0.0, 0.1, 0.5])
def test_dropout_is_counter_based_and_unbiased(dev, p):
    from lamp_amd import _native as N_
    n = 1 << 20
    x = torch.randn(n, generator=torch.Generator().manual_seed(3)).to(dev)
    y = N_.dropout(x, p, seed=1234)
    keep = N_.dropout_keep_mask(n, p, 1234).to(dev)
    assert torch.equal(y, torch.where(keep, x * (1.0 / (1.0 - p)) if p else x, torch.zeros_like(x)))
    assert abs(keep.float().mean().item() - (1 - p)) < 3e-3
    # another site (seed) gives an independent mask; the same seed on a gradient gives the same mask (= backward)
    keep2 = N_.dropout_keep_mask(n, p, 1235).to(dev)
    if p:
        assert abs((keep & keep2).float().mean().item() - (1 - p) ** 2) < 3e-3
    g = torch.ones(n, device=dev)
    assert torch.equal(N_.dropout(g, p, seed=1234) != 0, keep)
    # in place
    z = x.clone()
    N_.dropout(z, p, seed=1234, out=z)
    assert torch.equal(z, y)


@pytest.mark.parametrize('rows,lk', [(3, 5), (128 * 90, 300), (77, 90), (64, 4096)])
def test_softmax_bwd_vs_autograd(dev, rows, lk):
    from lamp_amd import _native as N_
    g = torch.Generator().manual_seed(rows + lk)
    s = _rand(g, rows, lk).double()
    blocked = torch.rand(rows, lk, generator=g) < 0.3
    blocked[:, 0] = False
    s = s.masked_fill(blocked, float('-inf')).requires_grad_()
    p = torch.softmax(s * 0.25, -1)
    dP = _rand(g, rows, lk)
    p.backward(dP.double())
    dS = N_.softmax_bwd(p.detach().float().to(dev), dP.to(dev), 0.25)
    assert max_abs_diff(dS, s.grad) < 1e-5
    assert (dS.cpu()[blocked] == 0).all()


def test_diag_logits_bwd_and_embed_bwd_vs_autograd(dev):
    from lamp_amd import _native as N_
    g = torch.Generator().manual_seed(8)
    B, L, d, V = 5, 37, 64, 50
    y, w, dl = _rand(g, B, L, d), _rand(g, L, d), _rand(g, B, L)
    yd, wd = y.double().requires_grad_(), w.double().requires_grad_()
    (yd * wd.unsqueeze(0)).sum(-1).backward(dl.double())
    dy, dw = N_.diag_logits_bwd(y.to(dev), w.to(dev), dl.to(dev))
    assert max_abs_diff(dy, yd.grad) < 1e-5 and max_abs_diff(dw, wd.grad) < 1e-5
    seq = torch.randint(0, V, (B, 23), generator=g)
    seq[:, -4:] = 0
    emb = torch.nn.Embedding(V, d, padding_idx=0).double()
    dout = _rand(g, B, 23, d)
    emb(seq).backward(dout.double())
    d_emb = N_.embed_bwd(seq.to(dev), dout.to(dev), V, pad_idx=0)
    assert max_abs_diff(d_emb, emb.weight.grad) < 1e-4
    assert (d_emb[0] == 0).all()


@pytest.mark.parametrize('B,H,lq,lk,dk,kind', [(3, 4, 90, 300, 128, 'keys'), (2, 4, 90, 90, 128, 'shared'), (2, 2, 37, 70, 32, 'none'),
                                               (1, 8, 200, 983, 64, 'shared'), (2, 1, 33, 5, 16, 'keys')])
def test_fast_maps_attention_equals_exact_two_pass(dev, B, H, lq, lk, dk, kind):
    """lamp_sdpa_fwd_fast_maps (single pass: scores + row log-sum-exp, normalised in place) against lamp_sdpa_fwd's exact
    two-pass maps and output, for every mask kind, with a key split (reuters' shape) and without, incl. a fully
    blocked row (NaN in both)."""
    from lamp_amd import _native as N_
    g = torch.Generator().manual_seed(B * 100 + lq)
    q, k, v = (_rand(g, B, l, H * dk).to(dev) for l in (lq, lk, lk))
    mask, keep = None, None
    if kind == 'keys':
        seq = torch.randint(1, 50, (B, lk), generator=g)
        seq[0, lk // 2:] = 0
        if B > 1:
            seq[1, :] = 0          # a sample whose keys are ALL padding: NaN rows
        mask, keep = N_.key_token_mask(seq.to(dev), lk)
    elif kind == 'shared':
        blocked = torch.rand(lq, lk, generator=g) < 0.6
        blocked[:, 0] = False
        mask, keep = N_.make_mask(blocked.to(dev), B, lq, lk)
    o1, p1 = N_.sdpa_fused(q, k, v, H, mask, 1.0 / dk ** 0.5, need_attn=True)
    o2, p2 = N_.sdpa_fused(q, k, v, H, mask, 1.0 / dk ** 0.5, need_attn=True, fast_maps=True)
    assert torch.equal(torch.isnan(p1), torch.isnan(p2)) and torch.equal(torch.isnan(o1), torch.isnan(o2))
    assert max_abs_diff(torch.nan_to_num(p2), torch.nan_to_num(p1)) < 2e-6
    assert max_abs_diff(torch.nan_to_num(o2), torch.nan_to_num(o1)) < 2e-5
    ok = ~torch.isnan(p2.sum(-1))
    assert max_abs_diff(p2.sum(-1)[ok], torch.ones_like(p2.sum(-1)[ok])) < 1e-5
