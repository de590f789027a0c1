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
