"""Host logic of lamp_amd/training.py's deferred weight gradients (train.py:40 `loss.backward()`): what is queued, when it is
flushed and where the results land -- on CPU, with the two grouped launches replaced by torch stand-ins (the kernels themselves
are tested on the device in test_gpu_backward.py / test_gpu_training.py)."""
import pytest
import torch

from lamp_amd import training


class _Lin(torch.autograd.Function):
    """y = x W^T whose weight gradient goes through the queue, as _FFNFn / _MHAFn do."""

    @staticmethod
    def forward(ctx, x, w, param):
        ctx.save_for_backward(x, w)
        ctx.param = param
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        training._weight_grads.add(ctx.param, dy, x)
        return dy @ w, None, None


@pytest.fixture
def grouped(monkeypatch):
    launches = []

    def fake(problems):
        launches.append(len(problems))
        outs = set()
        for dy2, x2, out, accumulate in problems:
            assert id(out) not in outs and out.data_ptr() not in outs, 'two products of one launch write one output'
            outs.add(out.data_ptr())
            r = dy2.t() @ x2
            out.copy_(out + r if accumulate else r)

    monkeypatch.setattr(training.N, 'wgrad_grouped', fake)
    training._weight_grads.tasks.clear()
    yield launches
    training._weight_grads.tasks.clear()


def test_queue_is_flushed_once_per_backward_and_fills_grad(grouped):
    g = torch.Generator().manual_seed(0)
    w1 = torch.nn.Parameter(torch.randn(5, 4, generator=g))
    w2 = torch.nn.Parameter(torch.randn(3, 5, 1, generator=g))   # Conv1d(k=1) layout
    x = torch.randn(7, 4, generator=g, requires_grad=True)

    def loss():
        h = _Lin.apply(x, w1, w1)
        return _Lin.apply(h, w2.view(3, 5), w2).square().sum()

    loss().backward()
    assert grouped == [2] and training._weight_grads.pending() == 0
    ref1, ref2, refx = torch.autograd.grad((x @ w1.t() @ w2.view(3, 5).t()).square().sum(), (w1, w2, x))
    assert torch.allclose(w1.grad, ref1, atol=1e-5) and w1.grad.shape == w1.shape
    assert torch.allclose(w2.grad, ref2, atol=1e-5) and w2.grad.shape == w2.shape
    assert torch.allclose(x.grad, refx, atol=1e-5)
    # a second pass without zero_grad accumulates in place, like autograd's AccumulateGrad
    loss().backward()
    assert grouped == [2, 2]
    assert torch.allclose(w1.grad, 2 * ref1, atol=1e-5) and torch.allclose(w2.grad, 2 * ref2, atol=1e-5)


def test_a_parameter_used_twice_accumulates_over_two_launches(grouped):
    g = torch.Generator().manual_seed(1)
    w = torch.nn.Parameter(torch.randn(4, 4, generator=g))
    x = torch.randn(6, 4, generator=g)
    _Lin.apply(_Lin.apply(x.requires_grad_(), w, w), w, w).sum().backward()
    assert grouped == [1, 1]   # never two writers of one output in a launch
    ref, = torch.autograd.grad((x.detach() @ w.t() @ w.t()).sum(), w)
    assert torch.allclose(w.grad, ref, atol=1e-5)


def test_non_contiguous_existing_grad_takes_the_detour(grouped):
    g = torch.Generator().manual_seed(2)
    w = torch.nn.Parameter(torch.randn(4, 6, generator=g))
    w.grad = torch.ones(6, 4).t()   # a gradient somebody assigned with another memory layout
    x = torch.randn(5, 6, generator=g, requires_grad=True)
    _Lin.apply(x, w, w).sum().backward()
    ref, = torch.autograd.grad((x.detach() @ w.t()).sum(), w)
    assert torch.allclose(w.grad, 1 + ref, atol=1e-5)


def test_reentrant_backward_has_its_own_queue(grouped):
    """torch.utils.checkpoint (re-entrant) runs a backward pass inside a backward pass: the inner pass flushes only what it
    queued, the outer pass's entries wait for the outer pass's end."""
    from torch.utils.checkpoint import checkpoint
    g = torch.Generator().manual_seed(3)
    w1 = torch.nn.Parameter(torch.randn(4, 4, generator=g))
    w2 = torch.nn.Parameter(torch.randn(4, 4, generator=g))
    x = torch.randn(3, 4, generator=g, requires_grad=True)
    inner = lambda t: _Lin.apply(t, w2, w2)   # noqa: E731
    y = checkpoint(inner, _Lin.apply(x, w1, w1), use_reentrant=True)
    _Lin.apply(y, w1, w1).sum().backward()
    assert sorted(grouped) == [1, 1, 1] and training._weight_grads.pending() == 0   # w1 twice (outer, two launches), w2 inner
    r1, r2 = torch.autograd.grad((x.detach() @ w1.t() @ w2.t() @ w1.t()).sum(), (w1, w2))
    assert torch.allclose(w1.grad, r1, atol=1e-5) and torch.allclose(w2.grad, r2, atol=1e-5)


def test_only_leaf_parameters_are_deferred():
    w = torch.nn.Parameter(torch.randn(3, 3))
    assert training._deferrable(w) is None          # not on the device
    assert training._deferrable(w * 2, None) is None    # a non-leaf (what nn.DataParallel replicas hold)
    old = training.DEFER_WEIGHT_GRADS
    try:
        training.DEFER_WEIGHT_GRADS = False
        assert training._deferrable() is None
    finally:
        training.DEFER_WEIGHT_GRADS = old
