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


class _LinSafe(torch.autograd.Function):
    """y = x W^T deferring only when the pass accumulates into .grad -- the decision _FFNFn / _MHAFn take (_Deferral.live)."""

    @staticmethod
    def forward(ctx, x, w, defer):
        ctx.save_for_backward(x, w)
        ctx.defer = defer
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        defer = ctx.defer.live() if ctx.defer is not None else None
        if defer is not None:
            training._weight_grads.add(defer[0], dy, x)
            return dy @ w, None, None
        return dy @ w, dy.t() @ x, None


def _deferral_on_cpu(*params):
    """_deferrable without its `is_cuda` requirement (there is no device here): hooks and switches still decide."""
    real = torch.Tensor.is_cuda
    try:
        torch.Tensor.is_cuda = property(lambda self: True)
        return training._deferrable(*params)
    finally:
        torch.Tensor.is_cuda = real


def test_deferral_never_hides_a_gradient_from_autograd(grouped):
    """torch.autograd.grad and backward(inputs=...) get their gradients through autograd (a deferred gradient is
    returned as None and written into .grad afterwards); tensors with gradient hooks are never deferred."""
    g = torch.Generator().manual_seed(4)
    w = torch.nn.Parameter(torch.randn(4, 3, generator=g))
    x = torch.randn(5, 3, generator=g, requires_grad=True)
    d = _deferral_on_cpu(w)
    assert d is not None and d[0] is w
    ref_w, ref_x = torch.autograd.grad((x @ w.t()).square().sum(), (w, x))
    # loss.backward(): deferred, .grad filled by the end-of-backward flush
    _LinSafe.apply(x, w, d).square().sum().backward()
    assert grouped == [1] and torch.allclose(w.grad, ref_w, atol=1e-5)
    # torch.autograd.grad: the gradient comes back as a tensor and .grad is not touched
    w.grad = None
    gw, = torch.autograd.grad(_LinSafe.apply(x, w, d).square().sum(), [w])
    assert grouped == [1] and torch.allclose(gw, ref_w, atol=1e-5) and w.grad is None
    # backward(inputs=[x]): nothing lands in w.grad
    x.grad = None
    _LinSafe.apply(x, w, d).square().sum().backward(inputs=[x])
    assert grouped == [1] and w.grad is None and torch.allclose(x.grad, ref_x, atol=1e-5)
    # backward(inputs=[w]) accumulates into w.grad: deferred again
    _LinSafe.apply(x, w, d).square().sum().backward(inputs=[w])
    assert grouped == [1, 1] and torch.allclose(w.grad, ref_w, atol=1e-5)
    # hooks see real gradients: such a parameter is not deferred at all
    seen = []
    h = w.register_hook(lambda gr: seen.append(gr.clone()))
    assert _deferral_on_cpu(w) is None
    w.grad = None
    _LinSafe.apply(x, w, _deferral_on_cpu(w)).square().sum().backward()
    assert grouped == [1, 1] and len(seen) == 1 and torch.allclose(seen[0], ref_w, atol=1e-5)
    h.remove()
    h2 = w.register_post_accumulate_grad_hook(lambda p: seen.append(p.grad.clone()))
    assert _deferral_on_cpu(w) is None
    h2.remove()
    assert _deferral_on_cpu(w) is not None
    with training.weight_grad_deferral(False):
        assert _deferral_on_cpu(w) is None
    assert _deferral_on_cpu(w) is not None


def test_eviction_of_a_live_queue_is_loud(grouped):
    w = torch.nn.Parameter(torch.randn(2, 2))
    wg = training._weight_grads
    for t in range(wg.MAX_TASKS + 1):
        wg.tasks[1000 + t] = ([(w, torch.zeros(1, 2), torch.zeros(1, 2))], [])
    x = torch.randn(3, 2, requires_grad=True)
    with pytest.warns(RuntimeWarning, match='dropped'):
        _Lin.apply(x, w, w).sum().backward()


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
