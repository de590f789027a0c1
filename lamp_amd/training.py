"""Training-mode forward and backward of the label-graph path (reference train.py:36-48).

The eval path is one ``lamp_forward`` call; training needs the intermediate activations, so here the forward is
composed per sub-layer -- one ``torch.autograd.Function`` per MultiHeadAttention / PositionwiseFeedForward /
embedding / read-out -- and ``loss.backward()`` + ``optimizer.step()`` of the reference's train loop work unchanged.
PyTorch supplies the autograd tape, memory and streams; every arithmetic op of forward AND backward is a HIP kernel
behind the C ABI (``lamp_linear_fwd``, ``lamp_sdpa_fwd``, ``lamp_gemm``, ``lamp_layernorm_bwd``, ...).

Dropout (lamp/SubLayers.py:40,113,138) is counter-based (``lamp_dropout``): a site's mask is a pure function of
(element index, seed), recomputed in backward instead of stored.  Seeds are drawn from torch's global CPU generator
once per forward, so ``torch.manual_seed`` makes a run reproducible; the random stream itself necessarily differs
from torch's Philox stream (the reference is not reproducible across devices either).

Attention backward uses the probability maps the forward kernel writes (its exact two-pass variant) and five
``lamp_gemm`` products on head-split VIEWS of the fused [B, l, h*d] projections -- no head split/merge copies.

Weight gradients are DEFERRED (``DEFER_WEIGHT_GRADS``): dW = dY^T.X of a projection is not needed by anything else in the
backward pass, and alone it is 64 output tiles at d_model = 512 -- a K split over partial buffers plus a reduce launch
each.  The sub-layers' backward functions therefore only queue (parameter, dY, X); when the engine has run the whole
graph, one ``lamp_gemm_grouped`` launch computes every queued product (28 of them for the 2+2-layer model) and the results
are stored in (or accumulated into) the parameters' ``.grad`` -- what ``optimizer.step()`` reads next (train.py:40-48).
A deferred gradient BYPASSES autograd (the sub-layer's backward returns None for it), so it is taken only where nothing can
observe the difference -- decided twice:
  * in the forward (``_deferrable``): every parameter is a leaf CUDA tensor that requires grad and carries no tensor hook
    (``register_hook`` / ``register_post_accumulate_grad_hook``), and the forward does not run inside a
    ``DistributedDataParallel`` wrapper (whose reducer hooks sit on the AccumulateGrad nodes, invisible from Python);
  * in the backward (``_Deferral.live``): the engine is going to EXECUTE each parameter's AccumulateGrad node in this pass,
    i.e. this is ``loss.backward()`` accumulating into ``.grad``.  Under ``torch.autograd.grad(loss, [w])`` and
    ``loss.backward(inputs=[x])`` it is not, and the gradient goes through autograd as a tensor -- as it does for
    ``nn.DataParallel`` replicas (non-leaf weights) and for everything when ``DEFER_WEIGHT_GRADS`` is False or inside
    ``with weight_grad_deferral(False):`` (the switch for hook-based reducers / clippers this module cannot see).
"""
import contextlib
import threading
import warnings

import torch

from . import Constants
from . import _native as N

DEFER_WEIGHT_GRADS = True
# One library call per sub-layer and direction (lamp_ffn_train_fwd / lamp_ffn_bwd / lamp_mha_train_fwd / lamp_mha_bwd) instead
# of one Python round trip per launch: the reuters step is ~190 launches and was bound by the issuing thread
# (tools/bench_train.py: host_issue_ms_per_step).  Same kernels, same bits; False = the per-launch route below.
COMPOSITE_CALLS = True


def _plain(*tensors):
    """Composite calls take contiguous fp32 device tensors as they are."""
    for t in tensors:
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            return False
    return True


def _on(dev):
    return torch.cuda.device(dev) if dev.type == 'cuda' else contextlib.nullcontext()


class _WeightGrads(object):
    """The queues of deferred weight gradients, one per running backward pass (autograd graph task: a re-entrant backward
    inside a checkpointed region is its own task); each is flushed by its task's end-of-backward callback."""

    # a pass that raised never runs its callback: its queue (dY / X activations) is dropped once this many newer passes
    # exist.  Live passes beyond that (deeply nested re-entrant checkpointing, many concurrent backward threads) would lose
    # gradients: the eviction of a non-empty queue is therefore loud.
    MAX_TASKS = 16

    def __init__(self):
        self.tasks = {}   # graph-task id -> (items, jobs)
        self.lock = threading.Lock()

    def _queue(self):
        """Called under the lock from inside a backward pass: this pass's queue, its flush registered on first use."""
        task = torch._C._current_graph_task_id()
        q = self.tasks.get(task)
        if q is None:
            q = self.tasks[task] = ([], [])
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)
            while len(self.tasks) > self.MAX_TASKS:
                oldest = min(self.tasks)
                items, jobs = self.tasks.pop(oldest)
                if items or jobs:
                    warnings.warn('lamp_amd.training: dropped %d deferred weight gradients of backward pass %d (more than %d '
                                  'passes in flight, or passes that raised); set training.DEFER_WEIGHT_GRADS = False if that '
                                  'pass was alive' % (len(items) + len(jobs), oldest, self.MAX_TASKS), RuntimeWarning)
        return q

    def pending(self):
        return sum(len(i) + len(j) for i, j in self.tasks.values())

    def add_reductions(self, pending, grads):
        """pending = (jobs, buffers) from N.ffn_bwd / N.mha_bwd(defer_reduce=True); grads = [(parameter, tensor the jobs fill)]."""
        with self.lock:
            self._queue()[1].append((pending, grads))

    def add(self, param, dy2, x2):
        """param.grad (+)= dy2^T x2  (dy2 (rows, out), x2 (rows, in); param is (out, in) or Conv1d's (out, in, 1))."""
        with self.lock:
            self._queue()[0].append((param, dy2, x2))

    def flush(self):
        with self.lock:
            items, jobs = self.tasks.pop(torch._C._current_graph_task_id(), ([], []))
        jobs_by_dev = {}
        for pending, grads in jobs:
            jobs_by_dev.setdefault(grads[0][1].device, []).append((pending, grads))
        for dev, todo in jobs_by_dev.items():   # the second stages of every LayerNorm-parameter / bias gradient: one launch
            with _on(dev), torch.no_grad():
                N.reduce_partials_grouped([j for pending, _ in todo for j in pending[0]])
                for _, grads in todo:
                    for param, t in grads:
                        if t.shape != param.shape:
                            t = t.view(param.shape)
                        if param.grad is None:
                            param.grad = t
                        else:
                            param.grad.add_(t)
        by_dev = {}
        for it in items:
            by_dev.setdefault(it[1].device, []).append(it)
        for dev, todo in by_dev.items():
            with _on(dev), torch.no_grad():
                while todo:
                    seen, batch, rest = set(), [], []
                    for it in todo:   # a parameter used twice: its second product accumulates in a later launch
                        (rest if id(it[0]) in seen else batch).append(it)
                        seen.add(id(it[0]))
                    problems, fresh, detour = [], [], []
                    for param, dy2, x2 in batch:
                        shape2 = (dy2.size(1), x2.size(1))
                        g = param.grad
                        if g is None:
                            out = torch.empty(shape2, dtype=torch.float32, device=dev)
                            fresh.append((param, out))
                            problems.append((dy2, x2, out, False))
                        elif g.is_contiguous() and g.dtype == torch.float32:
                            problems.append((dy2, x2, g.view(shape2), True))
                        else:
                            out = torch.empty(shape2, dtype=torch.float32, device=dev)
                            detour.append((g, out))
                            problems.append((dy2, x2, out, False))
                    N.wgrad_grouped(problems)
                    for param, out in fresh:
                        param.grad = out.view(param.shape)
                    for g, out in detour:
                        g.add_(out.view(g.shape))
                    todo = rest


_weight_grads = _WeightGrads()


@contextlib.contextmanager
def weight_grad_deferral(enabled):
    """``with weight_grad_deferral(False): loss = model(...); loss.backward()`` -- every weight gradient of forwards run
    inside goes through autograd (for gradient hooks this module cannot see: DDP-style reducers on AccumulateGrad nodes)."""
    global DEFER_WEIGHT_GRADS
    before = DEFER_WEIGHT_GRADS
    DEFER_WEIGHT_GRADS = bool(enabled)
    try:
        yield
    finally:
        DEFER_WEIGHT_GRADS = before


def _inside_ddp_forward():
    ddp = getattr(torch.nn.parallel, 'DistributedDataParallel', None)
    return getattr(ddp, '_active_ddp_module', None) is not None


class _Deferral(tuple):
    """The leaf Parameters of one sub-layer whose gradients MAY bypass autograd, with their AccumulateGrad nodes."""

    def __new__(cls, params):
        self = super().__new__(cls, params)
        with torch.enable_grad():   # the accumulator node of a leaf: what loss.backward() runs to fill .grad
            self.nodes = [p.view_as(p).grad_fn.next_functions[0][0] for p in params if p is not None]
        return self

    def live(self):
        """Called inside the backward pass: self if this pass accumulates into every parameter's .grad, else None."""
        try:
            for n in self.nodes:
                if not torch._C._will_engine_execute_node(n):
                    return None    # backward(inputs=[...]) that leaves this parameter out
        except RuntimeError:
            return None            # torch.autograd.grad(): gradients are captured, not accumulated
        return self


def _deferrable(*params):
    """A _Deferral of the parameters if their gradients may bypass autograd (see the module docstring), else None."""
    if not DEFER_WEIGHT_GRADS or _inside_ddp_forward():
        return None
    for p in params:
        if p is None:
            continue
        if not (p.is_leaf and p.requires_grad and p.is_cuda):
            return None
        if p._backward_hooks or getattr(p, '_post_accumulate_grad_hooks', None):
            return None   # a hook would fire with grad=None before the real gradient exists
    return _Deferral(params)


class _Seeds(object):
    """Per-forward dropout seeds: base from torch's CPU generator, one odd-multiple step per site."""

    def __init__(self):
        self.base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        self.count = 0

    def next(self):
        self.count += 1
        return (self.base + 0x9E3779B1 * self.count) & 0xffffffff


def _w2d(conv_w):
    return conv_w.view(conv_w.size(0), conv_w.size(1))


class _EmbedFn(torch.autograd.Function):
    """lamp/Encoders.py:66,75: token (+ frozen sinusoid position) embedding."""

    @staticmethod
    def forward(ctx, src_seq, src_pos, emb_w, pos_w):
        ctx.save_for_backward(src_seq)
        ctx.n_vocab = emb_w.size(0)
        return N.embed(src_seq, src_pos, emb_w, pos_w)

    @staticmethod
    def backward(ctx, dout):
        (src_seq,) = ctx.saved_tensors
        d_emb = N.embed_bwd(src_seq, dout, ctx.n_vocab, pad_idx=Constants.PAD) if ctx.needs_input_grad[2] else None
        return None, None, d_emb, None  # the position table is frozen (lamp/Models.py:97-107)


class _LabelRowsFn(torch.autograd.Function):
    """lamp/Decoders.py:132-134: every sample's decoder input is the whole label table."""

    @staticmethod
    def forward(ctx, table, B):
        ctx.B = B
        return table.unsqueeze(0).expand(B, -1, -1).contiguous()

    @staticmethod
    def backward(ctx, dy):
        L, d = dy.size(1), dy.size(2)
        return N.colsum(dy.reshape(ctx.B, L * d)).view(L, d), None


class _FFNFn(torch.autograd.Function):
    """LayerNorm(dropout(W2 relu(W1 x + b1) + b2) + x)   (lamp/SubLayers.py:133-142)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, ln_g, ln_b, p, seed, defer=None):
        ctx.defer = defer   # (w1, w2) as the leaf Parameters, or None
        x2 = x.reshape(-1, x.size(-1))
        ctx.composite = COMPOSITE_CALLS and _plain(x2, w1, b1, w2, b2, ln_g, ln_b)
        if ctx.composite:
            N.require_device(x2)
            h, o, y = N.ffn_train_fwd(x2, _w2d(w1), b1, _w2d(w2), b2, ln_g, ln_b, p, seed)
        else:
            h = N.linear(x2, _w2d(w1), b1, relu=True)
            o = N.linear(h, _w2d(w2), b2)
            y = N.layernorm_residual(o, x2, ln_g, ln_b, dropout_p=p, seed=seed)  # dropout + add & norm in one kernel
        ctx.save_for_backward(x2, h, o, w1, w2, ln_g)
        ctx.p, ctx.seed, ctx.shape = p, seed, x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, h, o, w1, w2, ln_g = ctx.saved_tensors
        W1, W2 = _w2d(w1), _w2d(w2)
        defer = ctx.defer.live() if ctx.defer is not None else None
        if ctx.composite:
            wait1, wait2 = defer is not None, defer is not None and ctx.p > 0
            dx, d_o, dh, dW1, dW2, db1, db2, dg, db, pending = N.ffn_bwd(
                x2, h, o, dy.reshape(x2.shape).contiguous(), W1, W2, ln_g, ctx.p, ctx.seed, not wait1, not wait2,
                defer_reduce=defer is not None)
            if wait1:
                _weight_grads.add(defer[0], dh, x2)
            if wait2:
                _weight_grads.add(defer[1], d_o, h)
            if pending is not None:
                _weight_grads.add_reductions(pending, [(defer[2], db1), (defer[3], db2), (defer[4], dg), (defer[5], db)])
                db1 = db2 = dg = db = None
            return (dx.view(ctx.shape), None if wait1 else dW1.view_as(w1), db1, None if wait2 else dW2.view_as(w2), db2,
                    dg, db, None, None, None)
        dz, do, dg, db, db2 = N.layernorm_bwd(o, x2, ln_g, dy.reshape(x2.shape), dropout_p=ctx.p, seed=ctx.seed,
                                              want_dbias=True)
        # do IS dz when there is no dropout, and dz is accumulated into below: that product cannot wait
        if defer is not None and do is not dz:
            _weight_grads.add(defer[1], do, h)
            dW2 = None
        else:
            dW2 = N.matmul_nt(do.t(), h.t()).view_as(w2)
        dh = N.matmul_nt(do, W2.t(), relu_mask=h)
        db1 = N.colsum(dh)
        if defer is not None:
            _weight_grads.add(defer[0], dh, x2)
            dW1 = None
        else:
            dW1 = N.matmul_nt(dh.t(), x2.t()).view_as(w1)
        dx = N.matmul_nt(dh, W1.t(), out=dz, accumulate=True)  # + the residual branch (do is dead by now, or queued)
        return dx.view(ctx.shape), dW1, db1, dW2, db2, dg, db, None, None, None


class _MHAFn(torch.autograd.Function):
    """LayerNorm(dropout(concat_heads(dropout(softmax(mask(q k^T / t))) v) Wfc^T) + xq)   (lamp/SubLayers.py:77-121)."""

    @staticmethod
    def forward(ctx, xq, xkv, wq, wk, wv, fc, ln_g, ln_b, n_head, mask, keep, p_attn, p_out, seed_attn, seed_out,
                xv=None, defer=None):
        """xkv: the key source and, unless ``xv`` is given, the value source too (every layer of the reference passes one
        tensor for both, lamp/Layers.py:16,35,40; the module itself accepts two, lamp/SubLayers.py:77-93)."""
        ctx.defer = defer   # (wq, wk, wv, fc, ln_g, ln_b) as the leaf Parameters, or None
        # self-attention: one tensor is query, key and value source -- its three gradient branches are summed in place by the
        # backward call instead of by an autograd add launch
        ctx.shared_qk = xq is xkv and xv is None
        B, lq, d = xq.shape
        lk = xkv.size(1)
        H = n_head
        dk, dv = wq.size(0) // H, wv.size(0) // H
        inv_t = 1.0 / float(dk) ** 0.5
        ctx.composite = (COMPOSITE_CALLS and dk <= 128 and dv <= 128 and
                         _plain(xq, xkv, xv, wq, wk, wv, fc, ln_g, ln_b) and (fc is not None or H * dv == d))
        if ctx.composite:
            N.require_device(xq, xkv)
            desc = N.MhaTrainDesc(B, lq, lk, d, H, dk, dv, inv_t, p_attn, p_out, seed_attn & 0xffffffff,
                                  seed_out & 0xffffffff)
            q, k, v, a, P, Pd, o, y = N.mha_train_fwd(desc, xq, xkv, xkv if xv is None else xv, wq, wk, wv, fc, ln_g, ln_b,
                                                     mask)
            attn = Pd if p_attn > 0 else P.clone()
            # the dropped map is what the value product used: kept for the backward instead of recomputed -- and it is the map
            # the caller gets, so it is SAVED (version-checked: an in-place edit by the caller raises in backward instead of
            # corrupting the gradients silently)
            ctx.save_for_backward(xq, xkv, wq, wk, wv, fc if fc is not None else wq.new_empty(0), ln_g, q, k, v, a, P,
                                  o if o is not None else a, xv if xv is not None else wq.new_empty(0),
                                  Pd if p_attn > 0 else wq.new_empty(0))
            ctx.cfg = (B, lq, lk, H, dk, dv, inv_t, p_attn, p_out, seed_attn, seed_out, fc is not None, xv is not None)
            ctx.mark_non_differentiable(attn)
            return y, attn
        q = N.linear(xq, wq)
        k = N.linear(xkv, wk)
        v = N.linear(xkv if xv is None else xv, wv)
        a, P = N.sdpa_fused(q, k, v, H, mask, inv_t, need_attn=True, fast_maps=True)
        Pd = P
        if p_attn > 0:  # the reference drops probabilities AFTER the softmax; the value product uses the dropped map
            Pd = N.dropout(P, p_attn, seed_attn)
            N.matmul_nt(Pd.view(H, B, lq, lk), v.view(B, lk, H, dv).permute(2, 0, 3, 1),
                        out=a.view(B, lq, H, dv).permute(2, 0, 1, 3))
        o = N.linear(a, fc) if fc is not None else a
        y = N.layernorm_residual(o, xq, ln_g, ln_b, dropout_p=p_out, seed=seed_out)
        ctx.save_for_backward(xq, xkv, wq, wk, wv, fc if fc is not None else wq.new_empty(0), ln_g, q, k, v, a, P, o,
                              xv if xv is not None else wq.new_empty(0), wq.new_empty(0))
        ctx.cfg = (B, lq, lk, H, dk, dv, inv_t, p_attn, p_out, seed_attn, seed_out, fc is not None, xv is not None)
        attn = Pd if p_attn > 0 else P.clone()  # what the reference returns (lamp/SubLayers.py:40-43): the dropped map
        ctx.mark_non_differentiable(attn)
        return y, attn

    @staticmethod
    def backward(ctx, dy, _dP_unused):
        xq, xkv, wq, wk, wv, fc, ln_g, q, k, v, a, P, o, xv, Pd_saved = ctx.saved_tensors
        B, lq, lk, H, dk, dv, inv_t, p_attn, p_out, seed_attn, seed_out, has_fc, has_xv = ctx.cfg
        d = xq.size(-1)
        defer = ctx.defer.live() if ctx.defer is not None else None
        if ctx.composite:
            wait, wait_fc = defer is not None, defer is not None and p_out > 0
            desc = N.MhaTrainDesc(B, lq, lk, d, H, dk, dv, inv_t, p_attn, p_out, seed_attn & 0xffffffff,
                                  seed_out & 0xffffffff)
            r = N.mha_bwd(desc, xq, xkv, xv if has_xv else xkv, q, k, v, a, P, Pd_saved if p_attn > 0 else None, o if has_fc else None,
                          dy.reshape(B * lq, d).contiguous(), wq, wk, wv, fc if has_fc else None, ln_g, has_xv, not wait,
                          not wait_fc, defer_reduce=wait, shared_qk=ctx.shared_qk)
            if r['pending'] is not None:
                _weight_grads.add_reductions(r['pending'], [(defer[4], r['dgamma']), (defer[5], r['dbeta'])])
                r['dgamma'] = r['dbeta'] = None
            if wait:
                xq2, xkv2 = xq.view(-1, d), xkv.view(-1, d)
                _weight_grads.add(defer[0], r['dq'], xq2)
                _weight_grads.add(defer[1], r['dk'], xkv2)
                _weight_grads.add(defer[2], r['dv'], xv.view(-1, d) if has_xv else xkv2)
            if wait_fc and has_fc:
                _weight_grads.add(defer[3], r['d_o'], a.view(-1, H * dv))
            return (r['dxq'].view(xq.shape), None if ctx.shared_qk else r['dxk'].view(xkv.shape), r['dwq'], r['dwk'], r['dwv'], r['dfc'], r['dgamma'],
                    r['dbeta']) + (None,) * 7 + (r['dxv'].view(xv.shape) if has_xv else None, None)
        xq2, xkv2 = xq.reshape(-1, d), xkv.reshape(-1, d)
        xv2 = xv.reshape(-1, d) if has_xv else xkv2
        dz, do, dg, db, _ = N.layernorm_bwd(o.view(xq2.shape), xq2, ln_g, dy.reshape(xq2.shape), dropout_p=p_out,
                                            seed=seed_out)
        a2 = a.view(-1, H * dv)
        if has_fc:
            if defer is not None and do is not dz:   # do IS dz without dropout, and dz is accumulated into below
                _weight_grads.add(defer[3], do, a2)
                dfc = None
            else:
                dfc = N.matmul_nt(do.t(), a2.t())
            da = N.matmul_nt(do, fc.t())
        else:
            dfc, da = None, do
        heads = lambda t, l, w: t.view(B, l, H, w).permute(2, 0, 1, 3)  # noqa: E731  (H, B, l, w) view
        dah, qh, kh, vh = heads(da, lq, dv), heads(q, lq, dk), heads(k, lk, dk), heads(v, lk, dv)
        P4 = P.view(H, B, lq, lk)
        Pd = N.dropout(P, p_attn, seed_attn).view(H, B, lq, lk) if p_attn > 0 else P4
        dv_buf = torch.empty_like(v)
        N.matmul_nt(Pd.transpose(-1, -2), dah.transpose(-1, -2), out=heads(dv_buf, lk, dv))      # dV = Pd^T dA
        dP = N.matmul_nt(dah, vh)                                                                 # dPd = dA V^T
        if p_attn > 0:
            N.dropout(dP, p_attn, seed_attn, out=dP)
        N.softmax_bwd(P4, dP, inv_t, out=dP)                                                      # dS (in place)
        dq_buf, dk_buf = torch.empty_like(q), torch.empty_like(k)
        N.matmul_nt(dP, kh.transpose(-1, -2), out=heads(dq_buf, lq, dk))                          # dQ = dS K
        N.matmul_nt(dP.transpose(-1, -2), qh.transpose(-1, -2), out=heads(dk_buf, lk, dk))        # dK = dS^T Q
        dq2, dk2, dv2 = dq_buf.view(-1, H * dk), dk_buf.view(-1, H * dk), dv_buf.view(-1, H * dv)
        if defer is not None:
            _weight_grads.add(defer[0], dq2, xq2)
            _weight_grads.add(defer[1], dk2, xkv2)
            _weight_grads.add(defer[2], dv2, xv2)
            dwq = dwk = dwv = None
        else:
            dwq = N.matmul_nt(dq2.t(), xq2.t())
            dwk = N.matmul_nt(dk2.t(), xkv2.t())
            dwv = N.matmul_nt(dv2.t(), xv2.t())
        dxq = N.matmul_nt(dq2, wq.t(), out=dz, accumulate=True)  # + the residual branch
        if ctx.shared_qk:   # one tensor behind query, key and value: the three branches summed in place (as lamp_mha_bwd does)
            N.matmul_nt(dk2, wk.t(), out=dxq, accumulate=True)
            N.matmul_nt(dv2, wv.t(), out=dxq, accumulate=True)
            return (dxq.view(xq.shape), None, dwq, dwk, dwv, dfc, dg, db) + (None,) * 7 + (None, None)
        dxkv = N.matmul_nt(dk2, wk.t())
        dxv = None
        if has_xv:
            dxv = N.matmul_nt(dv2, wv.t()).view(xv.shape)
        else:
            N.matmul_nt(dv2, wv.t(), out=dxkv, accumulate=True)
        return (dxq.view(xq.shape), dxkv.view(xkv.shape), dwq, dwk, dwv, dfc, dg, db) + (None,) * 7 + (dxv, None)


class _ReadoutFn(torch.autograd.Function):
    """lamp/Models.py:124-126: logits[b, i] = <y[b, i, :], w_out[i, :]>."""

    @staticmethod
    def forward(ctx, y, w_out):
        ctx.save_for_backward(y, w_out)
        return N.diag_logits(y, w_out)

    @staticmethod
    def backward(ctx, dl):
        y, w = ctx.saved_tensors
        return N.diag_logits_bwd(y, w, dl)


def ffn_train(mod, x, seeds):
    return _FFNFn.apply(x, mod.w_1.weight, mod.w_1.bias, mod.w_2.weight, mod.w_2.bias, mod.layer_norm.weight,
                        mod.layer_norm.bias, float(mod.dropout.p), seeds.next(),
                        _deferrable(mod.w_1.weight, mod.w_2.weight, mod.w_1.bias, mod.w_2.bias, mod.layer_norm.weight,
                                    mod.layer_norm.bias))


def mha_train(mod, xq, xkv, mask, keep, seeds, xv=None):
    fc = mod.fc.weight if hasattr(mod, 'fc') else None
    return _MHAFn.apply(xq, xkv, mod.w_qs.weight, mod.w_ks.weight, mod.w_vs.weight, fc, mod.layer_norm.weight,
                        mod.layer_norm.bias, mod.n_head, mask, keep, float(mod.attention.dropout.p),
                        float(mod.dropout.p), seeds.next(), seeds.next(), xv,
                        _deferrable(mod.w_qs.weight, mod.w_ks.weight, mod.w_vs.weight, fc, mod.layer_norm.weight,
                                    mod.layer_norm.bias))


class _LinearFn(torch.autograd.Function):
    """XavierLinear on its own (lamp/SubLayers.py:7-13): y = x W^T + b."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return N.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        x2, dy2 = x.reshape(-1, x.size(-1)), dy.reshape(-1, dy.size(-1)).contiguous()
        dx = N.matmul_nt(dy2, w.t()).view(x.shape) if ctx.needs_input_grad[0] else None
        dw = N.matmul_nt(dy2.t(), x2.t()) if ctx.needs_input_grad[1] else None
        db = N.colsum(dy2) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


def linear_train(mod, x):
    N.require_device(x)
    return _LinearFn.apply(x, mod.linear.weight, mod.linear.bias)


class _SDPAFn(torch.autograd.Function):
    """ScaledDotProductAttention on its own (lamp/SubLayers.py:27-43) on head-major batches q (n, lq, dk), k (n, lk, dk),
    v (n, lk, dv): out = dropout(softmax(mask(q k^T / t))) v; returns (out, the dropped map) like the reference."""

    @staticmethod
    def forward(ctx, q, k, v, mask, inv_t, p, seed):
        out, P = N.sdpa(q, k, v, mask, inv_t, need_attn=True)
        Pd = P
        if p > 0:
            Pd = N.dropout(P, p, seed)
            out = N.matmul_nt(Pd, v.transpose(-1, -2))
        ctx.save_for_backward(q, k, v, P)
        ctx.cfg = (inv_t, p, seed)
        attn = Pd if p > 0 else P.clone()
        ctx.mark_non_differentiable(attn)
        return out, attn

    @staticmethod
    def backward(ctx, do, _dattn_unused):
        q, k, v, P = ctx.saved_tensors
        inv_t, p, seed = ctx.cfg
        do = do.contiguous()
        Pd = N.dropout(P, p, seed) if p > 0 else P
        dv = N.matmul_nt(Pd.transpose(-1, -2), do.transpose(-1, -2))    # dV = Pd^T dO
        dP = N.matmul_nt(do, v)                                         # dPd = dO V^T
        if p > 0:
            N.dropout(dP, p, seed, out=dP)
        N.softmax_bwd(P, dP, inv_t, out=dP)                             # dS (in place)
        dq = N.matmul_nt(dP, k.transpose(-1, -2))                       # dQ = dS K
        dk = N.matmul_nt(dP.transpose(-1, -2), q.transpose(-1, -2))     # dK = dS^T Q
        return dq, dk, dv, None, None, None, None


def sdpa_train(mod, q, k, v, attn_mask):
    N.require_device(q, k, v)
    seed = _Seeds().next()
    return _SDPAFn.apply(q, k, v, attn_mask, 1.0 / float(mod.temperature), float(mod.dropout.p), seed)


def forward_train(model, src_seq, src_pos, return_attns=False, int_preds=False):
    """LAMP.forward (lamp/Models.py:110-137) in training mode, attached to the autograd graph; same return tuples as
    the reference: (logits, enc_output, None) | (..., intermediate_preds) | (..., [enc_attns], [slf_attns, enc_dec]).
    The encoder self-attention is skipped unless its maps are requested: the reference discards its output
    (lamp/Layers.py:16-18), so its parameters receive no gradient there either.  Intermediate predictions read out
    through a detached copy of the projection, as the reference does (lamp/Models.py:129-132)."""
    enc, dec = model.encoder, model.decoder
    seq = src_seq.long().contiguous()
    pos = src_pos.long().contiguous()
    seeds = _Seeds()
    B, T = seq.shape
    pad_mask, keep = N.key_token_mask(seq, T)
    pos_w = enc.position_enc.weight if hasattr(enc, 'position_enc') else None
    x = _EmbedFn.apply(seq, pos, enc.src_word_emb.weight, pos_w)
    enc_attns = []
    for layer in enc.layer_stack:
        if return_attns:
            enc_attns.append(mha_train(layer.slf_attn, x, x, pad_mask, keep, seeds)[1])
        x = ffn_train(layer.pos_ffn, x, seeds)
    y = _LabelRowsFn.apply(dec.tgt_word_emb.weight, B)
    label_mask = dec.label_mask_struct()
    if label_mask is not None:  # the map-writing attention variant visits every tile
        label_mask = N.Mask(label_mask.kind, 0, label_mask.ptr, label_mask.stride_b, label_mask.stride_q, None, 0, 0)
    int_outs, slf_attns, enc_dec_attns = [], [], []
    for layer in dec.layer_stack:
        y, a_enc = mha_train(layer.enc_attn, y, x, pad_mask, keep, seeds)
        y = ffn_train(layer.pos_ffn1, y, seeds)
        a_slf = None
        if hasattr(layer, 'slf_attn'):
            int_outs.append(y)
            y, a_slf = mha_train(layer.slf_attn, y, y, label_mask, None, seeds)
        y = ffn_train(layer.pos_ffn2, y, seeds)
        int_outs.append(y)
        slf_attns.append(a_slf)
        enc_dec_attns.append(a_enc)
    w_out = model.tgt_word_proj.linear.weight
    if w_out.size(0) != model.n_labels:
        raise NotImplementedError('proj_share_weight=False read-out is not on the graph path')
    logits = _ReadoutFn.apply(y, w_out)
    if int_preds:
        w_copy = w_out.detach()
        return logits, x, [_ReadoutFn.apply(o, w_copy) for o in int_outs[:-1]]
    if return_attns:
        return logits, x, [enc_attns], [slf_attns, enc_dec_attns]
    return logits, x, None
