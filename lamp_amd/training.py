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
"""
import torch

from . import Constants
from . import _native as N


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
    def forward(ctx, x, w1, b1, w2, b2, ln_g, ln_b, p, seed):
        x2 = x.reshape(-1, x.size(-1))
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
        dz, do, dg, db, db2 = N.layernorm_bwd(o, x2, ln_g, dy.reshape(x2.shape), dropout_p=ctx.p, seed=ctx.seed,
                                              want_dbias=True)
        dW2 = N.matmul_nt(do.t(), h.t())
        dh = N.matmul_nt(do, W2.t(), relu_mask=h)
        db1 = N.colsum(dh)
        dW1 = N.matmul_nt(dh.t(), x2.t())
        dx = N.matmul_nt(dh, W1.t(), out=dz, accumulate=True)  # + the residual branch (do is dead by now)
        return dx.view(ctx.shape), dW1.view_as(w1), db1, dW2.view_as(w2), db2, dg, db, None, None


class _MHAFn(torch.autograd.Function):
    """LayerNorm(dropout(concat_heads(dropout(softmax(mask(q k^T / t))) v) Wfc^T) + xq)   (lamp/SubLayers.py:77-121)."""

    @staticmethod
    def forward(ctx, xq, xkv, wq, wk, wv, fc, ln_g, ln_b, n_head, mask, keep, p_attn, p_out, seed_attn, seed_out,
                xv=None):
        """xkv: the key source and, unless ``xv`` is given, the value source too (every layer of the reference passes one
        tensor for both, lamp/Layers.py:16,35,40; the module itself accepts two, lamp/SubLayers.py:77-93)."""
        B, lq, d = xq.shape
        lk = xkv.size(1)
        H = n_head
        dk, dv = wq.size(0) // H, wv.size(0) // H
        q = N.linear(xq, wq)
        k = N.linear(xkv, wk)
        v = N.linear(xkv if xv is None else xv, wv)
        inv_t = 1.0 / float(dk) ** 0.5
        a, P = N.sdpa_fused(q, k, v, H, mask, inv_t, need_attn=True, fast_maps=True)
        Pd = P
        if p_attn > 0:  # the reference drops probabilities AFTER the softmax; the value product uses the dropped map
            Pd = N.dropout(P, p_attn, seed_attn)
            N.matmul_nt(Pd.view(H, B, lq, lk), v.view(B, lk, H, dv).permute(2, 0, 3, 1),
                        out=a.view(B, lq, H, dv).permute(2, 0, 1, 3))
        o = N.linear(a, fc) if fc is not None else a
        y = N.layernorm_residual(o, xq, ln_g, ln_b, dropout_p=p_out, seed=seed_out)
        ctx.save_for_backward(xq, xkv, wq, wk, wv, fc if fc is not None else wq.new_empty(0), ln_g, q, k, v, a, P, o,
                              xv if xv is not None else wq.new_empty(0))
        ctx.cfg = (B, lq, lk, H, dk, dv, inv_t, p_attn, p_out, seed_attn, seed_out, fc is not None, xv is not None)
        attn = Pd if p_attn > 0 else P.clone()  # what the reference returns (lamp/SubLayers.py:40-43): the dropped map
        ctx.mark_non_differentiable(attn)
        return y, attn

    @staticmethod
    def backward(ctx, dy, _dP_unused):
        xq, xkv, wq, wk, wv, fc, ln_g, q, k, v, a, P, o, xv = ctx.saved_tensors
        B, lq, lk, H, dk, dv, inv_t, p_attn, p_out, seed_attn, seed_out, has_fc, has_xv = ctx.cfg
        d = xq.size(-1)
        xq2, xkv2 = xq.reshape(-1, d), xkv.reshape(-1, d)
        xv2 = xv.reshape(-1, d) if has_xv else xkv2
        dz, do, dg, db, _ = N.layernorm_bwd(o.view(xq2.shape), xq2, ln_g, dy.reshape(xq2.shape), dropout_p=p_out,
                                            seed=seed_out)
        a2 = a.view(-1, H * dv)
        if has_fc:
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
        dwq = N.matmul_nt(dq2.t(), xq2.t())
        dwk = N.matmul_nt(dk2.t(), xkv2.t())
        dwv = N.matmul_nt(dv2.t(), xv2.t())
        dxq = N.matmul_nt(dq2, wq.t(), out=dz, accumulate=True)  # + the residual branch
        dxkv = N.matmul_nt(dk2, wk.t())
        dxv = None
        if has_xv:
            dxv = N.matmul_nt(dv2, wv.t()).view(xv.shape)
        else:
            N.matmul_nt(dv2, wv.t(), out=dxkv, accumulate=True)
        return (dxq.view(xq.shape), dxkv.view(xkv.shape), dwq, dwk, dwv, dfc, dg, db) + (None,) * 7 + (dxv,)


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
                        mod.layer_norm.bias, float(mod.dropout.p), seeds.next())


def mha_train(mod, xq, xkv, mask, keep, seeds, xv=None):
    fc = mod.fc.weight if hasattr(mod, 'fc') else None
    return _MHAFn.apply(xq, xkv, mod.w_qs.weight, mod.w_ks.weight, mod.w_vs.weight, fc, mod.layer_norm.weight,
                        mod.layer_norm.bias, mod.n_head, mask, keep, float(mod.attention.dropout.p),
                        float(mod.dropout.p), seeds.next(), seeds.next(), xv)


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
        label_mask = N.Mask(label_mask.kind, 0, label_mask.ptr, label_mask.stride_b, label_mask.stride_q, None, 0)
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
