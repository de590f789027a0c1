"""Sub-layers of the LaMP graph encoder/decoder, eval-mode forward on MI355X.

Same classes, constructor signatures, parameter names/shapes and initialisers as the reference's
lamp/SubLayers.py, so reference checkpoints load.  ``forward`` does no
arithmetic in PyTorch: every op goes through the C ABI of liblamp_hip.so (lamp_amd/_native.py).

Scope: a HIP device; a CPU tensor raises -- there is no fallback.  ``module.eval()`` is the fused inference
path.  In training mode MultiHeadAttention and PositionwiseFeedForward (and LAMP.forward as a whole) run the
autograd-recording path of lamp_amd/training.py, whose backward is HIP kernels as well; so do the
bare XavierLinear / ScaledDotProductAttention wrappers when called on their own.
"""
import numpy as np
import torch.nn as nn

from . import _native as N


class XavierLinear(nn.Module):
    """nn.Linear with Xavier-normal weight, wrapped so the parameter is ``linear.weight``
    (reference: lamp/SubLayers.py:7-13)."""

    def __init__(self, d_in, d_out, bias=True):
        super().__init__()
        self.linear = nn.Linear(d_in, d_out, bias=bias)
        nn.init.xavier_normal_(self.linear.weight)

    def forward(self, x):
        if self.training:
            from . import training
            return training.linear_train(self, x)
        return N.linear(x, self.linear.weight, self.linear.bias)


class ScaledDotProductAttention(nn.Module):
    """softmax(mask(q k^T / temperature)) v on head-major batches (reference: lamp/SubLayers.py:16-43).

    ``attn_type`` is accepted like in the reference; anything but 'softmax' is never reached by the
    reference's own call sites and is rejected here.  ``need_attn=False`` skips the
    write-out of the (N, lq, lk) probability tensor and returns ``None`` in its place.
    """

    def __init__(self, temperature, dropout=0.1, attn_type='softmax'):
        super().__init__()
        self.temperature = temperature
        self.dropout = nn.Dropout(dropout)  # identity in eval mode; kept for module-tree parity
        if attn_type != 'softmax':
            raise NotImplementedError("attn_type=%r: only 'softmax' is on the path" % (attn_type,))
        self.attn_type = nn.Softmax(dim=2)
        self.need_attn = True

    def forward(self, q, k, v, attn_mask=None, stop_sig=False):
        if self.training:
            from . import training
            return training.sdpa_train(self, q, k, v, attn_mask)
        return N.sdpa(q, k, v, attn_mask, 1.0 / float(self.temperature), need_attn=self.need_attn)


class MultiHeadAttention(nn.Module):
    """Bias-free Q/K/V projections, per-head masked attention, bias-free output projection (absent
    for a single head), residual, post-LayerNorm (reference: lamp/SubLayers.py:46-121).

    Returns ``(out (B, lq, d_model), attn (n_head*B, lq, lk))`` with attn index = head*B + b.
    """

    def __init__(self, n_head, d_model, d_k, d_v, dropout=0.1, dropout2=False, attn_type='softmax'):
        super().__init__()
        self.n_head, self.d_k, self.d_v = n_head, d_k, d_v
        self.w_qs = nn.Linear(d_model, n_head * d_k, bias=False)
        self.w_ks = nn.Linear(d_model, n_head * d_k, bias=False)
        self.w_vs = nn.Linear(d_model, n_head * d_v, bias=False)
        for lin, width in ((self.w_qs, d_k), (self.w_ks, d_k), (self.w_vs, d_v)):
            nn.init.normal_(lin.weight, mean=0.0, std=float(np.sqrt(2.0 / (d_model + width))))
        self.attention = ScaledDotProductAttention(
            temperature=np.power(d_k, 0.5), attn_type=attn_type, dropout=dropout2 if dropout2 else dropout)
        self.dropout = nn.Dropout(dropout)
        self.layer_norm = nn.LayerNorm(d_model)
        if n_head > 1:
            self.fc = nn.Linear(n_head * d_v, d_model, bias=False)
            nn.init.xavier_normal_(self.fc.weight)
        self.need_attn = True

    def _mask_struct(self, attn_mask, B, lq, lk):
        if attn_mask is None:
            return None, None
        if isinstance(attn_mask, N.Mask):  # pre-built descriptor handed down by the encoder/decoder
            return attn_mask, None
        return N.make_mask(attn_mask, B, lq, lk)

    def forward(self, q, k, v, attn_mask=None, dec_self=False):
        B, lq, d = q.shape
        lk = k.size(1)
        mstruct, keep = self._mask_struct(attn_mask, B, lq, lk)
        same_kv = (k is v) or (k.data_ptr() == v.data_ptr() and k.shape == v.shape and k.stride() == v.stride())
        if self.training:
            from . import training
            N.require_device(q, k, v)
            return training.mha_train(self, q, k, mstruct, keep, training._Seeds(), xv=None if same_kv else v)
        if same_kv:
            out, attn = N.mha(q, k, N.mha_weights(self), self.d_k, self.d_v, mstruct, self.need_attn)
            del keep
            return out, attn
        # Distinct key and value sources (lamp/SubLayers.py:77-93 projects them independently; no layer of the reference
        # does this, lamp/Layers.py:16,35,40): the same kernels, launched piecewise through the C ABI -- three
        # projections, the fused-layout attention, output projection + residual, LayerNorm.
        N.require_device(q, k, v)
        H = self.n_head
        a, attn = N.sdpa_fused(N.linear(q, self.w_qs.weight), N.linear(k, self.w_ks.weight), N.linear(v, self.w_vs.weight),
                               H, mstruct, 1.0 / float(self.d_k) ** 0.5, need_attn=self.need_attn)
        del keep
        if H > 1:
            o = N.linear(a, self.fc.weight, residual=q)
            return N.layernorm(o, self.layer_norm.weight, self.layer_norm.bias).view(q.shape), attn
        return N.layernorm_residual(a, q, self.layer_norm.weight, self.layer_norm.bias).view(q.shape), attn


class PositionwiseFeedForward(nn.Module):
    """LayerNorm(W2 relu(W1 x + b1) + b2 + x); the two maps are Conv1d(k=1) modules so that the
    parameters keep the reference's (out, in, 1) shapes (reference: lamp/SubLayers.py:125-142)."""

    def __init__(self, d_in, d_hid, dropout=0.1):
        super().__init__()
        self.w_1 = nn.Conv1d(d_in, d_hid, 1)
        self.w_2 = nn.Conv1d(d_hid, d_in, 1)
        self.layer_norm = nn.LayerNorm(d_in)
        self.dropout = nn.Dropout(dropout)
        self.d_hid = d_hid

    def forward(self, x):
        if self.training:
            from . import training
            N.require_device(x)
            return training.ffn_train(self, x, training._Seeds())
        return N.ffn(x, N.ffn_weights(self), self.d_hid)
