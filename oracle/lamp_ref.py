"""CPU oracle for LaMP's label-graph message-passing forward path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product path (``lamp_amd``) never routes through this file; it
fails loudly when the HIP library is missing.

What it is: a functional (no ``nn.Module``) restatement, in plain PyTorch CPU
ops, of the arithmetic the reference performs on the graph-encoder /
graph-decoder forward path.  It consumes a ``state_dict`` with the reference's
key layout and plain tensors.

Pinning: the reference ships no tests, golden vectors or known-answer files
for this path ("parity unpinned" by the reference itself).  The oracle is
therefore pinned against outputs of the reference *run in the build
container*: ``tests/golden/make_golden.py`` imports ``/root/reference`` and
writes small ``.npz`` fixtures (inputs, weights, expected outputs) that are
committed under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this
file against every one of them, and ``tests/test_oracle_vs_reference.py``
checks it against the live reference at the BASELINE sizes whenever
``/root/reference`` is present.

Reference lines restated (paths relative to the reference root):
    layer_norm / mha  -> lamp/SubLayers.py:77-121 (MultiHeadAttention.forward)
    sdpa              -> lamp/SubLayers.py:27-43  (ScaledDotProductAttention.forward)
    ffn               -> lamp/SubLayers.py:133-142 (PositionwiseFeedForward.forward)
    encoder           -> lamp/Encoders.py:64-110, lamp/Layers.py:15-20
    decoder           -> lamp/Decoders.py:127-163, lamp/Layers.py:34-48
    label mask        -> lamp/Decoders.py:105-120, lamp/utils.py:46-50
    read-out          -> lamp/Models.py:110-137
    sinusoid table    -> lamp/utils.py:9-19
    prior adjacency   -> utils/data_loader.py:37-47
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

PAD = 0  # lamp/Constants.py:1

LN_EPS = 1e-5  # nn.LayerNorm default, lamp/SubLayers.py:68,130


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------
def sinusoid_table(n_position, d):
    """lamp/utils.py:9-19 -- row 0 is zeros; sin on even dims, cos on odd dims.

    Computed in float64 then rounded to float32, exactly as the reference does.
    """
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d, dtype=np.float64)[None, :]
    tab = pos / np.power(10000.0, 2.0 * np.floor(j / 2.0) / d)
    tab[0, :] = 0.0
    tab[1:, 0::2] = np.sin(tab[1:, 0::2])
    tab[1:, 1::2] = np.cos(tab[1:, 1::2])
    return torch.from_numpy(tab).to(torch.float32)


def prior_adjacency(label_lists, n_labels, index_offset=4):
    """utils/data_loader.py:37-47 -- identity plus symmetric co-occurrence.

    ``label_lists`` holds, per training sample, the target-vocabulary ids of
    its labels (specials already stripped); the reference subtracts 4.
    """
    adj = torch.eye(n_labels)
    for labels in label_lists:
        ids = [int(x) - index_offset for x in labels]
        for a in ids:
            for b in ids:
                if a != b:
                    adj[a, b] = 1.0
                    adj[b, a] = 1.0
    return adj


def label_block_mask(label_adj_matrix, label_mask, n_labels):
    """lamp/Decoders.py:105-120.  Returns an (L, L) bool tensor, True = blocked,
    or None for the fully connected label graph ('none')."""
    if label_adj_matrix is not None:
        adj = label_adj_matrix.clone().to(torch.float32)
        for i in range(adj.size(0)):
            if adj[i].sum().item() < 1:
                adj[i, i] = 1.0
        return adj == 0
    if label_mask == 'inveye':
        return ~torch.eye(n_labels, dtype=torch.bool)
    if label_mask == 'none':
        return None
    # The reference merely evaluates the name NotImplementedError here and
    # carries on without a label_mask attribute; forward then raises.
    raise NotImplementedError(label_mask)


def layer_norm(x, g, b):
    return F.layer_norm(x, (x.size(-1),), g, b, LN_EPS)


# --------------------------------------------------------------------------
# sub-layers
# --------------------------------------------------------------------------
def sdpa(q, k, v, blocked=None, temperature=None):
    """lamp/SubLayers.py:27-43.  q,k,v: (N, l, dk); blocked: (N, lq, lk) bool."""
    if temperature is None:
        temperature = np.power(q.size(-1), 0.5)
    attn = torch.bmm(q, k.transpose(1, 2))
    attn = attn / temperature
    if blocked is not None:
        attn = attn.masked_fill(blocked.bool(), float('-inf'))
    attn = torch.softmax(attn, dim=2)
    return torch.bmm(attn, v), attn


def mha(xq, xkv, blocked, wq, wk, wv, wfc, g, b, n_head, as_written=False, xv=None):
    """lamp/SubLayers.py:77-121.  ``xv``: a value source distinct from the key source xkv (the module projects k and v
    independently, lamp/SubLayers.py:91-93; every layer of the reference passes the same tensor for both).

    xq (B, lq, d), xkv (B, lk, d); blocked broadcastable to (B, lq, lk) bool or
    None; wq/wk (h*dk, d), wv (h*dv, d), wfc (d, h*dv) or None when h == 1.
    Returns (out (B, lq, d), attn (h*B, lq, lk)) with attn index = head*B + b.
    ``as_written`` reproduces the reference's permute/contiguous copies and the
    per-head mask repeat instead of broadcasting.
    """
    B, lq, _ = xq.shape
    lk = xkv.size(1)
    dk = wq.size(0) // n_head
    dv = wv.size(0) // n_head
    q = F.linear(xq, wq).view(B, lq, n_head, dk)
    k = F.linear(xkv, wk).view(B, lk, n_head, dk)
    v = F.linear(xkv if xv is None else xv, wv).view(B, lk, n_head, dv)
    q = q.permute(2, 0, 1, 3).contiguous().view(-1, lq, dk)
    k = k.permute(2, 0, 1, 3).contiguous().view(-1, lk, dk)
    v = v.permute(2, 0, 1, 3).contiguous().view(-1, lk, dv)
    m = None
    if blocked is not None:
        m = blocked.bool().expand(B, lq, lk)
        m = m.repeat(n_head, 1, 1) if as_written else m.unsqueeze(0).expand(
            n_head, B, lq, lk).reshape(n_head * B, lq, lk)
    out, attn = sdpa(q, k, v, m, np.power(dk, 0.5))
    out = out.view(n_head, B, lq, dv).permute(1, 2, 0, 3).contiguous().view(B, lq, -1)
    if wfc is not None:
        out = F.linear(out, wfc)
    return layer_norm(out + xq, g, b), attn


def ffn(x, w1, b1, w2, b2, g, b, as_written=False):
    """lamp/SubLayers.py:133-142.  w1 (dff, d, 1), w2 (d, dff, 1) Conv1d weights."""
    if as_written:
        h = x.transpose(1, 2)
        h = F.conv1d(F.relu(F.conv1d(h, w1, b1)), w2, b2).transpose(1, 2)
    else:
        h = F.linear(F.relu(F.linear(x, w1.squeeze(-1), b1)), w2.squeeze(-1), b2)
    return layer_norm(h + x, g, b)


def _mha_params(sd, prefix):
    return (sd[prefix + 'w_qs.weight'], sd[prefix + 'w_ks.weight'], sd[prefix + 'w_vs.weight'],
            sd.get(prefix + 'fc.weight'), sd[prefix + 'layer_norm.weight'],
            sd[prefix + 'layer_norm.bias'])


def _ffn_params(sd, prefix):
    return (sd[prefix + 'w_1.weight'], sd[prefix + 'w_1.bias'], sd[prefix + 'w_2.weight'],
            sd[prefix + 'w_2.bias'], sd[prefix + 'layer_norm.weight'],
            sd[prefix + 'layer_norm.bias'])


def count_layers(sd, stack):
    n = 0
    while any(key.startswith('%s.layer_stack.%d.' % (stack, n)) for key in sd):
        n += 1
    return n


# --------------------------------------------------------------------------
# whole forward
# --------------------------------------------------------------------------
def encoder_forward(sd, src_seq, src_pos, n_head, return_attns=False, as_written=False):
    """lamp/Encoders.py:64-110 + lamp/Layers.py:15-20 (graph branch only)."""
    x = F.embedding(src_seq, sd['encoder.src_word_emb.weight'])
    if 'encoder.position_enc.weight' in sd:
        x = x + F.embedding(src_pos, sd['encoder.position_enc.weight'])
    attns = []
    pad = None
    if return_attns or as_written:
        T = src_seq.size(1)
        pad = src_seq.eq(PAD).unsqueeze(1).expand(-1, T, T)
    for i in range(count_layers(sd, 'encoder')):
        p = 'encoder.layer_stack.%d.' % i
        if return_attns or as_written:
            # The reference computes this self-attention and throws its output
            # away (lamp/Layers.py:16-18); only the attention map survives.
            _, a = mha(x, x, pad, *_mha_params(sd, p + 'slf_attn.'), n_head=n_head,
                       as_written=as_written)
            attns.append(a)
        x = ffn(x, *_ffn_params(sd, p + 'pos_ffn.'), as_written=as_written)
    return x, attns


def decoder_forward(sd, src_seq, enc, label_blocked, n_head, n_head2=None,
                    return_attns=False, int_preds=False, as_written=False):
    """lamp/Decoders.py:127-163 + lamp/Layers.py:34-48."""
    n_head2 = n_head if n_head2 is None else n_head2
    B = src_seq.size(0)
    emb = sd['decoder.tgt_word_emb.weight']
    L = emb.size(0)
    if as_written:
        tgt = torch.arange(L).view(-1, 1).repeat(1, B).transpose(0, 1)
        y = F.embedding(tgt, emb)
    else:
        y = emb.unsqueeze(0).expand(B, L, -1)
    T = enc.size(1)
    pad = src_seq[:, :T].eq(PAD).unsqueeze(1).expand(B, L, T)
    blk = None
    if label_blocked is not None:
        blk = label_blocked.view(1, L, L)
        if as_written:
            blk = blk.to(torch.float32).repeat(B, 1, 1).to(torch.uint8)
    slf_attns, enc_attns, int_outs = [], [], []
    for i in range(count_layers(sd, 'decoder')):
        p = 'decoder.layer_stack.%d.' % i
        y, a_enc = mha(y, enc, pad, *_mha_params(sd, p + 'enc_attn.'), n_head=n_head,
                       as_written=as_written)
        y = ffn(y, *_ffn_params(sd, p + 'pos_ffn1.'), as_written=as_written)
        a_slf = None
        if (p + 'slf_attn.w_qs.weight') in sd:
            int_outs.append(y)
            y, a_slf = mha(y, y, blk, *_mha_params(sd, p + 'slf_attn.'), n_head=n_head2,
                           as_written=as_written)
        y = ffn(y, *_ffn_params(sd, p + 'pos_ffn2.'), as_written=as_written)
        int_outs.append(y)
        slf_attns.append(a_slf)
        enc_attns.append(a_enc)
    return y, slf_attns, enc_attns, int_outs


def readout(y, w_out, as_written=False):
    """lamp/Models.py:124-126: diag(y @ W^T) -> (B, L)."""
    if as_written:
        return torch.diagonal(F.linear(y, w_out), 0, 1, 2)
    return (y * w_out.unsqueeze(0)).sum(-1)


def forward(sd, src_seq, src_pos, n_head, label_blocked, n_head2=None,
            return_attns=False, int_preds=False, as_written=False):
    """lamp/Models.py:110-137 for encoder='graph', decoder='graph'.

    Returns the same tuple structure as the reference.
    ``label_blocked``: (L, L) bool (True = blocked) or None.
    """
    enc, enc_attns = encoder_forward(sd, src_seq, src_pos, n_head, return_attns, as_written)
    y, slf_attns, enc_dec_attns, int_outs = decoder_forward(
        sd, src_seq, enc, label_blocked, n_head, n_head2, return_attns, int_preds, as_written)
    w_out = sd['tgt_word_proj.linear.weight']
    logits = readout(y, w_out, as_written)
    logits = logits.reshape(-1, logits.size(-1))
    if int_preds:
        return logits, enc, [readout(o, w_out, as_written) for o in int_outs[:-1]]
    if return_attns:
        return logits, enc, [enc_attns], [slf_attns, enc_dec_attns]
    return logits, enc, None


def to_dtype(sd, dtype):
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


# --------------------------------------------------------------------------
# synthetic model / input construction: lives in lamp_amd/synthetic.py (plain data generation, no reference
# arithmetic); re-exported here for the tests' convenience
# --------------------------------------------------------------------------
from lamp_amd.synthetic import make_adjacency, make_batch, make_state_dict  # noqa: E402,F401
