#!/usr/bin/env python3
"""Generate golden vectors by running the *reference* (QData/LaMP) on CPU.

Runs ONLY in the build container, where /root/reference exists.  It imports the
reference's own ``lamp`` package (never this repo's), applies the three
semantics-preserving shims of SURVEY.md section 8(c) (``.cuda()`` -> identity,
uint8 mask -> bool for ``masked_fill``), drives the reference modules with
seeded inputs and writes small ``.npz`` fixtures next to this script.  Weights
and inputs are stored *inside* each fixture so that nothing depends on RNG
reproducibility across torch builds.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

Fixtures are data only (inputs, weights, expected outputs).  No reference source
text is copied anywhere.
"""
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get('LAMP_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
# keep this repo's own packages out of the way: the name `lamp` must resolve to the reference
sys.path = [REF] + [p for p in sys.path if os.path.abspath(p or '.') not in
                    (os.path.abspath(os.path.join(HERE, '..', '..')),
                     os.path.abspath(os.path.join(HERE, '..', '..', 'dropin')))]

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self
_mf = torch.Tensor.masked_fill
torch.Tensor.masked_fill = lambda self, m, v: _mf(self, m.bool() if m.dtype == torch.uint8 else m, v)

from lamp.Models import LAMP  # noqa: E402
from lamp.SubLayers import (MultiHeadAttention, PositionwiseFeedForward,  # noqa: E402
                            ScaledDotProductAttention)

assert os.path.abspath(sys.modules['lamp'].__file__).startswith(os.path.abspath(REF))


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('%-40s %7.1f KB' % (name, os.path.getsize(path) / 1024.0))


def randomize_(module, gen, scale=0.05):
    """Perturb LayerNorm affine params / biases so they are not the trivial 1/0 defaults."""
    for n, p in module.named_parameters():
        if 'layer_norm' in n:
            p.data.add_(torch.randn(p.shape, generator=gen) * scale)


# ---------------------------------------------------------------- module level
def gen_sdpa():
    g = torch.Generator().manual_seed(11)
    N, lq, lk, dk = 6, 7, 9, 16
    q = torch.randn(N, lq, dk, generator=g)
    k = torch.randn(N, lk, dk, generator=g)
    v = torch.randn(N, lk, dk, generator=g)
    mod = ScaledDotProductAttention(temperature=np.power(dk, 0.5)).eval()
    masks = {}
    masks['none'] = None
    kp = torch.zeros(N, lq, lk, dtype=torch.bool)
    for n in range(N):
        kp[n, :, lk - (n % 4):] = True
    masks['keypad'] = kp
    shared = (torch.rand(lq, lk, generator=g) < 0.5)
    shared[:, 0] = False
    masks['shared'] = shared.unsqueeze(0).expand(N, lq, lk).clone()
    full = masks['shared'].clone()
    full[2, 3, :] = True  # one fully masked row -> NaN row in the reference
    masks['fullrow'] = full
    out = {'q': npy(q), 'k': npy(k), 'v': npy(v)}
    for name, m in masks.items():
        o, a = mod(q, k, v, attn_mask=m)
        if m is not None:
            out['mask_' + name] = npy(m)
        out['out_' + name] = npy(o)
        out['attn_' + name] = npy(a)
    save('sdpa', **out)


def gen_mha():
    for h in (1, 4):
        g = torch.Generator().manual_seed(20 + h)
        d, B, lq, lk = 64, 3, 10, 13
        dk = d // h
        torch.manual_seed(100 + h)
        mod = MultiHeadAttention(h, d, dk, dk).eval()
        randomize_(mod, g)
        xq = torch.randn(B, lq, d, generator=g)
        xkv = torch.randn(B, lk, d, generator=g)
        pad = torch.zeros(B, lq, lk, dtype=torch.bool)
        pad[1, :, 9:] = True
        pad[2, :, 4:] = True
        o_cross, a_cross = mod(xq, xkv, xkv, attn_mask=pad)
        slf = (torch.rand(lq, lq, generator=g) < 0.4)
        slf.fill_diagonal_(False)
        slf = slf.unsqueeze(0).expand(B, lq, lq).clone()
        o_self, a_self = mod(xq, xq, xq, attn_mask=slf)
        o_nomask, a_nomask = mod(xq, xkv, xkv, attn_mask=None)
        arrs = {'sd__' + k_: npy(v_) for k_, v_ in mod.state_dict().items()}
        arrs.update(xq=npy(xq), xkv=npy(xkv), pad=npy(pad), slf=npy(slf), n_head=np.int64(h),
                    out_cross=npy(o_cross), attn_cross=npy(a_cross),
                    out_self=npy(o_self), attn_self=npy(a_self),
                    out_nomask=npy(o_nomask), attn_nomask=npy(a_nomask))
        save('mha_h%d' % h, **arrs)


def gen_ffn():
    g = torch.Generator().manual_seed(31)
    torch.manual_seed(131)
    d, dff, B, l = 64, 128, 3, 11
    mod = PositionwiseFeedForward(d, dff).eval()
    randomize_(mod, g)
    x = torch.randn(B, l, d, generator=g)
    y = mod(x)
    arrs = {'sd__' + k_: npy(v_) for k_, v_ in mod.state_dict().items()}
    arrs.update(x=npy(x), out=npy(y))
    save('ffn', **arrs)


# ----------------------------------------------------------------- model level
def build_model(V, L, n_max_seq, d, dff, h, n_enc, n_dec, label_mask, pos_emb, adj,
                no_dec_self_att=False, seed=0):
    torch.manual_seed(seed)
    kw = dict(proj_share_weight=True, embs_share_weight=True, d_k=d // h, d_v=d // h, d_model=d,
              d_word_vec=d, d_inner_hid=dff, n_layers_enc=n_enc, n_layers_dec=n_dec, n_head=h,
              n_head2=h, dropout=0.1, dec_dropout=0.1, dec_dropout2=False, encoder='graph',
              decoder='graph', enc_transform='', onehot=False, no_enc_pos_embedding=not pos_emb,
              no_dec_self_att=no_dec_self_att, loss='ce',
              label_adj_matrix=adj.clone() if adj is not None else None, attn_type='softmax',
              label_mask=label_mask, matching_mlp=False, graph_conv=False, int_preds=False)
    m = LAMP(V, L, n_max_seq, L, **kw).eval()
    randomize_(m, torch.Generator().manual_seed(seed + 1))
    return m


def make_adj(L, p, gen):
    a = (torch.rand(L, L, generator=gen) < p).float()
    a = ((a + a.t()) > 0).float()
    a.fill_diagonal_(1.0)
    return a


def make_inputs(B, V, lengths, gen):
    T = max(lengths)
    seq = torch.zeros(B, T, dtype=torch.int64)
    pos = torch.zeros(B, T, dtype=torch.int64)
    for b, n in enumerate(lengths):
        if n:
            seq[b, :n] = torch.randint(4, V, (n,), generator=gen)
            pos[b, :n] = torch.arange(1, n + 1)
    return seq, pos


def flatten_attns(enc_self_attns, dec_output2):
    out = {}
    for i, a in enumerate(enc_self_attns[0]):
        out['attn_enc_%d' % i] = npy(a)
    for i, a in enumerate(dec_output2[0]):
        if a is not None:
            out['attn_dec_slf_%d' % i] = npy(a)
    for i, a in enumerate(dec_output2[1]):
        out['attn_dec_enc_%d' % i] = npy(a)
    return out


def gen_models():
    V, d, dff, n_max_seq = 40, 64, 128, 12
    case = 0
    for label_mask in ('prior', 'none', 'inveye'):
        for pos_emb in (True, False):
            for h in (1, 4):
                case += 1
                gen = torch.Generator().manual_seed(1000 + case)
                L = 12 + 4 * (case % 4)
                adj = make_adj(L, 0.25, gen) if label_mask == 'prior' else None
                if adj is not None and case % 2 == 0:
                    # a label with an empty adjacency row: the ctor adds a self-loop (Decoders.py:110-112)
                    adj[3, :] = 0
                    adj[:, 3] = 0
                m = build_model(V, L, n_max_seq, d, dff, h, 2, 2, label_mask, pos_emb, adj, seed=case)
                seq, pos = make_inputs(5, V, [12, 7, 3, 9, 1], gen)
                with torch.no_grad():
                    logits, enc, _ = m((seq, pos), None, None, None)
                    lg2, enc2, enc_attns, dec2 = m((seq, pos), None, None, None, return_attns=True)
                assert torch.equal(logits, lg2)
                arrs = {'sd__' + k_: npy(v_) for k_, v_ in m.state_dict().items()}
                arrs.update(src_seq=npy(seq), src_pos=npy(pos), n_head=np.int64(h),
                            label_mask=np.array(label_mask), pos_emb=np.bool_(pos_emb),
                            logits=npy(logits), enc_output=npy(enc))
                if adj is not None:
                    arrs['label_adj_matrix'] = npy(adj)
                    arrs['ref_label_mask'] = npy(m.decoder.label_mask)
                elif m.decoder.label_mask is not None:
                    arrs['ref_label_mask'] = npy(m.decoder.label_mask)
                arrs.update(flatten_attns(enc_attns, dec2))
                save('model_%s_pos%d_h%d' % (label_mask, int(pos_emb), h), **arrs)

    # int_preds (Models.py:127-133) and no_dec_self_att
    gen = torch.Generator().manual_seed(2001)
    L = 14
    adj = make_adj(L, 0.3, gen)
    m = build_model(V, L, n_max_seq, d, dff, 4, 2, 2, 'prior', True, adj, seed=77)
    seq, pos = make_inputs(4, V, [11, 12, 5, 8], gen)
    with torch.no_grad():
        logits, enc, ips = m((seq, pos), None, None, None, int_preds=True)
    arrs = {'sd__' + k_: npy(v_) for k_, v_ in m.state_dict().items()}
    arrs.update(src_seq=npy(seq), src_pos=npy(pos), n_head=np.int64(4), label_mask=np.array('prior'),
                pos_emb=np.bool_(True), label_adj_matrix=npy(adj), logits=npy(logits),
                enc_output=npy(enc))
    for i, p in enumerate(ips):
        arrs['int_pred_%d' % i] = npy(p)
    save('model_int_preds', **arrs)

    gen = torch.Generator().manual_seed(2002)
    m = build_model(V, L, n_max_seq, d, dff, 4, 2, 2, 'none', True, None, no_dec_self_att=True, seed=78)
    seq, pos = make_inputs(4, V, [6, 12, 12, 2], gen)
    with torch.no_grad():
        logits, enc, _ = m((seq, pos), None, None, None)
    arrs = {'sd__' + k_: npy(v_) for k_, v_ in m.state_dict().items()}
    arrs.update(src_seq=npy(seq), src_pos=npy(pos), n_head=np.int64(4), label_mask=np.array('none'),
                pos_emb=np.bool_(True), logits=npy(logits), enc_output=npy(enc))
    save('model_no_dec_self_att', **arrs)

    # a batch holding an all-PAD row (test.py:35-39 creates these): NaN logits for that row only
    gen = torch.Generator().manual_seed(2003)
    adj = make_adj(L, 0.3, gen)
    m = build_model(V, L, n_max_seq, d, dff, 4, 2, 2, 'prior', True, adj, seed=79)
    seq, pos = make_inputs(4, V, [9, 0, 12, 4], gen)
    with torch.no_grad():
        logits, enc, _ = m((seq, pos), None, None, None)
    assert torch.isnan(logits[1]).all() and not torch.isnan(logits[[0, 2, 3]]).any()
    arrs = {'sd__' + k_: npy(v_) for k_, v_ in m.state_dict().items()}
    arrs.update(src_seq=npy(seq), src_pos=npy(pos), n_head=np.int64(4), label_mask=np.array('prior'),
                pos_emb=np.bool_(True), label_adj_matrix=npy(adj), logits=npy(logits),
                enc_output=npy(enc))
    save('model_allpad_row', **arrs)

    # conditioning sweep: decoder Q/K weights scaled; expected values in fp64 and
    # the reference's own fp32-vs-fp64 gap recorded alongside
    for scale in (1, 3, 10):
        gen = torch.Generator().manual_seed(3000 + scale)
        adj = make_adj(L, 0.3, gen)
        m = build_model(V, L, n_max_seq, d, dff, 4, 2, 2, 'prior', True, adj, seed=80 + scale)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if n.startswith('decoder.') and ('w_qs' in n or 'w_ks' in n):
                    p.mul_(float(scale))
        seq, pos = make_inputs(4, V, [12, 10, 5, 8], gen)
        with torch.no_grad():
            lg32, enc32, _ = m((seq, pos), None, None, None)
        sd32 = {k_: v_.clone() for k_, v_ in m.state_dict().items()}
        m64 = m.double()
        m64.decoder.label_mask = m64.decoder.label_mask.double()
        with torch.no_grad():
            lg64, enc64, _ = m64((seq, pos), None, None, None)
        arrs = {'sd__' + k_: npy(v_) for k_, v_ in sd32.items()}
        arrs.update(src_seq=npy(seq), src_pos=npy(pos), n_head=np.int64(4),
                    label_mask=np.array('prior'), pos_emb=np.bool_(True),
                    label_adj_matrix=npy(adj), logits=npy(lg32), enc_output=npy(enc32),
                    logits_fp64=npy(lg64), ref_gap=np.float64((lg32.double() - lg64).abs().max().item()))
        save('model_qkscale_%d' % scale, **arrs)


if __name__ == '__main__':
    gen_sdpa()
    gen_mha()
    gen_ffn()
    gen_models()
