"""Synthetic weights, label graphs and token batches in the shapes SURVEY.md section 8(d) prescribes
(real datasets are not available offline).  Pure data generation: distributions follow the reference's
initialisers (lamp/SubLayers.py:57-59,74,11 and torch defaults), nothing here computes the forward path.
Used by bench.py, __graft_entry__.smoke() and the tests."""
import math

import torch

from . import Constants
from .utils import position_encoding_init


def make_state_dict(n_src_vocab, n_labels, n_max_seq, d_model, d_inner, n_head, n_layers_enc,
                    n_layers_dec, pos_emb=True, seed=0, qk_scale=1.0, no_dec_self_att=False):
    g = torch.Generator().manual_seed(seed)
    dk = d_model // n_head

    def normal(shape, std):
        return torch.randn(shape, generator=g) * std

    def uniform(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    sd = {}
    emb = normal((n_src_vocab, d_model), 1.0)
    emb[Constants.PAD] = 0
    sd['encoder.src_word_emb.weight'] = emb
    if pos_emb:
        sd['encoder.position_enc.weight'] = position_encoding_init(n_max_seq + 1, d_model)

    def add_mha(p, scale=1.0):
        sd[p + 'w_qs.weight'] = normal((n_head * dk, d_model), math.sqrt(2.0 / (d_model + dk))) * scale
        sd[p + 'w_ks.weight'] = normal((n_head * dk, d_model), math.sqrt(2.0 / (d_model + dk))) * scale
        sd[p + 'w_vs.weight'] = normal((n_head * dk, d_model), math.sqrt(2.0 / (d_model + dk)))
        sd[p + 'layer_norm.weight'] = torch.ones(d_model) + normal((d_model,), 0.05)
        sd[p + 'layer_norm.bias'] = normal((d_model,), 0.05)
        if n_head > 1:
            sd[p + 'fc.weight'] = normal((d_model, n_head * dk), math.sqrt(2.0 / (d_model + n_head * dk)))

    def add_ffn(p):
        sd[p + 'w_1.weight'] = uniform((d_inner, d_model, 1), 1.0 / math.sqrt(d_model))
        sd[p + 'w_1.bias'] = uniform((d_inner,), 1.0 / math.sqrt(d_model))
        sd[p + 'w_2.weight'] = uniform((d_model, d_inner, 1), 1.0 / math.sqrt(d_inner))
        sd[p + 'w_2.bias'] = uniform((d_model,), 1.0 / math.sqrt(d_inner))
        sd[p + 'layer_norm.weight'] = torch.ones(d_model) + normal((d_model,), 0.05)
        sd[p + 'layer_norm.bias'] = normal((d_model,), 0.05)

    for i in range(n_layers_enc):
        add_mha('encoder.layer_stack.%d.slf_attn.' % i)
        add_ffn('encoder.layer_stack.%d.pos_ffn.' % i)
    sd['decoder.tgt_word_emb.weight'] = normal((n_labels, d_model), 1.0)
    for i in range(n_layers_dec):
        add_mha('decoder.layer_stack.%d.enc_attn.' % i, qk_scale)
        add_ffn('decoder.layer_stack.%d.pos_ffn1.' % i)
        if not no_dec_self_att:
            add_mha('decoder.layer_stack.%d.slf_attn.' % i, qk_scale)
        add_ffn('decoder.layer_stack.%d.pos_ffn2.' % i)
    sd['tgt_word_proj.weight'] = sd['decoder.tgt_word_emb.weight']
    sd['tgt_word_proj.linear.weight'] = normal((n_labels, d_model), math.sqrt(2.0 / (d_model + n_labels)))
    return sd


def make_adjacency(n_labels, p, seed=0):
    """Symmetric Bernoulli(p) OR identity."""
    g = torch.Generator().manual_seed(seed)
    a = (torch.rand((n_labels, n_labels), generator=g) < p).float()
    a = ((a + a.t()) > 0).float()
    a.fill_diagonal_(1.0)
    return a


def make_batch(batch, n_src_vocab, t_max, lengths=None, seed=0):
    """Tokens i.i.d. U{4..V-1}; src_pos = 1..len then 0 on pads (utils/data_loader.py:261-279)."""
    g = torch.Generator().manual_seed(seed)
    if lengths is None:
        lengths = [t_max] * batch
    T = max(max(lengths), 1)
    src_seq = torch.zeros((batch, T), dtype=torch.int64)
    src_pos = torch.zeros((batch, T), dtype=torch.int64)
    for b, n in enumerate(lengths):
        if n > 0:
            src_seq[b, :n] = torch.randint(4, n_src_vocab, (n,), generator=g)
            src_pos[b, :n] = torch.arange(1, n + 1)
    return src_seq, src_pos
