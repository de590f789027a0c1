"""Graph encoder of LaMP (reference: lamp/Encoders.py:31-110).

Only the branch the label-graph path uses is built: token (+ sinusoid position) embeddings and a
stack of EncoderLayers.  The genomics one-hot/conv branch, the per-sample ``adj`` branch and
``enc_transform`` pooling, and the MLP/RNN baseline encoders are outside the hot path (SURVEY.md
section 2) and raise at construction.
"""
import torch.nn as nn

from . import Constants, utils
from . import _native as N
from .Layers import EncoderLayer
from .SubLayers import _eval_only


class GraphEncoder(nn.Module):
    def __init__(self, n_src_vocab, n_max_seq, n_layers=6, n_head=8, d_k=64, d_v=64, d_word_vec=512,
                 d_model=512, d_inner_hid=1024, onehot=False, enc_transform='', dropout=0.1,
                 no_enc_pos_embedding=False):
        super().__init__()
        if onehot or enc_transform != '':
            raise NotImplementedError('onehot / enc_transform encoders are outside the label-graph hot path')
        self.n_max_seq = n_max_seq
        self.d_model = d_model
        self.onehot = onehot
        self.enc_transform = enc_transform
        self.dropout = nn.Dropout(dropout)
        self.src_word_emb = nn.Embedding(n_src_vocab, d_word_vec, padding_idx=Constants.PAD)
        if no_enc_pos_embedding is False:
            n_position = n_max_seq + 1
            self.position_enc = nn.Embedding(n_position, d_word_vec, padding_idx=Constants.PAD)
            self.position_enc.weight.data = utils.position_encoding_init(n_position, d_word_vec)
        self.layer_stack = nn.ModuleList(
            EncoderLayer(d_model, d_inner_hid, n_head, d_k, d_v, dropout=dropout) for _ in range(n_layers))

    def forward(self, src_seq, adj, src_pos, return_attns=False):
        _eval_only(self)
        if adj:
            raise NotImplementedError('per-sample adjacency for the encoder is outside the hot path')
        pos_table = self.position_enc.weight if hasattr(self, 'position_enc') else None
        x = N.embed(src_seq, src_pos, self.src_word_emb.weight, pos_table)
        attns = []
        pad_mask, keep = (N.key_token_mask(src_seq, src_seq.size(1)) if return_attns else (None, None))
        for layer in self.layer_stack:
            x, a = layer(x, slf_attn_mask=pad_mask, need_attn=return_attns)
            if return_attns:
                attns.append(a)
        del keep
        return (x, attns) if return_attns else (x, None)


class MLPEncoder(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("encoder='mlp' is a baseline model outside the label-graph hot path")


class RNNEncoder(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("encoder='rnn' is a baseline model outside the label-graph hot path")
