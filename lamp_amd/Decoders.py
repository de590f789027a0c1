"""Graph decoder of LaMP: label nodes attend to the encoded input and to each other through the
label-adjacency mask (reference: lamp/Decoders.py:96-163).

Differences from the reference that do not change results: the (L, L) label mask is kept ONCE on
the device as uint8 and broadcast over batch and heads by the kernel, instead of being repeated to
(B, L, L) on the host, copied to the device and repeated again per head on every forward
(SURVEY.md G5); the key-padding mask is read straight from ``src_seq`` inside the kernel.
"""
import numpy as np
import torch
import torch.nn as nn

from . import utils
from . import _native as N
from .Layers import DecoderLayer
from .SubLayers import _eval_only


def build_label_mask(n_tgt_vocab, label_adj_matrix, label_mask):
    """-> float mask in the reference's own format (1 = blocked): (1, L, L) for a prior adjacency,
    (L, L) for 'inveye', None for 'none' (reference: lamp/Decoders.py:109-120)."""
    if label_adj_matrix is not None:
        empty_rows = (label_adj_matrix.sum(dim=1) < 1).nonzero().flatten().tolist()
        for i in empty_rows:  # an isolated label still sees itself, else its softmax row would be NaN
            label_adj_matrix[i, i] = 1
        return utils.swap_0_1(label_adj_matrix, 1, 0).unsqueeze(0)
    if label_mask == 'inveye':
        return 1 - torch.eye(n_tgt_vocab)
    if label_mask == 'none':
        return None
    raise NotImplementedError('label_mask=%r without a label_adj_matrix' % (label_mask,))


class GraphDecoder(nn.Module):
    def __init__(self, n_tgt_vocab, n_max_seq, n_layers=6, n_head=8, n_head2=8, d_k=64, d_v=64,
                 d_word_vec=512, d_model=512, d_inner_hid=1024, dropout=0.1, dropout2=0.1,
                 no_dec_self_att=False, label_adj_matrix=None, label_mask=None, enc_vec=True,
                 graph_conv=False, attn_type='softmax'):
        super().__init__()
        if enc_vec:
            raise NotImplementedError('vector encoders (mlp / enc_transform) are outside the hot path')
        self.enc_vec = enc_vec
        self.dropout = nn.Dropout(dropout)
        self.constant_input = torch.from_numpy(np.arange(n_tgt_vocab)).view(-1, 1)
        self.tgt_word_emb = nn.Embedding(n_tgt_vocab, d_word_vec)
        self.label_mask = build_label_mask(n_tgt_vocab, label_adj_matrix, label_mask)
        # device-resident uint8 copy, moved by .cuda()/.to() but (like the reference's plain
        # attribute) absent from the state_dict
        blocked = None
        if self.label_mask is not None:
            blocked = (self.label_mask.reshape(n_tgt_vocab, n_tgt_vocab) != 0).to(torch.uint8)
        self.register_buffer('label_mask_u8', blocked, persistent=False)
        # the same mask bit-packed per row: the attention kernel then fetches one 32-bit word per 32-key tile
        self.register_buffer('label_mask_bits', N.pack_mask_bits(blocked) if blocked is not None else None,
                             persistent=False)
        # block structure of the label graph: per 32-label query block, the 32-label key tiles with at least
        # one edge -- lets the attention kernel skip fully blocked tiles (large sparse / clustered graphs)
        self.register_buffer('label_tiles', N.active_tile_list(blocked) if blocked is not None else None,
                             persistent=False)
        self.layer_stack = nn.ModuleList(
            DecoderLayer(d_model, d_inner_hid, n_head, n_head2, d_k, d_v, dropout=dropout, dropout2=dropout2,
                         no_dec_self_att=no_dec_self_att, attn_type=attn_type) for _ in range(n_layers))

    def label_mask_struct(self):
        m = self.label_mask_u8
        if m is None:
            return None
        N.require_device(m)
        L = m.size(0)
        t, bits = self.label_tiles, self.label_mask_bits
        tl = (t.data_ptr() if t is not None else None, t.size(1) if t is not None else 0)
        if bits is not None:
            return N.Mask(N.LAMP_MASK_BITS_U32, 0, bits.data_ptr(), 0, bits.size(1), *tl)
        return N.Mask(N.LAMP_MASK_U8, 0, m.data_ptr(), 0, L, *tl)

    def forward(self, tgt, src_seq, enc_output, return_attns=False, int_preds=False):
        _eval_only(self)
        B = src_seq.size(0)
        T = enc_output.size(1)
        y = self.tgt_word_emb.weight.unsqueeze(0).expand(B, -1, -1).contiguous()
        pad_mask, keep = N.key_token_mask(src_seq[:, :T], T)
        label_mask = self.label_mask_struct()
        int_outs, slf_attns, enc_attns = [], [], []
        for layer in self.layer_stack:
            y, y_int, slf_attn, enc_attn = layer(y, enc_output, slf_attn_mask=label_mask,
                                                 dec_enc_attn_mask=pad_mask, need_attn=return_attns)
            if int_preds:
                if y_int is not None:
                    int_outs.append(y_int)
                int_outs.append(y)
            if return_attns:
                slf_attns.append(slf_attn)
                enc_attns.append(enc_attn)
        del keep
        if int_preds:
            return y, int_outs
        if return_attns:
            return y, slf_attns, enc_attns
        return y, None


class MLPDecoder(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("decoder='mlp' is a baseline model outside the label-graph hot path")


class RNNDecoder(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("decoder='rnn_m' is a baseline model outside the label-graph hot path")
