"""Encoder / decoder layers of the LaMP graph model (reference: lamp/Layers.py:9-48)."""
import torch.nn as nn

from .SubLayers import MultiHeadAttention, PositionwiseFeedForward


class EncoderLayer(nn.Module):
    """Token-state update.  In the reference the self-attention's OUTPUT is computed and then
    overwritten by ``pos_ffn(enc_input)`` (lamp/Layers.py:16-18): only its attention map is ever
    observable.  ``need_attn=False`` therefore skips the block entirely and returns ``None`` for the
    map; the default keeps the reference's return value."""

    def __init__(self, d_model, d_inner_hid, n_head, d_k, d_v, dropout=0.1):
        super().__init__()
        self.slf_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_inner_hid, dropout=dropout)

    def forward(self, enc_input, slf_attn_mask=None, need_attn=True):
        attn = None
        if need_attn:
            _, attn = self.slf_attn(enc_input, enc_input, enc_input, attn_mask=slf_attn_mask)
        return self.pos_ffn(enc_input), attn


class DecoderLayer(nn.Module):
    """Label-state update: input->label attention, FFN, label->label attention over the label graph,
    FFN (reference: lamp/Layers.py:22-48).  ``attn_type`` / ``ffn`` are accepted and unused, as in
    the reference."""

    def __init__(self, d_model, d_inner_hid, n_head, n_head2, d_k, d_v, dropout=0.1, dropout2=False,
                 no_dec_self_att=False, ffn=True, attn_type='softmax'):
        super().__init__()
        self.enc_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout)
        self.pos_ffn1 = PositionwiseFeedForward(d_model, d_inner_hid, dropout=dropout)
        if not no_dec_self_att:
            self.slf_attn = MultiHeadAttention(n_head2, d_model, d_k, d_v, dropout=dropout, dropout2=dropout2)
        self.pos_ffn2 = PositionwiseFeedForward(d_model, d_inner_hid, dropout=dropout)

    def forward(self, dec_input, enc_output, slf_attn_mask=None, dec_enc_attn_mask=None, need_attn=True):
        self.enc_attn.need_attn = need_attn
        out, enc_attn = self.enc_attn(dec_input, enc_output, enc_output, attn_mask=dec_enc_attn_mask)
        out = self.pos_ffn1(out)
        out_int, slf_attn = None, None
        if hasattr(self, 'slf_attn'):
            out_int = out
            self.slf_attn.need_attn = need_attn
            out, slf_attn = self.slf_attn(out, out, out, attn_mask=slf_attn_mask, dec_self=True)
        out = self.pos_ffn2(out)
        return out, out_int, slf_attn, enc_attn
