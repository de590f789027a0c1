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
import torch.nn.functional as F

from . import Constants, utils
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
        self.enc_vec = enc_vec   # the encoder hands over ONE vector per sample (mlp / enc_transform): no key-padding mask
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
        # lamp/Decoders.py:136-138: with a vector encoder there is nothing to pad-mask
        pad_mask, keep = (None, None) if self.enc_vec else N.key_token_mask(src_seq[:, :T], T)
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
    """Binary-relevance baseline (lamp/Decoders.py:76-93): Linear -> ReLU -> Dropout -> Linear over the encoder's
    vector; returns ((B, 1, n_tgt_vocab),).  Plain PyTorch, parameter names as in the reference."""

    def __init__(self, n_tgt_vocab, n_max_seq_e, n_max_seq_d, n_layers=6, n_head=8, d_k=64, d_v=64, d_word_vec=512,
                 d_model=512, d_inner_hid=1024, dropout=0.1, enc_transform='mean'):
        super().__init__()
        self.n_max_seq = n_max_seq_e
        self.d_model = d_model
        self.dropout = nn.Dropout(dropout)
        self.enc_transform = enc_transform
        if enc_transform in ['flatten']:
            raise NotImplementedError
        self.linear1 = nn.Linear(d_model, d_model)
        self.linear4 = nn.Linear(d_model, n_tgt_vocab)

    def forward(self, tgt_seq, src_seq, enc_output, return_attns=False, int_preds=False):
        batch_size = src_seq.size(0)
        out1 = self.dropout(F.relu(self.linear1(enc_output.float())))
        return self.linear4(out1).view(batch_size, 1, -1),


def _dot_attention(q, k, v, temperature, blocked):
    """lamp/SubLayers.py:27-43 in plain PyTorch (eval-mode dropout = identity is applied by the caller's module):
    the RNN decoder's one-query attention over the encoder states."""
    attn = torch.bmm(q, k.transpose(1, 2)) / temperature
    if blocked is not None:
        attn = attn.masked_fill(blocked.bool(), float('-inf'))
    attn = torch.softmax(attn, dim=2)
    return attn


class _PlainAttention(nn.Module):
    """ScaledDotProductAttention as the RNN decoder holds it (no parameters; temperature = d_model as the reference
    passes it, lamp/Decoders.py:30), differentiable PyTorch so that train.py works for the baseline too."""

    def __init__(self, temperature, dropout=0.1):
        super().__init__()
        self.temperature = temperature
        self.dropout = nn.Dropout(dropout)
        self.attn_type = nn.Softmax(dim=2)

    def forward(self, q, k, v, attn_mask=None, stop_sig=False):
        attn = self.dropout(_dot_attention(q, k, v, self.temperature, attn_mask))
        return torch.bmm(attn, v), attn


class RNNDecoder(nn.Module):
    """Autoregressive GRU baseline with attention over the encoder states (lamp/Decoders.py:16-72): every step feeds
    the arg-max label of the previous one; returns ((B, len, n_tgt_vocab),).  Plain PyTorch."""

    def __init__(self, n_tgt_vocab, n_max_seq, n_layers=6, n_head=8, d_k=64, d_v=64, d_word_vec=512, d_model=512,
                 d_inner_hid=1024, dropout=0.1):
        super().__init__()
        self.n_max_seq = n_max_seq
        self.d_model = d_model
        self.n_tgt_vocab = n_tgt_vocab
        self.tgt_word_emb = nn.Embedding(n_tgt_vocab, d_word_vec, padding_idx=Constants.PAD)
        self.dropout = nn.Dropout(dropout)
        self.attention_stack = nn.ModuleList([_PlainAttention(d_model, dropout=dropout) for _ in range(n_layers)])
        self.rnn_layer_stack = nn.ModuleList(
            [nn.GRU(d_model + d_word_vec, d_model, batch_first=True, dropout=dropout) for _ in range(n_layers)])
        self.U = nn.Linear(self.d_model, self.n_tgt_vocab)
        self.V = nn.Linear(self.d_model, self.n_tgt_vocab)
        self.C = nn.Linear(self.d_model, self.n_tgt_vocab)

    def forward_step(self, input_var, decoder_hidden, encoder_outputs, dec_enc_attn_pad_mask=None):
        batch_size = input_var.size(0)
        embedded = self.tgt_word_emb(input_var)
        decoder_hidden = decoder_hidden.view(batch_size, 1, -1)
        if encoder_outputs.size(1) == 1:
            dec_enc_attn_pad_mask = None
        for idx, dec_layer in enumerate(self.rnn_layer_stack):
            context, attn = self.attention_stack[idx](decoder_hidden.view(batch_size, 1, -1), encoder_outputs,
                                                      encoder_outputs, dec_enc_attn_pad_mask)
            rnn_input = torch.cat((embedded, context), 2)
            embedded, decoder_hidden = dec_layer(rnn_input, decoder_hidden.view(1, batch_size, -1).contiguous())
        output = self.U(decoder_hidden)
        output = output + self.V(embedded.view(batch_size, -1))
        output = output + self.C(context.view(batch_size, -1))
        return output, decoder_hidden, attn

    def forward(self, tgt_seq, src_seq, enc_output, return_attns=False, int_preds=False):
        batch_size = enc_output.size(0)
        pad_mask = utils.get_attn_padding_mask(tgt_seq, src_seq, unsqueeze=False)
        dec_output = torch.zeros(tgt_seq.size(0), tgt_seq.size(1), self.n_tgt_vocab, device=enc_output.device)
        dec_input = tgt_seq[:, 0].unsqueeze(1)
        decoder_hidden = enc_output.mean(1)
        for di in range(tgt_seq.size(1)):
            decoder_output, decoder_hidden, _ = self.forward_step(dec_input, decoder_hidden, enc_output, pad_mask)
            dec_output[:, di, :] = decoder_output
            dec_input = F.log_softmax(decoder_output.view(batch_size, -1), dim=1).topk(1)[1].view(batch_size, -1)
        return dec_output,
