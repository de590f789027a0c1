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

from . import Constants, utils
from . import _native as N
from .Layers import DecoderLayer


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
    TILE_HINT_MAX_DENSITY = 0.9   # active 32x32 tiles / all tiles from which the tile-list hint is not handed to the kernels
    SPARSE_ROWS_MAX_DENSITY = 0.20   # allowed pairs / L^2 up to which an unstructured graph is flagged LAMP_MASK_SPARSE_ROWS

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
        # one edge -- lets the attention kernel skip fully blocked tiles (large sparse / clustered graphs).  The hint is
        # dropped when it cannot help: on an unstructured graph (BASELINE configs[4]'s Bernoulli(0.05) prior: every tile
        # holds an edge) walking the list only costs (-0.7 % attention at L = 4096, profiles/r05_sparse_label_attention.txt).
        tiles = N.active_tile_list(blocked) if blocked is not None else None
        self.label_tile_density = None
        if tiles is not None:
            self.label_tile_density = float(tiles[:, 0].sum()) / float(tiles.size(0) * (tiles.size(1) - 1))
            if self.label_tile_density >= self.TILE_HINT_MAX_DENSITY:
                tiles = None
        self.register_buffer('label_tiles', tiles, persistent=False)
        # ... and an unstructured graph whose ROWS are sparse (configs[4]: ~5 % of the keys per label) is flagged for the pair
        # kernel (csrc/attention_sparse.hip computes the allowed (query, key) pairs only; the library takes it for >= 1024 labels
        # with 128-wide heads).  Measured break-even against the dense tile kernel: profiles/r06_sparse_label_attention.txt.
        self.label_allowed_pairs = int((blocked == 0).sum()) if blocked is not None else 0
        self.label_rows_sparse = bool(blocked is not None and tiles is None and
                                      self.label_allowed_pairs <= self.SPARSE_ROWS_MAX_DENSITY * blocked.numel())
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
            sparse = self.label_rows_sparse and self.use_sparse_rows
            return N.Mask(N.LAMP_MASK_BITS_U32, N.LAMP_MASK_SPARSE_ROWS if sparse else 0, bits.data_ptr(), 0, bits.size(1), *tl,
                          self.label_allowed_pairs if sparse else 0)
        return N.Mask(N.LAMP_MASK_U8, 0, m.data_ptr(), 0, L, *tl)

    use_sparse_rows = True   # A/B switch of the pair kernel (False: the dense tile kernels visit every key)

    def forward(self, tgt, src_seq, enc_output, return_attns=False, int_preds=False):
        B = src_seq.size(0)
        T = enc_output.size(1)
        label_mask = self.label_mask_struct()
        if self.training:
            # module-by-module training (graph decoder over a vector encoder): the label-table broadcast records autograd
            # here, the layers dispatch to lamp_amd/training.py; its map-writing attention visits every key tile
            from . import training
            y = training._LabelRowsFn.apply(self.tgt_word_emb.weight, B)
            if label_mask is not None:
                label_mask = N.Mask(label_mask.kind, 0, label_mask.ptr, label_mask.stride_b, label_mask.stride_q, None, 0, 0)
        else:
            y = self.tgt_word_emb.weight.unsqueeze(0).expand(B, -1, -1).contiguous()
        # lamp/Decoders.py:136-138: with a vector encoder there is nothing to pad-mask
        pad_mask, keep = (None, None) if self.enc_vec else N.key_token_mask(src_seq[:, :T], T)
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
    """Binary-relevance baseline: a two-layer perceptron (hidden width d_model, ReLU, dropout) scoring every label from
    the encoder's single vector per sample; output ``((B, 1, n_tgt_vocab),)``.  Behaviour of lamp/Decoders.py:76-93;
    only the attribute names ``linear1`` / ``linear4`` / ``dropout`` are kept, for checkpoint compatibility.  Plain
    PyTorch: not on the hot path."""

    def __init__(self, n_tgt_vocab, n_max_seq_e, n_max_seq_d, n_layers=6, n_head=8, d_k=64, d_v=64, d_word_vec=512,
                 d_model=512, d_inner_hid=1024, dropout=0.1, enc_transform='mean'):
        super().__init__()
        if enc_transform == 'flatten':
            raise NotImplementedError("the mlp decoder takes one d_model vector per sample, not a flattened sequence")
        self.n_max_seq, self.d_model, self.enc_transform = n_max_seq_e, d_model, enc_transform
        self.linear1 = nn.Linear(d_model, d_model)
        self.dropout = nn.Dropout(dropout)
        self.linear4 = nn.Linear(d_model, n_tgt_vocab)

    def forward(self, tgt_seq, src_seq, enc_output, return_attns=False, int_preds=False):
        scorer = nn.Sequential(self.linear1, nn.ReLU(), self.dropout, self.linear4)
        return (scorer(enc_output.to(torch.float32)).reshape(src_seq.size(0), 1, -1),)


class _MemoryReader(nn.Module):
    """What the reference's parameter-free ScaledDotProductAttention does for the RNN decoder (lamp/SubLayers.py:27-43
    with one query row): a softmax-weighted summary of the encoder states.  Scores are divided by ``temperature``
    (the reference passes d_model itself, lamp/Decoders.py:30), blocked keys get -inf, dropout acts on the weights."""

    def __init__(self, temperature, dropout=0.1):
        super().__init__()
        self.temperature = temperature
        self.dropout = nn.Dropout(dropout)

    def forward(self, query, memory, blocked=None):
        """query (B, d), memory (B, T, d), blocked (B, T) bool or None -> summary (B, d), weights (B, T)."""
        weights = (memory @ query.unsqueeze(2)).squeeze(2) / self.temperature
        if blocked is not None:
            weights = weights.masked_fill(blocked.reshape(weights.shape).bool(), float('-inf'))
        weights = self.dropout(torch.softmax(weights, dim=1))
        return (weights.unsqueeze(1) @ memory).squeeze(1), weights


class RNNDecoder(nn.Module):
    """Autoregressive GRU baseline (behaviour of lamp/Decoders.py:16-72): the recurrent state starts at the mean
    encoder state; every step embeds the previous label, lets each layer read the encoder states with the current
    state as the query, feeds [input, summary] through that layer's one-step GRU, and scores the labels as
    U(state) + V(top layer output) + C(last summary).  The next input is the arg-max label (no teacher forcing).
    Returns ``((B, len, n_tgt_vocab),)``.  Attribute names follow the reference's state_dict; plain PyTorch."""

    def __init__(self, n_tgt_vocab, n_max_seq, n_layers=6, n_head=8, d_k=64, d_v=64, d_word_vec=512, d_model=512,
                 d_inner_hid=1024, dropout=0.1):
        super().__init__()
        self.n_max_seq, self.d_model, self.n_tgt_vocab = n_max_seq, d_model, n_tgt_vocab
        self.tgt_word_emb = nn.Embedding(n_tgt_vocab, d_word_vec, padding_idx=Constants.PAD)
        self.dropout = nn.Dropout(dropout)
        self.attention_stack = nn.ModuleList(_MemoryReader(d_model, dropout) for _ in range(n_layers))
        self.rnn_layer_stack = nn.ModuleList(
            nn.GRU(d_model + d_word_vec, d_model, batch_first=True, dropout=dropout) for _ in range(n_layers))
        self.U, self.V, self.C = (nn.Linear(d_model, n_tgt_vocab) for _ in range(3))

    def step(self, token, state, memory, pad=None):
        """One decoding step.  token (B,) int64, state (B, d), memory (B, T, d), pad (B, T) True on PAD keys
        -> label scores (B, n_tgt_vocab), new state (B, d), the last layer's read weights (B, T)."""
        if memory.size(1) == 1:      # a vector encoder hands over one state per sample: nothing to mask
            pad = None
        feed = self.tgt_word_emb(token.reshape(-1))
        for read, cell in zip(self.attention_stack, self.rnn_layer_stack):
            summary, weights = read(state, memory, pad)
            y, h = cell(torch.cat([feed, summary], dim=1).unsqueeze(1), state.unsqueeze(0).contiguous())
            feed, state = y[:, 0], h[0]
        return self.U(state) + self.V(feed) + self.C(summary), state, weights

    def forward_step(self, token, state, memory, pad=None):
        """The reference's calling convention for ``step`` (its Translator uses it): scores and state come back with
        a leading length-1 axis."""
        scores, state, weights = self.step(token, state.reshape(token.size(0), -1), memory, pad)
        return scores.unsqueeze(0), state.unsqueeze(0), weights.unsqueeze(1)

    def forward(self, tgt_seq, src_seq, enc_output, return_attns=False, int_preds=False):
        pad = src_seq.eq(Constants.PAD)
        token, state = tgt_seq[:, 0], enc_output.mean(dim=1)
        per_step = []
        for _ in range(tgt_seq.size(1)):
            scores, state, _ = self.step(token, state, enc_output, pad)
            per_step.append(scores)
            token = scores.argmax(dim=1)
        return (torch.stack(per_step, dim=1),)
