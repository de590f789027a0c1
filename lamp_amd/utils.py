"""Host-side helpers of the path (reference: lamp/utils.py).

These run once per model or are pure index bookkeeping; none of them is on the per-batch compute
path (the key-padding test itself happens inside the attention kernel, straight from src_seq).
"""
import numpy as np
import torch

from . import Constants


def position_encoding_init(n_position, d_pos_vec):
    """Sinusoid table, row 0 (PAD position) all zero -- lamp/utils.py:9-19.

    angle[pos, j] = pos / 10000^(2*floor(j/2)/d); sin on even columns, cos on odd columns;
    evaluated in float64 and rounded once to float32 like the reference.
    """
    pos = np.arange(n_position, dtype=np.float64).reshape(-1, 1)
    exponent = 2.0 * (np.arange(d_pos_vec) // 2).astype(np.float64) / d_pos_vec
    table = pos / np.power(10000.0, exponent).reshape(1, -1)
    table[1:, 0::2] = np.sin(table[1:, 0::2])
    table[1:, 1::2] = np.cos(table[1:, 1::2])
    table[0, :] = 0.0
    return torch.from_numpy(table).type(torch.FloatTensor)


def get_attn_padding_mask(seq_q, seq_k, unsqueeze=True):
    """(B, len_q, len_k) bool, True where the KEY is PAD -- lamp/utils.py:26-34.

    Kept for API compatibility.  The fused path never materialises this: it hands seq_k itself to
    the attention kernel (LAMP_MASK_KEY_TOKENS_I64).
    """
    if seq_q.dim() != 2 or seq_k.dim() != 2:
        raise ValueError('expected 2-D index tensors')
    blocked = seq_k.eq(Constants.PAD).unsqueeze(1)
    if unsqueeze:
        blocked = blocked.expand(seq_k.size(0), seq_q.size(1), seq_k.size(1))
    return blocked


def get_attn_subsequent_mask(seq):
    """Strictly-upper-triangular uint8 mask (lamp/utils.py:36-44); unused by the graph decoder."""
    if seq.dim() != 2:
        raise ValueError('expected a 2-D index tensor')
    n = seq.size(1)
    tri = torch.triu(torch.ones((n, n), dtype=torch.uint8, device=seq.device), diagonal=1)
    return tri.unsqueeze(0).expand(seq.size(0), n, n).contiguous()


def swap_0_1(tensor, on_zero, on_non_zero):
    """Map zeros to `on_zero` and everything else to `on_non_zero` (lamp/utils.py:46-50)."""
    zero = tensor == 0
    out = torch.full_like(tensor, on_non_zero)
    out[zero] = on_zero
    return out
