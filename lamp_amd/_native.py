"""ctypes binding of liblamp_hip.so -- the only route from the Python modules to the GPU.

There is no fallback: if the shared library is missing or a tensor is not on a HIP device the
call raises.  PyTorch is used for device memory (``data_ptr()``), the current HIP stream and
nothing else.  Signatures mirror include/lamp_hip.h one to one.
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# LAMP_HIP_LIBRARY: load another build of the same library (tools/bench_kernels.py runs on the -DLAMP_TUNING build)
LIB_PATH = os.environ.get('LAMP_HIP_LIBRARY') or os.path.join(_HERE, 'liblamp_hip.so')
TUNING_LIB_PATH = os.path.join(_HERE, 'liblamp_hip_tuning.so')

ABI_VERSION = 5
LAMP_MASK_NONE, LAMP_MASK_U8, LAMP_MASK_KEY_TOKENS_I64, LAMP_MASK_BITS_U32 = 0, 1, 2, 3
LAMP_MASK_SPARSE_ROWS = 1   # lamp_mask.flags (include/lamp_hip.h)
K_EMBED, K_GEMM, K_ATTN, K_LAYERNORM, K_DIAG, K_COUNT = 0, 1, 2, 3, 4, 5
KERNEL_CLASS_NAMES = ('embed', 'gemm', 'attention', 'layernorm', 'diag_readout')


class LampError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = lib().lamp_strerror(status).decode() if _lib is not None else '?'
        super().__init__('%s failed with status %d: %s' % (where, status, msg))


_f32p = C.POINTER(C.c_float)
_vp = C.c_void_p


class Mask(C.Structure):
    _fields_ = [('kind', C.c_int32), ('flags', C.c_int32), ('ptr', _vp),
                ('stride_b', C.c_int64), ('stride_q', C.c_int64),
                ('tile_list', _vp), ('tile_list_stride', C.c_int64), ('allowed_pairs', C.c_int64)]


class AttnLayout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ('q_b', 'q_h', 'q_r', 'k_b', 'k_h', 'k_r',
                                         'v_b', 'v_h', 'v_r', 'o_b', 'o_h', 'o_r')]


class MhaWeights(C.Structure):
    _fields_ = [('w_qs', _vp), ('w_ks', _vp), ('w_vs', _vp), ('fc', _vp), ('ln_g', _vp), ('ln_b', _vp),
                ('n_head', C.c_int32), ('present', C.c_int32)]


class FfnWeights(C.Structure):
    _fields_ = [('w1', _vp), ('b1', _vp), ('w2', _vp), ('b2', _vp), ('ln_g', _vp), ('ln_b', _vp)]


class EncLayer(C.Structure):
    _fields_ = [('slf_attn', MhaWeights), ('pos_ffn', FfnWeights)]


class DecLayer(C.Structure):
    _fields_ = [('enc_attn', MhaWeights), ('pos_ffn1', FfnWeights), ('slf_attn', MhaWeights),
                ('pos_ffn2', FfnWeights)]


class ChainPack(C.Structure):  # include/lamp_hip.h: lamp_chain_pack
    _fields_ = [('fc', _vp), ('w1', _vp), ('w2', _vp), ('fc4', _vp), ('w14', _vp), ('w24', _vp)]


class Model(C.Structure):
    _fields_ = [('n_src_vocab', C.c_int32), ('n_position', C.c_int32), ('n_labels', C.c_int32),
                ('d_model', C.c_int32), ('d_inner', C.c_int32), ('d_k', C.c_int32), ('d_v', C.c_int32),
                ('n_layers_enc', C.c_int32), ('n_layers_dec', C.c_int32), ('label_mask_flags', C.c_int32),
                ('src_word_emb', _vp), ('position_enc', _vp), ('tgt_word_emb', _vp), ('w_out', _vp),
                ('label_mask', _vp), ('label_mask_bits', _vp), ('label_tiles', _vp), ('enc_layers', C.POINTER(EncLayer)), ('dec_layers', C.POINTER(DecLayer)),
                ('dec0_query', _vp), ('chain_packs', C.POINTER(ChainPack)),
                ('enc0_emb_w1', _vp), ('enc0_pos_w1', _vp), ('label_mask_allowed', C.c_int64)]


class GemmDesc(C.Structure):  # include/lamp_hip.h: lamp_gemm_desc
    _fields_ = [('A', _vp), ('B', _vp), ('C', _vp), ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32),
                ('batch0', C.c_int32), ('batch1', C.c_int32), ('accumulate', C.c_int32),
                ('a_row_stride', C.c_int64), ('a_col_stride', C.c_int64), ('a_batch0', C.c_int64), ('a_batch1', C.c_int64),
                ('b_row_stride', C.c_int64), ('b_col_stride', C.c_int64), ('b_batch0', C.c_int64), ('b_batch1', C.c_int64),
                ('ldc', C.c_int64), ('c_batch0', C.c_int64), ('c_batch1', C.c_int64),
                ('relu_mask', _vp), ('ld_mask', C.c_int64), ('alpha', C.c_float), ('reserved', C.c_int32)]


class ReduceJob(C.Structure):  # include/lamp_hip.h: lamp_reduce_job
    _fields_ = [('partial', _vp), ('n_total', C.c_int64), ('n_seg', C.c_int64), ('out', _vp * 3),
                ('n_partials', C.c_int32), ('reserved', C.c_int32)]


class MhaTrainDesc(C.Structure):  # include/lamp_hip.h: lamp_mha_train_desc
    _fields_ = [('B', C.c_int32), ('lq', C.c_int32), ('lk', C.c_int32), ('d_model', C.c_int32), ('n_head', C.c_int32),
                ('d_k', C.c_int32), ('d_v', C.c_int32), ('inv_temperature', C.c_float), ('p_attn', C.c_float),
                ('p_out', C.c_float), ('seed_attn', C.c_uint32), ('seed_out', C.c_uint32)]


class Aux(C.Structure):
    _fields_ = [('enc_self_attn', C.POINTER(_vp)), ('dec_self_attn', C.POINTER(_vp)),
                ('dec_enc_attn', C.POINTER(_vp)), ('int_preds', C.POINTER(_vp)),
                ('n_int_preds', C.c_int32), ('reserved', C.c_int32)]


# name -> (restype, argtypes); every function include/lamp_hip.h declares
_i32, _i64, _sz, _f = C.c_int32, C.c_int64, C.c_size_t, C.c_float
PROTOTYPES = {
    'lamp_version': (C.c_int, []),
    'lamp_strerror': (C.c_char_p, [C.c_int]),
    'lamp_linear_fwd': (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i32, _i64, _vp, _vp, _i64, _i32, _vp, _i64, _vp]),
    'lamp_layernorm_fwd': (C.c_int, [_vp, _i64, _i32, _vp, _vp, _f, _vp, _vp]),
    'lamp_sdpa_fwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f,
                                C.POINTER(Mask), C.POINTER(AttnLayout), _vp]),
    'lamp_sdpa_fwd_fast_maps': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f,
                                          C.POINTER(Mask), C.POINTER(AttnLayout), _vp]),
    'lamp_mha_workspace_bytes': (_sz, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    'lamp_mha_fwd': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(MhaWeights),
                               C.POINTER(Mask), _vp, _vp, _vp, _sz, _vp]),
    'lamp_ffn_workspace_bytes': (_sz, [_i64, _i32, _i32]),
    'lamp_ffn_fwd': (C.c_int, [_vp, _i64, _i32, _i32, C.POINTER(FfnWeights), _vp, _vp, _sz, _vp]),
    'lamp_embed_fwd': (C.c_int, [_vp, _vp, _i64, _vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    'lamp_diag_logits_fwd': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    'lamp_pack_weight': (C.c_int, [_vp, _i32, _i32, _i64, _i32, _vp, _vp]),
    'lamp_gemm_workspace_bytes': (_sz, [_i32, _i32, _i32, _i32]),
    'lamp_gemm': (C.c_int, [C.POINTER(GemmDesc), _vp, _sz, _vp]),
    'lamp_gemm_grouped': (C.c_int, [C.POINTER(GemmDesc), _i32, _vp]),
    'lamp_ffn_train_fwd': (C.c_int, [_vp, _i64, _i32, _i32, C.POINTER(FfnWeights), _f, C.c_uint32, _vp, _vp, _vp, _vp]),
    'lamp_reduce_partials_grouped': (C.c_int, [C.POINTER(ReduceJob), _i32, _vp]),
    'lamp_ffn_bwd_workspace_bytes': (_sz, [_i64, _i32, _i32]),
    'lamp_ffn_bwd_partials_bytes': (_sz, [_i64, _i32, _i32]),
    'lamp_ffn_bwd': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, C.POINTER(FfnWeights), _f, C.c_uint32] + [_vp] * 9 +
                     [_vp, _sz, _vp, _sz, C.POINTER(ReduceJob), _vp]),
    'lamp_mha_train_fwd': (C.c_int, [C.POINTER(MhaTrainDesc), C.POINTER(MhaWeights), _vp, _vp, _vp, C.POINTER(Mask)] +
                           [_vp] * 9 + [_vp]),
    'lamp_mha_bwd_workspace_bytes': (_sz, [C.POINTER(MhaTrainDesc)]),
    'lamp_mha_bwd_partials_bytes': (_sz, [C.POINTER(MhaTrainDesc)]),
    'lamp_mha_bwd': (C.c_int, [C.POINTER(MhaTrainDesc), C.POINTER(MhaWeights)] + [_vp] * 11 + [_vp] * 15 +
                     [_vp, _sz, _vp, _sz, C.POINTER(ReduceJob), _vp]),
    'lamp_layernorm_residual_fwd': (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _f, _f, C.c_uint32, _vp, _vp]),
    'lamp_layernorm_bwd_workspace_bytes': (_sz, [_i64, _i32]),
    'lamp_layernorm_bwd': (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp, _f, _f, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _sz, _vp]),
    'lamp_colsum_workspace_bytes': (_sz, [_i64, _i64]),
    'lamp_colsum': (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    'lamp_dropout': (C.c_int, [_vp, _i64, _f, C.c_uint32, _vp, _vp]),
    'lamp_softmax_bwd': (C.c_int, [_vp, _vp, _i64, _i32, _f, _vp, _vp]),
    'lamp_diag_logits_bwd': (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    'lamp_embed_bwd': (C.c_int, [_vp, _i64, _vp, _i32, _i32, _i64, _vp, _vp]),
    'lamp_prior_graph_build': (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    'lamp_sigmoid_bce_fwd': (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    'lamp_forward_workspace_bytes': (_sz, [C.POINTER(Model), _i32, _i32, _i32]),
    'lamp_forward': (C.c_int, [C.POINTER(Model), _vp, _vp, _i32, _i32, _vp, _vp, C.POINTER(Aux), _vp, _sz, _vp]),
    'lamp_prof_enable': (C.c_int, [_i32]),
    'lamp_prof_reset': (C.c_int, []),
    'lamp_prof_read': (C.c_int, [_i32, C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                 C.POINTER(C.c_double)]),
}

_lib = None
_lock = threading.Lock()


def load_library(path):
    """dlopen one build of the library and attach the header's prototypes.  Raises when it is absent."""
    if not os.path.exists(path):
        raise RuntimeError(
            'lamp_amd: %s is missing -- build it with `python -m lamp_amd.build` '
            '(hipcc, gfx950).  There is no CPU or PyTorch fallback.' % path)
    handle = C.CDLL(path)
    handle.lamp_version.restype = C.c_int
    version = handle.lamp_version()
    # A/B runs against a build of an earlier round (tools/build_variant.sh): ABI 2 and 3 are prefixes of ABI 4 -- lamp_model
    # grew one trailing member an older library never reads, lamp_pack_weight / the training composites are absent
    # (weight_pack() then returns None)
    old_ok = version in (2, 3) and os.environ.get('LAMP_ALLOW_OLD_ABI') == '1'
    if version != ABI_VERSION and not old_ok:
        raise RuntimeError('lamp_amd: ABI version mismatch in ' + path)
    for name, (res, args) in PROTOTYPES.items():
        if old_ok and not hasattr(handle, name):
            continue
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args
    return handle


def lib():
    """Load liblamp_hip.so once.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                _lib = load_library(LIB_PATH)
    return _lib


def check(status, where):
    if status != 0:
        raise LampError(status, where)


# ------------------------------------------------------------------ tensor plumbing
def require_device(*tensors):
    """Every tensor lives on a HIP device, and on the CURRENT one: launches go to the current device's stream
    (stream() below), so a tensor of another GPU would be read by kernels enqueued on the wrong device."""
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('lamp_amd runs on an MI355X HIP device only; got a %s tensor (%s). '
                               'There is no CPU path.' % (t.device, tuple(t.shape)))
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise RuntimeError('lamp_amd: tensor on %s but the current device is cuda:%d -- wrap the call in '
                               '`with torch.cuda.device(tensor.device):` (LAMP.forward does so itself)' % (t.device, cur))


def f32c(t):
    """fp32, contiguous view/copy of t (device resident)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def ptr(t):
    return 0 if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream():
    """Raw hipStream_t of torch's current stream on the current device (what every launch is enqueued on).
    torch.cuda.current_stream() builds a Python Stream object per call (~10 us); the raw getter is ~0.5 us."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_ws = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer per (device, HIP stream): forwards issued on different streams may be in
    flight at the same time and must not share scratch (the library itself never allocates)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), stream())
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def make_mask(mask, B, lq, lk):
    """Torch bool/uint8 mask broadcastable to (B, lq, lk) -> (Mask struct, keepalive tensor)."""
    if mask is None:
        return None, None
    require_device(mask)
    m = mask
    if m.dtype == torch.bool:
        m = m.view(torch.uint8) if m.is_contiguous() or m.stride(-1) == 1 else m.contiguous().view(torch.uint8)
    elif m.dtype != torch.uint8:
        m = (m != 0).view(torch.uint8)
    if m.dim() == 2:
        m = m.unsqueeze(0)
    if m.dim() != 3 or m.size(-1) != lk or m.size(1) not in (1, lq) or m.size(0) not in (1, B):
        raise ValueError('attention mask of shape %s does not broadcast to (%d, %d, %d)' %
                         (tuple(mask.shape), B, lq, lk))
    if m.stride(-1) != 1 and lk > 1:
        m = m.contiguous()
    sb = 0 if m.size(0) == 1 else m.stride(0)
    sq = 0 if m.size(1) == 1 else m.stride(1)
    return Mask(LAMP_MASK_U8, 0, m.data_ptr(), sb, sq, None, 0), m


def pack_mask_bits(blocked_u8):
    """(lq, lk) uint8 mask -> int32 [lq, ceil(lk/32)] rows, bit (k & 31) of word (k >> 5) = blocked
    (LAMP_MASK_BITS_U32).  Host-side, once per mask."""
    m = (blocked_u8.cpu() != 0)
    lq, lk = m.shape
    nw = (lk + 31) // 32
    pad = torch.zeros((lq, nw * 32), dtype=torch.int64)
    pad[:, :lk] = m.to(torch.int64)
    weights = (torch.ones(32, dtype=torch.int64) << torch.arange(32, dtype=torch.int64))
    words = (pad.view(lq, nw, 32) * weights).sum(dim=2)             # 0 .. 2^32 - 1
    words = torch.where(words >= 2 ** 31, words - 2 ** 32, words)    # reinterpret as int32
    return words.to(torch.int32).contiguous()


def active_tile_list(blocked_u8):
    """Sparsity hint for a shared (lq, lk) uint8 mask: int32 [ceil(lq/32), ceil(lk/32) + 1] rows
    [count, tile_0, tile_1, ...] of the 32-key tiles holding at least one unblocked entry (host-side, built
    once per mask)."""
    m = (blocked_u8.cpu() == 0)
    lq, lk = m.shape
    nq, nk = (lq + 31) // 32, (lk + 31) // 32
    pad = torch.zeros((nq * 32, nk * 32), dtype=torch.bool)
    pad[:lq, :lk] = m
    act = pad.view(nq, 32, nk, 32).any(dim=3).any(dim=1)        # (nq, nk)
    out = torch.zeros((nq, nk + 1), dtype=torch.int32)
    for i in range(nq):
        idx = act[i].nonzero().flatten().to(torch.int32)
        out[i, 0] = idx.numel()
        out[i, 1:1 + idx.numel()] = idx
    return out


def key_token_mask(src_seq, T):
    """Key-padding mask straight from the int64 token ids (blocked iff token == PAD == 0)."""
    require_device(src_seq)
    s = src_seq if src_seq.dtype == torch.int64 else src_seq.long()
    if s.stride(-1) != 1:
        s = s.contiguous()
    return Mask(LAMP_MASK_KEY_TOKENS_I64, 0, s.data_ptr(), s.stride(0), 0, None, 0), s


# ------------------------------------------------------------------ thin wrappers
def linear(x, weight, bias=None, residual=None, relu=False, _lib=None):
    require_device(x, weight, bias, residual)
    x2 = f32c(x).reshape(-1, x.size(-1))
    w = f32c(weight).reshape(weight.size(0), -1)
    M, K = x2.shape
    N = w.size(0)
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    r = f32c(residual).reshape(M, N) if residual is not None else None
    b = f32c(bias) if bias is not None else None
    check((_lib or lib()).lamp_linear_fwd(ptr(x2), M, K, K, ptr(w), N, K, ptr(b), ptr(r), N, int(relu), ptr(out), N,
                                stream()), 'lamp_linear_fwd')
    return out.view(*x.shape[:-1], N)


def layernorm(x, gamma, beta, eps=1e-5):
    require_device(x, gamma, beta)
    x2 = f32c(x).reshape(-1, x.size(-1))
    out = torch.empty_like(x2)
    check(lib().lamp_layernorm_fwd(ptr(x2), x2.size(0), x2.size(1), ptr(f32c(gamma)), ptr(f32c(beta)),
                                   float(eps), ptr(out), stream()), 'lamp_layernorm_fwd')
    return out.view(x.shape)


def sdpa(q, k, v, mask, inv_temperature, need_attn=True, _lib=None):
    """q (N, lq, dk), k (N, lk, dk), v (N, lk, dv) head-major batches as in the reference."""
    require_device(q, k, v)
    q, k, v = f32c(q), f32c(k), f32c(v)
    N, lq, dk = q.shape
    lk, dv = k.size(1), v.size(2)
    out = torch.empty((N, lq, dv), dtype=torch.float32, device=q.device)
    wide = dk > 128 or dv > 128  # the general path keeps its scores in the map buffer
    attn = torch.empty((N, lq, lk), dtype=torch.float32, device=q.device) if (need_attn or wide) else None
    mstruct, keep = make_mask(mask, N, lq, lk)
    lay = AttnLayout(lq * dk, 0, dk, lk * dk, 0, dk, lk * dv, 0, dv, lq * dv, 0, dv)
    check((_lib or lib()).lamp_sdpa_fwd(ptr(q), ptr(k), ptr(v), ptr(out), ptr(attn), N, 1, lq, lk, dk, dv,
                              float(inv_temperature), C.byref(mstruct) if mstruct is not None else None,
                              C.byref(lay), stream()), 'lamp_sdpa_fwd')
    del keep
    return out, (attn if need_attn else None)


def sdpa_fused(q, k, v, n_head, mask_struct, inv_temperature, need_attn=True, fast_maps=False):
    """Attention on head-fused projections: q (B, lq, H*dk), k (B, lk, H*dk), v (B, lk, H*dv) ->
    out (B, lq, H*dv), attn (H*B, lq, lk) [index head*B + b] or None.  No head split/merge copies: the
    kernel walks the heads through lamp_attn_layout strides.  fast_maps: single-pass map write-out
    (lamp_sdpa_fwd_fast_maps) instead of the exact two-pass variant."""
    require_device(q, k, v)
    q, k, v = f32c(q), f32c(k), f32c(v)
    B, lq, hq = q.shape
    lk, H = k.size(1), n_head
    dk, dv = hq // H, v.size(2) // H
    out = torch.empty((B, lq, H * dv), dtype=torch.float32, device=q.device)
    wide = dk > 128 or dv > 128  # the general path keeps its scores in the map buffer
    attn = torch.empty((H * B, lq, lk), dtype=torch.float32, device=q.device) if (need_attn or wide) else None
    lay = AttnLayout(lq * H * dk, dk, H * dk, lk * H * dk, dk, H * dk, lk * H * dv, dv, H * dv, lq * H * dv, dv, H * dv)
    m = C.byref(mask_struct) if mask_struct is not None else None
    if fast_maps and need_attn:
        lse = torch.empty((H * B * lq,), dtype=torch.float32, device=q.device)
        check(lib().lamp_sdpa_fwd_fast_maps(ptr(q), ptr(k), ptr(v), ptr(out), ptr(attn), ptr(lse), B, H, lq, lk, dk, dv,
                                            float(inv_temperature), m, C.byref(lay), stream()), 'lamp_sdpa_fwd_fast_maps')
        return out, attn
    check(lib().lamp_sdpa_fwd(ptr(q), ptr(k), ptr(v), ptr(out), ptr(attn), B, H, lq, lk, dk, dv,
                              float(inv_temperature), m, C.byref(lay), stream()), 'lamp_sdpa_fwd')
    return out, (attn if need_attn else None)


def mha_weights(mod):
    """MhaWeights struct for a lamp_amd MultiHeadAttention module (pointers into its parameters)."""
    fc = getattr(mod, 'fc', None)
    return MhaWeights(ptr(mod.w_qs.weight), ptr(mod.w_ks.weight), ptr(mod.w_vs.weight),
                      ptr(fc.weight) if fc is not None else 0, ptr(mod.layer_norm.weight),
                      ptr(mod.layer_norm.bias), mod.n_head, 1)


def ffn_weights(mod):
    return FfnWeights(ptr(mod.w_1.weight), ptr(mod.w_1.bias), ptr(mod.w_2.weight), ptr(mod.w_2.bias),
                      ptr(mod.layer_norm.weight), ptr(mod.layer_norm.bias))


def mha(xq, xkv, weights, d_k, d_v, mask_struct, need_attn):
    require_device(xq, xkv)
    xq, xkv = f32c(xq), f32c(xkv)
    B, lq, d = xq.shape
    lk = xkv.size(1)
    h = weights.n_head
    out = torch.empty_like(xq)
    attn = torch.empty((h * B, lq, lk), dtype=torch.float32, device=xq.device) if need_attn else None
    nbytes = lib().lamp_mha_workspace_bytes(B, lq, lk, d, h, d_k, d_v)
    ws = workspace(nbytes, xq.device)
    check(lib().lamp_mha_fwd(ptr(xq), ptr(xkv), B, lq, lk, d, d_k, d_v, C.byref(weights),
                             C.byref(mask_struct) if mask_struct is not None else None, ptr(out), ptr(attn),
                             ptr(ws), ws.numel(), stream()), 'lamp_mha_fwd')
    return out, attn


def ffn(x, weights, d_inner):
    require_device(x)
    x2 = f32c(x).reshape(-1, x.size(-1))
    M, d = x2.shape
    out = torch.empty_like(x2)
    nbytes = lib().lamp_ffn_workspace_bytes(M, d, d_inner)
    ws = workspace(nbytes, x.device)
    check(lib().lamp_ffn_fwd(ptr(x2), M, d, d_inner, C.byref(weights), ptr(out), ptr(ws), ws.numel(),
                             stream()), 'lamp_ffn_fwd')
    return out.view(x.shape)


def embed(src_seq, src_pos, emb, pos_table):
    require_device(src_seq, src_pos, emb, pos_table)
    seq = src_seq.long().contiguous()
    pos = src_pos.long().contiguous() if pos_table is not None else None
    d = emb.size(1)
    out = torch.empty(tuple(seq.shape) + (d,), dtype=torch.float32, device=emb.device)
    check(lib().lamp_embed_fwd(ptr(seq), ptr(pos), seq.numel(), ptr(f32c(emb)), emb.size(0),
                               ptr(f32c(pos_table)) if pos_table is not None else 0,
                               pos_table.size(0) if pos_table is not None else 0, d, ptr(out), stream()),
          'lamp_embed_fwd')
    return out


def weight_pack(w, fmt=0):
    """lamp_pack_weight: a [N, K] weight (Conv1d(k=1) weights [N, K, 1] are read as [N, K]) in the order the fused decoder
    chain streams it (format 0: 16x16x4 fragments, format 1: 64-column panels of the 4x4x1 kernel), or None when the shape
    has no packed form or the library predates it."""
    require_device(w)
    w = f32c(w.detach())
    n, k = w.size(0), w.size(1)
    fn = getattr(lib(), 'lamp_pack_weight', None)
    if fn is None or (n % 16 or k % 32 if fmt == 0 else n % 64 or k % 16):
        return None
    out = torch.empty(n * k, dtype=torch.float32, device=w.device)
    check(fn(ptr(w), n, k, k, fmt, ptr(out), stream()), 'lamp_pack_weight')
    return out


def diag_logits(y, w_out):
    require_device(y, w_out)
    y = f32c(y)
    B, L, d = y.shape
    out = torch.empty((B, L), dtype=torch.float32, device=y.device)
    check(lib().lamp_diag_logits_fwd(ptr(y), ptr(f32c(w_out)), B, L, d, ptr(out), stream()),
          'lamp_diag_logits_fwd')
    return out


def _dims4(t):
    """shape and strides of a (..., r, c) tensor padded to (b0, b1, r, c) -- plain ints, no view objects."""
    sh, st = tuple(t.shape), t.stride()
    n = len(sh)
    if n == 2:
        return (1, 1) + sh, (0, 0) + st
    if n == 3:
        return (1,) + sh, (0,) + st
    if n == 4:
        return sh, st
    raise ValueError('gemm operands must be 2-, 3- or 4-dimensional')


def _operand(t):
    """-> (tensor to keep alive, shape4, row stride, col stride, batch strides) with one unit stride among (row, col)."""
    sh, st = _dims4(t)
    rs, cs = st[2], st[3]
    if sh[3] == 1 and rs != 1:   # a size-1 dim may carry any stride
        cs = 1
    elif sh[2] == 1 and cs != 1:
        rs = 1
    if rs != 1 and cs != 1:
        t = t.contiguous()
        sh, st = _dims4(t)
        rs, cs = st[2], st[3]
    return t, sh, rs, cs, (0 if sh[0] == 1 else st[0], 0 if sh[1] == 1 else st[1])


def matmul_nt(a, b, out=None, alpha=1.0, accumulate=False, relu_mask=None):
    """out[..., m, n] (+)= alpha * sum_k a[..., m, k] * b[..., n, k]   (lamp_gemm).

    a and b are fp32 device VIEWS: any strides with a unit stride on one of the last two dims, so transposes
    (`w.t()`, `p.transpose(-1, -2)`), head-split views of [B, l, h*d] projections etc. are passed as they are.
    Batch dims (0, 1 or 2 of them) must match or be 1.  out, if given, is a view with unit last stride."""
    if not (a.is_cuda and b.is_cuda):
        require_device(a, b)
    if a.dtype != torch.float32 or b.dtype != torch.float32:
        raise TypeError('matmul_nt needs float32 operands')
    A, ash, ars, acs, abs_ = _operand(a)
    Bm, bsh, brs, bcs, bbs = _operand(b)
    M, K, Nn = ash[2], ash[3], bsh[2]
    if bsh[3] != K:
        raise ValueError('contraction sizes differ: %s vs %s' % (tuple(a.shape), tuple(b.shape)))
    b0, b1 = max(ash[0], bsh[0]), max(ash[1], bsh[1])
    if ash[0] not in (1, b0) or ash[1] not in (1, b1) or bsh[0] not in (1, b0) or bsh[1] not in (1, b1):
        raise ValueError('batch dims do not broadcast: %s vs %s' % (tuple(a.shape), tuple(b.shape)))
    if out is None:
        if accumulate:
            raise ValueError('accumulate needs an out tensor')
        lead = a.shape[:-2] if a.dim() >= b.dim() else b.shape[:-2]
        out = torch.empty(tuple(lead) + (M, Nn), dtype=torch.float32, device=a.device)
    csh, cst = _dims4(out)
    if cst[3] != 1 and csh[3] != 1:
        raise ValueError('out must have unit stride on its last dim')
    if csh != (b0, b1, M, Nn):
        raise ValueError('out has shape %s, expected %s' % (tuple(out.shape), (b0, b1, M, Nn)))
    d = GemmDesc(A.data_ptr(), Bm.data_ptr(), out.data_ptr(), M, Nn, K, b0, b1, 1 if accumulate else 0,
                 ars, acs, abs_[0], abs_[1], brs, bcs, bbs[0], bbs[1],
                 cst[2], 0 if csh[0] == 1 else cst[0], 0 if csh[1] == 1 else cst[1], None, 0, float(alpha), 0)
    keep = None
    if relu_mask is not None:
        keep = f32c(relu_mask)
        if tuple(keep.shape[-2:]) != (M, Nn) or keep.numel() != M * Nn:
            raise ValueError('relu_mask must be (M, N)')
        d.relu_mask, d.ld_mask = keep.data_ptr(), Nn
    L = lib()
    nb = L.lamp_gemm_workspace_bytes(M, Nn, K, b0 * b1)
    ws = workspace(nb, a.device) if nb else None
    check(L.lamp_gemm(C.byref(d), ptr(ws), nb, stream()), 'lamp_gemm')
    return out


def matmul_nt_grouped(problems):
    """[(a, b, out, accumulate)] -> every out[m, n] (+)= sum_k a[m, k] * b[n, k] in ONE launch per operand form
    (lamp_gemm_grouped): 2-D fp32 device views as in matmul_nt, distinct outs.  Deepest K first inside a launch."""
    groups = {}
    for a, b, out, accumulate in problems:
        if not (a.is_cuda and b.is_cuda and out.is_cuda):
            require_device(a, b, out)
        if a.dtype != torch.float32 or b.dtype != torch.float32 or out.dtype != torch.float32:
            raise TypeError('matmul_nt_grouped needs float32 operands')
        if a.dim() != 2 or b.dim() != 2 or out.dim() != 2:
            raise ValueError('matmul_nt_grouped takes 2-D operands')
        A, ash, ars, acs, _ = _operand(a)
        Bm, bsh, brs, bcs, _ = _operand(b)
        M, K, Nn = ash[2], ash[3], bsh[2]
        if bsh[3] != K or tuple(out.shape) != (M, Nn) or (out.stride(1) != 1 and Nn != 1):
            raise ValueError('shapes: a %s, b %s, out %s' % (tuple(a.shape), tuple(b.shape), tuple(out.shape)))
        d = GemmDesc(A.data_ptr(), Bm.data_ptr(), out.data_ptr(), M, Nn, K, 1, 1, 1 if accumulate else 0,
                     ars, acs, 0, 0, brs, bcs, 0, 0, out.stride(0), 0, 0, None, 0, 1.0, 0)
        form = (ars == 1 and acs != 1, brs == 1 and bcs != 1)
        groups.setdefault(form, []).append((K, d, A, Bm))
    L, st = lib(), stream()
    for form, items in groups.items():
        items.sort(key=lambda it: -it[0])
        arr = (GemmDesc * len(items))(*[it[1] for it in items])
        check(L.lamp_gemm_grouped(arr, len(items), st), 'lamp_gemm_grouped')


def wgrad_grouped(problems):
    """[(dy2 (rows, out), x2 (rows, in), grad (out, in), accumulate)] -> every grad (+)= dy2^T x2 in one lamp_gemm_grouped launch:
    matmul_nt_grouped for the one operand form a weight gradient has (both operands read transposed), without the view objects
    and stride analysis of the general wrapper -- this runs 28 times per training step on the issuing thread."""
    items = []
    for dy2, x2, out, accumulate in problems:
        if (dy2.dim() != 2 or x2.dim() != 2 or dy2.stride(1) != 1 or x2.stride(1) != 1 or out.stride(1) != 1 or
                dy2.dtype != torch.float32 or x2.dtype != torch.float32 or out.dtype != torch.float32 or
                not (dy2.is_cuda and x2.is_cuda and out.is_cuda) or dy2.size(1) == 1 or x2.size(1) == 1):
            return matmul_nt_grouped([(a.t(), b.t(), o, acc) for a, b, o, acc in problems])
        K, M = dy2.shape
        Nn = x2.size(1)
        if x2.size(0) != K or tuple(out.shape) != (M, Nn):
            raise ValueError('shapes: dy %s, x %s, grad %s' % (tuple(dy2.shape), tuple(x2.shape), tuple(out.shape)))
        items.append((K, GemmDesc(dy2.data_ptr(), x2.data_ptr(), out.data_ptr(), M, Nn, K, 1, 1, 1 if accumulate else 0,
                                  1, dy2.stride(0), 0, 0, 1, x2.stride(0), 0, 0, out.stride(0), 0, 0, None, 0, 1.0, 0)))
    if items:
        items.sort(key=lambda it: -it[0])
        arr = (GemmDesc * len(items))(*[it[1] for it in items])
        check(lib().lamp_gemm_grouped(arr, len(items), stream()), 'lamp_gemm_grouped')


def _dp(t):
    """data pointer of a contiguous fp32 device tensor (or 0): the composite calls below take no copies."""
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise TypeError('contiguous float32 device tensor expected, got %s %s' % (t.dtype, tuple(t.shape)))
    return t.data_ptr()


def ffn_train_fwd(x2, w1, b1, w2, b2, ln_g, ln_b, p, seed):
    """lamp_ffn_train_fwd: x2 (M, d) -> (h, o, y); see include/lamp_hip.h."""
    M, d = x2.shape
    dff = w1.size(0)
    e = torch.empty
    h = e((M, dff), dtype=torch.float32, device=x2.device)
    o = e((M, d), dtype=torch.float32, device=x2.device)
    y = e((M, d), dtype=torch.float32, device=x2.device)
    wts = FfnWeights(_dp(w1), _dp(b1), _dp(w2), _dp(b2), _dp(ln_g), _dp(ln_b))
    check(lib().lamp_ffn_train_fwd(_dp(x2), M, d, dff, C.byref(wts), float(p), int(seed) & 0xffffffff, h.data_ptr(),
                                   o.data_ptr(), y.data_ptr(), stream()), 'lamp_ffn_train_fwd')
    return h, o, y


def reduce_partials_grouped(jobs):
    """lamp_reduce_partials_grouped over a list of ReduceJob (their buffers kept alive by the caller)."""
    if jobs:
        arr = (ReduceJob * len(jobs))(*jobs)
        check(lib().lamp_reduce_partials_grouped(arr, len(jobs), stream()), 'lamp_reduce_partials_grouped')


def ffn_bwd(x2, h, o, dy, w1, w2, ln_g, p, seed, want_dw1, want_dw2, defer_reduce=False):
    """lamp_ffn_bwd -> (dx, d_o, dh, dW1 | None, dW2 | None, db1, db2, dgamma, dbeta, pending); d_o is dx when p == 0.
    defer_reduce: db1 / db2 / dgamma / dbeta are finished by reduce_partials_grouped(pending[0]) later; pending[1] keeps the
    partial sums alive until then.  pending is None otherwise."""
    M, d = x2.shape
    dff = w1.size(0)
    dev = x2.device
    e = torch.empty
    dx = e((M, d), dtype=torch.float32, device=dev)
    d_o = e((M, d), dtype=torch.float32, device=dev) if p > 0 else None
    dh = e((M, dff), dtype=torch.float32, device=dev)
    dW1 = e((dff, d), dtype=torch.float32, device=dev) if want_dw1 else None
    dW2 = e((d, dff), dtype=torch.float32, device=dev) if want_dw2 else None
    vec = e((2 * dff + 3 * d,), dtype=torch.float32, device=dev)   # db1 | db2 | dgamma | dbeta (one allocation)
    db1, db2, dg, db = vec[:dff], vec[dff:dff + d], vec[dff + d:dff + 2 * d], vec[dff + 2 * d:dff + 3 * d]
    L = lib()
    nb = L.lamp_ffn_bwd_workspace_bytes(M, d, dff)
    ws = workspace(nb, dev)
    wts = FfnWeights(_dp(w1), None, _dp(w2), None, _dp(ln_g), None)
    pending, part, npb, jobs = None, None, 0, None
    if defer_reduce:
        npb = L.lamp_ffn_bwd_partials_bytes(M, d, dff)
        part = e((npb,), dtype=torch.uint8, device=dev)
        jobs = (ReduceJob * 2)()
    check(L.lamp_ffn_bwd(_dp(x2), _dp(h), _dp(o), _dp(dy), M, d, dff, C.byref(wts), float(p), int(seed) & 0xffffffff,
                         dx.data_ptr(), _dp(d_o), dh.data_ptr(), _dp(dW1), _dp(dW2), db1.data_ptr(), db2.data_ptr(),
                         dg.data_ptr(), db.data_ptr(), ptr(ws), nb, ptr(part), npb, jobs, stream()), 'lamp_ffn_bwd')
    if defer_reduce:
        pending = ([jobs[0], jobs[1]], (part, vec))
    return dx, (d_o if d_o is not None else dx), dh, dW1, dW2, db1, db2, dg, db, pending


def mha_train_fwd(desc, xq, xk, xv, wq, wk, wv, fc, ln_g, ln_b, mask_struct):
    """lamp_mha_train_fwd -> (q, k, v, a, P, Pd | None, o | None, y)."""
    B, lq, lk, d, H, dk, dv = desc.B, desc.lq, desc.lk, desc.d_model, desc.n_head, desc.d_k, desc.d_v
    dev = xq.device
    e = torch.empty
    q = e((B, lq, H * dk), dtype=torch.float32, device=dev)
    k = e((B, lk, H * dk), dtype=torch.float32, device=dev)
    v = e((B, lk, H * dv), dtype=torch.float32, device=dev)
    a = e((B, lq, H * dv), dtype=torch.float32, device=dev)
    P = e((H * B, lq, lk), dtype=torch.float32, device=dev)
    Pd = e((H * B, lq, lk), dtype=torch.float32, device=dev) if desc.p_attn > 0 else None
    lse = e((H * B * lq,), dtype=torch.float32, device=dev)
    o = e((B * lq, d), dtype=torch.float32, device=dev) if fc is not None else None
    y = e((B, lq, d), dtype=torch.float32, device=dev)
    wts = MhaWeights(_dp(wq), _dp(wk), _dp(wv), _dp(fc), _dp(ln_g), _dp(ln_b), H, 1)
    m = C.byref(mask_struct) if mask_struct is not None else None
    check(lib().lamp_mha_train_fwd(C.byref(desc), C.byref(wts), _dp(xq), _dp(xk), _dp(xv), m, q.data_ptr(), k.data_ptr(),
                                   v.data_ptr(), a.data_ptr(), P.data_ptr(), _dp(Pd), lse.data_ptr(), _dp(o), y.data_ptr(),
                                   stream()), 'lamp_mha_train_fwd')
    return q, k, v, a, P, Pd, o, y


def mha_bwd(desc, xq, xk, xv, q, k, v, a, P, Pd, o, dy, wq, wk, wv, fc, ln_g, separate_value_source, want_dw, want_dfc,
            defer_reduce=False, shared_qk=False):
    """lamp_mha_bwd -> dict of gradients and of the buffers deferred weight gradients are computed from; shared_qk: xq and xk
    are one tensor (self-attention) -- r['dxq'] is then its whole gradient and r['dxk'] None; with defer_reduce,
    r['pending'] = ([job], buffers to keep alive): dgamma / dbeta are finished by reduce_partials_grouped later."""
    B, lq, lk, d, H, dk, dv = desc.B, desc.lq, desc.lk, desc.d_model, desc.n_head, desc.d_k, desc.d_v
    dev = xq.device
    e = torch.empty
    f = dict(dtype=torch.float32, device=dev)
    r = {}
    r['dxq'] = e((B * lq, d), **f)
    r['d_o'] = e((B * lq, d), **f) if desc.p_out > 0 else None
    da = e((B * lq, H * dv), **f) if fc is not None else None
    dP = e((H * B, lq, lk), **f)
    r['dq'], r['dk'], r['dv'] = e((B * lq, H * dk), **f), e((B * lk, H * dk), **f), e((B * lk, H * dv), **f)
    r['dxk'] = r['dxq'] if shared_qk else e((B * lk, d), **f)
    r['dxv'] = e((B * lk, d), **f) if separate_value_source else None
    vec = e((2 * d,), **f)
    r['dgamma'], r['dbeta'] = vec[:d], vec[d:]
    r['dwq'] = e((H * dk, d), **f) if want_dw else None
    r['dwk'] = e((H * dk, d), **f) if want_dw else None
    r['dwv'] = e((H * dv, d), **f) if want_dw else None
    r['dfc'] = e((d, H * dv), **f) if (want_dfc and fc is not None) else None
    L = lib()
    nb = L.lamp_mha_bwd_workspace_bytes(C.byref(desc))
    ws = workspace(nb, dev)
    wts = MhaWeights(_dp(wq), _dp(wk), _dp(wv), _dp(fc), _dp(ln_g), None, H, 1)
    part, npb, job = None, 0, None
    if defer_reduce:
        npb = L.lamp_mha_bwd_partials_bytes(C.byref(desc))
        part = e((npb,), dtype=torch.uint8, device=dev)
        job = (ReduceJob * 1)()
    check(L.lamp_mha_bwd(C.byref(desc), C.byref(wts), _dp(xq), _dp(xk), _dp(xv), _dp(q), _dp(k), _dp(v), _dp(a), _dp(P),
                         _dp(Pd), _dp(o), _dp(dy), r['dxq'].data_ptr(), _dp(r['d_o']), _dp(da), dP.data_ptr(),
                         r['dq'].data_ptr(), r['dk'].data_ptr(), r['dv'].data_ptr(), r['dxk'].data_ptr(), _dp(r['dxv']),
                         r['dgamma'].data_ptr(), r['dbeta'].data_ptr(), _dp(r['dwq']), _dp(r['dwk']), _dp(r['dwv']),
                         _dp(r['dfc']), ptr(ws), nb, ptr(part), npb, job, stream()), 'lamp_mha_bwd')
    r['pending'] = ([job[0]], (part, vec)) if defer_reduce else None
    if shared_qk:   # r['dxq'] is the gradient of the one tensor behind xq and xk
        r['dxk'] = None
    if r['d_o'] is None:
        r['d_o'] = r['dxq']
    return r


def layernorm_residual(x, residual, gamma, beta, eps=1e-5, dropout_p=0.0, seed=0):
    """LayerNorm(dropout(x) + residual); residual (rows, d) is broadcast over x's rows when it has fewer (row % rows).
    dropout_p > 0 applies lamp_dropout's counter-based mask for `seed` to x inside the kernel."""
    require_device(x, gamma, beta, residual)
    xc = f32c(x)
    d = xc.size(-1)
    M = xc.numel() // d
    r = f32c(residual) if residual is not None else None
    r_rows = 0
    if r is not None:
        rr = r.numel() // d
        if rr != M:
            if M % rr:
                raise ValueError('residual rows do not tile x rows')
            r_rows = rr
    y = torch.empty_like(xc)
    check(lib().lamp_layernorm_residual_fwd(ptr(xc), ptr(r), r_rows, M, d, ptr(f32c(gamma)), ptr(f32c(beta)), eps,
                                            float(dropout_p), int(seed) & 0xffffffff, ptr(y), stream()),
          'lamp_layernorm_residual_fwd')
    return y


def layernorm_bwd(x, residual, gamma, dy, eps=1e-5, dropout_p=0.0, seed=0, want_dbias=False):
    """Backward of y = LayerNorm(dropout(x) + residual) -> (dz, dx, dgamma, dbeta, dbias):
    dz = gradient of the residual branch, dx = gradient of x (the same tensor as dz when dropout_p == 0),
    dbias = column sums of dx (None unless want_dbias)."""
    require_device(x, gamma, dy, residual)
    xc, g = f32c(x), f32c(dy)
    d = xc.size(-1)
    M = xc.numel() // d
    r = f32c(residual) if residual is not None else None
    r_rows = 0
    if r is not None and r.numel() // d != M:
        r_rows = r.numel() // d
    dz = torch.empty_like(xc)
    dx = torch.empty_like(xc) if dropout_p > 0 else None
    dgamma = torch.empty(d, dtype=torch.float32, device=xc.device)
    dbeta = torch.empty_like(dgamma)
    dbias = torch.empty_like(dgamma) if want_dbias else None
    nb = lib().lamp_layernorm_bwd_workspace_bytes(M, d)
    ws = workspace(nb, xc.device)
    check(lib().lamp_layernorm_bwd(ptr(xc), ptr(r), r_rows, M, d, ptr(f32c(gamma)), eps, float(dropout_p),
                                   int(seed) & 0xffffffff, ptr(g), ptr(dz), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(dbias),
                                   ptr(ws), nb, stream()), 'lamp_layernorm_bwd')
    return dz, (dx if dx is not None else dz), dgamma, dbeta, dbias


def colsum(x):
    """(M, N) -> (N,): sum over rows (fixed order)."""
    require_device(x)
    xc = f32c(x)
    M, Nn = xc.shape
    out = torch.empty(Nn, dtype=torch.float32, device=xc.device)
    nb = lib().lamp_colsum_workspace_bytes(M, Nn)
    ws = workspace(nb, xc.device)
    check(lib().lamp_colsum(ptr(xc), M, Nn, Nn, ptr(out), ptr(ws), nb, stream()), 'lamp_colsum')
    return out


def dropout(x, p, seed, out=None):
    """Counter-based dropout (see include/lamp_hip.h: lamp_dropout); the same call on a gradient is its backward."""
    require_device(x)
    xc = f32c(x)
    y = torch.empty_like(xc) if out is None else out
    check(lib().lamp_dropout(ptr(xc), xc.numel(), float(p), int(seed) & 0xffffffff, ptr(y), stream()), 'lamp_dropout')
    return y


def dropout_keep_mask(n, p, seed, device='cpu'):
    """The library's keep mask for elements 0..n-1, restated with torch integer ops (tests / documentation)."""
    e = torch.arange(n, dtype=torch.int64, device=device)
    m32 = 0xffffffff
    h = ((e & m32) ^ (((e >> 32) * 0x9E3779B9) & m32) ^ (int(seed) & m32)) & m32
    h = h ^ (h >> 16)
    h = (h * 0x7feb352d) & m32
    h = h ^ (h >> 15)
    h = (h * 0x846ca68b) & m32
    h = h ^ (h >> 16)
    thr = min(int(float(torch.tensor(p, dtype=torch.float32)) * 4294967296.0), 4294967295)
    return h >= thr


def softmax_bwd(P, dP, scale, out=None):
    require_device(P, dP)
    p, g = f32c(P), f32c(dP)
    lk = p.size(-1)
    o = torch.empty_like(p) if out is None else out
    check(lib().lamp_softmax_bwd(ptr(p), ptr(g), p.numel() // lk, lk, float(scale), ptr(o), stream()), 'lamp_softmax_bwd')
    return o


def diag_logits_bwd(y, w_out, dlogits):
    require_device(y, w_out, dlogits)
    yc, w, g = f32c(y), f32c(w_out), f32c(dlogits)
    B, L, d = yc.shape
    dy, dw = torch.empty_like(yc), torch.empty_like(w)
    check(lib().lamp_diag_logits_bwd(ptr(yc), ptr(w), ptr(g), B, L, d, ptr(dy), ptr(dw), stream()), 'lamp_diag_logits_bwd')
    return dy, dw


def embed_bwd(src_seq, dout, n_vocab, pad_idx=-1):
    require_device(src_seq, dout)
    g = f32c(dout)
    d = g.size(-1)
    seq = src_seq.contiguous()
    d_emb = torch.zeros(n_vocab, d, dtype=torch.float32, device=g.device)
    check(lib().lamp_embed_bwd(ptr(seq), seq.numel(), ptr(g), d, n_vocab, int(pad_idx), ptr(d_emb), stream()),
          'lamp_embed_bwd')
    return d_emb


def prior_graph(label_ids, offsets, n_labels, want_blocked=False):
    """Co-occurrence label graph on the device (utils/data_loader.py:37-47).  label_ids int64 (nnz,) 0-based,
    offsets int64 (n_samples + 1,), both on the HIP device -> adj float (L, L) [, blocked uint8 (L, L)]."""
    require_device(label_ids, offsets)
    if label_ids.dtype != torch.int64 or offsets.dtype != torch.int64:
        raise TypeError('label_ids and offsets must be int64')
    ids, off = label_ids.contiguous(), offsets.contiguous()
    n_samples = off.numel() - 1
    if n_samples < 0:
        raise ValueError('offsets needs n_samples + 1 entries')
    if ids.numel():
        lo, hi = int(ids.min()), int(ids.max())
        if lo < 0 or hi >= n_labels:
            raise IndexError('label index %d outside [0, %d)' % (lo if lo < 0 else hi, n_labels))
        if int(off[-1]) != ids.numel() or int(off[0]) != 0 or bool((off[1:] < off[:-1]).any()):
            raise ValueError('offsets must rise from 0 to len(label_ids)')
    adj = torch.empty((n_labels, n_labels), dtype=torch.float32, device=ids.device)
    blocked = torch.empty((n_labels, n_labels), dtype=torch.uint8, device=ids.device) if want_blocked else None
    check(lib().lamp_prior_graph_build(ptr(ids), ptr(off), n_samples, n_labels, ptr(adj), ptr(blocked), stream()),
          'lamp_prior_graph_build')
    return (adj, blocked) if want_blocked else adj


def sigmoid_bce(logits, targets=None, probs_out=None, row_loss_out=None):
    """-> (sigmoid(logits), per-row summed BCE-with-logits or None)   (test.py:49-51).  probs_out (B, L) / row_loss_out (B,):
    contiguous fp32 device buffers to write into (an evaluation epoch keeps one of each for the whole split)."""
    require_device(logits, targets, probs_out, row_loss_out)
    x = f32c(logits)
    B, L = x.shape
    for o, shape in ((probs_out, (B, L)), (row_loss_out, (B,))):
        if o is not None and (tuple(o.shape) != shape or o.dtype != torch.float32 or not o.is_contiguous()):
            raise ValueError('sigmoid_bce: output buffer must be contiguous fp32 of shape %s' % (shape,))
    probs = probs_out if probs_out is not None else torch.empty_like(x)
    z = f32c(targets) if targets is not None else None
    row_loss = None
    if z is not None:
        row_loss = row_loss_out if row_loss_out is not None else torch.empty((B,), dtype=torch.float32, device=x.device)
    check(lib().lamp_sigmoid_bce_fwd(ptr(x), ptr(z), B, L, ptr(probs), ptr(row_loss), stream()),
          'lamp_sigmoid_bce_fwd')
    return probs, row_loss


# ------------------------------------------------------------------ profiling
def prof_enable(on=True):
    check(lib().lamp_prof_enable(int(bool(on))), 'lamp_prof_enable')


def prof_reset():
    check(lib().lamp_prof_reset(), 'lamp_prof_reset')


def prof_read():
    """-> {class_name: dict(launches, ms, flops, bytes)} for everything recorded since the last reset."""
    out = {}
    for cls, name in enumerate(KERNEL_CLASS_NAMES):
        n, ms, fl, by = C.c_int64(0), C.c_double(0), C.c_double(0), C.c_double(0)
        check(lib().lamp_prof_read(cls, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)), 'lamp_prof_read')
        out[name] = dict(launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value)
    return out
