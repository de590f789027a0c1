// Attention for head widths beyond the fused kernel's 128 (d_k or d_v up to any multiple of 4): the reference's own
// three steps (lamp/SubLayers.py:27-43) as three launches -- S = Q K^T / temperature (lamp_gemm, batched over
// (head, sample) straight on the fused [B, l, h*d] projections), masked softmax in place, O = P V (lamp_gemm).  The
// (H*B, lq, lk) score tensor is materialised: in the caller's map buffer when maps are requested, else in workspace
// scratch.  Slower than the fused kernel (scores go through memory); it exists so that any n_head the reference
// accepts runs (d_model 1024 with 4 heads -> d_k = 256).
#include "lamp_kernels.h"

namespace lamp {

namespace {

__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wsum2(float v) { return wave64_sum(v); }

struct SoftmaxParams {
    float* P;        // [(H * P_batch), lq, lk]; rows of sample (P_b0 + b), head h at (h * P_batch + P_b0 + b) * lq
    int B, H, lq, lk, P_batch, P_b0;
    const void* mask;
    int64_t m_sb, m_sq;
};

// One wave per (head, sample, query) row: blocked entries -> -inf, softmax over the keys, in place.
// A fully blocked row gives exp(-inf - -inf) = NaN, as torch's masked_fill + softmax does.
template <int MK>
__global__ __launch_bounds__(256) void masked_softmax_kernel(SoftmaxParams p) {
    const int lane = threadIdx.x & 63;
    const int64_t rr = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const int64_t rows = int64_t(p.H) * p.B * p.lq;
    if (rr >= rows) return;
    const int q = int(rr % p.lq);
    const int64_t hb = rr / p.lq;
    const int b = int(hb % p.B), h = int(hb / p.B);
    float* row = p.P + ((int64_t(h) * p.P_batch + p.P_b0 + b) * p.lq + q) * int64_t(p.lk);
    auto blocked = [&](int k) -> bool {
        if constexpr (MK == LAMP_MASK_U8)
            return static_cast<const unsigned char*>(p.mask)[int64_t(b) * p.m_sb + int64_t(q) * p.m_sq + k] != 0;
        else if constexpr (MK == LAMP_MASK_KEY_TOKENS_I64)
            return static_cast<const long long*>(p.mask)[int64_t(b) * p.m_sb + k] == 0;
        else if constexpr (MK == LAMP_MASK_BITS_U32)
            return (static_cast<const unsigned*>(p.mask)[int64_t(b) * p.m_sb + int64_t(q) * p.m_sq + (k >> 5)] >> (k & 31)) & 1u;
        else
            return false;
    };
    float m = -INFINITY;
    for (int k = lane; k < p.lk; k += 64) {
        const float s = blocked(k) ? -INFINITY : row[k];
        row[k] = s;
        m = fmaxf(m, s);
    }
    m = wmax(m);
    float l = 0.f;
    for (int k = lane; k < p.lk; k += 64) {
        const float e = expf(row[k] - m);  // NaN for every k when the whole row is blocked
        row[k] = e;
        l += e;
    }
    l = wsum2(l);
    const float inv = 1.0f / l;
    for (int k = lane; k < p.lk; k += 64) row[k] *= inv;
}

}  // namespace

int launch_attn_general(const AttnParams& p, hipStream_t s) {
    float* S = p.P ? p.P : p.scratch;
    if (!S) return LAMP_E_WORKSPACE;  // d_k / d_v > 128 needs the map buffer or scratch for the scores
    const int P_batch = p.P ? p.P_batch : p.B, P_b0 = p.P ? p.P_b0 : 0;
    const int64_t map = int64_t(p.lq) * p.lk;
    if ((p.dk & 3) || (p.dv & 3)) return LAMP_E_UNSUPPORTED;

    lamp_gemm_desc g{};
    g.A = p.Q; g.B = p.K; g.C = S + int64_t(P_b0) * map;
    g.M = p.lq; g.N = p.lk; g.K = p.dk;
    g.batch0 = p.H; g.batch1 = p.B;
    g.a_row_stride = p.lay.q_r; g.a_col_stride = 1; g.a_batch0 = p.lay.q_h; g.a_batch1 = p.lay.q_b;
    g.b_row_stride = p.lay.k_r; g.b_col_stride = 1; g.b_batch0 = p.lay.k_h; g.b_batch1 = p.lay.k_b;
    g.ldc = p.lk; g.c_batch0 = int64_t(P_batch) * map; g.c_batch1 = map;
    g.alpha = p.scale_log2e * 0.6931471805599453f;  // 1 / temperature
    if (int e = launch_gemm_gen(g, nullptr, 0, s)) return e;

    SoftmaxParams sp{S, p.B, p.H, p.lq, p.lk, P_batch, P_b0, p.mask, p.m_sb, p.m_sq};
    const int64_t rows = int64_t(p.H) * p.B * p.lq;
    const int64_t grid = (rows + 3) / 4;
    if (grid > 0x7fffffffLL) return LAMP_E_DIMS;
    switch (p.mask_kind) {
        case LAMP_MASK_U8:
            hipLaunchKernelGGL(masked_softmax_kernel<LAMP_MASK_U8>, dim3(unsigned(grid)), dim3(256), 0, s, sp);
            break;
        case LAMP_MASK_KEY_TOKENS_I64:
            hipLaunchKernelGGL(masked_softmax_kernel<LAMP_MASK_KEY_TOKENS_I64>, dim3(unsigned(grid)), dim3(256), 0, s, sp);
            break;
        case LAMP_MASK_BITS_U32:
            hipLaunchKernelGGL(masked_softmax_kernel<LAMP_MASK_BITS_U32>, dim3(unsigned(grid)), dim3(256), 0, s, sp);
            break;
        default:
            hipLaunchKernelGGL(masked_softmax_kernel<LAMP_MASK_NONE>, dim3(unsigned(grid)), dim3(256), 0, s, sp);
            break;
    }
    if (int e = int(hipGetLastError())) return e;
    if (!p.V || !p.O) return 0;  // maps only (the reference's dead encoder self-attention)

    lamp_gemm_desc o{};
    o.A = S + int64_t(P_b0) * map; o.B = p.V; o.C = p.O;
    o.M = p.lq; o.N = p.dv; o.K = p.lk;
    o.batch0 = p.H; o.batch1 = p.B;
    o.a_row_stride = p.lk; o.a_col_stride = 1; o.a_batch0 = int64_t(P_batch) * map; o.a_batch1 = map;
    o.b_row_stride = 1; o.b_col_stride = p.lay.v_r; o.b_batch0 = p.lay.v_h; o.b_batch1 = p.lay.v_b;  // B(n, k) = V[k][n]
    o.ldc = p.lay.o_r; o.c_batch0 = p.lay.o_h; o.c_batch1 = p.lay.o_b;
    o.alpha = 1.0f;
    return launch_gemm_gen(o, nullptr, 0, s);
}

}  // namespace lamp
