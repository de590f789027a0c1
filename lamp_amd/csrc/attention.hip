// Fused masked attention  O = softmax_k(mask(Q K^T * inv_temperature)) V  in exact fp32 on the
// CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).  Scores never leave the CU.
//
// Work decomposition: grid = (ceil(lq/128), H, B); a 256-thread workgroup owns 128 query rows of
// one (sample, head); each of its four 64-lane waves owns 32 of them.  Keys/values stream through
// LDS in tiles of 32 keys, register-staged and double buffered (the next tile's global loads are
// in flight under the current tile's MFMAs; one barrier per tile).
//
// Both products are computed TRANSPOSED so that the query index lands on the lane (MFMA C/D
// column = lane & 31) in both accumulators:
//     S^T[key][query] = K . Q^T      A = K rows (from LDS), B = Q rows (registers, pre-scaled)
//     O^T[dv ][query] = V^T . P^T    A = V columns (from LDS), B = P (the S^T accumulator itself)
// Lane (q = l&31, hi = l>>5) then holds, for ITS query, the 16 keys {(r&3)+8(r>>2)+4hi} of the
// tile.  Row max / row sum are 15 in-lane ops plus ONE exchange with lane l^32; the online-softmax
// rescale of O^T is a plain per-lane multiply; and -- because an MFMA may take its k index in any
// order as long as A and B agree -- accumulator register r of S^T is DIRECTLY the B operand of PV
// step r (key (r&3)+8(r>>2)+4hi for both operands).  P never moves: no LDS round trip, no
// permutes, no bf16 repack, exact fp32 throughout.
//
// Masking follows the reference: blocked scores become -inf BEFORE the softmax; a fully blocked
// row therefore has zero row-sum and comes out NaN (0 * inf), exactly like torch's
// softmax(-inf, ..., -inf) -- lamp/SubLayers.py:31-39, SURVEY.md G10.
#include "lamp_kernels.h"

namespace lamp {

__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

template <int DP, bool WRITE_P>
__global__ __launch_bounds__(256) void attn_kernel(AttnParams p) {
    constexpr int DKC = DP / 8;   // 8-wide k-dim chunks of Q.K
    constexpr int DVB = DP / 32;  // 32-wide column blocks of O
    constexpr int KS = DP + 4;    // K tile row stride: padded -> conflict-free ds_read_b128
    constexpr int VS = DP;        // V tile row stride (b32 reads of 32 consecutive columns)
    constexpr int LD = DP / 32;   // float4 loads per thread per 32xDP tile (256 threads)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                // [2][32][KS]
    float* Vs = smem + 2 * 32 * KS;  // [2][32][VS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int qi = q0 + l31;
    const bool wave_active = q0 < p.lq;  // wave-uniform
    const int qc = qi < p.lq ? qi : p.lq - 1;  // clamped row for mask reads

    const float* __restrict__ Kg = p.K + int64_t(b) * p.lay.k_b + int64_t(h) * p.lay.k_h;
    const float* __restrict__ Vg = p.V + int64_t(b) * p.lay.v_b + int64_t(h) * p.lay.v_h;

    // ---- Q fragments: lane holds Q[qi][8c + 4hi .. +3], pre-multiplied by scale*log2(e) ----
    float4 qf[DKC];
    {
        const float* Qrow = p.Q + int64_t(b) * p.lay.q_b + int64_t(h) * p.lay.q_h + int64_t(qc) * p.lay.q_r;
#pragma unroll
        for (int c = 0; c < DKC; ++c) {
            const int kd = c * 8 + hi * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qi < p.lq && kd < p.dk) v = *reinterpret_cast<const float4*>(Qrow + kd);
            qf[c] = make_float4(v.x * p.scale_log2e, v.y * p.scale_log2e, v.z * p.scale_log2e,
                                v.w * p.scale_log2e);
        }
    }

    const int nt = (p.lk + 31) / 32;
    float4 rk[LD], rv[LD];
    constexpr int C4 = DP / 4;

    auto gload = [&](int kt, bool with_v) {
#pragma unroll
        for (int i = 0; i < LD; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / C4, c4 = idx - row * C4;
            const int key = kt * 32 + row;
            const int c = c4 * 4;
            rk[i] = (key < p.lk && c < p.dk)
                        ? *reinterpret_cast<const float4*>(Kg + int64_t(key) * p.lay.k_r + c)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
            if (with_v)
                rv[i] = (key < p.lk && c < p.dv)
                            ? *reinterpret_cast<const float4*>(Vg + int64_t(key) * p.lay.v_r + c)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](int buf, bool with_v) {
#pragma unroll
        for (int i = 0; i < LD; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / C4, c4 = idx - row * C4;
            *reinterpret_cast<float4*>(Ks + buf * 32 * KS + row * KS + c4 * 4) = rk[i];
            if (with_v) *reinterpret_cast<float4*>(Vs + buf * 32 * VS + row * VS + c4 * 4) = rv[i];
        }
    };

    // S^T tile for this wave's 32 queries vs keys [kt*32, kt*32+32), masked, in the log2 domain.
    auto scores = [&](int kt, int buf, f32x16& s) {
        // mask bytes first, so their latency hides under the MFMAs
        unsigned blocked = 0;  // bit r set = blocked
        const int kbase = kt * 32 + 4 * hi;
        if (p.mask_kind == LAMP_MASK_U8) {
            const unsigned char* mrow = static_cast<const unsigned char*>(p.mask) + int64_t(b) * p.m_sb +
                                        int64_t(qc) * p.m_sq;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2);
                if (key < p.lk && mrow[key] != 0) blocked |= 1u << r;
            }
        } else if (p.mask_kind == LAMP_MASK_KEY_TOKENS_I64) {
            const long long* trow = static_cast<const long long*>(p.mask) + int64_t(b) * p.m_sb;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2);
                if (key < p.lk && trow[key] == 0) blocked |= 1u << r;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* kp = Ks + buf * 32 * KS + l31 * KS + hi * 4;
#pragma unroll
        for (int c = 0; c < DKC; ++c) {
            const float4 kf = *reinterpret_cast<const float4*>(kp + c * 8);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[c].x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[c].y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[c].z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[c].w, s, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kbase + (r & 3) + 8 * (r >> 2);
            if (key >= p.lk || ((blocked >> r) & 1u)) s[r] = -INFINITY;
        }
    };

    auto pv = [&](int buf, const f32x16& pr, f32x16 (&o)[DVB]) {
        const float* vp = Vs + buf * 32 * VS + (4 * hi) * VS + l31;
#pragma unroll
        for (int cb = 0; cb < DVB; ++cb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float vf = vp[((r & 3) + 8 * (r >> 2)) * VS + cb * 32];
                o[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, pr[r], o[cb], 0, 0, 0);
            }
        }
    };

    f32x16 o[DVB];
#pragma unroll
    for (int cb = 0; cb < DVB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[cb][r] = 0.f;

    float m_run = -INFINITY, l_run = 0.f;

    if constexpr (WRITE_P) {
        // ---- pass 1: exact row max and row sum (K only) ----
        gload(0, false);
        lstore(0, false);
        __syncthreads();
        for (int kt = 0; kt < nt; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nt) gload(kt + 1, false);
            if (wave_active) {
                f32x16 s;
                scores(kt, buf, s);
                float tmax = s[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
                tmax = fmaxf(tmax, xor32(tmax));
                const float m_new = fmaxf(m_run, tmax);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) psum += exp2f(s[r] - m_use);
                psum += xor32(psum);
                l_run = l_run * exp2f(m_run - m_use) + psum;
                m_run = m_new;
            }
            if (kt + 1 < nt) lstore(buf ^ 1, false);
            __syncthreads();
        }
        // ---- pass 2: normalised probabilities out, and O = P V ----
        const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
        const float inv_l = 1.0f / l_run;  // l == 0 (fully blocked row) -> inf -> P, O = NaN
        float* Prow = p.P + (int64_t(h) * p.B + b) * int64_t(p.lq) * p.lk + int64_t(qc) * p.lk;
        const bool with_v = p.O != nullptr;  // uniform: false = probabilities only
        gload(0, with_v);
        lstore(0, with_v);
        __syncthreads();
        for (int kt = 0; kt < nt; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nt) gload(kt + 1, with_v);
            if (wave_active) {
                f32x16 s;
                scores(kt, buf, s);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] = exp2f(s[r] - m_use) * inv_l;
                    const int key = kt * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
                    if (qi < p.lq && key < p.lk) Prow[key] = s[r];
                    // keys beyond lk hold exp2(-inf) * inv_l: 0, or NaN for a dead row -- their V rows are 0
                    if (key >= p.lk) s[r] = 0.f;
                }
                if (with_v) pv(buf, s, o);
            }
            if (kt + 1 < nt) lstore(buf ^ 1, with_v);
            __syncthreads();
        }
        if (l_run == 0.f) {
            // fully blocked row: the reference gets NaN from softmax; 0-padded keys must not hide it
#pragma unroll
            for (int cb = 0; cb < DVB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[cb][r] = __builtin_nanf("");
        }
    } else {
        // ---- single pass, online softmax ----
        gload(0, true);
        lstore(0, true);
        __syncthreads();
        for (int kt = 0; kt < nt; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nt) gload(kt + 1, true);
            if (wave_active) {
                f32x16 s;
                scores(kt, buf, s);
                float tmax = s[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
                tmax = fmaxf(tmax, xor32(tmax));
                const float m_new = fmaxf(m_run, tmax);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = exp2f(m_run - m_use);  // m_run = -inf -> 0
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] = exp2f(s[r] - m_use);
                    psum += s[r];
                }
                psum += xor32(psum);
                l_run = l_run * alpha + psum;
                m_run = m_new;
#pragma unroll
                for (int cb = 0; cb < DVB; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[cb][r] *= alpha;
                pv(buf, s, o);
            }
            if (kt + 1 < nt) lstore(buf ^ 1, true);
            __syncthreads();
        }
        const float inv_l = 1.0f / l_run;  // 0 -> inf; O is 0 there -> NaN, as the reference
#pragma unroll
        for (int cb = 0; cb < DVB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[cb][r] *= inv_l;
    }

    // ---- store O[b, qi, h, :]: lane owns columns 32cb + 8g + 4hi + {0..3} = registers 4g..4g+3 ----
    if (wave_active && qi < p.lq && p.O != nullptr) {
        float* Orow = p.O + int64_t(b) * p.lay.o_b + int64_t(h) * p.lay.o_h + int64_t(qi) * p.lay.o_r;
        const bool vec = ((p.lay.o_b | p.lay.o_h | p.lay.o_r) & 3) == 0 &&
                         (reinterpret_cast<uintptr_t>(p.O) & 15u) == 0;
#pragma unroll
        for (int cb = 0; cb < DVB; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = cb * 32 + g * 8 + hi * 4;
                if (col >= p.dv) continue;  // dv is a multiple of 4
                if (vec) {
                    *reinterpret_cast<float4*>(Orow + col) =
                        make_float4(o[cb][4 * g], o[cb][4 * g + 1], o[cb][4 * g + 2], o[cb][4 * g + 3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) Orow[col + e] = o[cb][4 * g + e];
                }
            }
    }
}

template <int DP, bool WRITE_P>
static int launch_attn_cfg(const AttnParams& p, hipStream_t s) {
    constexpr size_t lds = size_t(2) * 32 * ((DP + 4) + DP) * sizeof(float);
    auto kern = attn_kernel<DP, WRITE_P>;
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (lds > 65536 && dev >= 0 && dev < 64 && !attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return int(e);
        attr_done[dev] = true;
    }
    dim3 grid((p.lq + 127) / 128, p.H, p.B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    return int(hipGetLastError());
}

int launch_attn(const AttnParams& p, hipStream_t s) {
    if (p.B <= 0 || p.H <= 0 || p.lq <= 0 || p.lk <= 0 || p.dk <= 0 || p.dv <= 0) return LAMP_E_DIMS;
    if (p.B > 65535 || p.H > 65535) return LAMP_E_UNSUPPORTED;
    if (p.dk > 128 || p.dv > 128) return LAMP_E_UNSUPPORTED;
    if ((p.dk & 3) || (p.dv & 3)) return LAMP_E_UNSUPPORTED;
    if (!p.Q || !p.K) return LAMP_E_NULL;
    if ((!p.V || !p.O) && !(p.P && !p.V && !p.O)) return LAMP_E_NULL;  // V, O optional only with P
    if (p.mask_kind != LAMP_MASK_NONE && !p.mask) return LAMP_E_NULL;
    const lamp_attn_layout& L = p.lay;
    if (((L.q_b | L.q_h | L.q_r | L.k_b | L.k_h | L.k_r | L.v_b | L.v_h | L.v_r) & 3) ||
        !aligned16(p.Q) || !aligned16(p.K) || (p.V && !aligned16(p.V)))
        return LAMP_E_ALIGN;
    const double flops = 2.0 * p.B * p.H * double(p.lq) * p.lk * (p.dk + p.dv) * (p.P ? 1.5 : 1.0);
    const double bytes = 4.0 * p.B * p.H * (double(p.lq) * (p.dk + p.dv) + double(p.lk) * (p.dk + p.dv)) +
                         (p.P ? 4.0 * p.B * p.H * double(p.lq) * p.lk : 0.0);
    ProfScope prof(LAMP_K_ATTN, flops, bytes, s);
    const int dmax = p.dk > p.dv ? p.dk : p.dv;
    const bool wp = p.P != nullptr;
    if (dmax <= 32) return wp ? launch_attn_cfg<32, true>(p, s) : launch_attn_cfg<32, false>(p, s);
    if (dmax <= 64) return wp ? launch_attn_cfg<64, true>(p, s) : launch_attn_cfg<64, false>(p, s);
    return wp ? launch_attn_cfg<128, true>(p, s) : launch_attn_cfg<128, false>(p, s);
}

}  // namespace lamp
