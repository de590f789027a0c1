// Fused masked attention  O = softmax_k(mask(Q K^T * inv_temperature)) V  in exact fp32 on the
// CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).  Scores never leave the CU.
//
// Work decomposition: a 64-lane wave owns 32 query rows of one (sample, head) and streams 32-key tiles;
// the four waves of a 256-thread workgroup cover QB = 4/KSPLIT query blocks, the KSPLIT waves of one
// block taking interleaved key tiles (flash-decoding style, merged at the end) -- KSPLIT = 2 when there
// are at most 128 queries (reuters' 90 labels would otherwise leave most of the chip idle), else 1.
//
// Both products are computed TRANSPOSED so that the query index lands on the lane (MFMA C/D
// column = lane & 31) in both accumulators:
//     S^T[key][query] = K . Q^T      A = K rows,    B = Q rows (pre-scaled, from LDS)
//     O^T[dv ][query] = V^T . P^T    A = V columns, B = P (the S^T accumulator itself)
// Lane (q = l&31, hi = l>>5) then holds, for ITS query, the 16 keys {(r&3)+8(r>>2)+4hi} of the
// tile.  Row max / row sum are 15 in-lane ops plus ONE exchange with lane l^32; the online-softmax
// rescale of O^T is a plain per-lane multiply; and -- because an MFMA may take its k index in any
// order as long as A and B agree -- accumulator register r of S^T is DIRECTLY the B operand of PV
// step r (key (r&3)+8(r>>2)+4hi for both operands).  P never moves: no LDS round trip, no
// permutes, no bf16 repack, exact fp32 throughout.
//
// There is NO shared K/V tile and NO barrier in the main loop: every wave pulls its own MFMA
// fragments straight from global/L2 into registers through range-checked buffer descriptors --
//   K : lane (key = l&31, hi) reads 16 B at K[key][8c + 4hi], c = 0..DP/8-1
//   V : lane (i = l&31, hi) reads 16 B at V[key_r(hi)][4i .. 4i+3]: block e of O^T then holds the dv
//       columns {4i + e}, which turns the epilogue into 16-byte stores as well
// -- and software-prefetches the next tile's K and mask bytes right after the QK^T MFMAs and the next
// V right after the PV MFMAs, so the loads fly under ~4096 cycles of matrix work each (fp32 MFMA is
// slow enough, 64 cycles per 32x32x2, that the L2 traffic of unsynchronised waves is a non-issue; an
// LDS-tiled variant with a shared K/V tile measured equal or slower on every shape and was removed,
// profiles/r01_attn_variants.txt).  An earlier version lost half its time because hipcc put the wait
// for the mask bytes -- and with it for the whole V prefetch -- in front of the QK^T MFMAs; hence the
// one-tile-ahead mask registers.
//
// Occupancy: the loop is written with ONE K tile and ONE V tile in registers (64 + 64 beside the 64 accumulators), but
// with 512 registers on offer hipcc hoists the next tile's loads above the MFMAs and double-buffers both by itself
// (370-430 registers, one wave per SIMD, whose softmax VALU work then serialises with its own MFMAs).  The two
// sched_barrier(0) in the loop pin the written order; the variants without map output (PM == 0) then fit 240 registers and
// run two waves per SIMD: 118 -> 128 TFLOP/s at L = 4096 (profiles/r02_attn_variants.txt).  The map-writing variants
// spill at 256 registers and stay at one wave (measured both ways, profiles/r02_attn_maps.txt).
//
// Masking follows the reference: blocked scores become -inf BEFORE the softmax; a fully blocked
// row therefore has zero row-sum and comes out NaN (0 * inf), exactly like torch's
// softmax(-inf, ..., -inf) -- lamp/SubLayers.py:31-39, SURVEY.md G10.
#include "lamp_kernels.h"

namespace lamp {

__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

// MK = mask kind (LAMP_MASK_*), a compile-time parameter so that each variant carries only its own mask code.
// PM = what is written beside O:  0 nothing;  1 the probability maps, exact two-pass softmax (return_attns in eval);
// 2 the raw scaled scores (log2 domain, -inf where blocked) into the map buffer plus each row's log2-sum-exp --
// the single-pass kernel at full speed; softmax_from_scores_kernel then turns the scores into probabilities in
// place (training forward: the maps are needed for the backward pass).
// Two waves per SIMD for the variants that fit 240 registers.  The map-writing variants need the registers for their
// addresses, and the key-token variant (MK = LAMP_MASK_KEY_TOKENS_I64: 32 raw token registers per tile; only bare
// lamp_sdpa_fwd callers reach it -- lamp_forward and lamp_mha_fwd hand the kernels the plan's bit-packed copy) spilled
// 13-18 instructions to scratch under that bound through round 3: both are built for one wave per SIMD, spill-free.
template <int DP, int KSPLIT, int PM, int MK>
__global__ __launch_bounds__(256, (PM == 0 && MK != LAMP_MASK_KEY_TOKENS_I64) ? 2 : 1) void attn_kernel(AttnParams p) {
    constexpr bool WRITE_P = PM == 1;
    static_assert(!WRITE_P || KSPLIT == 1, "probability write-out uses unsplit keys");
    constexpr int DKC = DP / 8, DVB = DP / 32, QB = 4 / KSPLIT, QS = DP + 4;
    constexpr int QG = DKC >= 4 ? 4 : DKC;  // Q fragments read ahead per group
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int qb = wave / KSPLIT, ks = wave % KSPLIT;
    // 1-D grid, XCD-aware: consecutive work items = the query blocks of one (sample, head), kept on one XCD
    const int nqb = (p.lq + 32 * QB - 1) / (32 * QB);
    const int item = xcd_remap(blockIdx.x, gridDim.x);
    const int qblk = item % nqb;
    const int bh = item / nqb;
    const int h = bh % p.H, b = bh / p.H;
    const int q0 = (qblk * QB + qb) * 32;
    const int qi = q0 + l31;
    const bool wave_active = q0 < p.lq;
    const int qc = qi < p.lq ? qi : p.lq - 1;

    const int q_r = int(p.lay.q_r), k_r = int(p.lay.k_r), v_r = int(p.lay.v_r);
    // this sample's keys: all lk, or (ragged batches, AttnParams::kv_len) its own count and its first row in the packed
    // K / V matrices; the descriptors end at the sample's last key (rows past it read as zeros)
    const int lk_b = __builtin_amdgcn_readfirstlane(p.kv_len ? p.kv_len[b] : p.lk);   // wave-uniform: keep it scalar
    const int row0 = __builtin_amdgcn_readfirstlane(p.kv_len ? p.kv_off[b] : 0);
    const int64_t k_row0 = p.kv_len ? int64_t(row0) * k_r : int64_t(b) * p.lay.k_b;
    const int64_t v_row0 = p.kv_len ? int64_t(row0) * v_r : int64_t(b) * p.lay.v_b;
    const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(p.Q + int64_t(b) * p.lay.q_b + int64_t(h) * p.lay.q_h,
                                                 (uint64_t(p.lq - 1) * q_r + p.dk) * 4u);
    const __amdgpu_buffer_rsrc_t rsK = make_rsrc(p.K + k_row0 + int64_t(h) * p.lay.k_h,
                                                 lk_b > 0 ? (uint64_t(lk_b - 1) * k_r + p.dk) * 4u : 0);
    const bool has_v = p.V != nullptr;
    const __amdgpu_buffer_rsrc_t rsV =
        make_rsrc(has_v ? p.V + v_row0 + int64_t(h) * p.lay.v_h : p.K,
                  (has_v && lk_b > 0) ? (uint64_t(lk_b - 1) * v_r + p.dv) * 4u : 0);
    const __amdgpu_buffer_rsrc_t rsM =
        MK == LAMP_MASK_BITS_U32
            ? make_rsrc(static_cast<const unsigned*>(p.mask) + int64_t(b) * p.m_sb,
                        (uint64_t(p.lq - 1) * uint64_t(p.m_sq) + (p.lk + 31) / 32) * 4u)
        : MK == LAMP_MASK_U8
            ? make_rsrc(static_cast<const unsigned char*>(p.mask) + int64_t(b) * p.m_sb,
                        uint64_t(p.lq - 1) * uint64_t(p.m_sq) + p.lk)
        : MK == LAMP_MASK_KEY_TOKENS_I64
            ? make_rsrc(static_cast<const long long*>(p.mask) + int64_t(b) * p.m_sb, uint64_t(p.lk) * 8u)
            : make_rsrc(p.K, 0);

    // ---- Q block -> LDS (pre-scaled); the KSPLIT waves of a block share the copy work ----
    float* Qs = smem + qb * 32 * QS;
    {
        constexpr int C4 = DP / 4;
        constexpr int PER_WAVE = 32 * C4 / KSPLIT;  // float4 per wave
#pragma unroll
        for (int i = 0; i < PER_WAVE / 64; ++i) {
            const int idx = ks * PER_WAVE + i * 64 + lane;
            const int row = idx / C4, c = (idx - row * C4) * 4;
            const int q = q0 + row;
            const float4 v = bload4(rsQ, (q < p.lq && c < p.dk) ? unsigned(q * q_r + c) * 4u : OOB, 0);
            *reinterpret_cast<float4*>(Qs + row * QS + c) =
                make_float4(v.x * p.scale_log2e, v.y * p.scale_log2e, v.z * p.scale_log2e, v.w * p.scale_log2e);
        }
    }
    __syncthreads();

    // key tiles to visit: the sample's own (tiles past them hold PAD keys only: exp2(-inf) = 0 exactly); the map-writing
    // variants walk the padded length, every column of a map row has to be produced
    const int nt = PM == 0 ? (lk_b + 31) / 32 : (p.lk + 31) / 32;
    float4 kf[DKC];
    float vf[16][DVB];   // V[key_r(hi)][DVB*l31 + e]: block e of O^T holds dv columns {DVB*i + e}
    unsigned mraw[16];   // token-is-PAD flags of the tile, loaded one tile ahead
    // LAMP_MASK_U8: the tile's sixteen mask bytes of this lane, two per register (hipcc merges pairs of byte loads with
    // v_perm_b32): eight registers instead of sixteen -- the 128-wide, maps-off, unsplit instantiation behind a bare
    // lamp_sdpa_fwd with a byte mask was spilling 12 bytes per lane; four per register spills again
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    u16x2 mpair[8];
    unsigned mword = 0;  // LAMP_MASK_BITS_U32: this row's 32 mask bits of the tile (one load instead of sixteen)

    auto load_k = [&](int kt) {
        const unsigned base = unsigned((kt * 32 + l31) * k_r + hi * 4) * 4u;
#pragma unroll
        for (int c = 0; c < DKC; ++c)
            kf[c] = bload4(rsK, (c * 8 + hi * 4 < p.dk) ? base + unsigned(c) * 32u : OOB, 0);
    };
    auto load_v = [&](int kt) {
        const bool col_ok = DVB * l31 < p.dv;  // dv is a multiple of 4: the DVB-wide vector is all-or-nothing
        const unsigned base = unsigned((kt * 32 + 4 * hi) * v_r + DVB * l31) * 4u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned off = col_ok ? base + unsigned(((r & 3) + 8 * (r >> 2)) * v_r) * 4u : OOB;
            if constexpr (DVB == 4) {
                const float4 v = bload4(rsV, off, 0);
                vf[r][0] = v.x; vf[r][1] = v.y; vf[r][2] = v.z; vf[r][3] = v.w;
            } else if constexpr (DVB == 2) {
                const f32x2 v = bload2(rsV, off);
                vf[r][0] = v.x; vf[r][1] = v.y;
            } else {
                vf[r][0] = bload1(rsV, off);
            }
        }
    };
    auto load_mask = [&](int kt) {
        const int kbase = kt * 32 + 4 * hi;
        if constexpr (MK == LAMP_MASK_BITS_U32) {
            // this row's 32 mask bits of the tile, pre-shifted so that bit (r&3)+8(r>>2) belongs to register r
            mword = __builtin_amdgcn_raw_buffer_load_b32(rsM, unsigned(int64_t(qc) * p.m_sq + kt) * 4u, 0, 0);
        } else if constexpr (MK == LAMP_MASK_U8) {
            const unsigned mo = unsigned(int64_t(qc) * p.m_sq) + unsigned(kbase);
#pragma unroll
            for (int r = 0; r < 16; ++r) mpair[r >> 1][r & 1] = static_cast<unsigned short>(bload_u8(rsM, mo + (r & 3) + 8 * (r >> 2)));
        } else if constexpr (MK == LAMP_MASK_KEY_TOKENS_I64) {
#pragma unroll
            for (int r = 0; r < 16; ++r)  // past lk: reads 0 == PAD == blocked (forced to -inf below anyway)
                mraw[r] = bload_u64(rsM, unsigned(kbase + (r & 3) + 8 * (r >> 2)) * 8u) == 0 ? 1u : 0u;
        }
    };
    // S^T = K Q^T for the tile in kf, Q fragments read QG chunks ahead from LDS; then -inf where blocked.
    auto scores = [&](int kt, f32x16& s) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* qp = Qs + l31 * QS + hi * 4;
        float4 qa[QG], qn[QG];
#pragma unroll
        for (int j = 0; j < QG; ++j) qa[j] = *reinterpret_cast<const float4*>(qp + j * 8);
#ifdef LAMP_SETPRIO   // experiment (profiles/r04_setprio.txt)
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int g = 0; g < DKC / QG; ++g) {
            if (g + 1 < DKC / QG) {
#pragma unroll
                for (int j = 0; j < QG; ++j) qn[j] = *reinterpret_cast<const float4*>(qp + ((g + 1) * QG + j) * 8);
            }
#pragma unroll
            for (int j = 0; j < QG; ++j) {
                const float4 kk = kf[g * QG + j];
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.x, qa[j].x, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.y, qa[j].y, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.z, qa[j].z, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.w, qa[j].w, s, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < QG; ++j) qa[j] = qn[j];
        }
#ifdef LAMP_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        const int kbase = kt * 32 + 4 * hi;
        const unsigned mw = MK == LAMP_MASK_BITS_U32 ? mword >> (4 * hi) : 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kbase + (r & 3) + 8 * (r >> 2);
            bool blk = false;
            if constexpr (MK == LAMP_MASK_BITS_U32) blk = (mw & (1u << ((r & 3) + 8 * (r >> 2)))) != 0;
            if constexpr (MK == LAMP_MASK_U8) blk = mpair[r >> 1][r & 1] != 0;
            if constexpr (MK == LAMP_MASK_KEY_TOKENS_I64) blk = mraw[r] != 0;
            if (key >= lk_b || blk) s[r] = -INFINITY;
        }
    };
    auto pv = [&](const f32x16& pr, f32x16 (&o)[DVB], int k_next = -1) {
#ifdef LAMP_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int e = 0; e < DVB; ++e) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                o[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r][e], pr[r], o[e], 0, 0, 0);
#ifdef ATTN_K_AFTER_PV0   // experiment: the next tile's K / mask requested behind the first PV block instead of behind QK^T
            if (e == 0 && k_next >= 0) {
                __builtin_amdgcn_sched_barrier(0);
                load_k(k_next);
                load_mask(k_next);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        }
#ifdef LAMP_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };

    f32x16 o[DVB];
#pragma unroll
    for (int e = 0; e < DVB; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[e][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    if constexpr (WRITE_P) {
        if (wave_active) {
            // pass 1: exact row max / row sum
            for (int kt = 0; kt < nt; ++kt) {
                load_k(kt);
                load_mask(kt);
                f32x16 s;
                scores(kt, s);
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
            // pass 2: normalised probabilities out, O = P V
            const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
            const float inv_l = 1.0f / l_run;
            float* Prow = p.P + (int64_t(h) * p.P_batch + p.P_b0 + b) * int64_t(p.lq) * p.lk + int64_t(qc) * p.lk;
            for (int kt = 0; kt < nt; ++kt) {
                load_k(kt);
                load_mask(kt);
                if (has_v) load_v(kt);
                f32x16 s;
                scores(kt, s);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] = exp2f(s[r] - m_use) * inv_l;
                    const int key = kt * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
                    if (qi < p.lq && key < p.lk) Prow[key] = s[r];
                    if (key >= p.lk) s[r] = 0.f;
                }
                if (has_v) pv(s, o);
            }
            if (l_run == 0.f) {
#pragma unroll
                for (int e = 0; e < DVB; ++e)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[e][r] = __builtin_nanf("");
            }
        }
    } else {
        // Online softmax with a LAZY rescale: the running max is only advanced (and O^T, l rescaled) when
        // some row's tile max exceeds it by more than 2^RESCALE_THR; until then probabilities are taken
        // relative to the stale max (<= 2^32, harmless in fp32 and exactly cancelled by the final 1/l).
        // This keeps the 64-register accumulator out of the VALU on almost every tile.  The decision is
        // wave-uniform and depends only on this (sample, head, query block): batch-invariant.
        constexpr float RESCALE_THR = 32.0f;
        // Key tiles to visit: all of them, or -- for a shared mask with a sparsity hint -- only the tiles of this
        // 32-query block that hold at least one unblocked entry (skipped tiles would contribute exp2(-inf) = 0).
        const int* tl = (p.tiles && PM != 2) ? p.tiles + int64_t(q0 >> 5) * p.tiles_stride : nullptr;  // wave-uniform
        const int n_act = tl ? tl[0] : nt;
        auto tile_at = [&](int idx) { return idx < n_act ? (tl ? tl[1 + idx] : idx) : nt; };  // nt = past the end
        if (wave_active && ks < n_act) {
            int kt = tile_at(ks);
            load_k(kt);
            load_mask(kt);
            load_v(kt);
            for (int idx = ks; idx < n_act; idx += KSPLIT) {
                const int kn = tile_at(idx + KSPLIT);  // next tile of this wave (past-the-end: range-checked zeros)
                f32x16 s;
                scores(kt, s);
                __builtin_amdgcn_sched_barrier(0);
#if !(defined(ATTN_ABL) && (ATTN_ABL & 2)) && !defined(ATTN_K_AFTER_PV0)   // timing experiments (tools/build_variant.sh EXTRA=-DATTN_ABL=..): 2 = no loads in the loop
                load_k(kn);     // unconditional prefetch, flies under softmax + PV
                load_mask(kn);
#endif
                if constexpr (PM == 2) {
                    float* Srow = p.P + (int64_t(h) * p.P_batch + p.P_b0 + b) * int64_t(p.lq) * p.lk + int64_t(qi) * p.lk;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
                        if (qi < p.lq && key < p.lk) Srow[key] = s[r];
                    }
                }
#if defined(ATTN_ABL) && (ATTN_ABL & 1)   // 1 = no softmax arithmetic between the two products
                l_run += 1.0f;
                pv(s, o);
                __builtin_amdgcn_sched_barrier(0);
#if !(ATTN_ABL & 2)
                load_v(kn);
#endif
                kt = kn;
                continue;
#endif
                float tmax = s[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
                tmax = fmaxf(tmax, xor32(tmax));
                if (__any(tmax > m_run + RESCALE_THR)) {
                    const float m_new = fmaxf(m_run, tmax);
                    const float alpha = __builtin_amdgcn_exp2f(m_run - ((m_new == -INFINITY) ? 0.f : m_new));
                    l_run *= alpha;
                    m_run = m_new;
#pragma unroll
                    for (int e = 0; e < DVB; ++e)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[e][r] *= alpha;
                }
                const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
                // even and odd registers summed apart, then the two halves of the row: the order of attention_tile.hip, which
                // does it in packed adds (a vector instruction costs matrix-pipe time, profiles/r05_mfma_chain.txt)
                float pe = 0.f, po = 0.f;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    s[r] = __builtin_amdgcn_exp2f(s[r] - m_use);
                    s[r + 1] = __builtin_amdgcn_exp2f(s[r + 1] - m_use);
                    pe += s[r];
                    po += s[r + 1];
                }
                float psum = pe + po;
                psum += xor32(psum);
                l_run += psum;
#ifdef ATTN_K_AFTER_PV0
                pv(s, o, kn);
#else
                pv(s, o);
#endif
                __builtin_amdgcn_sched_barrier(0);
#if !(defined(ATTN_ABL) && (ATTN_ABL & 2))
                load_v(kn);  // flies under the next QK^T
#endif
                kt = kn;
            }
        }
        if constexpr (KSPLIT > 1) {
            // ---- merge the KSPLIT partial results of each query block (lane-local positions) ----
            constexpr int CW = (DP + 2) * 32;  // floats per wave: O^T [DP][32], m [32], l [32]
            __syncthreads();                   // every wave is done reading Qs: the region is reused
            float* mine = smem + wave * CW;
#pragma unroll
            for (int e = 0; e < DVB; ++e)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    mine[(e * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = o[e][r];
            if (hi == 0) {
                mine[DP * 32 + l31] = m_run;
                mine[(DP + 1) * 32 + l31] = l_run;
            }
            __syncthreads();
            if (ks == 0 && wave_active) {
                float m_all = m_run;
#pragma unroll
                for (int s2 = 1; s2 < KSPLIT; ++s2) m_all = fmaxf(m_all, smem[(wave + s2) * CW + DP * 32 + l31]);
                const float m_use = (m_all == -INFINITY) ? 0.f : m_all;
                const float w0 = exp2f(m_run - m_use);
                l_run *= w0;
#pragma unroll
                for (int e = 0; e < DVB; ++e)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[e][r] *= w0;
#pragma unroll
                for (int s2 = 1; s2 < KSPLIT; ++s2) {
                    const float* other = smem + (wave + s2) * CW;
                    const float ws = exp2f(other[DP * 32 + l31] - m_use);
                    l_run = fmaf(other[(DP + 1) * 32 + l31], ws, l_run);  // explicit fma: same bits in every variant
#pragma unroll
                    for (int e = 0; e < DVB; ++e)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            o[e][r] = fmaf(other[(e * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31], ws, o[e][r]);
                }
                m_run = m_all;  // l_run is now relative to the merged maximum
            }
        }
        if constexpr (PM == 2) {
            // row log2-sum-exp of the scaled scores: probabilities = exp2(score - lse).  A fully blocked row has
            // l = 0 -> lse = -inf -> exp2(-inf - -inf) = NaN, as the reference's softmax gives.
            if (wave_active && ks == 0 && hi == 0 && qi < p.lq)
                p.lse[(int64_t(h) * p.B + b) * int64_t(p.lq) + qi] =  // local to this call: [H][B][lq]
                    ((m_run == -INFINITY) ? 0.f : m_run) + log2f(l_run);
        }
        const float inv_l = 1.0f / l_run;
#pragma unroll
        for (int e = 0; e < DVB; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[e][r] *= inv_l;
    }

    // ---- store: lane (query, hi), register r, block e  <->  O[query][DVB*i(r,hi) + e] ----
    if (wave_active && ks == 0 && qi < p.lq && p.O != nullptr) {
        float* Orow = p.O + int64_t(b) * p.lay.o_b + int64_t(h) * p.lay.o_h + int64_t(qi) * p.lay.o_r;
        const bool vec = ((p.lay.o_b | p.lay.o_h | p.lay.o_r) & 3) == 0 &&
                         (reinterpret_cast<uintptr_t>(p.O) & 15u) == 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = DVB * ((r & 3) + 8 * (r >> 2) + 4 * hi);
            if (col >= p.dv) continue;
            if (DVB == 4 && vec) {
                *reinterpret_cast<float4*>(Orow + col) = make_float4(o[0][r], o[DVB > 1 ? 1 : 0][r],
                                                                     o[DVB > 2 ? 2 : 0][r], o[DVB > 3 ? 3 : 0][r]);
            } else {
#pragma unroll
                for (int e = 0; e < DVB; ++e) Orow[col + e] = o[e][r];
            }
        }
    }
}

template <int DP, int KSPLIT, int PM, int MK>
static int launch_attn_mk(const AttnParams& p, hipStream_t s) {
    constexpr int QB = 4 / KSPLIT;
    constexpr size_t lds_q = size_t(QB) * 32 * (DP + 4) * sizeof(float);
    constexpr size_t lds_c = KSPLIT > 1 ? size_t(4) * (DP + 2) * 32 * sizeof(float) : 0;
    constexpr size_t lds = lds_q > lds_c ? lds_q : lds_c;
    auto kern = attn_kernel<DP, KSPLIT, PM, MK>;
    if constexpr (lds > 65536) {
        static AttrOnce once;
        if (int e = once.set(reinterpret_cast<const void*>(kern), lds)) return e;
    }
    const int64_t nwg = int64_t((p.lq + 32 * QB - 1) / (32 * QB)) * p.H * p.B;
    if (nwg > 0x7fffffffLL) return LAMP_E_DIMS;
    hipLaunchKernelGGL(kern, dim3(unsigned(nwg)), dim3(256), lds, s, p);
    return int(hipGetLastError());
}

template <int DP, int KSPLIT, int PM>
static int launch_attn_ks(const AttnParams& p, hipStream_t s) {
    switch (p.mask_kind) {
        case LAMP_MASK_U8: return launch_attn_mk<DP, KSPLIT, PM, LAMP_MASK_U8>(p, s);
        case LAMP_MASK_KEY_TOKENS_I64: return launch_attn_mk<DP, KSPLIT, PM, LAMP_MASK_KEY_TOKENS_I64>(p, s);
        case LAMP_MASK_BITS_U32: return launch_attn_mk<DP, KSPLIT, PM, LAMP_MASK_BITS_U32>(p, s);
        default: return launch_attn_mk<DP, KSPLIT, PM, LAMP_MASK_NONE>(p, s);
    }
}

template <int DP>
static int launch_attn_dp(const AttnParams& p, int ksplit, hipStream_t s) {
    if (p.P && p.lse) return ksplit == 2 ? launch_attn_ks<DP, 2, 2>(p, s) : launch_attn_ks<DP, 1, 2>(p, s);
    if (p.P) return launch_attn_ks<DP, 1, 1>(p, s);
    if (ksplit >= 4) return launch_attn_ks<DP, 4, 0>(p, s);
    if (ksplit == 2) return launch_attn_ks<DP, 2, 0>(p, s);
    return launch_attn_ks<DP, 1, 0>(p, s);
}

// P[row][k] = exp2(S[row][k] - lse[row]) in place: the second half of the single-pass map write-out (PM == 2).
__global__ __launch_bounds__(256) void softmax_from_scores_kernel(float* __restrict__ P, const float* __restrict__ lse,
                                                                  int64_t rows, int lk) {
    const int lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float l = lse[row];
    float* p = P + row * lk;
    for (int c = lane; c < lk; c += 64) p[c] = __builtin_amdgcn_exp2f(p[c] - l);
}

#ifdef LAMP_TUNING
// Tuning build only (liblamp_hip_tuning.so): 0 = heuristic; bits 0-2: force that key split (1/2/4); bits 4-6: query
// blocks per workgroup of the small-shape kernel (attention_small.hip); bit 7: that kernel for any query count; bit 8: no
// LDS-tile kernel (attention_tile.hip).
static int g_force_attn = 0;
extern "C" __attribute__((visibility("default"))) void lamp_debug_force_attn(int v) { g_force_attn = v; }
#else
constexpr int g_force_attn = 0;
#endif

int launch_attn(const AttnParams& p, hipStream_t s) {
    if (p.B <= 0 || p.H <= 0 || p.lq <= 0 || p.lk <= 0 || p.dk <= 0 || p.dv <= 0) return LAMP_E_DIMS;
    if ((p.dk & 3) || (p.dv & 3)) return LAMP_E_UNSUPPORTED;
    if (!p.Q || !p.K) return LAMP_E_NULL;
    if ((!p.V || !p.O) && !(p.P && !p.V && !p.O)) return LAMP_E_NULL;  // V, O optional only with P
    if (p.lse && (!p.P || !p.V || !p.O)) return LAMP_E_NULL;             // single-pass map write-out needs everything
    if (p.mask_kind != LAMP_MASK_NONE && !p.mask) return LAMP_E_NULL;
    const lamp_attn_layout& L = p.lay;
    if (((L.q_b | L.q_h | L.q_r | L.k_b | L.k_h | L.k_r | L.v_b | L.v_h | L.v_r) & 3) ||
        !aligned16(p.Q) || !aligned16(p.K) || (p.V && !aligned16(p.V)))
        return LAMP_E_ALIGN;
    if (p.dk > 128 || p.dv > 128) return launch_attn_general(p, s);  // beyond the fused kernel's register budget
    const bool sparse = !(g_force_attn & 0x200) && attn_sparse_applies(p);   // bit 9 of the tuning hook: never the pair kernel
    // the pair kernel executes the allowed pairs only: count what is executed (a roofline fraction must not exceed 1)
    const double pairs = sparse && p.allowed_pairs > 0 ? double(p.allowed_pairs) : double(p.lq) * p.lk;
    const double flops = 2.0 * p.B * p.H * pairs * (p.dk + p.dv) * (p.P ? 1.5 : 1.0);
    const double bytes = 4.0 * p.B * p.H * (double(p.lq) * (p.dk + p.dv) + double(p.lk) * (p.dk + p.dv)) +
                         (p.P ? 4.0 * p.B * p.H * double(p.lq) * p.lk : 0.0);
    ProfScope prof(LAMP_K_ATTN, flops, bytes, s);
    const int dmax = p.dk > p.dv ? p.dk : p.dv;
    if (int64_t(p.lq) * L.q_r * 4 >= 0x7fffffffLL || int64_t(p.lk) * L.k_r * 4 >= 0x7fffffffLL ||
        int64_t(p.lk) * L.v_r * 4 >= 0x7fffffffLL || (int64_t(p.lq) * p.m_sq + p.lk) * (p.mask_kind == LAMP_MASK_BITS_U32 ? 4 : 1) >= 0x7fffffffLL)
        return LAMP_E_UNSUPPORTED;  // 32-bit offsets inside one (sample, head) slice
    // Key split: must NOT depend on the batch size (a split sums in a different order than the
    // sequential online softmax, and samples must come out bit-identical for every batch / shard), so it
    // is chosen from the per-sample shape only.  At most 128 queries and >= 3 key tiles -> 2-way split
    // (enc-dec attention of reuters: 34 vs 53 us unsplit; its 90 x 90 label self-attention: 18 vs 21 us); otherwise none.
    const int nt = (p.lk + 31) / 32;
    if (p.tiles && (p.m_sb != 0 || (p.mask_kind != LAMP_MASK_U8 && p.mask_kind != LAMP_MASK_BITS_U32)))
        return LAMP_E_UNSUPPORTED;  // the sparsity hint belongs to shared masks
    int ksplit = g_force_attn & 7;
    if (ksplit != 1 && ksplit != 2 && ksplit != 4) ksplit = (p.lq <= 128 && nt >= 3) ? 2 : 1;
    if (ksplit == 4 && p.lse) ksplit = 2;
    int rc;
    // at most 256 queries: 16-query blocks on 16x16x4 (attention_small.hip); bit 7 of the tuning hook lifts the limit
    if (sparse)
        rc = launch_attn_sparse(p, s);
    else if (attn_small_applies(p, (g_force_attn & 0x80) != 0))
        rc = launch_attn_small(p, g_force_attn, s);
    else if (ksplit == 1 && !(g_force_attn & 0x100) && attn_tile_applies(p))   // bit 8 of the tuning hook: attn_kernel instead
        rc = launch_attn_tile(p, s);
    else if (dmax <= 32)
        rc = launch_attn_dp<32>(p, ksplit, s);
    else if (dmax <= 64)
        rc = launch_attn_dp<64>(p, ksplit, s);
    else
        rc = launch_attn_dp<128>(p, ksplit, s);
    if (rc || !p.lse) return rc;
    // NB: only the rows of THIS call's (P_b0 .. P_b0 + B) samples are normalised, head by head
    const int64_t rows_per_head = int64_t(p.B) * p.lq;
    const int64_t g = (rows_per_head + 3) / 4;
    if (g > 0x7fffffffLL) return LAMP_E_DIMS;
    if (p.P_batch == p.B && p.P_b0 == 0) {  // the maps of all heads are one contiguous block of rows
        const int64_t g_all = (rows_per_head * p.H + 3) / 4;
        if (g_all > 0x7fffffffLL) return LAMP_E_DIMS;
        hipLaunchKernelGGL(softmax_from_scores_kernel, dim3(unsigned(g_all)), dim3(256), 0, s, p.P, p.lse,
                           rows_per_head * p.H, p.lk);
        return int(hipGetLastError());
    }
    for (int h = 0; h < p.H; ++h) {
        const int64_t off = (int64_t(h) * p.P_batch + p.P_b0) * p.lq;
        hipLaunchKernelGGL(softmax_from_scores_kernel, dim3(unsigned(g)), dim3(256), 0, s, p.P + off * p.lk,
                           p.lse + int64_t(h) * rows_per_head, rows_per_head, p.lk);
    }
    return int(hipGetLastError());
}

}  // namespace lamp
