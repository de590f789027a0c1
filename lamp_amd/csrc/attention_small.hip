// Fused masked attention for SMALL label / token counts (lq <= 256: reuters' 90 and bibtex's 159 labels), exact
// fp32 on v_mfma_f32_16x16x4_f32.  Same mathematics, masks and NaN behaviour as attention.hip (which keeps the large
// shapes: 32-query blocks on 32x32x2 are the better tile once a (sample, head) alone fills the chip).
//
// Why a second kernel.  With 90 queries the 32-row kernel has 3 query blocks per (sample, head) -- at batch 32 x 4
// heads 384 blocks x 2 key halves = 768 waves of ~20 us for 1024 SIMDs: a quarter of the chip idle, the rest running
// ONE wave per SIMD with nothing to cover its softmax and load latencies (round 1: 0.26 of the fp32-MFMA peak).
// Here the unit is a 16-query x 16-key block (64 MFMAs of 32 cycles = 2048 cycles at d_k = d_v = 128):
//   * a wave owns ONE 16-query block and a share of its key tiles; a workgroup = QB query blocks x KSPLIT key shares
//     (merged lane-locally through LDS at the end).  reuters enc-dec (90 x 302): 3072 waves of 4-5 tiles -- three waves
//     per SIMD, so one wave's exp2 / loads overlap another's MFMAs.
//   * the Q block (16 x d_k, pre-scaled by log2(e)/temperature) sits in LDS and is re-read per key tile as b128
//     fragments: keeping it in registers (32 per lane at d_k = 128) put the kernel at 212 registers = two waves per
//     SIMD; from LDS it fits three, and all 768 workgroups of the reuters shape are resident at once.
//   * both products are TRANSPOSED as in attention.hip, so the query sits on the lane (column = lane & 15) in both
//     accumulators and register r of S^T (key 4*(lane>>4) + r) is directly the B operand of PV step r.
//   * row statistics: a lane sees 4 of a tile's 16 keys.  The running maximum must agree across the four lane groups of
//     a query (they feed one PV product), so it is exchanged across groups -- but only inside the lazy-rescale branch
//     (first tile, or a tile maximum 2^32 above the running one); the row SUM stays a per-lane partial until the end.
// Variant (KSPLIT) and kernel choice depend on the per-sample shape only, never on the batch: a sample's bits do not
// depend on the batch it is in.
#include "lamp_kernels.h"

namespace lamp {

// LDS row padding of the Q block / K tiles: 8 floats.  With 4 (rounds 2-3) the sixteen lanes ds_read_b128 serves together
// -- rows 0-3 and 12-15 at one k-quad, rows 4-11 at the next (MI355X_MICROARCH.md, LDS lane groups) -- met two by two on a
// 16-byte slot: 29 % of the kernel's LDS cycles were bank conflicts (profiles/r02_attn_ta_pmc.txt); with 8 none do.
#ifndef LAMP_LDS_PAD16
#define LAMP_LDS_PAD16 8
#endif
constexpr int ATTN16_PAD = LAMP_LDS_PAD16;

namespace {
// Exchange across the four lane groups of a query (lanes l, l^16, l^32, l^48) with gfx950's row-swap instructions instead
// of __shfl_xor (= ds_bpermute through the LDS crossbar, ~100 cycles a step):
//   v_permlane16_swap a, b  swaps the odd 16-lane rows of a with the even rows of b;  v_permlane32_swap a, b swaps the
//   upper half of a with the lower half of b.  Starting from a = b = v, a and b then hold the two partners of every lane.
// (Inline asm: the builtins return their second result equal to the first in hipcc 7.2 -- DESIGN.md 4.4.  Only used in
// the rescale branch and after the key loop, where the scheduling barrier an asm statement implies costs nothing.)
// The s_nop pairs: hipcc cannot see inside an asm statement, so the wait states it would insert between a VALU write and
// a permlane swap reading it (and between the swap and its consumers) are spelled out.
__device__ __forceinline__ void swap16(float& a, float& b) {
    asm volatile("s_nop 3\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap32(float& a, float& b) {
    asm volatile("s_nop 3\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float group_max(float v) {  // max over the 4 lane groups
    float a = v, b = v;
    swap16(a, b);
    a = fmaxf(a, b);
    b = a;
    swap32(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float group_sum(float v) {  // (v[l] + v[l^16]) + (v[l^32] + v[l^48]), the pairing of the xor butterfly
    float a = v, b = v;
    swap16(a, b);
    a = a + b;
    b = a;
    swap32(a, b);
    return a + b;
}
}  // namespace

// PM = 0: O only.  PM = 2: additionally the scaled scores (log2 domain, -inf where blocked) into the map buffer and each
// row's log2-sum-exp into lse; softmax_from_scores_kernel (attention.hip) then normalises in place.
// A workgroup = QB query blocks x KSPLIT key shares = QB * KSPLIT waves (4 .. 12): the QB waves that hold the same key
// share walk the same K / V tiles in step, so all but the first of them find the tile in the CU's L1 -- with one query
// block per workgroup every wave pulled its own 16 KB per tile from L2 and the kernel sat on the L2 -> L1 path
// (reuters enc-dec: 28 us against 15 us of MFMA issue).
template <int DP, int QB, int KSPLIT, int PM, int MK>
__global__ __launch_bounds__(QB* KSPLIT * 64, 3) void attn16_kernel(AttnParams p) {
    constexpr int DKC = DP / 16;   // 16-wide k chunks of the QK^T product (one b128 fragment each)
    constexpr int DV8 = DP / 16;   // floats of a V row per lane = number of 16-row blocks of O^T
    constexpr int QS = DP + ATTN16_PAD;   // LDS row stride of the Q block and the K tiles (floats)
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // Key share of this wave.  With ragged keys a short sample uses fewer shares than the launch has waves for; which
    // waves idle rotates with the workgroup, else the idle ones would always be the same SIMDs' (wave w runs on SIMD w % 4)
    // and the busy SIMDs would be as loaded as without the skipping.
    const int rot = (KSPLIT > 1 && p.kv_len) ? int(blockIdx.x % KSPLIT) : 0;
    // readfirstlane: tell the compiler the share index is wave-uniform (it otherwise wraps every buffer access that depends
    // on it in a waterfall loop: +35 branches, +9 us on the reuters shapes)
    const int qb = wave / KSPLIT, ks = __builtin_amdgcn_readfirstlane((wave % KSPLIT + KSPLIT - rot) % KSPLIT);
    const int slot = qb * KSPLIT + ks;   // position among the workgroup's partial results
#ifdef LAMP_TUNING
    const unsigned long long t_entry = p.trace ? wall_clock64() : 0ull;
    unsigned long long t_loop = 0, t_merge = 0;
#endif
    const int nqg = (p.lq + 16 * QB - 1) / (16 * QB);
    const int item = xcd_remap(blockIdx.x, gridDim.x);  // the query groups of one (sample, head) stay on one XCD
    const int qgrp = item % nqg;
    const int bh = item / nqg;
    const int h = bh % p.H, b = bh / p.H;
    const int q0 = (qgrp * QB + qb) * 16;
    const int qi = q0 + l15;
    const bool wave_active = q0 < p.lq;
    const int qc = qi < p.lq ? qi : p.lq - 1;

    const int q_r = int(p.lay.q_r), k_r = int(p.lay.k_r), v_r = int(p.lay.v_r);
    // this sample's keys: all lk, or (ragged batches) its own count and its first row in the packed K / V matrices.
    // The descriptors end at the sample's last key: rows past it read as zeros (never another sample's, never NaNs).
    // (readfirstlane: the loaded extents are wave-uniform; without the hint hipcc keeps the descriptors built from them in
    // vector registers and wraps every K / V load in a waterfall loop)
    const int lk_b = __builtin_amdgcn_readfirstlane(p.kv_len ? p.kv_len[b] : p.lk);
    const int row0 = __builtin_amdgcn_readfirstlane(p.kv_len ? p.kv_off[b] : 0);
    const int64_t k_row0 = p.kv_len ? int64_t(row0) * k_r : int64_t(b) * p.lay.k_b;
    const int64_t v_row0 = p.kv_len ? int64_t(row0) * v_r : int64_t(b) * p.lay.v_b;
    const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(p.Q + int64_t(b) * p.lay.q_b + int64_t(h) * p.lay.q_h,
                                                 (uint64_t(p.lq - 1) * q_r + p.dk) * 4u);
    const __amdgpu_buffer_rsrc_t rsK = make_rsrc(p.K + k_row0 + int64_t(h) * p.lay.k_h,
                                                 lk_b > 0 ? (uint64_t(lk_b - 1) * k_r + p.dk) * 4u : 0);
    const __amdgpu_buffer_rsrc_t rsV = make_rsrc(p.V + v_row0 + int64_t(h) * p.lay.v_h,
                                                 lk_b > 0 ? (uint64_t(lk_b - 1) * v_r + p.dv) * 4u : 0);
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
    float* Qs = smem + qb * 16 * QS;
    // this wave's K tile (16 keys x d_k), staged through a wave-private LDS block: the tile is fetched with COALESCED loads
    // (whole 512-byte rows per instruction) and re-read as MFMA fragments; fragment-shaped global loads (16 rows x 64 B per
    // instruction) run at about a quarter of the rate (profiles/r02_rejected_experiments.txt #9)
    float* Ks = smem + (QB + wave) * 16 * QS;
    {
        constexpr int C4 = DP / 4;
        constexpr int PER_WAVE = 16 * C4 / KSPLIT;  // float4 per wave
#pragma unroll
        for (int i = 0; i < (PER_WAVE + 63) / 64; ++i) {
            const int idx = ks * PER_WAVE + i * 64 + lane;
            const int row = idx / C4, c = (idx - row * C4) * 4;
            const int q = q0 + row;
            if (i * 64 + lane < PER_WAVE) {
                const float4 v = bload4(rsQ, (q < p.lq && c < p.dk) ? unsigned(q * q_r + c) * 4u : OOB, 0);
                *reinterpret_cast<float4*>(Qs + row * QS + c) =
                    make_float4(v.x * p.scale_log2e, v.y * p.scale_log2e, v.z * p.scale_log2e, v.w * p.scale_log2e);
            }
        }
    }
    __syncthreads();

    // key tiles to visit: the sample's own (the rest holds PAD keys only: exp2(-inf) = 0 exactly); with the map
    // write-out every column of the map row has to be produced, so all tiles of the padded length are walked
    const int nt_b = (lk_b + 15) / 16;
    const int nt = PM == 2 ? (p.lk + 15) / 16 : nt_b;
    // key shares in use: the launch's KSPLIT, or -- ragged batches -- the share count launch_attn_small would choose for
    // THIS sample's key count (<= KSPLIT: the launch was sized for the padded length); the other waves idle
    const int ks_eff = (KSPLIT > 1 && p.kv_len) ? (nt_b >= 12 ? (KSPLIT < 4 ? KSPLIT : 4) : (nt_b >= 4 ? 2 : 1)) : KSPLIT;
    float4 kg[DKC];        // the NEXT tile's K rows in flight (coalesced layout: row = i * RPI + lane / C4K, float4 lane % C4K)
    constexpr int C4K = DP / 4, RPI = 64 / C4K;   // float4 per K row; rows per load instruction
    float vf[4][DV8];      // V[kt*16 + 4g + r][.]: block e of O^T holds the d_v columns given at load_v below
    unsigned mbits = 0;    // bit r = key (kt*16 + 4g + r) is blocked for this lane's query (one tile ahead)

    // Loads of the key loop carry NO per-tile vector arithmetic (a vector instruction costs matrix-pipe time on gfx950,
    // profiles/r05_mfma_chain.txt): the lane's offset inside a tile is a loop invariant with the head-dimension check folded in
    // (OOB + anything < 2^31 stays out of range), the tile's / row's base rides in the SCALAR offset, which the descriptor's
    // range check covers (tools/probes/lds_dma_oob.hip) -- rows past the sample's last key read as zeros without a test.
    const unsigned k_voff = (lane % C4K) * 4 < p.dk ? unsigned((lane / C4K) * k_r + (lane % C4K) * 4) * 4u : OOB;
    auto load_k = [&](int kt) {
#pragma unroll
        for (int i = 0; i < DKC; ++i) kg[i] = bload4(rsK, k_voff, unsigned((kt * 16 + i * RPI) * k_r) * 4u);
    };
    auto stage_k = [&]() {   // registers -> this wave's LDS block (its reads of the previous tile are behind us: in order)
        const int c = (lane % C4K) * 4;
#pragma unroll
        for (int i = 0; i < DKC; ++i) *reinterpret_cast<float4*>(Ks + (i * RPI + lane / C4K) * QS + c) = kg[i];
    };
    // V columns of a lane: DVW consecutive floats per load, the 16 lanes of a group side by side (whole cache lines per
    // row and instruction).  d = 128 takes two such loads 64 columns apart: block e of O^T then holds the d_v columns
    // {4 i + e} (e < 4) and {64 + 4 i + e - 4} (e >= 4), i = the block's row index -- see the store at the end.
    constexpr int DVW = DV8 == 8 ? 4 : DV8;
    // d_v is a multiple of 4: all-or-nothing per load; second half (d = 128 only) 64 columns on
    const unsigned v_voff = DVW * l15 < p.dv ? unsigned(4 * g * v_r + DVW * l15) * 4u : OOB;
    const unsigned v_voff2 = 64 + DVW * l15 < p.dv ? unsigned(4 * g * v_r + DVW * l15) * 4u + 256u : OOB;
    auto load_v = [&](int kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned off = v_voff, so = unsigned((kt * 16 + r) * v_r) * 4u;
            if constexpr (DV8 == 8) {
                const float4 a = bload4(rsV, off, so);
                const float4 c2 = bload4(rsV, v_voff2, so);
                vf[r][0] = a.x; vf[r][1] = a.y; vf[r][2] = a.z; vf[r][3] = a.w;
                vf[r][4] = c2.x; vf[r][5] = c2.y; vf[r][6] = c2.z; vf[r][7] = c2.w;
            } else if constexpr (DV8 == 4) {
                const float4 a = bload4(rsV, off, so);
                vf[r][0] = a.x; vf[r][1] = a.y; vf[r][2] = a.z; vf[r][3] = a.w;
            } else {
                const f32x2 a = bload2(rsV, off + so);
                vf[r][0] = a.x; vf[r][1] = a.y;
            }
        }
    };
    const unsigned m_voff = unsigned(int64_t(qc) * p.m_sq) * 4u;
    auto load_mask = [&](int kt) {
        const int kbase = kt * 16 + 4 * g;
        if constexpr (MK == LAMP_MASK_BITS_U32) {
            // the row's mask word (two tiles per word); the lane group's four bits are taken in scores()
            mbits = __builtin_amdgcn_raw_buffer_load_b32(rsM, m_voff, unsigned(kt >> 1) * 4u, 0);
        } else if constexpr (MK == LAMP_MASK_U8) {
            const unsigned mo = unsigned(int64_t(qc) * p.m_sq) + unsigned(kbase);
            unsigned m = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) m |= (bload_u8(rsM, kt < nt ? mo + r : OOB) != 0 ? 1u : 0u) << r;
            mbits = m;
        } else if constexpr (MK == LAMP_MASK_KEY_TOKENS_I64) {
            unsigned m = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r)  // past lk: reads 0 == PAD == blocked (forced to -inf below anyway)
                m |= (bload_u64(rsM, unsigned(kbase + r) * 8u) == 0 ? 1u : 0u) << r;
            mbits = m;
        }
    };
    // S^T = K Q^T for the tile in kf (two accumulator chains, summed: the dependent-issue latency of the 16x16x4 MFMA
    // is 40 cycles against 32 of issue), then -inf where blocked / past lk.
    auto scores = [&](int kt, f32x4& s) {
        // Blocked keys -- the mask's, and those past the sample's last key, OR-ed in as a scalar -- enter as the first chain's
        // INITIAL value: -inf + anything finite = -inf, 0 + x as before (two instructions per score, none after the product).
        const int valid = lk_b - kt * 16;
        const unsigned tail = valid >= 16 ? 0u : (0xffffu << (valid > 0 ? valid : 0)) & 0xffffu;   // scalar
        unsigned word = tail;
        if constexpr (MK == LAMP_MASK_BITS_U32) word |= mbits >> ((kt & 1) * 16);
        const int mine = MK == LAMP_MASK_BITS_U32 || MK == LAMP_MASK_NONE ? int(word >> (4 * g)) : int(mbits | (tail >> (4 * g)));
        // Blocked keys enter as the accumulator's INITIAL value (-inf; 0 elsewhere): two instructions per score and none behind
        // the product.  DELIBERATE DEVIATION from masked_fill (lamp/SubLayers.py:32), which overwrites whatever the product was:
        // -inf + x is -inf for every FINITE x, but NaN for x = +inf or NaN -- a non-finite K (or Q) row behind a BLOCKED key
        // poisons that query's row here, while the reference stays finite.  Finite inputs (every model this path serves) are
        // unaffected; attention_tile.hip does the same; attention_sparse.hip never touches a blocked key at all.  Pinned by
        // tests/test_gpu_parity.py::test_non_finite_key_behind_a_blocked_key_is_a_documented_deviation.
        f32x4 s0, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int t;
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(t) : "v"(mine), "n"(r));
            s0[r] = __int_as_float(t & int(0xff800000u));
        }
#ifdef LAMP_SETPRIO   // experiment (profiles/r04_setprio.txt): raised wave priority around the QK^T and PV MFMA runs
        __builtin_amdgcn_s_setprio(1);
#endif
        const float* qp = Qs + l15 * QS + 4 * g;   // lane (query l15, group g): Q[q][16c + 4g + j]
        const float* kp = Ks + l15 * QS + 4 * g;   // lane (key   l15, group g): K[k][16c + 4g + j]
#pragma unroll
        for (int c = 0; c < DKC; c += 2) {
            const float4 qa = *reinterpret_cast<const float4*>(qp + 16 * c);
            const float4 qb2 = *reinterpret_cast<const float4*>(qp + 16 * c + 16);
            const float4 ka = *reinterpret_cast<const float4*>(kp + 16 * c);
            const float4 kb = *reinterpret_cast<const float4*>(kp + 16 * c + 16);
            s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.x, qa.x, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kb.x, qb2.x, s1, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.y, qa.y, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kb.y, qb2.y, s1, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.z, qa.z, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kb.z, qb2.z, s1, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ka.w, qa.w, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kb.w, qb2.w, s1, 0, 0, 0);
        }
#ifdef LAMP_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] = s0[r] + s1[r];
    };

    f32x4 o[DV8];
#pragma unroll
    for (int e = 0; e < DV8; ++e) o[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_part = 0.f;   // m_run: this query's running maximum (equal in its 4 lane groups)
    constexpr float RESCALE_THR = 32.0f;

#ifdef LAMP_TUNING
    if (p.trace) t_loop = wall_clock64();
#endif
    if (wave_active && ks < ks_eff && ks < nt) {
        int kt = ks;
        load_k(kt);
        load_mask(kt);
        load_v(kt);
        for (; kt < nt; kt += ks_eff) {
            const int kn = kt + ks_eff;   // past the end: range-checked zeros
            f32x4 s;
            stage_k();       // the tile requested one iteration ago: registers -> LDS, read back as fragments by scores()
            scores(kt, s);
            // pin the order "MFMAs of this tile, THEN the next tile's loads into the registers they just freed": hoisted
            // above the MFMAs the loads need a second register set (K and V double-buffered = +64 VGPRs, spills at 3 waves)
            __builtin_amdgcn_sched_barrier(0);
            load_k(kn);      // flies under softmax + PV
            load_mask(kn);
            if constexpr (PM == 2) {
                float* Srow = p.P + (int64_t(h) * p.P_batch + p.P_b0 + b) * int64_t(p.lq) * p.lk + int64_t(qi) * p.lk;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + 4 * g + r;
                    if (qi < p.lq && key < p.lk) Srow[key] = s[r];
                }
            }
            const float tmax = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            if (__any(tmax > m_run + RESCALE_THR)) {
                const float m_new = fmaxf(m_run, group_max(tmax));
                const float alpha = __builtin_amdgcn_exp2f(m_run - ((m_new == -INFINITY) ? 0.f : m_new));
                l_part *= alpha;
                m_run = m_new;
#pragma unroll
                for (int e = 0; e < DV8; ++e)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[e][r] *= alpha;
            }
            const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] = __builtin_amdgcn_exp2f(s[r] - m_use);
            l_part += (s[0] + s[1]) + (s[2] + s[3]);
#ifdef LAMP_SETPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int e = 0; e < DV8; ++e)
                    o[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r][e], s[r], o[e], 0, 0, 0);
#ifdef LAMP_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            __builtin_amdgcn_sched_barrier(0);
            load_v(kn);      // flies under the next QK^T
        }
    }

#ifdef LAMP_TUNING
    if (p.trace) t_merge = wall_clock64();
#endif
    if constexpr (KSPLIT > 1) {
        // ---- merge the KSPLIT partial results of each query block (lane-local positions, fixed order) ----
        constexpr int CW = (DV8 * 4 + 4) * 64;  // floats per wave: o blocks as float4 per lane, then (m, l, -, -) per lane
        __syncthreads();                        // every wave is done reading its Q block: the region is reused
        float* mine = smem + slot * CW;
#pragma unroll
        for (int e = 0; e < DV8; ++e)
            *reinterpret_cast<float4*>(mine + (e * 64 + lane) * 4) = make_float4(o[e][0], o[e][1], o[e][2], o[e][3]);
        *reinterpret_cast<float4*>(mine + (DV8 * 64 + lane) * 4) = make_float4(m_run, l_part, 0.f, 0.f);
        __syncthreads();
        if (ks == 0 && wave_active) {
            float m_all = m_run;
#pragma unroll
            for (int s2 = 1; s2 < KSPLIT; ++s2)
                if (s2 < ks_eff) m_all = fmaxf(m_all, smem[(slot + s2) * CW + (DV8 * 64 + lane) * 4]);
            const float m_use = (m_all == -INFINITY) ? 0.f : m_all;
            const float w0 = exp2f(m_run - m_use);
            l_part *= w0;
#pragma unroll
            for (int e = 0; e < DV8; ++e)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[e][r] *= w0;
#pragma unroll
            for (int s2 = 1; s2 < KSPLIT; ++s2) {
                if (s2 >= ks_eff) break;
                const float* other = smem + (slot + s2) * CW;
                const float4 ml = *reinterpret_cast<const float4*>(other + (DV8 * 64 + lane) * 4);
                const float ws = exp2f(ml.x - m_use);
                l_part = fmaf(ml.y, ws, l_part);  // explicit fma: same bits in every variant
#pragma unroll
                for (int e = 0; e < DV8; ++e) {
                    const float4 v = *reinterpret_cast<const float4*>(other + (e * 64 + lane) * 4);
                    o[e][0] = fmaf(v.x, ws, o[e][0]);
                    o[e][1] = fmaf(v.y, ws, o[e][1]);
                    o[e][2] = fmaf(v.z, ws, o[e][2]);
                    o[e][3] = fmaf(v.w, ws, o[e][3]);
                }
            }
            m_run = m_all;
        }
    }
    if (!(wave_active && ks == 0)) return;
    const float l_run = group_sum(l_part);   // the four lane groups of a query hold disjoint keys
    if constexpr (PM == 2) {
        // row log2-sum-exp of the scaled scores: probabilities = exp2(score - lse).  A fully blocked row has
        // l = 0 -> lse = -inf -> exp2(-inf - -inf) = NaN, as the reference's softmax gives.
        if (g == 0 && qi < p.lq)
            p.lse[(int64_t(h) * p.B + b) * int64_t(p.lq) + qi] = ((m_run == -INFINITY) ? 0.f : m_run) + log2f(l_run);
    }
    const float inv_l = 1.0f / l_run;   // l = 0 (fully blocked row): 0 * inf = NaN, like torch

    // ---- store: lane (query, g), register r, block e  <->  O[query][DVW*(4g + r) + (e % DVW) + 64*(e / DVW)] ----
    if (qi < p.lq) {
        float* Orow = p.O + int64_t(b) * p.lay.o_b + int64_t(h) * p.lay.o_h + int64_t(qi) * p.lay.o_r;
        const bool vec = ((p.lay.o_b | p.lay.o_h | p.lay.o_r) & 3) == 0 && (reinterpret_cast<uintptr_t>(p.O) & 15u) == 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int e0 = 0; e0 < DV8; e0 += DVW) {
                const int col = DVW * (4 * g + r) + 64 * (e0 / DVW);
                if (col >= p.dv) continue;
                if (DVW == 4 && vec) {
                    *reinterpret_cast<float4*>(Orow + col) =
                        make_float4(o[e0][r] * inv_l, o[e0 + (DVW > 1 ? 1 : 0)][r] * inv_l,
                                    o[e0 + (DVW > 2 ? 2 : 0)][r] * inv_l, o[e0 + (DVW > 3 ? 3 : 0)][r] * inv_l);
                } else {
#pragma unroll
                    for (int e = 0; e < DVW; ++e) Orow[col + e] = o[e0 + e][r] * inv_l;
                }
            }
        }
    }
#ifdef LAMP_TUNING
    if (p.trace && tid == 0) {   // wave 0: entry, loop start, loop end, exit -- its ROLE (key share, query block) rotates with the
                                 // workgroup when keys are ragged, so the role goes into word 7 for the analysis to split on
        unsigned long long* t = p.trace + size_t(blockIdx.x) * 8;
        t[0] = t_entry; t[1] = t_loop; t[2] = t_merge; t[3] = wall_clock64();
        t[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        t[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        t[6] = unsigned(item);
        t[7] = unsigned(ks) | (unsigned(qb) << 8);
    }
#endif
}

template <int DP, int QB, int KSPLIT, int PM, int MK>
static int launch_small_mk(const AttnParams& p, hipStream_t s) {
    constexpr int NW = QB * KSPLIT;
    constexpr size_t lds_q = size_t(QB + NW) * 16 * (DP + ATTN16_PAD) * sizeof(float);   // Q blocks + one K tile per wave
    constexpr size_t lds_c = KSPLIT > 1 ? size_t(NW) * (DP / 16 * 4 + 4) * 64 * sizeof(float) : 0;
    constexpr size_t lds = lds_q > lds_c ? lds_q : lds_c;
    auto kern = attn16_kernel<DP, QB, KSPLIT, PM, MK>;
    if constexpr (lds > 65536) {
        static AttrOnce once;
        if (int e = once.set(reinterpret_cast<const void*>(kern), lds)) return e;
    }
    const int64_t nwg = int64_t((p.lq + 16 * QB - 1) / (16 * QB)) * p.H * p.B;
    if (nwg > 0x7fffffffLL) return LAMP_E_DIMS;
    hipLaunchKernelGGL(kern, dim3(unsigned(nwg)), dim3(NW * 64), lds, s, p);
    return int(hipGetLastError());
}

template <int DP, int QB, int KSPLIT, int PM>
static int launch_small_ks(const AttnParams& p, hipStream_t s) {
    switch (p.mask_kind) {
        case LAMP_MASK_U8: return launch_small_mk<DP, QB, KSPLIT, PM, LAMP_MASK_U8>(p, s);
        case LAMP_MASK_KEY_TOKENS_I64: return launch_small_mk<DP, QB, KSPLIT, PM, LAMP_MASK_KEY_TOKENS_I64>(p, s);
        case LAMP_MASK_BITS_U32: return launch_small_mk<DP, QB, KSPLIT, PM, LAMP_MASK_BITS_U32>(p, s);
        default: return launch_small_mk<DP, QB, KSPLIT, PM, LAMP_MASK_NONE>(p, s);
    }
}

template <int DP, int QB, int KSPLIT>
static int launch_small_pm(const AttnParams& p, hipStream_t s) {
    return p.lse ? launch_small_ks<DP, QB, KSPLIT, 2>(p, s) : launch_small_ks<DP, QB, KSPLIT, 0>(p, s);
}

template <int DP>
static int launch_small_dp(const AttnParams& p, int qb, int ksplit, hipStream_t s) {
    switch (qb * 8 + ksplit) {
        case 1 * 8 + 1: return launch_small_pm<DP, 4, 1>(p, s);   // one key share: always four query blocks per workgroup
        case 2 * 8 + 1: return launch_small_pm<DP, 4, 1>(p, s);
        case 3 * 8 + 1: return launch_small_pm<DP, 4, 1>(p, s);
        case 4 * 8 + 1: return launch_small_pm<DP, 4, 1>(p, s);
        case 1 * 8 + 2: return launch_small_pm<DP, 2, 2>(p, s);   // at least four waves per workgroup
        case 2 * 8 + 2: return launch_small_pm<DP, 2, 2>(p, s);
        case 3 * 8 + 2: return launch_small_pm<DP, 3, 2>(p, s);
        case 4 * 8 + 2: return launch_small_pm<DP, 4, 2>(p, s);
        case 1 * 8 + 4: return launch_small_pm<DP, 1, 4>(p, s);
        case 2 * 8 + 4: return launch_small_pm<DP, 2, 4>(p, s);
        case 3 * 8 + 4: return launch_small_pm<DP, 3, 4>(p, s);
        default: return launch_small_pm<DP, 2, 4>(p, s);           // 4 x 4 = 16 waves do not fit three per SIMD
    }
}

// The shapes this kernel takes (a function of the per-sample shape ONLY): at most 256 queries -- or any number of queries
// over at most 64 keys (delicious' 983 labels x 40 tokens: 135 -> 121 us; 300 x 100 would be 52 -> 35, but 983 x 100 is a
// tie and the self-attention shapes beyond 256 belong to attention.hip) -- with V and O given and either no maps or the
// single-pass map write-out.  The exact two-pass maps and map-only calls stay in attention.hip.
// With ragged keys (kv_len) lk is only the PADDED length of the batch: the choice then rests on lq alone, so that a sample's
// bits do not depend on how far its batch was padded (the fuzz campaign found the <= 64-key rule switching kernels between
// a 300-label sample run alone and the same sample inside a longer batch).  lamp_forward and lamp_mha_fwd always bring
// per-sample key counts for key-token masks, so on the product path the <= 64-key rule serves only calls without one
// (bare lamp_sdpa_fwd, shared masks): delicious' enc-dec attention (983 labels x 40 tokens) runs in attention.hip at
// 135 us instead of 121 us here -- 28 us of a 15.6 ms forward, the price of padding-invariant bits.
bool attn_small_applies(const AttnParams& p, bool any_lq) {
    const bool few_keys = p.lk <= 64 && !p.kv_len;
    return (p.lq <= 256 || few_keys || any_lq) && p.V && p.O && (!p.P || p.lse) && p.dk <= 128 && p.dv <= 128;
}

#ifdef LAMP_TUNING
static unsigned long long* g_attn_trace = nullptr;   // 8 words per workgroup of the NEXT small-shape launch(es)
extern "C" __attribute__((visibility("default"))) void lamp_debug_set_attn_trace(unsigned long long* buf) { g_attn_trace = buf; }
#endif

// force: 0 = heuristic; else (tuning build) key shares in bits 0-2, query blocks per workgroup in bits 4-6.
int launch_attn_small(const AttnParams& p_in, int force, hipStream_t s) {
    AttnParams p = p_in;
    p.trace = nullptr;
#ifdef LAMP_TUNING
    p.trace = g_attn_trace;
#endif
    const int nt = (p.lk + 15) / 16;
    const int nqb = (p.lq + 15) / 16;
    // Key shares (measured, profiles/r02_attn_variants.txt): four from 12 sixteen-key tiles on (reuters enc-dec: 19), two
    // from 4 (reuters' 90-label self-attention: 6; bibtex 7 / 10), else one.  Always four waves per workgroup: larger
    // workgroups (several query blocks sharing their K / V tiles through L1) measured equal or slower on every shape
    // and exist for the tuning build's every-variant tests only.
    int ksplit = force & 7;
    if (ksplit != 1 && ksplit != 2 && ksplit != 4) ksplit = nt >= 12 ? 4 : (nt >= 4 ? 2 : 1);
    int qb = (force >> 4) & 7;
    if (qb < 1 || qb > 4) qb = 4 / ksplit;
    (void)nqb;
    const int dmax = p.dk > p.dv ? p.dk : p.dv;
    if (dmax <= 32) return launch_small_dp<32>(p, qb, ksplit, s);
    if (dmax <= 64) return launch_small_dp<64>(p, qb, ksplit, s);
    return launch_small_dp<128>(p, qb, ksplit, s);
}

}  // namespace lamp
