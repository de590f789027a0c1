// Label self-attention over a SPARSE, UNSTRUCTURED label graph (lamp/Decoders.py:109-113, utils/data_loader.py:
// 37-47): every query row of the shared L x L mask allows only a few keys (BASELINE configs[4]: Bernoulli(0.05) prior, ~206 of
// 4096), but no 32 x 32 tile of the mask is empty, so the tile lists of attention_tile.hip skip nothing and 95 % of the dense
// kernel's 4 L^2 d FLOPs go to blocked pairs.  Here only the ALLOWED (query, key) pairs are computed:
//
//   * one workgroup = 256 queries of one (sample, head), 16 waves; a wave owns 16 queries, 4 lanes per query ("slot"), a lane holds
//     32 of the query's 128 dimensions (q pre-scaled by 1/temperature * log2 e, the running output o: 32 + 32 registers).  (An
//     8-lanes-per-query instantiation -- 128 queries per workgroup, 16 dimensions per lane -- is kept for the tuning build: the
//     fixed part of a step is shared by half as many pairs, 14-16 % slower on every density measured);
//   * K / V stream through LDS in 64-key tiles shared by the whole workgroup (double buffered, 2 x (32 + 32) KiB), fetched by
//     LDS-DMA through the compiler's own builtin (tracked: no hand-counted waits, no inline-assembly loads): every wave requests
//     1/16 of the next tile at the top of a step, one barrier per tile;
//   * per tile a slot takes its row's 64 mask bits (two words of the bit-packed shared mask) and walks the SET bits: per pair
//     8 x ds_read_b128 of the key row, 16 packed FMAs + a 2-step DPP sum over the slot's 4 lanes = the score, lazy online softmax
//     (rescale only when the score exceeds the running maximum by 2^32, as attention_tile.hip), 8 x ds_read_b128 of the value
//     row, 16 packed FMAs.  Slots without a pair left in the tile run along with a score of -inf (probability 0) until the
//     wave's longest list ends: the price of lock-step, ~57 % useful slots at configs[4]'s density (a SIMD's other waves fill
//     the issue slots of a wave that waits at the barrier);
//   * LDS reads are conflict-free for ARBITRARY key rows: a ds_read_b128 is served in lane groups {0-3, 12-15, 20-27},
//     {4-11, 16-19, 28-31} (+32), i.e. a quarter of four different slots each; slot s reads the 128-byte halves of its row in
//     the order r ^ ((s >> 1) & 1), so the four quarters always fall on four different 16-bank quarters (rows are 512 bytes:
//     every row starts at bank 0).
//
// Arithmetic: algorithmic work 2 nnz (d_k + d_v) per (sample, head) instead of 2 L^2 (d_k + d_v); exact masked softmax (a blocked
// key has probability 0 in the reference too), different summation order than the dense kernels -- within 1e-4 of the oracle, not
// bit-identical to attention_tile.hip.  The choice dense / sparse is made from the mask alone (lamp_mask.flags, set by the
// caller from the graph's density), never from the batch.
// Reference: lamp/SubLayers.py:27-43 (ScaledDotProductAttention.forward).
#include "lamp_kernels.h"

namespace lamp {
namespace {

constexpr int SP_WAVES = 16;
constexpr int SP_LPQ_DEFAULT = 4;          // lanes per query of the product route (profiles/r06_sparse_label_attention.txt)
constexpr int SP_TILE = 64;                // keys per tile: one 64-bit mask word pair per query
constexpr int SP_D = 128;                  // d_k = d_v
constexpr int SP_BUF = SP_TILE * SP_D;     // floats of one K or V tile (32 KiB)

typedef float f32x2v __attribute__((ext_vector_type(2)));

template <int LPQ>
__device__ __forceinline__ float slot_sum(float v) {   // sum over the LPQ lanes of a slot, result in all of them
    v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
    if constexpr (LPQ == 8) v += dpp_move<0x141>(v);   // row_half_mirror: lane i <-> 7 - i of the 8-lane half row (the other quad's total)
    return v;
}

// the two K | V tile buffers are SEPARATE variables: the compiler's wait-count pass then knows that an LDS-DMA into one
// does not feed the reads of the other (alias scopes of distinct LDS variables), and the next tile's requests fly under
// this tile's pair loop without a wait in front of every LDS read
__shared__ __attribute__((aligned(16))) float sp_buf0[2][SP_BUF];   // [K | V][key][dim]
__shared__ __attribute__((aligned(16))) float sp_buf1[2][SP_BUF];

typedef __attribute__((address_space(3))) void* sp_lds_ptr;

// LPQ = lanes per query: 8 (8 queries per wave, 128 per workgroup, 16 dimensions per lane) or 4 (16 / 256 / 32: the fixed part of
// a step -- bit scan, softmax bookkeeping, loop control -- is shared by twice the pairs, at twice the registers per lane)
template <int LPQ>
__global__ __launch_bounds__(SP_WAVES * 64) void attn_sparse_kernel(AttnParams p) {
    static_assert(LPQ == 8 || LPQ == 4, "lanes per query");
    constexpr int SP_QPW = 64 / LPQ, SP_QBLK = SP_QPW * SP_WAVES, NR = 32 / LPQ;   // NR: 16-byte reads per row and lane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32): four slots' 64-byte quarters (LPQ = 4:
    // slots {0, 3, 5, 6} / {1, 2, 4, 7}) or quarters of four slots' 128-byte halves (LPQ = 8).  Slot s reads its row's pieces in
    // the order r ^ rot(s) with rot distinct (mod 4 resp. 2) inside every group: conflict-free for arbitrary rows.
    const int slot = lane / LPQ, t = lane % LPQ, rot = LPQ == 8 ? (slot >> 1) & 1 : (slot & 7) >> 1;
    const int nqb = (p.lq + SP_QBLK - 1) / SP_QBLK;
    const int item = xcd_remap(blockIdx.x, gridDim.x);   // the query blocks of one (sample, head) on ONE XCD: its K / V stay in that L2
    const int qblk = item % nqb, bh = item / nqb;
    const int h = bh % p.H, b = bh / p.H;
    const int qi = qblk * SP_QBLK + wave * SP_QPW + slot;
    const bool q_live = qi < p.lq;
    const int q_r = int(p.lay.q_r), k_r = int(p.lay.k_r), v_r = int(p.lay.v_r);

    const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(p.Q + int64_t(b) * p.lay.q_b + int64_t(h) * p.lay.q_h, (uint64_t(p.lq - 1) * q_r + SP_D) * 4u);
    const __amdgpu_buffer_rsrc_t rsK = make_rsrc(p.K + int64_t(b) * p.lay.k_b + int64_t(h) * p.lay.k_h, (uint64_t(p.lk - 1) * k_r + SP_D) * 4u);
    const __amdgpu_buffer_rsrc_t rsV = make_rsrc(p.V + int64_t(b) * p.lay.v_b + int64_t(h) * p.lay.v_h, (uint64_t(p.lk - 1) * v_r + SP_D) * 4u);
    const int words = (p.lk + 31) / 32;
    const __amdgpu_buffer_rsrc_t rsM = make_rsrc(static_cast<const unsigned*>(p.mask), (uint64_t(p.lq - 1) * uint64_t(p.m_sq) + words) * 4u);

    // this lane's 16-byte chunks of a 512-byte row, in the order its slot reads them (conflict-free, see the header)
    unsigned lane_off[NR];
    f32x2v q[2 * NR], o[2 * NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int chunk = t + LPQ * (r ^ rot);
        lane_off[r] = unsigned(chunk) * 16u;
        const float4 v = bload4(rsQ, q_live ? unsigned(qi * q_r + 4 * chunk) * 4u : OOB, 0);
        q[2 * r] = f32x2v{v.x * p.scale_log2e, v.y * p.scale_log2e};
        q[2 * r + 1] = f32x2v{v.z * p.scale_log2e, v.w * p.scale_log2e};
        o[2 * r] = o[2 * r + 1] = f32x2v{0.f, 0.f};
    }
    float m_run = -1e30f, l_run = 0.f;   // finite "minus infinity": exp2(-inf - m_run) stays 0 for a slot that has not seen a key yet
    constexpr float RESCALE_THR = 32.0f;

    // LDS-DMA: a wave requests 2 KiB of the K and 2 KiB of the V tile (four rows each), 1 KiB = two rows per request; lane l's
    // 16 bytes land at the request's LDS address + 16 l.  Rows past the last key come back as zeros (descriptor range check).
    const int d_row = 4 * wave + (lane >> 5), d_col = (lane & 31) * 4;
    auto request = [&](int kt, auto* dst) {   // dst: sp_buf0 or sp_buf1
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = kt * SP_TILE + d_row + 2 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (sp_lds_ptr)&dst[0][(4 * wave + 2 * i) * SP_D], 16,
                                                     unsigned(key * k_r + d_col) * 4u, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (sp_lds_ptr)&dst[1][(4 * wave + 2 * i) * SP_D], 16,
                                                     unsigned(key * v_r + d_col) * 4u, 0, 0, 0);
        }
    };
    // this query's mask bits of a tile: bit j set = key 64 kt + j is ALLOWED (the packed mask holds 1 = blocked; keys past lk,
    // words past the row and queries past lq allow nothing).  Unconditional loads from clamped offsets (OOB reads 0).
    auto mask_words = [&](int kt, unsigned& w0, unsigned& w1) {
        const unsigned base = unsigned(int64_t(qi) * p.m_sq + 2 * kt) * 4u;
        const unsigned o0 = (q_live && 2 * kt < words) ? base : OOB, o1 = (q_live && 2 * kt + 1 < words) ? base + 4u : OOB;
        w0 = __builtin_bit_cast(unsigned, bload1(rsM, o0));
        w1 = __builtin_bit_cast(unsigned, bload1(rsM, o1));
    };
    auto allowed_of = [&](int kt, unsigned w0, unsigned w1) -> unsigned long long {
        const int valid = p.lk - kt * SP_TILE;   // keys of this tile (>= 1)
        const unsigned long long in_range = valid >= 64 ? ~0ull : ((1ull << valid) - 1ull);
        const unsigned long long blocked = (static_cast<unsigned long long>(w1) << 32) | w0;
        return q_live ? (~blocked & in_range) : 0ull;
    };

    // the pairs of one tile (its K | V in `cur`), while the next tile's requests and mask words are in flight
    auto walk = [&](const float (*cur)[SP_BUF], unsigned long long rem) {
        const char* kbase = reinterpret_cast<const char*>(&cur[0][0]);
        const char* vbase = reinterpret_cast<const char*>(&cur[1][0]);
        if (!__any(rem != 0ull)) return;
        do {   // bottom-tested: with the exit test in the loop header hipcc parks o in a second register set every trip
            const bool active = rem != 0ull;
            const unsigned j = active ? unsigned(__builtin_ctzll(rem)) : 0u;   // an idle slot reads row 0: harmless, its score is -inf
            rem &= rem - 1ull;
            const unsigned row = j * (SP_D * 4u);
            f32x2v acc = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const float4 k4 = *reinterpret_cast<const float4*>(kbase + row + lane_off[r]);
                acc = __builtin_elementwise_fma(f32x2v{k4.x, k4.y}, q[2 * r], acc);
                acc = __builtin_elementwise_fma(f32x2v{k4.z, k4.w}, q[2 * r + 1], acc);
            }
            float s = slot_sum<LPQ>(acc.x + acc.y);
            s = active ? s : -INFINITY;
            if (__any(s > m_run + RESCALE_THR)) {
                const float m_new = fmaxf(m_run, s);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
                m_run = m_new;
                const f32x2v aa = {alpha, alpha};
#pragma unroll
                for (int e = 0; e < 2 * NR; ++e) asm("v_pk_mul_f32 %0, %0, %1" : "+v"(o[e]) : "v"(aa));   // in place: no register shuffle
            }
            const float pr = __builtin_amdgcn_exp2f(s - m_run);
            l_run += pr;
            const f32x2v pp = {pr, pr};
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const float4 v4 = *reinterpret_cast<const float4*>(vbase + row + lane_off[r]);
                asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o[2 * r]) : "v"(f32x2v{v4.x, v4.y}), "v"(pp));
                asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(o[2 * r + 1]) : "v"(f32x2v{v4.z, v4.w}), "v"(pp));
            }
        } while (__any(rem != 0ull));
    };

    const int nt = (p.lk + SP_TILE - 1) / SP_TILE;
    unsigned a0, a1, b0 = 0, b1 = 0;
    request(0, sp_buf0);
    mask_words(0, a0, a1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nt; kt += 2) {   // two tiles per trip: the buffer of a step is a compile-time variable
        // the mask words of THIS tile first (requested a tile ago: landed before the last barrier) -- hipcc waits vmcnt(0) for a
        // load it issued before a loop back edge, and behind the requests below that wait would expose their whole latency
        unsigned long long rem = allowed_of(kt, a0, a1);
        asm volatile("" : "+v"(rem));
        if (kt + 1 < nt) {
            request(kt + 1, sp_buf1);
            mask_words(kt + 1, b0, b1);
        }
        walk(sp_buf0, rem);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next tile has landed ...
        __syncthreads();                                     // ... and everybody's; everybody is done with this tile
        if (kt + 1 >= nt) break;
        rem = allowed_of(kt + 1, b0, b1);
        asm volatile("" : "+v"(rem));
        if (kt + 2 < nt) {
            request(kt + 2, sp_buf0);
            mask_words(kt + 2, a0, a1);
        }
        walk(sp_buf1, rem);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (!q_live) return;
    const float inv_l = 1.0f / l_run;   // a row without one allowed key: 0 / 0 = NaN, as the reference's softmax of an all -inf row
    float* Orow = p.O + int64_t(b) * p.lay.o_b + int64_t(h) * p.lay.o_h + int64_t(qi) * p.lay.o_r;
#pragma unroll
    for (int r = 0; r < NR; ++r)
        *reinterpret_cast<float4*>(Orow + lane_off[r] / 4) =
            make_float4(o[2 * r].x * inv_l, o[2 * r].y * inv_l, o[2 * r + 1].x * inv_l, o[2 * r + 1].y * inv_l);
}

}  // namespace

// Shape + mask rule only (a sample's bits must not depend on its batch): the caller flagged the shared bit-packed mask as sparse
// per row (lamp_mask.flags & LAMP_MASK_SPARSE_ROWS), heads of exactly 128 dimensions, no map output, and enough keys for the
// per-tile walk to pay (>= 16 tiles).
bool attn_sparse_applies(const AttnParams& p) {
    if (!p.sparse_rows || p.P || p.lse || !p.V || !p.O) return false;
    if (p.mask_kind != LAMP_MASK_BITS_U32 || p.m_sb != 0 || p.kv_len) return false;
    if (p.dk != SP_D || p.dv != SP_D) return false;
    if (p.lk < 16 * SP_TILE) return false;
    if (((p.lay.o_b | p.lay.o_h | p.lay.o_r) & 3) || !aligned16(p.O)) return false;
    return true;
}

#ifdef LAMP_TUNING
int g_sparse_lpq = 0;   // 4 / 8: force the lanes-per-query variant (tools/bench_kernels.py sparse_rows)
extern "C" __attribute__((visibility("default"))) void lamp_debug_sparse_lpq(int v) { g_sparse_lpq = v; }
#else
constexpr int g_sparse_lpq = 0;
#endif

int launch_attn_sparse(const AttnParams& p, hipStream_t s) {
    const int lpq = g_sparse_lpq == 4 || g_sparse_lpq == 8 ? g_sparse_lpq : SP_LPQ_DEFAULT;
    const int qblk = (64 / lpq) * SP_WAVES;
    const int64_t nwg = int64_t((p.lq + qblk - 1) / qblk) * p.H * p.B;
    if (nwg > 0x7fffffffLL) return LAMP_E_DIMS;
    if (lpq == 4)
        hipLaunchKernelGGL(attn_sparse_kernel<4>, dim3(unsigned(nwg)), dim3(SP_WAVES * 64), 0, s, p);
    else
        hipLaunchKernelGGL(attn_sparse_kernel<8>, dim3(unsigned(nwg)), dim3(SP_WAVES * 64), 0, s, p);
    return int(hipGetLastError());
}

}  // namespace lamp
