// fp32 GEMM  C = act(A . W^T + bias) + R  on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Both operands are K-contiguous ("NT"): A is [M, K] activations, W is nn.Linear's native
// [N, K].  A BMxBNxBK block tile is staged global -> registers -> LDS (double buffered, one
// barrier per K step, next tile's global loads in flight under the MFMAs); each 64-lane wave owns
// a (MI*32)x(NI*32) sub-tile held in MI*NI 32x32 accumulators.
//
// Fragment trick: the 32x32x2 MFMA takes ONE f32 per lane for A and B -- lane l supplies
// A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31].  Any permutation of k is legal as long as A and B use
// the same one, so each lane reads 4 CONSECUTIVE k of its row with one ds_read_b128
// (k = 8c + 4*(l>>5) + j) and feeds component j to MFMA step j: one 16-byte LDS read per four
// MFMAs per operand instead of four 4-byte reads.  LDS rows are padded (GemmTile::LDS_STRIDE) so that the
// b128 fragment reads of a 16-lane group land on 16 distinct 16-byte slots.
//
// All global traffic goes through per-tile buffer descriptors (buffer_load/store ... offen): the
// hardware range check returns 0 for rows past M / N and drops stores to them, so the hot loop has
// no bounds branches, no clamps and only 32-bit offsets (guarded plain loads compile to one branch
// plus a full vmcnt(0) drain per element; 64-bit per-element addresses spill).
//
// The k-accumulation order of every output element is fixed by (K, BK) alone -- never by M, the
// grid or the batch -- so a sample's result is bit-identical however the batch is sharded.
#include <type_traits>

#include "lamp_kernels.h"

namespace lamp {

// MF = edge of the MFMA block a wave tile is built from: 16 (v_mfma_f32_16x16x4_f32, 4 accumulator registers
// per block; the production tiles -- see launch_gemm) or 32 (v_mfma_f32_32x32x2_f32, 16 registers per block;
// kept as forced configurations for comparison).  Both issue 64 FLOP/cycle/SIMD.
// Tile-index arithmetic without integer division.  A runtime 32-bit division is ~40 scalar / vector instructions on gfx950;
// the five of them in the tile walk were ~0.4 us of every launch's prologue, before its first load is even requested.
// n / d for 0 <= n < 2^31 with host-made magic numbers: mul = ceil(2^(31 + l) / d), l = ceil(log2 d):  q = umulhi(n, mul) >> (l - 1).
struct FastDiv {
    unsigned mul, shift;   // mul == 0: d == 1
};
static inline FastDiv make_fastdiv(unsigned d) {
    if (d <= 1) return FastDiv{0u, 0u};
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    return FastDiv{unsigned(((1ull << (31 + l)) + d - 1) / d), l - 1};
}
__device__ __forceinline__ int fdiv(int n, FastDiv f) { return f.mul ? int(__umulhi(unsigned(n), f.mul) >> f.shift) : n; }
// divisor 1 .. 8 (the row-panels of a group), same scheme with the magic numbers as literals
__device__ __forceinline__ int div_1_to_8(int n, int d) {
    if ((d & (d - 1)) == 0) return n >> (31 - __builtin_clz(unsigned(d)));
    const unsigned mul = d == 5 ? 0xCCCCCCCDu : (d == 7 ? 0x92492493u : 0xAAAAAAABu);   // 3 and 6 share 0xAAAAAAAB
    return int(__umulhi(unsigned(n), mul) >> (d == 3 ? 1 : 2));
}

constexpr int gemm_min_waves(int bm, int bn, int bk, int mf) { return (mf == 16 && bm == 64 && bn == 64 && bk == 16) ? 5 : 1; }

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int MF = 32>
struct GemmTile {
    static constexpr int NT = WAVES_M * WAVES_N * 64;
    static constexpr int WTM = BM / WAVES_M;
    static constexpr int WTN = BN / WAVES_N;
    static constexpr int MI = WTM / MF;
    static constexpr int NI = WTN / MF;
    // LDS row padding.  ds_read_b128 serves a wave in four groups of sixteen lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19,
    // 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- not in four quarters.  For the 16x16x4 fragments (lane = row
    // l & 15, k-quad l >> 4) a group therefore mixes eight rows at quad q with eight rows at quad q + 1: with rows padded by
    // 4 floats (round 1-3: an odd number of 16-byte slots per row) two of them always meet on a slot -- SQ_LDS_BANK_CONFLICT
    // was a third of the LDS cycles of the 32x64x32 tile (profiles/r04_chain_pmc.txt) -- with 8 floats none do.  The
    // 32x32x2 fragments (row l & 31) are conflict-free at 4.
#ifndef LAMP_LDS_PAD16
#define LAMP_LDS_PAD16 8
#endif
    static constexpr int LDS_STRIDE = BK + (MF == 16 ? LAMP_LDS_PAD16 : 4);
    static constexpr int A_LD = BM * BK / 4 / NT;  // float4 loads per thread per tile
    static constexpr int B_LD = BN * BK / 4 / NT;
    static constexpr size_t LDS_BYTES = size_t(2) * (BM + BN) * LDS_STRIDE * sizeof(float);
    // Waves per SIMD the register allocator must leave room for.  The 64x64x16 tile serves launches of ~1200 tiles
    // (encoder FFN at batch 32: 1208): five workgroups per CU hold them all at once, four leave a second, mostly empty
    // round (44 -> 51 us when the 16-byte epilogue operands pushed the kernel from 92 to 100 registers).
    static constexpr int MIN_WAVES = gemm_min_waves(BM, BN, BK, MF);
    static_assert(MF == 32 || MF == 16, "MFMA block edge");
    static_assert(WTM % MF == 0 && WTN % MF == 0, "wave tile must be a multiple of the MFMA block");
    static_assert(BK % (MF == 32 ? 8 : 16) == 0, "BK must cover whole fragment reads");
    static_assert((BM * BK / 4) % NT == 0 && (BN * BK / 4) % NT == 0, "staging must divide evenly");
    static_assert(BK % 8 == 0, "BK must be a multiple of 8");
};

// RPRE: the epilogue's bias and residual values are fetched BEFORE the main loop (their round trip to L2 / the
// Infinity Cache then hides under the MFMAs instead of standing between the last k-step and the stores).  Used by the
// small tiles, whose accumulators -- hence the prefetched values -- are few registers; the large tiles amortise the
// round trip over 8-16x more matrix work per wave and keep the registers for occupancy.
// VEC: bias / residual / C move as 16-byte accesses (N, ldc, ldr multiples of 4 and 16-byte aligned bases -- every shape
// of the forward); the scalar instantiation serves odd widths.

__device__ __forceinline__ const float* uniform_ptr(const float* q) {
    const uint64_t b = reinterpret_cast<uint64_t>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(b)), hi = __builtin_amdgcn_readfirstlane(unsigned(b >> 32));
    return reinterpret_cast<const float*>((uint64_t(hi) << 32) | lo);
}
// The whole tile program: one tile per workgroup, derived from blockIdx.  (Staging variants that lost their measurements --
// LDS-DMA with compiler-scheduled / inline-assembly fragment reads, W fragments straight from global memory or from a packed
// copy, two dependent GEMMs in one persistent launch -- are not carried here any more: csrc/experiments/README.md.)
// RGATHER: the residual is gathered from two tables through per-row indices (GemmParams::rg_tok) instead of read from R.
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool KTAIL, int MF, bool RPRE, bool VEC, bool RGATHER>
__device__ __forceinline__ void gemm_body(const GemmParams& p, int tiles_n_seg, int tiles_n, int tiles_m, int panel_split,
                                          FastDiv fd_group, FastDiv fd_seg) {
    using T = GemmTile<BM, BN, BK, WAVES_M, WAVES_N, MF>;
    // lane -> (row within an MFMA block, which group of 4 consecutive k this lane's b128 read covers)
    constexpr int KQ = 64 / MF;             // 2 for 32x32x2, 4 for 16x16x4
    constexpr int KCH = 4 * KQ;             // k covered by one round of fragment reads: 8 or 16
    constexpr int NACC = MF == 32 ? 16 : 4; // accumulator registers per block
    using acc_t = typename std::conditional<MF == 32, f32x16, f32x4>::type;
    constexpr int S = T::LDS_STRIDE;
    constexpr int C4 = BK / 4;  // float4 per tile row
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;               // [2][BM][S]
    float* Bs = smem + 2 * BM * S;  // [2][BN][S]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & (MF - 1), hi = lane / MF;  // row-in-block, k-group

#ifdef LAMP_TUNING
    const unsigned long long t_entry = p.trace ? wall_clock64() : 0ull;
#endif
    // Work-item order.  (1) xcd_remap: every XCD (workgroup b runs on XCD b % 8) gets a CONTIGUOUS range of items, so
    // neighbours in item order share one 4 MiB L2.  (2) Inside that order the tiles are walked in groups of GROUP_M
    // row-panels, column-panel by column-panel: the ~32-64 tiles an XCD has in flight then cover GROUP_M row-panels of
    // A x a few column-panels of W (a working set of 2-3 MiB at K = 512), so A is fetched about once and W about once
    // per group -- instead of the whole W once per ROW-PANEL, which is what the plain row-major walk cost as soon as W
    // alone filled the L2 (K/V projection of reuters: 291 MB fetched for 24 MB of compulsory traffic, round 1).
    // Placement only: every output element is computed by exactly one tile with a k-order fixed by (K, BK).
    // (3) panel_split: the launch gave every XCD WHOLE row-panels (all their column tiles; workgroups past an XCD's
    // share exit), so a row of C is written by one XCD only -- the LayerNorm / attention / GEMM that reads it next
    // walks rows in the same XCD order and finds it in its own L2.  Used when it costs no extra round of tiles.
    constexpr int GROUP_M = 8;
    // Row count from device memory (ragged batches: the packed token rows of this micro-batch, known only on the device):
    // the launch was sized for the host's upper bound p.M; the tile walk below is rebuilt from the real count so that
    // the live tiles still spread over all eight XCDs, and workgroups past it exit.
    int64_t M = p.M;
    if (p.m_dev) {
        M = *p.m_dev;
        if (M > p.M) M = p.M;
        tiles_m = int((M + BM - 1) / BM);
        const int nwg = tiles_m * tiles_n, pan_xcd = (tiles_m + 7) >> 3;
        panel_split = tiles_m >= 16 && (pan_xcd * tiles_n + 31) / 32 <= ((nwg + 7) / 8 + 31) / 32;
    }
    int item = 0, pan0 = 0, pan1 = tiles_m;
    int tn_all = 0, tm = 0;
    if (panel_split) {
        const int q = tiles_m >> 3, r = tiles_m & 7, xcd = blockIdx.x & 7;
        pan0 = xcd * q + (xcd < r ? xcd : r);
        pan1 = pan0 + q + (xcd < r ? 1 : 0);
        item = blockIdx.x >> 3;
        if (item >= (pan1 - pan0) * tiles_n) return;
    } else {
        const int nwg = p.m_dev ? tiles_m * tiles_n : int(gridDim.x);
        if (int(blockIdx.x) >= nwg) return;
        item = xcd_remap(blockIdx.x, nwg);
        pan0 = 0;
        pan1 = tiles_m;
    }
    if (p.walk_gn > 0) {
        // (4) W-resident walk, for weight matrices that crowd the 4 MiB L2 on their own (delicious' 8 MB FFN weights):
        // groups of walk_gn COLUMN panels (~2 MiB of W), and inside a group row-panel by row-panel -- the group's W
        // stays in the L2 while the XCD's A panels stream past it once per group, instead of every group of 8 row-panels
        // re-fetching all of W and, with A + W over 4 MiB in flight, its own A panels once per column step
        // (profiles/r02_hbm_traffic.txt: 1231 MB fetched for 137 MB of operands).  Placement only, as above.
        const int rows = pan1 - pan0;
        const int group_sz = rows * p.walk_gn;
        const int grp = item / group_sz;
        const int in_grp = item - grp * group_sz;
        const int first_n = grp * p.walk_gn;
        const int gn = tiles_n - first_n < p.walk_gn ? tiles_n - first_n : p.walk_gn;
        const int tml = in_grp / gn;
        tm = pan0 + tml;
        tn_all = first_n + (in_grp - tml * gn);
    } else {
        const int group_sz = GROUP_M * tiles_n;
        const int grp = fdiv(item, fd_group);   // item / group_sz
        const int in_grp = item - grp * group_sz;
        const int first_m = pan0 + grp * GROUP_M;
        const int gm = pan1 - first_m < GROUP_M ? pan1 - first_m : GROUP_M;
        tn_all = div_1_to_8(in_grp, gm);
        tm = first_m + (in_grp - tn_all * gm);
    }
    const int seg = fdiv(tn_all, fd_seg);       // tn_all / tiles_n_seg
    const int tn = tn_all - seg * tiles_n_seg;
    const int64_t m0 = int64_t(tm) * BM;
    const int n0 = tn * BN;
    const int rows_m = int(M - m0 < BM ? M - m0 : BM);  // valid rows / columns of this tile
    const int rows_n = p.N - n0 < BN ? p.N - n0 : BN;

    const int lda = int(p.lda), ldw = int(p.ldw);
    const float* Abase = p.A;
    if (p.A_dense) {
        if (p.m_dev[0] == p.m_dev[1]) Abase = p.A_dense;
    }
    // (the tile origin is wave-uniform, but parts of its arithmetic run on the vector ALU; an LDS-DMA load through a descriptor
    // held in vector registers compiles to a waterfall loop -- hand the compiler scalars)
    const __amdgpu_buffer_rsrc_t rsA =
        make_rsrc(uniform_ptr(Abase + m0 * p.lda), __builtin_amdgcn_readfirstlane(unsigned((uint64_t(rows_m - 1) * lda + p.K) * 4u)));
    const __amdgpu_buffer_rsrc_t rsW =
        make_rsrc(uniform_ptr(p.W[seg] + int64_t(n0) * p.ldw), __builtin_amdgcn_readfirstlane(unsigned((uint64_t(rows_n - 1) * ldw + p.K) * 4u)));

    acc_t acc[T::MI][T::NI];
#pragma unroll
    for (int i = 0; i < T::MI; ++i)
#pragma unroll
        for (int j = 0; j < T::NI; ++j)
#pragma unroll
            for (int r = 0; r < NACC; ++r) acc[i][j][r] = 0.f;

    // Two register sets: tiles kt+1 and kt+2 are in flight while tile kt is multiplied (prefetch distance
    // of two K steps -- with a single workgroup on a CU, as on the M = B*L decoder shapes, one step of MFMAs
    // does not cover an L2/MALL round trip).
    float4 ra0[T::A_LD], rb0[T::B_LD], ra1[T::A_LD], rb1[T::B_LD];
    unsigned voa[T::A_LD], vob[T::B_LD];  // byte offsets of this thread's float4s inside the tile
#pragma unroll
    for (int i = 0; i < T::A_LD; ++i) {
        const int idx = tid + i * T::NT;
        const int row = idx / C4, c4 = idx - row * C4;
        voa[i] = unsigned(row * lda + c4 * 4) * 4u;
    }
#pragma unroll
    for (int i = 0; i < T::B_LD; ++i) {
        const int idx = tid + i * T::NT;
        const int row = idx / C4, c4 = idx - row * C4;
        vob[i] = unsigned(row * ldw + c4 * 4) * 4u;
    }

    auto gload = [&](int k0, float4 (&ra)[T::A_LD], float4 (&rb)[T::B_LD]) {
        if constexpr (KTAIL) {
            // K is not a multiple of BK: columns past K must read as 0 (the row range check cannot see them)
#pragma unroll
            for (int i = 0; i < T::A_LD; ++i) {
                const int k = k0 + ((tid + i * T::NT) % C4) * 4;
                ra[i] = bload4(rsA, k < p.K ? voa[i] + unsigned(k0) * 4u : OOB, 0);
            }
#pragma unroll
            for (int i = 0; i < T::B_LD; ++i) {
                const int k = k0 + ((tid + i * T::NT) % C4) * 4;
                rb[i] = bload4(rsW, k < p.K ? vob[i] + unsigned(k0) * 4u : OOB, 0);
            }
        } else {
            const unsigned so = unsigned(k0) * 4u;  // uniform -> soffset
#pragma unroll
            for (int i = 0; i < T::A_LD; ++i) ra[i] = bload4(rsA, voa[i], so);
#pragma unroll
            for (int i = 0; i < T::B_LD; ++i) rb[i] = bload4(rsW, vob[i], so);
        }
    };
    auto lstore = [&](int buf, const float4 (&ra)[T::A_LD], const float4 (&rb)[T::B_LD]) {
        float* a = As + buf * BM * S;
        float* b = Bs + buf * BN * S;
#pragma unroll
        for (int i = 0; i < T::A_LD; ++i) {
            const int idx = tid + i * T::NT;
            const int row = idx / C4, c4 = idx - row * C4;
            *reinterpret_cast<float4*>(a + row * S + c4 * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < T::B_LD; ++i) {
            const int idx = tid + i * T::NT;
            const int row = idx / C4, c4 = idx - row * C4;
            *reinterpret_cast<float4*>(b + row * S + c4 * 4) = rb[i];
        }
    };
    auto mfma_chunk = [&](const float4 (&fa)[T::MI], const float4 (&fb)[T::NI]) {
#ifdef LAMP_SETPRIO   // experiment (profiles/r04_setprio.txt): raised wave priority around the MFMA run
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int i = 0; i < T::MI; ++i)
#pragma unroll
            for (int j = 0; j < T::NI; ++j) {
                if constexpr (MF == 32) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j].x, fa[i].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j].y, fa[i].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j].z, fa[i].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j].w, fa[i].w, acc[i][j], 0, 0, 0);
                } else {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j].x, fa[i].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j].y, fa[i].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j].z, fa[i].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j].w, fa[i].w, acc[i][j], 0, 0, 0);
                }
            }
#ifdef LAMP_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    auto compute = [&](int buf) {
        const float* a = As + buf * BM * S + (wm * T::WTM + l31) * S + hi * 4;
        const float* b = Bs + buf * BN * S + (wn * T::WTN + l31) * S + hi * 4;
#pragma unroll
        for (int c = 0; c < BK / KCH; ++c) {
            float4 fa[T::MI], fb[T::NI];
#pragma unroll
            for (int i = 0; i < T::MI; ++i) fa[i] = *reinterpret_cast<const float4*>(a + i * MF * S + c * KCH);
#pragma unroll
            for (int j = 0; j < T::NI; ++j) fb[j] = *reinterpret_cast<const float4*>(b + j * MF * S + c * KCH);
            mfma_chunk(fa, fb);
        }
    };

    const int nk = (p.K + BK - 1) / BK;
    gload(0, ra0, rb0);

    // Epilogue operands.  The products are issued TRANSPOSED -- W fragment as the MFMA's A operand, activation fragment as
    // its B operand -- so the accumulator block holds C^T: lane (m = lane & (MF-1), hi) owns, for ITS output row m, four
    // CONSECUTIVE output columns per register quad (16x16 block: columns 4 hi + r, r < 4; 32x32 block: 8 q + 4 hi + (r & 3)
    // for quad q = r >> 2).  Same products, same k-order, same bits as the untransposed issue; but bias, residual and C
    // now move as 16-byte accesses and the four lane groups of a row write 64 contiguous bytes per instruction, the
    // next block of the same wave the adjacent 64 (round 2 stored one float per lane: four rows x 64 B per
    // instruction; the 128x64 tile's WRITE_SIZE was 1.31x its output, profiles/r02_hbm_traffic.txt).
    // Stores to rows past M fall outside the descriptor and are dropped by the hardware; columns past N are steered
    // to an out-of-range offset.  N, ldc (and ldr) not multiples of 4, or unaligned bases, take the scalar path.
    const int ldc = int(p.ldc), ldr = int(p.ldr);
    const bool has_r = RGATHER || p.R != nullptr;
    const bool read_r = !RGATHER && p.R != nullptr;
    const __amdgpu_buffer_rsrc_t rsR =
        make_rsrc(read_r ? p.R + m0 * p.ldr + n0 : p.A, read_r ? (uint64_t(rows_m - 1) * ldr + rows_n) * 4u : 0);
    const float* bias = p.bias[seg];
    const __amdgpu_buffer_rsrc_t rsBias = make_rsrc(bias ? bias + n0 : p.A, bias ? uint64_t(rows_n) * 4u : 0);
    constexpr int NQ = NACC / 4;  // register quads (= float4 of consecutive columns) per block
    constexpr bool vec = VEC;
    const int lrow0 = wm * T::WTM + l31;
    const int lcol0 = wn * T::WTN + 4 * hi;
    auto load4 = [&](__amdgpu_buffer_rsrc_t rs, unsigned off, int lcol) -> float4 {  // off in floats; lcol: first column
        if constexpr (vec) return bload4(rs, lcol < rows_n ? off * 4u : OOB, 0);
        float4 v;
        v.x = bload1(rs, lcol + 0 < rows_n ? (off + 0) * 4u : OOB);
        v.y = bload1(rs, lcol + 1 < rows_n ? (off + 1) * 4u : OOB);
        v.z = bload1(rs, lcol + 2 < rows_n ? (off + 2) * 4u : OOB);
        v.w = bload1(rs, lcol + 3 < rows_n ? (off + 3) * 4u : OOB);
        return v;
    };
    // gathered residual: this lane's rows' table rows (row indices past the tile's rows read table row 0: their stores are dropped)
    const float* ge[RGATHER ? T::MI : 1];
    const float* gp[RGATHER ? T::MI : 1];
    if constexpr (RGATHER) {
#pragma unroll
        for (int i = 0; i < T::MI; ++i) {
            const int lrow = lrow0 + i * MF;
            const bool in = lrow < rows_m;
            const int tk = in ? p.rg_tok[m0 + lrow] : 0, ps = (in && p.rg_pos_table) ? p.rg_pos[m0 + lrow] : 0;
            ge[i] = p.rg_emb + int64_t(tk) * p.N + n0;
            gp[i] = p.rg_pos_table ? p.rg_pos_table + int64_t(ps) * p.N + n0 : nullptr;
        }
    }
    auto gathered = [&](int i, int lcol) -> float4 {   // emb[tok][col] (+ pos[p][col]): the add the gather kernel would have done
        if constexpr (RGATHER) {
            if (lcol >= rows_n) return make_float4(0.f, 0.f, 0.f, 0.f);
            float4 v = *reinterpret_cast<const float4*>(ge[i] + lcol);
            if (gp[i]) {
                const float4 w = *reinterpret_cast<const float4*>(gp[i] + lcol);
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            return v;
        }
        return make_float4(0.f, 0.f, 0.f, 0.f);
    };
    constexpr int NPRE = RPRE ? T::MI * T::NI * NQ : 1;
    float4 pre_r[NPRE], pre_b[RPRE ? T::NI * NQ : 1];
    if constexpr (RPRE) {
#pragma unroll
        for (int j = 0; j < T::NI; ++j)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int lcol = lcol0 + j * MF + 8 * q;
                pre_b[j * NQ + q] = load4(rsBias, unsigned(lcol), lcol);
#pragma unroll
                for (int i = 0; i < T::MI; ++i)
                    pre_r[(j * T::MI + i) * NQ + q] = RGATHER ? gathered(i, lcol) : load4(rsR, unsigned((lrow0 + i * MF) * ldr + lcol), lcol);
            }
    }

    lstore(0, ra0, rb0);
    if (nk > 1) gload(BK, ra0, rb0);      // tile 1 -> set 0
    if (nk > 2) gload(2 * BK, ra1, rb1);  // tile 2 -> set 1
    __syncthreads();
#ifdef LAMP_TUNING
    const unsigned long long t_loop = p.trace ? wall_clock64() : 0ull;
    const unsigned long long c_loop = p.trace ? __builtin_readcyclecounter() : 0ull;   // shader-clock cycles (s_memtime)
#endif

    for (int kt = 0; kt < nk; kt += 2) {
        // even step: tile kt in LDS[0]; tile kt+1 in set 0, tile kt+2 in set 1
        compute(0);
        if (kt + 1 < nk) lstore(1, ra0, rb0);
        if (kt + 3 < nk) gload((kt + 3) * BK, ra0, rb0);
        __syncthreads();
        if (kt + 1 >= nk) break;
        // odd step: tile kt+1 in LDS[1]; tile kt+2 in set 1, tile kt+3 in set 0
        compute(1);
        if (kt + 2 < nk) lstore(0, ra1, rb1);
        if (kt + 4 < nk) gload((kt + 4) * BK, ra1, rb1);
        __syncthreads();
    }
#ifdef LAMP_TUNING
    const unsigned long long t_epi = p.trace ? wall_clock64() : 0ull;
    const unsigned long long c_epi = p.trace ? __builtin_readcyclecounter() : 0ull;
#endif

    const __amdgpu_buffer_rsrc_t rsC =
        make_rsrc(p.C[seg] + m0 * p.ldc + n0, (uint64_t(rows_m - 1) * ldc + rows_n) * 4u);
    // The epilogue in four straight-line copies (residual or not, ReLU or not), chosen by two scalar branches: as selects on the
    // two flags it carried 16 v_cndmask and 16 canonicalising v_max per tile and wave, and a vector instruction is matrix-pipe
    // time on gfx950 (profiles/r05_mfma_chain.txt).  relu1: one v_max_f32 (fmaxf() quiets its operand with a second one first).
    auto relu1 = [](float x) {
        float y;
        asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
        return y;
    };
    auto emit = [&](auto with_r, auto with_relu) {
        constexpr bool WITH_R = decltype(with_r)::value, WITH_RELU = decltype(with_relu)::value;
    #pragma unroll
        for (int i = 0; i < T::MI; ++i) {
            const int lrow = lrow0 + i * MF;
    #pragma unroll
            for (int j = 0; j < T::NI; ++j)   // the blocks of one row back to back: adjacent 64-byte pieces of its lines
    #pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int lcol = lcol0 + j * MF + 8 * q;
                    float4 bv, res = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (RPRE) {
                        bv = pre_b[j * NQ + q];
                        res = pre_r[(j * T::MI + i) * NQ + q];
                    } else {
                        bv = load4(rsBias, unsigned(lcol), lcol);
                        if constexpr (WITH_R) res = RGATHER ? gathered(i, lcol) : load4(rsR, unsigned(lrow * ldr + lcol), lcol);
                    }
                    float4 v = make_float4(acc[i][j][4 * q + 0] + bv.x, acc[i][j][4 * q + 1] + bv.y,
                                           acc[i][j][4 * q + 2] + bv.z, acc[i][j][4 * q + 3] + bv.w);
                    if constexpr (WITH_RELU) v = make_float4(relu1(v.x), relu1(v.y), relu1(v.z), relu1(v.w));
                    if constexpr (WITH_R) v = make_float4(v.x + res.x, v.y + res.y, v.z + res.z, v.w + res.w);
                    const unsigned off = unsigned(lrow * ldc + lcol);
                    if constexpr (vec) {
                        bstore4(rsC, lcol < rows_n ? off * 4u : OOB, v);
                    } else {
                        bstore1(rsC, lcol + 0 < rows_n ? (off + 0) * 4u : OOB, v.x);
                        bstore1(rsC, lcol + 1 < rows_n ? (off + 1) * 4u : OOB, v.y);
                        bstore1(rsC, lcol + 2 < rows_n ? (off + 2) * 4u : OOB, v.z);
                        bstore1(rsC, lcol + 3 < rows_n ? (off + 3) * 4u : OOB, v.w);
                    }
                }
        }
    };
    if (has_r) {
        if (p.relu) emit(std::true_type{}, std::true_type{});
        else emit(std::true_type{}, std::false_type{});
    } else {
        if (p.relu) emit(std::false_type{}, std::true_type{});
        else emit(std::false_type{}, std::false_type{});
    }
#ifdef LAMP_TUNING
    if (p.trace && tid == 0) {
        unsigned long long* t = p.trace + size_t(blockIdx.x) * 8;
        t[0] = t_entry; t[1] = t_loop; t[2] = t_epi; t[3] = wall_clock64();
        t[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID (wave / SIMD / CU / SH / SE ids)
        t[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        t[6] = unsigned(item);
        t[7] = c_epi - c_loop;   // main loop in shader cycles: / ((t[2] - t[1]) x 10 ns) = the clock the loop ran at
    }
#endif
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool KTAIL, int MF, bool RPRE, bool VEC, bool RGATHER = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, (gemm_min_waves(BM, BN, BK, MF))) void gemm_nt_kernel(GemmParams p, int tiles_n_seg,
                                                                          int tiles_n, int tiles_m,
                                                                          int panel_split, FastDiv fd_group,
                                                                          FastDiv fd_seg) {
    gemm_body<BM, BN, BK, WAVES_M, WAVES_N, KTAIL, MF, RPRE, VEC, RGATHER>(p, tiles_n_seg, tiles_n, tiles_m, panel_split, fd_group, fd_seg);
}

#ifdef LAMP_TUNING
static int g_force_walk = -1;   // -1 = heuristic; 0 = row-panel groups; n > 0 = W-resident walk with n column panels per group
extern "C" __attribute__((visibility("default"))) void lamp_debug_force_gemm_walk(int gn) { g_force_walk = gn; }
static long long g_trace_slab_words = 0;   // capacity of one timeline slab (8 words per workgroup), see lamp_debug_set_gemm_trace
static size_t g_extra_lds = 0;
extern "C" __attribute__((visibility("default"))) void lamp_debug_set_gemm_extra_lds(int bytes) { g_extra_lds = size_t(bytes); }
#endif

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool KTAIL, int MF, bool VEC, bool RGATHER = false>
static int launch_cfg2(const GemmParams& p, hipStream_t s) {
    using T = GemmTile<BM, BN, BK, WAVES_M, WAVES_N, MF>;
    constexpr bool RPRE = T::MI * T::NI * (MF == 32 ? 16 : 4) <= 16;
    auto kern = gemm_nt_kernel<BM, BN, BK, WAVES_M, WAVES_N, KTAIL, MF, RPRE, VEC, RGATHER>;
    size_t LDS = T::LDS_BYTES;
    static AttrOnce once;
#ifdef LAMP_TUNING
    LDS += g_extra_lds;   // residency experiments: more LDS per workgroup = fewer workgroups per CU
    if (int e = once.set(reinterpret_cast<const void*>(kern), 160 * 1024)) return e;
#else
    if (int e = once.set(reinterpret_cast<const void*>(kern), LDS)) return e;
#endif
    // 32-bit in-tile byte offsets
    const int64_t ldmax = p.lda > p.ldw ? (p.lda > p.ldc ? p.lda : p.ldc) : (p.ldw > p.ldc ? p.ldw : p.ldc);
    if (ldmax * (BM > BN ? BM : BN) * 4 >= 0x7fffffffLL || (p.R && p.ldr * BM * 4 >= 0x7fffffffLL))
        return LAMP_E_UNSUPPORTED;
    const int64_t tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n_seg = (p.N + BN - 1) / BN;
    const int tiles_n = tiles_n_seg * p.nseg;
    int64_t nwg = tiles_m * tiles_n;
    // whole row-panels per XCD when the fullest XCD then needs no more rounds of tiles (32 CUs each) than an even split
    const int64_t pan_xcd = (tiles_m + 7) / 8;
    int panel_split = tiles_m >= 16 && (pan_xcd * tiles_n + 31) / 32 <= ((nwg + 7) / 8 + 31) / 32;
    if (panel_split) nwg = 8 * pan_xcd * tiles_n;
    if (p.m_dev) {
        // the kernel takes this decision again from the device-side row count: size the grid for either outcome
        const int64_t split_grid = 8 * pan_xcd * tiles_n;
        if (split_grid > nwg) nwg = split_grid;
        panel_split = 0;
    }
    if (nwg > 0x7fffffffLL) return LAMP_E_DIMS;
    GemmParams q = p;
    // W-resident walk (8 column panels per group) once the weight matrices of the launch reach 8 MiB -- twice an XCD's L2:
    // delicious' FFN and fused Q/K/V weights.  Measured (profiles/r03_gemm_walk.txt): 3-22 % fewer bytes fetched, the same
    // time (the kernel is MFMA-bound; everything hits the Infinity Cache).  Below that size the row-panel groups fetch less.
    q.walk_gn = (int64_t(p.nseg) * p.N * p.K * 4 >= (8ll << 20) && tiles_n > 8) ? 8 : 0;
#ifdef LAMP_TUNING
    if (p.trace && nwg * 8 > g_trace_slab_words) return LAMP_E_WORKSPACE;  // the timeline is indexed by blockIdx.x
    if (g_force_walk >= 0) q.walk_gn = g_force_walk < tiles_n ? g_force_walk : tiles_n;
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(T::NT), LDS, s, q, tiles_n_seg, tiles_n, int(tiles_m),
                       panel_split, make_fastdiv(unsigned(8 * tiles_n)), make_fastdiv(unsigned(tiles_n_seg)));
    return int(hipGetLastError());
}

// a K that is not a multiple of BK takes the KTAIL instantiation (columns past K read as zeros): same bits
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int MF = 32>
static int launch_cfg(const GemmParams& p, hipStream_t s) {
    if (p.rg_tok) {   // gathered residual: the 16-byte epilogue of a whole-k-tile product (every shape the forward uses it on)
        if (!p.vec_epilogue || (p.K % BK)) return LAMP_E_UNSUPPORTED;
        return launch_cfg2<BM, BN, BK, WAVES_M, WAVES_N, false, MF, true, true>(p, s);
    }
    if (p.vec_epilogue) {
        if (p.K % BK) return launch_cfg2<BM, BN, BK, WAVES_M, WAVES_N, true, MF, true>(p, s);
        return launch_cfg2<BM, BN, BK, WAVES_M, WAVES_N, false, MF, true>(p, s);
    }
    if (p.K % BK) return launch_cfg2<BM, BN, BK, WAVES_M, WAVES_N, true, MF, false>(p, s);
    return launch_cfg2<BM, BN, BK, WAVES_M, WAVES_N, false, MF, false>(p, s);
}

#ifdef LAMP_TUNING
// Tuning build only (liblamp_hip_tuning.so: tools/bench_kernels.py and the every-variant tests): force a tile
// configuration (0 = heuristic) and collect a per-workgroup timeline.  The production library has neither.
static int g_force_tile = 0;
static unsigned long long* g_gemm_trace = nullptr;  // n_slabs slabs of slab_words u64: launch i records into slab i % n_slabs,
static long long g_trace_slab = 0;                  // 8 words per workgroup (entry, loop start, loop end, exit: wall_clock64
static int g_trace_slabs = 0, g_trace_count = 0;    // ticks; HW_ID; XCC_ID; work item; main loop in shader cycles)
extern "C" __attribute__((visibility("default"))) void lamp_debug_force_gemm_tile(int cfg) { g_force_tile = cfg; }
extern "C" __attribute__((visibility("default"))) void lamp_debug_set_gemm_trace(unsigned long long* buf, long long slab_words, int n_slabs) {
    g_gemm_trace = buf;
    g_trace_slab = slab_words;
    g_trace_slab_words = slab_words;
    g_trace_slabs = n_slabs;
    g_trace_count = 0;
}
#endif

bool gemm_gathered_residual_ok(int N, int K, int64_t ldc, const float* bias, const float* C, const float* emb, const float* pos_table) {
    return N > 0 && !(N & 3) && K > 0 && (K % 32) == 0 && !(ldc & 3) && aligned16(C) && (!bias || aligned16(bias)) && emb && aligned16(emb) &&
           (!pos_table || aligned16(pos_table));
}

int launch_gemm(const GemmParams& p_in, hipStream_t s) {
    GemmParams p = p_in;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.nseg < 1 || p.nseg > GEMM_MAX_SEG) return LAMP_E_DIMS;
    if ((p.K & 3) || (p.lda & 3) || (p.ldw & 3)) return LAMP_E_ALIGN;
    if (!p.A || (p.A_dense && !p.m_dev)) return LAMP_E_NULL;
    if (p.rg_tok && (p.R || !p.rg_emb || p.nseg != 1 || (p.rg_pos_table && !p.rg_pos))) return LAMP_E_UNSUPPORTED;
    if (p.rg_tok && (!aligned16(p.rg_emb) || (p.rg_pos_table && !aligned16(p.rg_pos_table)))) return LAMP_E_ALIGN;
    if (!aligned16(p.A) || (p.A_dense && !aligned16(p.A_dense))) return LAMP_E_ALIGN;
    for (int i = 0; i < p.nseg; ++i) {
        if (!p.W[i] || !p.C[i]) return LAMP_E_NULL;
        if (!aligned16(p.W[i])) return LAMP_E_ALIGN;
    }
    const double flops = 2.0 * double(p.M) * p.N * p.nseg * p.K;
    const double bytes = 4.0 * (double(p.M) * p.K + double(p.N) * p.nseg * p.K +
                                double(p.M) * p.N * p.nseg * ((p.R || p.rg_tok) ? 2 : 1));
    ProfScope prof(LAMP_K_GEMM, flops, bytes, s);
    p.trace = nullptr;
    bool vec = !(p.N & 3) && !(p.ldc & 3) && (!p.R || (!(p.ldr & 3) && aligned16(p.R)));
    for (int i = 0; i < p.nseg; ++i) vec = vec && aligned16(p.C[i]) && (!p.bias[i] || aligned16(p.bias[i]));
    p.vec_epilogue = vec ? 1 : 0;
#ifdef LAMP_TUNING
    if (g_gemm_trace && g_trace_slabs > 0) {
        // upper bound of the grid over the tile menu: 32x64 tiles
        const long long wg_max = ((p.M + 31) / 32) * ((p.N + 63) / 64) * p.nseg;
        if (wg_max * 8 <= g_trace_slab) p.trace = g_gemm_trace + (g_trace_count % g_trace_slabs) * g_trace_slab;
        ++g_trace_count;
    }
    switch (g_force_tile) {
        case 1: return launch_cfg<128, 128, 32, 2, 2>(p, s);
        case 2: return launch_cfg<64, 64, 32, 2, 2>(p, s);
        case 3: return launch_cfg<128, 64, 32, 2, 2>(p, s);
        case 4: return launch_cfg<64, 128, 32, 2, 2>(p, s);
        case 5: return launch_cfg<128, 128, 16, 2, 2>(p, s);
        case 6: return launch_cfg<64, 64, 16, 2, 2>(p, s);
        case 7: return launch_cfg<128, 64, 16, 2, 2>(p, s);
        case 8: return launch_cfg<256, 128, 16, 4, 2>(p, s);
        case 9: return launch_cfg<32, 64, 32, 1, 4, 16>(p, s);   // waves 32x16 (2 blocks of 16x16)
        case 10: return launch_cfg<64, 32, 32, 4, 1, 16>(p, s);  // waves 16x32
        case 11: return launch_cfg<64, 64, 16, 2, 2, 16>(p, s);
        case 12: return launch_cfg<64, 64, 32, 2, 2, 16>(p, s);  // waves 32x32 as 2x2 blocks of 16x16
        case 13: return launch_cfg<32, 128, 32, 1, 4, 16>(p, s); // waves 32x32 as 2x2 blocks
        case 14: return launch_cfg<128, 128, 32, 2, 2, 16>(p, s);  // waves 64x64 as 4x4 blocks of 16x16
        case 15: return launch_cfg<128, 128, 16, 2, 2, 16>(p, s);
        case 16: return launch_cfg<128, 64, 32, 2, 2, 16>(p, s);
        case 17: return launch_cfg<64, 128, 32, 2, 2, 16>(p, s);
        case 18: return launch_cfg<128, 64, 16, 2, 2, 16>(p, s);
        default: break;
    }
#endif
    // Tile choice, from tools/bench_kernels.py on MI355X (profiles/r01_gemm_tiles.txt).  Every configuration the
    // heuristic may pick is built on the 16x16x4 MFMA, whose k-accumulation order (16c + {j, 4+j, 8+j, 12+j} for
    // j = 0..3 in every 16-chunk) does not depend on BM/BN/BK -- so the choice, which depends on M (i.e. on the
    // batch size), never changes a result bit: samples stay bit-identical across batch sizes and shards.
    // Why 16x16x4 and not 32x32x2 (same peak rate): the unit of serial work is one wave's accumulator chain over K.
    // A 32x32 block at K = 512 is an 8.2 us chain; the M = B*L decoder shapes have 1440 of them for 1024 SIMDs, so
    // some SIMD runs two in sequence (16.4 us) however the blocks are grouped into workgroups -- measured 21.5 us for
    // every 32x32x2 tile from 32x32 (1 wave) to 128x64.  16x16 blocks are 2 us chains: 5760 / 1024 -> 12.3 us,
    // measured 17.5 us.  On the large shapes 128x64x16 with 64x32 wave tiles (4x2 blocks) reaches 135-144 TFLOP/s,
    // also ahead of the best 32x32x2 tile (128x128x32: 128-139).  The 32x32x2 tiles exist in the tuning build only.
    auto tiles = [&](int bm, int bn) { return ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * p.nseg; };
    const int64_t t64 = tiles(64, 64);
    if (tiles(128, 64) >= 2048 && p.K >= 512) return launch_cfg<128, 64, 16, 2, 2, 16>(p, s);  // short K: fewer, deeper steps
    if (t64 >= 2048) return launch_cfg<64, 64, 32, 2, 2, 16>(p, s);
    if (t64 >= 1200) return launch_cfg<64, 64, 16, 2, 2, 16>(p, s);
    return launch_cfg<32, 64, 32, 1, 4, 16>(p, s);  // 4 waves of 32x16 (2 blocks each)
}

}  // namespace lamp
