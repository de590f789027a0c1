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
    // Direct-to-LDS staging (DMA = true, see gemm_nt_kernel): two or three unpadded [BM + BN][BK] images, filled by
    // buffer_load_dwordx4 ... lds in 1 KiB pieces (one wave instruction = 64 lanes x 16 B = RPP whole rows).
    static constexpr int QPR = BK / 4;            // 16-byte quads per row
    static constexpr int RPP = 64 / QPR;          // rows per 1 KiB piece
    static constexpr int A_PW = BM / RPP / (NT / 64);   // pieces per wave and k-step
    static constexpr int B_PW = BN / RPP / (NT / 64);
    static constexpr size_t DMA_STAGE_BYTES = size_t(BM + BN) * BK * sizeof(float);   // x 2 (DMA mode 1) or 3 (mode 2)
    static constexpr bool DMA_OK = (BK == 16 || BK == 32 || BK == 64) && BM % (RPP * (NT / 64)) == 0 && BN % (RPP * (NT / 64)) == 0;
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
//
// DMA (round 4): the K-step tiles go global -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction), no
// staging registers and no ds_write pass.  The LDS-DMA destination is wave-uniform base + lane * 16, so the image cannot be
// padded; rows are BK floats, stored back to back, and bank conflicts are avoided by an XOR on the 16-byte quad index:
// quad q of row r sits in slot q ^ f(r & 15), applied to the SOURCE address of the filling lane and to the fragment read
// (f: see dma_swz; checked against the ds_read_b128 lane groups of MI355X_MICROARCH.md: conflict-free for BK = 16 / 32 / 64
// and both MFMA shapes, where the padded image is 2-way).  A fragment still holds k = 16c + 4 hi + j: same products, same
// k-order, same bits as the register-staged kernel.
//   DMA = 1: fragment reads are ordinary loads.  hipcc (ROCm 7.2) orders every LDS read behind ALL earlier LDS-DMA
//            (s_waitcnt vmcnt(0) in front of the first ds_read that follows one -- it cannot tell the stages apart), so the
//            step is: drain, barrier, read ALL fragments of tile t, request tile t + 1, multiply.  Two stages; the request
//            is covered by one step of MFMAs (and by the other workgroups of the CU).
//   DMA = 2: fragment reads are inline-asm ds_read_b128 (invisible to that pass; waited for by hand, lgkmcnt + sched_barrier).
//            Three stages: while tile t is multiplied, t + 1 has landed or is landing and t + 2 is requested; ONE raw
//            s_barrier per K step, and the only vector-memory wait in the loop is a counted s_waitcnt vmcnt(pieces of one
//            tile) -- __syncthreads() would drain the DMA queue at every step.
template <int BK>
__device__ __forceinline__ int dma_swz(int r) {   // r = row & 15
    if constexpr (BK == 16) return (0x1230 >> (r & 12)) & 3;   // rows 0-3 / 4-7 / 8-11 / 12-15 -> 0, 3, 2, 1
    else if constexpr (BK == 32) return (r >> 1) & 7;
    else return r & 15;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ const float* uniform_ptr(const float* q) {
    const uint64_t b = reinterpret_cast<uint64_t>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(b)), hi = __builtin_amdgcn_readfirstlane(unsigned(b >> 32));
    return reinterpret_cast<const float*>((uint64_t(hi) << 32) | lo);
}
// One LDS-DMA instruction: 64 lanes x 16 bytes from per-lane buffer offsets to LDS [dst, dst + 1 KiB) in lane order (dst
// wave-uniform).  (A plain function, not code inside the kernel template: with the builtin spelled in the template hipcc's
// host pass silently drops the kernel's launch stub -- the library then fails to load with an undefined symbol.)
typedef __attribute__((address_space(3))) void* lds_ptr;
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, float* dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)dst, 16, voff, soff, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);   // hipcc moves register-only MFMAs across an asm wait otherwise
}
__device__ __forceinline__ f32x4 lds_read16(unsigned addr) {   // LDS byte address
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// The whole tile program.  EXPL = false: the tile is derived from blockIdx (gemm_nt_kernel: one tile per workgroup);
// EXPL = true: the caller names the tile (x_tm, x_tn) -- a persistent workgroup running one tile after another (gemm_pair_kernel).
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool KTAIL, int MF, bool RPRE, bool VEC, int DMA, bool EXPL>
__device__ __forceinline__ void gemm_body(const GemmParams& p, int tiles_n_seg, int tiles_n, int tiles_m, int panel_split,
                                          FastDiv fd_group, FastDiv fd_seg, int x_tm, int x_tn) {
    using T = GemmTile<BM, BN, BK, WAVES_M, WAVES_N, MF>;
    // lane -> (row within an MFMA block, which group of 4 consecutive k this lane's b128 read covers)
    constexpr int KQ = 64 / MF;             // 2 for 32x32x2, 4 for 16x16x4
    constexpr int KCH = 4 * KQ;             // k covered by one round of fragment reads: 8 or 16
    constexpr int NACC = MF == 32 ? 16 : 4; // accumulator registers per block
    using acc_t = typename std::conditional<MF == 32, f32x16, f32x4>::type;
    constexpr int S = T::LDS_STRIDE;
    constexpr int C4 = BK / 4;  // float4 per tile row
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;               // [2][BM][S]                    (DMA: stage s = smem + s * (BM + BN) * BK, A then W rows)
    float* Bs = smem + 2 * BM * S;  // [2][BN][S]
    // DMA = 3 (experiment): the W fragments come straight from global memory in MFMA layout (a lane's b128 = the four
    // consecutive k of ITS weight row: exactly its fragment of an NT product) -- no LDS pass for W at all; A is staged as ever.
    constexpr bool WDIR = DMA == 3;
    constexpr int LDMA = WDIR ? 0 : DMA;   // the LDS-DMA mode proper
    static_assert(!DMA || (T::DMA_OK && !KTAIL), "direct-to-LDS staging: whole 1 KiB pieces per wave, K a multiple of BK");

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
    int tn_all = x_tn, tm = x_tm;
    if constexpr (!EXPL) {
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
    }   // !EXPL
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
    // WDIR with a packed copy (lamp_pack_weight format 0: per 16 columns and 32 k the two fragment chunks lane by lane): a fragment
    // load is one contiguous KiB instead of 16 rows x 64 bytes
    const bool wpk = WDIR && MF == 16 && p.Wp[seg] != nullptr;
    const __amdgpu_buffer_rsrc_t rsW = wpk
        ? make_rsrc(uniform_ptr(p.Wp[seg]), __builtin_amdgcn_readfirstlane(unsigned(uint64_t(p.N) * uint64_t(p.K) * 4u)))
        : make_rsrc(uniform_ptr(p.W[seg] + int64_t(n0) * p.ldw), __builtin_amdgcn_readfirstlane(unsigned((uint64_t(rows_n - 1) * ldw + p.K) * 4u)));

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
    constexpr int NWF = WDIR ? (BK / KCH) * T::NI : 1;
    float4 wf0[NWF], wf1[NWF];            // WDIR: W fragments of the even / odd k-step, [chunk][block]
    unsigned vow[NWF];
    if constexpr (WDIR) {
#pragma unroll
        for (int c = 0; c < BK / KCH; ++c)
#pragma unroll
            for (int j = 0; j < T::NI; ++j)
                vow[c * T::NI + j] = wpk ? unsigned((n0 + wn * T::WTN + j * MF) / 16) * unsigned(p.K / 32) * 2048u + unsigned(c) * 1024u + unsigned(lane) * 16u
                                         : unsigned((wn * T::WTN + j * MF + l31) * ldw + c * KCH + hi * 4) * 4u;
    }
    auto wload = [&](int k0, float4 (&wf)[NWF]) {
        const unsigned so = wpk ? unsigned(k0) * 64u : unsigned(k0) * 4u;
#pragma unroll
        for (int x = 0; x < NWF; ++x) wf[x] = bload4(rsW, vow[x], so);
    };
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
            if constexpr (!WDIR) {
#pragma unroll
                for (int i = 0; i < T::B_LD; ++i) rb[i] = bload4(rsW, vob[i], so);
            }
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
        if constexpr (!WDIR) {
#pragma unroll
            for (int i = 0; i < T::B_LD; ++i) {
                const int idx = tid + i * T::NT;
                const int row = idx / C4, c4 = idx - row * C4;
                *reinterpret_cast<float4*>(b + row * S + c4 * 4) = rb[i];
            }
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
    auto compute = [&](int buf, const float4 (&wf)[NWF]) {
        const float* a = As + buf * BM * S + (wm * T::WTM + l31) * S + hi * 4;
        const float* b = Bs + buf * BN * S + (wn * T::WTN + l31) * S + hi * 4;
#pragma unroll
        for (int c = 0; c < BK / KCH; ++c) {
            float4 fa[T::MI], fb[T::NI];
#pragma unroll
            for (int i = 0; i < T::MI; ++i) fa[i] = *reinterpret_cast<const float4*>(a + i * MF * S + c * KCH);
#pragma unroll
            for (int j = 0; j < T::NI; ++j) {
                if constexpr (WDIR) fb[j] = wf[c * T::NI + j];
                else fb[j] = *reinterpret_cast<const float4*>(b + j * MF * S + c * KCH);
            }
            mfma_chunk(fa, fb);
        }
    };

    // ---- direct-to-LDS staging (DMA) ----
    constexpr int STAGE = (BM + BN) * BK;   // floats per LDS stage
    constexpr int PW = T::A_PW + T::B_PW;   // LDS-DMA instructions per wave and k-step
    unsigned dva[LDMA ? T::A_PW : 1], dvb[LDMA ? T::B_PW : 1];   // source byte offsets of this lane's quads inside the tile
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    if constexpr (LDMA) {
        const int prow = lane / T::QPR, pq = lane % T::QPR;   // row inside a piece, LDS slot inside the row
#pragma unroll
        for (int i = 0; i < T::A_PW; ++i) {
            const int row = (wave_u * T::A_PW + i) * T::RPP + prow;
            dva[i] = unsigned(row * lda + ((pq ^ dma_swz<BK>(row & 15)) << 2)) * 4u;
        }
#pragma unroll
        for (int i = 0; i < T::B_PW; ++i) {
            const int row = (wave_u * T::B_PW + i) * T::RPP + prow;
            dvb[i] = unsigned(row * ldw + ((pq ^ dma_swz<BK>(row & 15)) << 2)) * 4u;
        }
    }
    auto dma_stage = [&](int kt, int st) {
        const unsigned so = unsigned(kt * BK) * 4u;   // uniform -> soffset
        float* base = smem + st * STAGE;
#pragma unroll
        for (int i = 0; i < T::A_PW; ++i)
            lds_dma16(rsA, base + (wave_u * T::A_PW + i) * 256, dva[i], so);
#pragma unroll
        for (int i = 0; i < T::B_PW; ++i)
            lds_dma16(rsW, base + BM * BK + (wave_u * T::B_PW + i) * 256, dvb[i], so);
    };
    int qoff[BK / KCH];   // float offset of this lane's quad of chunk c inside its (swizzled) row
#pragma unroll
    for (int c = 0; c < BK / KCH; ++c) qoff[c] = ((c * KQ + hi) ^ dma_swz<BK>(l31 & 15)) << 2;
    constexpr int NCH = BK / KCH;
    // DMA = 1: all fragments of a stage into registers (ordinary loads), then -- by the caller -- the next request, then the MFMAs
    float4 fra[LDMA == 1 ? NCH : 1][T::MI], frb[LDMA == 1 ? NCH : 1][T::NI];
    auto dma_read = [&](int st) {
        const float* a = smem + st * STAGE + (wm * T::WTM + l31) * BK;
        const float* b = smem + st * STAGE + BM * BK + (wn * T::WTN + l31) * BK;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int i = 0; i < T::MI; ++i) fra[c][i] = *reinterpret_cast<const float4*>(a + i * MF * BK + qoff[c]);
#pragma unroll
            for (int j = 0; j < T::NI; ++j) frb[c][j] = *reinterpret_cast<const float4*>(b + j * MF * BK + qoff[c]);
        }
    };
    // DMA = 2: inline-asm reads, chunk c + 1 requested before chunk c is multiplied
    const unsigned lds0 = unsigned(reinterpret_cast<uintptr_t>((lds_ptr)smem));
    auto dma_compute_asm = [&](int st) {
        const unsigned a = lds0 + unsigned(st * STAGE + (wm * T::WTM + l31) * BK) * 4u;
        const unsigned b = lds0 + unsigned(st * STAGE + BM * BK + (wn * T::WTN + l31) * BK) * 4u;
        f32x4 ra[2][T::MI], rb[2][T::NI];
        auto rd = [&](int c, f32x4 (&xa)[T::MI], f32x4 (&xb)[T::NI]) {
#pragma unroll
            for (int i = 0; i < T::MI; ++i) xa[i] = lds_read16(a + unsigned(i * MF * BK + qoff[c]) * 4u);
#pragma unroll
            for (int j = 0; j < T::NI; ++j) xb[j] = lds_read16(b + unsigned(j * MF * BK + qoff[c]) * 4u);
        };
        rd(0, ra[0], rb[0]);
        static_for<0, NCH>([&](auto C) {
            constexpr int c = decltype(C)::value;
            if constexpr (c + 1 < NCH) {
                rd(c + 1, ra[(c + 1) & 1], rb[(c + 1) & 1]);
                wait_lgkmcnt<T::MI + T::NI>();
            } else {
                wait_lgkmcnt<0>();
            }
            float4 fa[T::MI], fb[T::NI];
#pragma unroll
            for (int i = 0; i < T::MI; ++i) fa[i] = make_float4(ra[c & 1][i].x, ra[c & 1][i].y, ra[c & 1][i].z, ra[c & 1][i].w);
#pragma unroll
            for (int j = 0; j < T::NI; ++j) fb[j] = make_float4(rb[c & 1][j].x, rb[c & 1][j].y, rb[c & 1][j].z, rb[c & 1][j].w);
            mfma_chunk(fa, fb);
            if constexpr (c + 1 < NCH) __builtin_amdgcn_sched_barrier(0);   // the next chunk's wait stays behind these MFMAs
        });
    };

    const int nk = (p.K + BK - 1) / BK;
    if constexpr (!LDMA) gload(0, ra0, rb0);
    if constexpr (WDIR) {
        wload(0, wf0);
        if (nk > 1) wload(BK, wf1);
    }

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
    const bool has_r = p.R != nullptr;
    const __amdgpu_buffer_rsrc_t rsR =
        make_rsrc(has_r ? p.R + m0 * p.ldr + n0 : p.A, has_r ? (uint64_t(rows_m - 1) * ldr + rows_n) * 4u : 0);
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
                    pre_r[(j * T::MI + i) * NQ + q] = load4(rsR, unsigned((lrow0 + i * MF) * ldr + lcol), lcol);
            }
    }

    if constexpr (LDMA) {
        // after the epilogue operands' loads: vector-memory operations retire in order, and the counted wait of step 0 must
        // not have younger register loads between itself and tile 0
        dma_stage(0, 0);
        if (LDMA == 2 && nk > 1) dma_stage(1, 1);
    } else {
        lstore(0, ra0, rb0);
        if (nk > 1) gload(BK, ra0, rb0);      // tile 1 -> set 0
        if (nk > 2) gload(2 * BK, ra1, rb1);  // tile 2 -> set 1
        __syncthreads();
    }
#ifdef LAMP_TUNING
    const unsigned long long t_loop = p.trace ? wall_clock64() : 0ull;
    const unsigned long long c_loop = p.trace ? __builtin_readcyclecounter() : 0ull;   // shader-clock cycles (s_memtime)
#endif

    if constexpr (LDMA == 1) {
        for (int kt = 0; kt < nk; ++kt) {
            wait_vmcnt<0>();                 // tile kt has landed (this wave's pieces; the barrier makes it everyone's)
            __builtin_amdgcn_s_barrier();    // ... and every wave is past its reads of tile kt - 1, whose stage is refilled below
            asm volatile("" ::: "memory");
            dma_read(kt & 1);
            if (kt + 1 < nk) dma_stage(kt + 1, (kt + 1) & 1);
#pragma unroll
            for (int c = 0; c < NCH; ++c) mfma_chunk(fra[c], frb[c]);
        }
    } else if constexpr (LDMA == 2) {
        // step kt: tile kt in stage kt % 3 (requested two steps ago), tile kt + 1 in flight, tile kt + 2 requested here --
        // into the stage tile kt - 1 was read from: every wave is past those reads once it has passed this step's barrier
        // (its MFMAs of step kt - 1 consumed them).  Loads retire in order, so "at most one tile's pieces outstanding" =
        // this wave's pieces of tile kt have landed; the barrier then makes that true for every wave's pieces.
        int st = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) wait_vmcnt<PW>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 2 < nk) dma_stage(kt + 2, st >= 1 ? st - 1 : 2);
            dma_compute_asm(st);
            st = st == 2 ? 0 : st + 1;
        }
    } else
    for (int kt = 0; kt < nk; kt += 2) {
        // even step: tile kt in LDS[0]; tile kt+1 in set 0, tile kt+2 in set 1
        compute(0, wf0);
        if (kt + 1 < nk) lstore(1, ra0, rb0);
        if (kt + 3 < nk) gload((kt + 3) * BK, ra0, rb0);
        if constexpr (WDIR) { if (kt + 2 < nk) wload((kt + 2) * BK, wf0); }   // into the set this step's MFMAs have read
        __syncthreads();
        if (kt + 1 >= nk) break;
        // odd step: tile kt+1 in LDS[1]; tile kt+2 in set 1, tile kt+3 in set 0
        compute(1, wf1);
        if (kt + 2 < nk) lstore(0, ra1, rb1);
        if (kt + 4 < nk) gload((kt + 4) * BK, ra1, rb1);
        if constexpr (WDIR) { if (kt + 3 < nk) wload((kt + 3) * BK, wf1); }
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
                        if constexpr (WITH_R) res = load4(rsR, unsigned(lrow * ldr + lcol), lcol);
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

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool KTAIL, int MF, bool RPRE, bool VEC, int DMA = 0>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, (gemm_min_waves(BM, BN, BK, MF))) void gemm_nt_kernel(GemmParams p, int tiles_n_seg,
                                                                          int tiles_n, int tiles_m,
                                                                          int panel_split, FastDiv fd_group,
                                                                          FastDiv fd_seg) {
    gemm_body<BM, BN, BK, WAVES_M, WAVES_N, KTAIL, MF, RPRE, VEC, DMA, false>(p, tiles_n_seg, tiles_n, tiles_m, panel_split, fd_group,
                                                                            fd_seg, 0, 0);
}

#ifdef LAMP_TUNING
// ---- EXPERIMENT (tuning build): two dependent GEMMs in ONE launch, row-panel-local hand-off through counters ----
//   H = act(X . W1^T + b1)   then   Y = H . W2^T + b2 (+ R)        (the encoder's FFN pair, lamp/SubLayers.py:133-142 before the LayerNorm)
// Persistent workgroups (at most the resident capacity), one task queue per XCD: row-panel tm belongs to XCD tm % 8 -- a workgroup
// reads its XCD from HW_REG_XCC_ID and only ever takes that XCD's tasks, so a panel's H rows are written and read through ONE L2
// (no cross-XCD coherence traffic: the producer waits for its stores' acknowledgements, vmcnt(0), and bumps the panel's counter;
// the consumer sees the counter reach the panel's tile count, invalidates its L1 and reads).  Queue order per XCD: all stage-0
// tiles (panel-major), then all stage-1 tiles in the same panel order -- when a stage-1 tile is taken every stage-0 tile of that XCD
// has been taken by a running workgroup that waits for nothing: no deadlock, whatever the residency.  The tiles are the ordinary
// tile program (gemm_body): same products, same k-order, same bits as two launches.
#ifndef PAIR_POLL_SLEEP
#define PAIR_POLL_SLEEP 32   // x 64 cycles between two looks at a panel counter
#endif
struct PairParams {
    const GemmParams* g;   // [2] in device memory (a kernarg array indexed at run time would be copied to scratch)
    int* queue;            // [8] task cursors + [8] = workgroups that have left: the last one out zeroes the cursors for the next launch
    int* done;             // [tiles_m] stage-0 tiles finished per row-panel, one counter per 128-byte line (640 pollers on two lines
                           // slowed every tile 3x); zeroed by the last workgroup out, like the cursors
    int tiles_m, tiles_n0, tiles_n1;
    FastDiv fd0, fd1;      // / tiles_n0, / tiles_n1
    unsigned long long* trace;   // nullable: per workgroup [wait cycles, tasks, first task start, last task end]
};
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int MF, bool RPRE, bool TRACE, int OCC>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, OCC) void gemm_pair_kernel(PairParams q) {
    __shared__ int task_s;
    const int tid = threadIdx.x;
    const int xcc = __builtin_amdgcn_readfirstlane(int(__builtin_amdgcn_s_getreg((31 << 11) | 20)) & 7);
    const int npan = (q.tiles_m - xcc + 7) >> 3;          // row-panels xcc, xcc + 8, ...
    const int n0 = npan * q.tiles_n0, n1 = npan * q.tiles_n1;
    const FastDiv fd0 = q.fd0, fd1 = q.fd1;
    const int want = q.tiles_n0;
    unsigned long long waited = 0, t_first = 0, t_last = 0;
    int ntask = 0;
    // Loop shape (matters): ONE single-thread region per iteration, between two workgroup barriers, and nothing divergent next to
    // the back edge.  With the panel counter's increment as a second `if (tid == 0)` at the END of the body hipcc merged it with the
    // next iteration's task fetch across the back edge and let the other lanes run ahead into the next barrier -- wave 0 then
    // arrives at that s_barrier twice per iteration and the workgroup hangs.
    int finished_tm = -1;   // stage-0 tile whose stores the closing barrier of the last iteration has seen complete
    for (;;) {
        if (tid == 0) {
            if (finished_tm >= 0) __hip_atomic_fetch_add(q.done + finished_tm * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            task_s = __hip_atomic_fetch_add(q.queue + xcc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const int task = __builtin_amdgcn_readfirstlane(task_s);
        if (task >= n0 + n1) break;
        const int stage = task >= n0 ? 1 : 0;
        const int t = stage ? task - n0 : task;
        const int tn_per = stage ? q.tiles_n1 : q.tiles_n0;
        const int pl = fdiv(t, stage ? fd1 : fd0);
        const int tn = t - pl * tn_per, tm = pl * 8 + xcc;
        if (stage) {
            if (tid == 0) {
                const unsigned long long c0 = TRACE ? __builtin_readcyclecounter() : 0ull;
                // (bounded: a broken hand-off must show up as a wrong result in the experiment's check, not as a hung GPU)
                for (int spin = 0; spin < (1 << 16) && __hip_atomic_load(q.done + tm * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++spin)
                    __builtin_amdgcn_s_sleep(PAIR_POLL_SLEEP);
                if constexpr (TRACE) waited += __builtin_readcyclecounter() - c0;
            }
            __syncthreads();
            asm volatile("buffer_inv sc0" ::: "memory");   // L1 only (H lines this CU may hold from an earlier launch / layer); producer and consumer share the L2
        }
        if constexpr (TRACE) { if (tid == 0 && ntask == 0) t_first = wall_clock64(); }
        const GemmParams& gp = q.g[stage];
        gemm_body<BM, BN, BK, WAVES_M, WAVES_N, false, MF, RPRE, true, 0, true>(gp, tn_per, tn_per, q.tiles_m, 0, fd0, stage ? fd1 : fd0, tm, tn);
        finished_tm = stage ? -1 : tm;
        if constexpr (TRACE) {
            ++ntask;
            if (tid == 0) t_last = wall_clock64();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's stores are in the L2 ...
        __syncthreads();                                    // ... everyone's are; and the next tile's staging may overwrite the LDS
    }
    if constexpr (TRACE) {
        if (tid == 0) {
            unsigned long long* tr = q.trace + size_t(blockIdx.x) * 4;
            tr[0] = waited; tr[1] = unsigned(ntask); tr[2] = t_first; tr[3] = t_last;
        }
    }
    // last workgroup out: every cursor and counter has been read for the last time -- zero them for the next launch
    if (tid == 0) task_s = __hip_atomic_fetch_add(q.queue + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (task_s == int(gridDim.x) - 1) {
        for (int i = tid; i < q.tiles_m; i += int(blockDim.x)) __hip_atomic_store(q.done + i * 32, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 9) __hip_atomic_store(q.queue + tid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// scratch (device, 256-byte aligned, zeroed ONCE by the caller): [0, 1024) two GemmParams, [1024, 1088) cursors, [2048, ...) panel counters
extern "C" __attribute__((visibility("default"))) int lamp_debug_ffn_pair_prepare(
    const float* x, long long M, int d, const float* w1, const float* b1, int dff, const float* w2, const float* b2, const float* r,
    float* H, float* Y, void* scratch, void* stream) {
    static_assert(2 * sizeof(GemmParams) <= 1024, "scratch layout");
    GemmParams g[2] = {};
    g[0].A = x; g[0].lda = d; g[0].M = M; g[0].K = d; g[0].N = dff; g[0].nseg = 1; g[0].W[0] = w1; g[0].ldw = d; g[0].bias[0] = b1;
    g[0].C[0] = H; g[0].ldc = dff; g[0].relu = 1; g[0].vec_epilogue = 1;
    g[1].A = H; g[1].lda = dff; g[1].M = M; g[1].K = dff; g[1].N = d; g[1].nseg = 1; g[1].W[0] = w2; g[1].ldw = dff; g[1].bias[0] = b2;
    g[1].C[0] = Y; g[1].ldc = d; g[1].R = r; g[1].ldr = d; g[1].vec_epilogue = 1;
    if (int e = int(hipMemsetAsync(scratch, 0, 2048 + 128 * size_t((M + 31) / 32), hipStream_t(stream)))) return e;
    if (int e = int(hipMemcpyAsync(scratch, g, sizeof g, hipMemcpyHostToDevice, hipStream_t(stream)))) return e;
    return int(hipStreamSynchronize(hipStream_t(stream)));
}
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int OCC>
static int launch_pair(long long M, int d, int dff, void* scratch, int wg_per_cu, unsigned long long* trace, hipStream_t s) {
    using T = GemmTile<BM, BN, BK, WAVES_M, WAVES_N, 16>;
    constexpr bool RPRE = T::MI * T::NI * 4 <= 16;
    auto kern = trace ? gemm_pair_kernel<BM, BN, BK, WAVES_M, WAVES_N, 16, RPRE, true, OCC> : gemm_pair_kernel<BM, BN, BK, WAVES_M, WAVES_N, 16, RPRE, false, OCC>;
    static AttrOnce once, once_t;
    if (int e = (trace ? once_t : once).set(reinterpret_cast<const void*>(kern), T::LDS_BYTES)) return e;
    PairParams q{};
    q.g = reinterpret_cast<const GemmParams*>(scratch);
    q.queue = reinterpret_cast<int*>(static_cast<char*>(scratch) + 1024);
    q.done = reinterpret_cast<int*>(static_cast<char*>(scratch) + 2048);
    q.tiles_m = int((M + BM - 1) / BM);
    q.tiles_n0 = (dff + BN - 1) / BN;
    q.tiles_n1 = (d + BN - 1) / BN;
    q.fd0 = make_fastdiv(unsigned(q.tiles_n0));
    q.fd1 = make_fastdiv(unsigned(q.tiles_n1));
    q.trace = trace;
    hipLaunchKernelGGL(kern, dim3(unsigned(256 * wg_per_cu)), dim3(T::NT), T::LDS_BYTES, s, q);
    return int(hipGetLastError());
}
// tile: 0 = 64x64x16 (the encoder FFN's tile; 5 waves per SIMD, a few spilled dwords), 3 = the same at 4 waves, 1 = 32x64x32, 2 = 128x64x16.  K must be a multiple of the tile's BK, N of 4.
extern "C" __attribute__((visibility("default"))) int lamp_debug_ffn_pair_launch(long long M, int d, int dff, void* scratch, int tile,
                                                                              int wg_per_cu, unsigned long long* trace, void* stream) {
    hipStream_t s = hipStream_t(stream);
    if ((d % 32) || (dff % 32)) return LAMP_E_UNSUPPORTED;
    switch (tile) {
        case 0: return launch_pair<64, 64, 16, 2, 2, 5>(M, d, dff, scratch, wg_per_cu, trace, s);
        case 1: return launch_pair<32, 64, 32, 1, 4, 4>(M, d, dff, scratch, wg_per_cu, trace, s);
        case 2: return launch_pair<128, 64, 16, 2, 2, 3>(M, d, dff, scratch, wg_per_cu, trace, s);
        case 3: return launch_pair<64, 64, 16, 2, 2, 4>(M, d, dff, scratch, wg_per_cu, trace, s);
    }
    return LAMP_E_UNSUPPORTED;
}
#endif

#ifdef LAMP_TUNING
static int g_force_walk = -1;   // -1 = heuristic; 0 = row-panel groups; n > 0 = W-resident walk with n column panels per group
extern "C" __attribute__((visibility("default"))) void lamp_debug_force_gemm_walk(int gn) { g_force_walk = gn; }
static long long g_trace_slab_words = 0;   // capacity of one timeline slab (8 words per workgroup), see lamp_debug_set_gemm_trace
static size_t g_extra_lds = 0;
extern "C" __attribute__((visibility("default"))) void lamp_debug_set_gemm_extra_lds(int bytes) { g_extra_lds = size_t(bytes); }
#endif

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool KTAIL, int MF, bool VEC, int DMA>
static int launch_cfg2(const GemmParams& p, hipStream_t s) {
    using T = GemmTile<BM, BN, BK, WAVES_M, WAVES_N, MF>;
    constexpr bool RPRE = T::MI * T::NI * (MF == 32 ? 16 : 4) <= 16;
    auto kern = gemm_nt_kernel<BM, BN, BK, WAVES_M, WAVES_N, KTAIL, MF, RPRE, VEC, DMA>;
    size_t LDS = (DMA == 1 || DMA == 2) ? T::DMA_STAGE_BYTES * (DMA == 1 ? 2 : 3) : T::LDS_BYTES;
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

// DMA: direct-to-LDS staging; a K that is not a multiple of BK (columns past K must read as zeros, which only the
// register path's per-element offsets can arrange) takes the register-staged kernel of the same tile -- same bits.
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int MF = 32, int DMA = 0>
static int launch_cfg(const GemmParams& p, hipStream_t s) {
    if (p.vec_epilogue) {
        if (p.K % BK) return launch_cfg2<BM, BN, BK, WAVES_M, WAVES_N, true, MF, true, 0>(p, s);
        return launch_cfg2<BM, BN, BK, WAVES_M, WAVES_N, false, MF, true, DMA>(p, s);
    }
    if (p.K % BK) return launch_cfg2<BM, BN, BK, WAVES_M, WAVES_N, true, MF, false, 0>(p, s);
    return launch_cfg2<BM, BN, BK, WAVES_M, WAVES_N, false, MF, false, DMA>(p, s);
}

#ifdef LAMP_TUNING
// Tuning build only (liblamp_hip_tuning.so: tools/bench_kernels.py and the every-variant tests): force a tile
// configuration (0 = heuristic) and collect a per-workgroup timeline.  The production library has neither.
static int g_force_tile = 0;
static const float* g_gemm_wp = nullptr;   // packed copy of the NEXT launches' weight matrix (single segment), for the W-direct tiles
extern "C" __attribute__((visibility("default"))) void lamp_debug_gemm_packed_w(const float* wp) { g_gemm_wp = wp; }
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

int launch_gemm(const GemmParams& p_in, hipStream_t s) {
    GemmParams p = p_in;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.nseg < 1 || p.nseg > GEMM_MAX_SEG) return LAMP_E_DIMS;
    if ((p.K & 3) || (p.lda & 3) || (p.ldw & 3)) return LAMP_E_ALIGN;
    if (!p.A || (p.A_dense && !p.m_dev)) return LAMP_E_NULL;
    if (!aligned16(p.A) || (p.A_dense && !aligned16(p.A_dense))) return LAMP_E_ALIGN;
    for (int i = 0; i < p.nseg; ++i) {
        if (!p.W[i] || !p.C[i]) return LAMP_E_NULL;
        if (!aligned16(p.W[i])) return LAMP_E_ALIGN;
    }
    const double flops = 2.0 * double(p.M) * p.N * p.nseg * p.K;
    const double bytes = 4.0 * (double(p.M) * p.K + double(p.N) * p.nseg * p.K +
                                double(p.M) * p.N * p.nseg * (p.R ? 2 : 1));
    ProfScope prof(LAMP_K_GEMM, flops, bytes, s);
    p.trace = nullptr;
    bool vec = !(p.N & 3) && !(p.ldc & 3) && (!p.R || (!(p.ldr & 3) && aligned16(p.R)));
    for (int i = 0; i < p.nseg; ++i) vec = vec && aligned16(p.C[i]) && (!p.bias[i] || aligned16(p.bias[i]));
    p.vec_epilogue = vec ? 1 : 0;
#ifdef LAMP_TUNING
    if (g_gemm_wp && p.nseg == 1 && p.N % 16 == 0 && p.K % 32 == 0) p.Wp[0] = g_gemm_wp;
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
        // direct-to-LDS staging: 20-29 inline-asm reads + counted vmcnt (DMA = 2), 40-49 the same tiles with compiler-scheduled reads (DMA = 1)
        case 20: return launch_cfg<32, 64, 32, 1, 4, 16, 2>(p, s);
        case 21: return launch_cfg<64, 64, 16, 2, 2, 16, 2>(p, s);
        case 22: return launch_cfg<64, 64, 32, 2, 2, 16, 2>(p, s);
        case 23: return launch_cfg<128, 64, 16, 2, 2, 16, 2>(p, s);
        case 24: return launch_cfg<128, 64, 32, 2, 2, 16, 2>(p, s);
        case 25: return launch_cfg<128, 128, 16, 2, 2, 16, 2>(p, s);
        case 26: return launch_cfg<128, 128, 32, 2, 2, 16, 2>(p, s);
        case 27: return launch_cfg<128, 128, 32, 2, 2, 32, 2>(p, s);
        case 28: return launch_cfg<32, 64, 64, 1, 4, 16, 2>(p, s);
        case 29: return launch_cfg<64, 64, 64, 2, 2, 16, 2>(p, s);
        case 30: return launch_cfg<64, 64, 16, 2, 2, 16, 3>(p, s);     // W fragments straight from global memory
        case 31: return launch_cfg<64, 64, 32, 2, 2, 16, 3>(p, s);
        case 32: return launch_cfg<32, 64, 32, 1, 4, 16, 3>(p, s);
        case 33: return launch_cfg<128, 64, 16, 2, 2, 16, 3>(p, s);
        case 40: return launch_cfg<32, 64, 32, 1, 4, 16, 1>(p, s);
        case 41: return launch_cfg<64, 64, 16, 2, 2, 16, 1>(p, s);
        case 42: return launch_cfg<64, 64, 32, 2, 2, 16, 1>(p, s);
        case 43: return launch_cfg<128, 64, 16, 2, 2, 16, 1>(p, s);
        case 44: return launch_cfg<128, 64, 32, 2, 2, 16, 1>(p, s);
        case 45: return launch_cfg<128, 128, 16, 2, 2, 16, 1>(p, s);
        case 46: return launch_cfg<128, 128, 32, 2, 2, 16, 1>(p, s);
        case 47: return launch_cfg<128, 128, 32, 2, 2, 32, 1>(p, s);
        case 48: return launch_cfg<32, 64, 64, 1, 4, 16, 1>(p, s);
        case 49: return launch_cfg<64, 64, 64, 2, 2, 16, 1>(p, s);
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
