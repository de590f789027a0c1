// General strided / batched fp32 GEMM for the backward pass:
//
//     C_z[m, n] (+)= alpha * sum_k A_z(m, k) * B_z(n, k)          z = (z0, z1) batch index
//
// where each operand is addressed through element strides, A_z(m, k) = A[z0*a_b0 + z1*a_b1 + m*a_rs + k*a_cs],
// and ONE of (a_rs, a_cs) is 1: either k is the contiguous index ("row" operand: activations times nn.Linear
// weights, as in the forward kernel) or m is ("column" operand: the transposed reads that dgrad / wgrad /
// attention-backward need -- dX = dY.W, dW = dY^T.X, dV = P^T.dO, dQ = dS.K, dK = dS^T.Q -- without any transpose
// copy).  Column operands are staged k-major in LDS and their MFMA fragments are read as four ds_read_b32 per
// 16-chunk; row operands use the forward kernel's b128 fragment trick.  Same 16x16x4 MFMA and the same
// k-accumulation order as the forward tiles.  Split-K (for the deep-K, small-output weight gradients) writes
// per-split partial tiles to the caller's workspace and a second kernel sums them in split order: deterministic.
#include "lamp_kernels.h"

namespace lamp {

namespace {

constexpr int GK = 16;        // k-tile
constexpr int S_ROW = GK + 8;  // LDS row stride of a row operand   [rows][24]: conflict-free b128 fragment reads under the real
                               // ds_read_b128 lane groups (gemm.hip: GemmTile::LDS_STRIDE); 20 was 2-way
// Block tiles (4 waves): 64 x 64 as 2 x 2 waves of 32 x 32 (2 x 2 MFMA blocks), and -- for the M = B*L shapes that
// would leave SIMDs idle, exactly as in the forward menu -- 32 x 64 as 1 x 4 waves of 32 x 16 (2 x 1 blocks).
// A column operand of R rows is staged as [16][R + 4].
constexpr int lds_operand(int rows) { return rows * S_ROW > GK * (rows + 4) ? rows * S_ROW : GK * (rows + 4); }

struct Operand {
    const float* p;
    int64_t rs, cs;  // strides of the (m or n) index and of k
    int64_t b0, b1;
};

struct GenParams {
    Operand A, B;
    float* C;
    int64_t ldc, c_b0, c_b1;
    int M, N, K;
    int nb1;          // z = z0 * nb1 + z1
    float alpha;
    int accumulate;
    const float* mask;  // relu'(mask > 0) applied to the result, same indexing as C (ld = ldm), or nullptr
    int64_t ldm;
    int k_chunk;        // K range per split (multiple of GK); K when not split
    float* part;        // split-K partials [nsplit][M][N], or nullptr
};

// One thread stages (at most) one float4 of each operand per k-tile; ROWS = the operand's tile extent (32 or 64).
template <bool COL, int ROWS>
__device__ __forceinline__ float4 stage_load(__amdgpu_buffer_rsrc_t rs, int tid, int k0, int k_end, int64_t ld, bool vec) {
    if (ROWS < 64 && tid >= ROWS * 4) return make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (!COL) {
        const int row = tid >> 2, k = k0 + 4 * (tid & 3);
        const unsigned off = unsigned(row * int(ld) + k) * 4u;
        if (vec) return bload4(rs, k < k_end ? off : OOB, 0);
        float4 v;
        v.x = bload1(rs, k + 0 < k_end ? off : OOB);
        v.y = bload1(rs, k + 1 < k_end ? off + 4u : OOB);
        v.z = bload1(rs, k + 2 < k_end ? off + 8u : OOB);
        v.w = bload1(rs, k + 3 < k_end ? off + 12u : OOB);
        return v;
    } else {
        constexpr int V4 = ROWS / 4;  // float4 per k-row
        const int krow = k0 + tid / V4, c = 4 * (tid % V4);
        const unsigned off = krow < k_end ? unsigned(krow * int(ld) + c) * 4u : OOB;
        if (vec) return bload4(rs, off, 0);
        float4 v;  // columns past the tile edge only feed output rows that are never stored
        v.x = bload1(rs, off);
        v.y = bload1(rs, off == OOB ? OOB : off + 4u);
        v.z = bload1(rs, off == OOB ? OOB : off + 8u);
        v.w = bload1(rs, off == OOB ? OOB : off + 12u);
        return v;
    }
}

// The same load with NOTHING recomputed per k-tile: this thread's byte offset inside the operand's first k-tile is a loop
// invariant (OOB for the threads without a share of a 32-row operand), the k-tile's base rides in the instruction's SCALAR offset
// -- the descriptor's range check covers voffset + soffset (tools/probes/lds_dma_oob.hip).  The per-tile offset arithmetic and
// bound selects of stage_load() were 36-66 vector instructions per two k-tiles, i.e. +17...+63 % on the MFMAs' time: on gfx950
// a vector instruction is not hidden under fp32 MFMAs (profiles/r05_mfma_chain.txt).  Any dword-aligned operand (see TAIL below); a
// row operand's LAST, partial k-tile still takes stage_load() (its k bound is not the descriptor's).
template <bool COL, int ROWS>
__device__ __forceinline__ unsigned stage_voff(int tid, int64_t ld) {
    if (ROWS < 64 && tid >= ROWS * 4) return OOB;
    if constexpr (!COL) return unsigned((tid >> 2) * int(ld) + 4 * (tid & 3)) * 4u;
    constexpr int V4 = ROWS / 4;
    return unsigned((tid / V4) * int(ld) + 4 * (tid % V4)) * 4u;
}

template <bool COL, int ROWS>
__device__ __forceinline__ void stage_store(float* lds, int tid, float4 v) {
    if (ROWS < 64 && tid >= ROWS * 4) return;
    if constexpr (!COL)
        *reinterpret_cast<float4*>(lds + (tid >> 2) * S_ROW + 4 * (tid & 3)) = v;
    else
        *reinterpret_cast<float4*>(lds + (tid / (ROWS / 4)) * (ROWS + 4) + 4 * (tid % (ROWS / 4))) = v;
}

// Fragment of one 16-row MFMA block: the 4 k-values {4*hi + j} of row `row` (tile-local).
template <bool COL, int ROWS>
__device__ __forceinline__ float4 frag(const float* lds, int row, int hi) {
    if constexpr (!COL) return *reinterpret_cast<const float4*>(lds + row * S_ROW + 4 * hi);
    constexpr int S = ROWS + 4;
    float4 v;
    const float* q = lds + (4 * hi) * S + row;
    v.x = q[0];
    v.y = q[S];
    v.z = q[2 * S];
    v.w = q[3 * S];
    return v;
}

// TAIL: K leaves a partial last k-tile AND at least one operand is a row operand (k contiguous): that one tile of that operand
// takes stage_load()'s per-element bounds (its k bound is not the descriptor's); a column operand's rows past K fail the range
// check by themselves.  Every other load is stage_voff()'s: no per-tile vector arithmetic.  16-byte loads need only dword
// alignment on gfx950 (tools/probes/unaligned_b128.hip, profiles/r05_unaligned_b128.txt) and are range-checked dword by dword
// (a float4 that straddles the end of the descriptor returns its valid part: profiles/r05_lds_dma_oob.txt), so there is no
// alignment condition left: the probability maps of 302 keys (1208-byte rows), head-split views, ragged edges all load this way.
// A float4 of a column operand that overshoots the tile's valid columns reads its in-range neighbours: they only feed output rows
// / columns that are never stored.
template <int GM, int GN, int WM, int WN, bool TA, bool TB, bool TAIL>
__device__ __forceinline__ void gemm_gen_tile(const GenParams& p, int tiles_n, int tile, int zy, int split) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int WTM = GM / WM, WTN = GN / WN, MI = WTM / 16, NI = WTN / 16;
    __shared__ __attribute__((aligned(16))) float As[2][lds_operand(GM)];
    __shared__ __attribute__((aligned(16))) float Bs[2][lds_operand(GN)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, l15 = lane & 15, hi = lane >> 4;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * GM, n0 = tn * GN;
    const int rows_m = p.M - m0 < GM ? p.M - m0 : GM;
    const int rows_n = p.N - n0 < GN ? p.N - n0 : GN;
    const int z0 = zy / p.nb1, z1 = zy - z0 * p.nb1;
    const int k_begin = split * p.k_chunk;
    const int k_end = k_begin + p.k_chunk < p.K ? k_begin + p.k_chunk : p.K;

    const float* Az = p.A.p + z0 * p.A.b0 + z1 * p.A.b1;
    const float* Bz = p.B.p + z0 * p.B.b0 + z1 * p.B.b1;
    const int64_t lda = TA ? p.A.cs : p.A.rs, ldb = TB ? p.B.cs : p.B.rs;
    // row operand: base at the tile's first row, range = valid rows x K; column operand: base at the tile's first
    // column, range = k_end rows x valid columns.
    const __amdgpu_buffer_rsrc_t rsA =
        TA ? make_rsrc(Az + m0, (uint64_t(k_end - 1) * lda + rows_m) * 4u)
           : make_rsrc(Az + int64_t(m0) * lda, (uint64_t(rows_m - 1) * lda + p.K) * 4u);
    const __amdgpu_buffer_rsrc_t rsB =
        TB ? make_rsrc(Bz + n0, (uint64_t(k_end - 1) * ldb + rows_n) * 4u)
           : make_rsrc(Bz + int64_t(n0) * ldb, (uint64_t(rows_n - 1) * ldb + p.K) * 4u);

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (k_end - k_begin + GK - 1) / GK;
    const unsigned voffA = stage_voff<TA, GM>(tid, lda), voffB = stage_voff<TB, GN>(tid, ldb);
    const unsigned kstepA = TA ? unsigned(lda) * 4u : 4u, kstepB = TB ? unsigned(ldb) * 4u : 4u;   // bytes per unit of k
    auto gload = [&](int kt, float4& ra, float4& rb) {
        const int k0 = k_begin + kt * GK;
        if (TAIL && k0 + GK > k_end) {   // wave-uniform: the one partial k-tile (the last tile of the last split)
            ra = TA ? bload4(rsA, voffA, unsigned(k0) * kstepA) : stage_load<TA, GM>(rsA, tid, k0, k_end, lda, false);
            rb = TB ? bload4(rsB, voffB, unsigned(k0) * kstepB) : stage_load<TB, GN>(rsB, tid, k0, k_end, ldb, false);
        } else {
            ra = bload4(rsA, voffA, unsigned(k0) * kstepA);
            rb = bload4(rsB, voffB, unsigned(k0) * kstepB);
        }
    };
    auto lstore = [&](int buf, const float4& ra, const float4& rb) {
        stage_store<TA, GM>(As[buf], tid, ra);
        stage_store<TB, GN>(Bs[buf], tid, rb);
    };
    auto compute = [&](int buf) {
        float4 fa[MI], fb[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = frag<TA, GM>(As[buf], wm * WTM + 16 * i + l15, hi);
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[j] = frag<TB, GN>(Bs[buf], wn * WTN + 16 * j + l15, hi);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
            }
    };
    // Two register sets, as in the forward kernel: tiles kt+1 and kt+2 are in flight while tile kt is multiplied.
    float4 ra0, rb0, ra1, rb1;
    gload(0, ra0, rb0);
    lstore(0, ra0, rb0);
    if (nk > 1) gload(1, ra0, rb0);
    if (nk > 2) gload(2, ra1, rb1);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        compute(0);  // tile kt in LDS[0]; tile kt+1 in set 0, tile kt+2 in set 1
        if (kt + 1 < nk) lstore(1, ra0, rb0);
        if (kt + 3 < nk) gload(kt + 3, ra0, rb0);
        __syncthreads();
        if (kt + 1 >= nk) break;
        compute(1);  // tile kt+1 in LDS[1]; tile kt+2 in set 1, tile kt+3 in set 0
        if (kt + 2 < nk) lstore(0, ra1, rb1);
        if (kt + 4 < nk) gload(kt + 4, ra1, rb1);
        __syncthreads();
    }

    // C/D layout of the 16x16 block: col = lane & 15, row = 4 * (lane >> 4) + r.
    if (p.part) {  // split-K partial: dense [split][M][N]
        float* base = p.part + (int64_t(split) * p.M + m0) * p.N + n0;
        const __amdgpu_buffer_rsrc_t rsP = make_rsrc(base, (uint64_t(rows_m - 1) * p.N + rows_n) * 4u);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int col = wn * WTN + 16 * j + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * WTM + 16 * i + 4 * hi + r;
                    bstore1(rsP, col < rows_n ? unsigned(row * p.N + col) * 4u : OOB, acc[i][j][r]);
                }
            }
        return;
    }
    float* Cz = p.C + z0 * p.c_b0 + z1 * p.c_b1 + int64_t(m0) * p.ldc + n0;
    const int ldc = int(p.ldc), ldm = int(p.ldm);
    const __amdgpu_buffer_rsrc_t rsC = make_rsrc(Cz, (uint64_t(rows_m - 1) * ldc + rows_n) * 4u);
    const bool has_m = p.mask != nullptr;
    const __amdgpu_buffer_rsrc_t rsM =
        make_rsrc(has_m ? p.mask + int64_t(m0) * p.ldm + n0 : Cz, has_m ? (uint64_t(rows_m - 1) * ldm + rows_n) * 4u : 0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = wn * WTN + 16 * j + l15;
            const bool ok = col < rows_n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * WTM + 16 * i + 4 * hi + r;
                float v = p.alpha * acc[i][j][r];
                if (has_m && !(bload1(rsM, ok ? unsigned(row * ldm + col) * 4u : OOB) > 0.f)) v = 0.f;
                const unsigned off = ok ? unsigned(row * ldc + col) * 4u : OOB;
                if (p.accumulate) v += bload1(rsC, off);
                bstore1(rsC, off, v);
            }
        }
}

template <int GM, int GN, int WM, int WN, bool TA, bool TB, bool TAIL>
__global__ __launch_bounds__(256) void gemm_gen_kernel(GenParams p, int tiles_n) {
    gemm_gen_tile<GM, GN, WM, WN, TA, TB, TAIL>(p, tiles_n, xcd_remap(blockIdx.x, gridDim.x), blockIdx.y, blockIdx.z);
}

// Grouped launch: up to GROUP_MAX independent products (no batch dims, no K split, no ReLU mask) share ONE grid of 64 x 64
// tiles -- the weight gradients of a whole backward pass (dW = dY^T.X: 64 tiles each at d_model = 512) fill the chip
// together instead of one by one through split-K partials and a reduce launch each.  The problem table rides in the kernel
// arguments.  Workgroup b runs on XCD b % 8 (round-robin dispatch): every XCD takes an eighth of EVERY problem's tiles
// (consecutive tiles, i.e. whole row panels sharing their A columns in that XCD's L2), so the XCDs stay balanced whatever
// the mix of K depths, and within an XCD the host's order (deepest K first) is the dispatch order.  Every output tile is owned
// by exactly one workgroup and accumulates k in ascending order: results do not depend on the grouping.
constexpr int GROUP_MAX = 32;
struct GroupProblem {
    const float* A;
    const float* B;
    float* C;
    int lda, ldb, ldc;   // lda / ldb: the non-unit stride of the operand
    int M, N, K;
    int tiles_n, tiles;
    float alpha;
    int accumulate;
};
struct GroupParams {
    GroupProblem prob[GROUP_MAX];
    int share_begin[GROUP_MAX + 1];   // prefix sums of ceil(tiles / 8): a problem's tiles per XCD
    int n;
};

template <bool TA, bool TB, bool TAIL>
__global__ __launch_bounds__(256) void gemm_group_kernel(GroupParams g) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    int idx = 0;
    for (int i = 1; i < g.n; ++i)
        if (j >= g.share_begin[i]) idx = i;
    const GroupProblem& q = g.prob[idx];
    const int share = g.share_begin[idx + 1] - g.share_begin[idx];
    const int tile = xcd * share + (j - g.share_begin[idx]);
    if (tile >= q.tiles) return;
    GenParams p;
    p.A = Operand{q.A, TA ? 1 : q.lda, TA ? q.lda : 1, 0, 0};
    p.B = Operand{q.B, TB ? 1 : q.ldb, TB ? q.ldb : 1, 0, 0};
    p.C = q.C;
    p.ldc = q.ldc;
    p.c_b0 = p.c_b1 = 0;
    p.M = q.M;
    p.N = q.N;
    p.K = q.K;
    p.nb1 = 1;
    p.alpha = q.alpha;
    p.accumulate = q.accumulate;
    p.mask = nullptr;
    p.ldm = 0;
    p.k_chunk = q.K;
    p.part = nullptr;
    gemm_gen_tile<64, 64, 2, 2, TA, TB, TAIL>(p, q.tiles_n, tile, 0, 0);
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int nsplit, int M, int N,
                                                            float alpha, const float* __restrict__ mask, int64_t ldm,
                                                            int accumulate, float* __restrict__ C, int64_t ldc) {
    const int64_t total = int64_t(M) * N;
    for (int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x; e < total; e += int64_t(gridDim.x) * 256) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += part[k * total + e];
        const int64_t m = e / N, n = e - m * N;
        float v = alpha * s;
        if (mask && !(mask[m * ldm + n] > 0.f)) v = 0.f;
        if (accumulate) v += C[m * ldc + n];
        C[m * ldc + n] = v;
    }
}

}  // namespace

namespace {
// tile choice and split-K depth: functions of the problem shape only
inline int pick_splits(int M, int N, int K, int64_t batch, int gm);
inline int pick_gm(int M, int N, int K, int64_t batch) {
    const int64_t t64 = int64_t((M + 63) / 64) * ((N + 63) / 64) * batch;
    if (pick_splits(M, N, K, batch, 64) > 1) return 64;  // deep-K weight gradients: parallelism comes from the K split
    return t64 < 1200 ? 32 : 64;
}
inline int pick_splits(int M, int N, int K, int64_t batch, int gm) {
    const int64_t tiles = int64_t((M + gm - 1) / gm) * ((N + 63) / 64);
    if (batch != 1 || tiles >= 512 || K < 1024) return 1;
    int64_t ns = (1024 + tiles - 1) / tiles;
    if (ns > K / 256) ns = K / 256;
    if (ns > 64) ns = 64;
    return ns < 2 ? 1 : int(ns);
}
}  // namespace

size_t gemm_gen_workspace_bytes(int M, int N, int K, int batch) {
    if (M <= 0 || N <= 0 || K <= 0 || batch < 1) return 0;
    const int ns = pick_splits(M, N, K, batch, pick_gm(M, N, K, batch));
    return ns < 2 ? 0 : size_t(ns) * M * N * sizeof(float);
}

int launch_gemm_gen(const lamp_gemm_desc& d, void* ws, size_t ws_bytes, hipStream_t s) {
    if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.batch0 < 1 || d.batch1 < 1) return LAMP_E_DIMS;
    if (!d.A || !d.B || !d.C) return LAMP_E_NULL;
    const bool ta = d.a_row_stride == 1 && d.a_col_stride != 1, tb = d.b_row_stride == 1 && d.b_col_stride != 1;
    if ((!ta && d.a_col_stride != 1) || (!tb && d.b_col_stride != 1)) return LAMP_E_UNSUPPORTED;
    const int64_t lda = ta ? d.a_col_stride : d.a_row_stride, ldb = tb ? d.b_col_stride : d.b_row_stride;
    if (lda < 1 || ldb < 1 || d.ldc < d.N) return LAMP_E_DIMS;
    // 32-bit in-tile byte offsets
    const int64_t span_a = ta ? int64_t(d.K) * lda : int64_t(64) * lda + d.K;
    const int64_t span_b = tb ? int64_t(d.K) * ldb : int64_t(64) * ldb + d.K;
    if (span_a * 4 >= 0x7fffffffLL || span_b * 4 >= 0x7fffffffLL || int64_t(64) * d.ldc * 4 >= 0x7fffffffLL ||
        (d.relu_mask && int64_t(64) * d.ld_mask * 4 >= 0x7fffffffLL))
        return LAMP_E_UNSUPPORTED;
    const int64_t batch = int64_t(d.batch0) * d.batch1;
    if (batch > 65535) return LAMP_E_DIMS;
    if (d.relu_mask && batch != 1) return LAMP_E_UNSUPPORTED;

    GenParams p;
    p.A = Operand{d.A, d.a_row_stride, d.a_col_stride, d.a_batch0, d.a_batch1};
    p.B = Operand{d.B, d.b_row_stride, d.b_col_stride, d.b_batch0, d.b_batch1};
    p.C = d.C;
    p.ldc = d.ldc;
    p.c_b0 = d.c_batch0;
    p.c_b1 = d.c_batch1;
    p.M = d.M;
    p.N = d.N;
    p.K = d.K;
    p.nb1 = d.batch1;
    p.alpha = d.alpha;
    p.accumulate = d.accumulate;
    p.mask = d.relu_mask;
    p.ldm = d.ld_mask;
    p.k_chunk = d.K;
    p.part = nullptr;

    const int gm = pick_gm(d.M, d.N, d.K, batch);
    const int tiles_m = (d.M + gm - 1) / gm, tiles_n = (d.N + 63) / 64;
    const int64_t tiles = int64_t(tiles_m) * tiles_n;
    if (tiles > 0x7fffffffLL) return LAMP_E_DIMS;
    int nsplit = pick_splits(d.M, d.N, d.K, batch, gm);
    if (nsplit > 1) {
        const int64_t fit = ws ? int64_t(ws_bytes / (size_t(d.M) * d.N * sizeof(float))) : 0;
        if (nsplit > fit) nsplit = int(fit);
        if (nsplit >= 2) {
            p.k_chunk = ((d.K + nsplit - 1) / nsplit + GK - 1) / GK * GK;
            nsplit = (d.K + p.k_chunk - 1) / p.k_chunk;
            p.part = static_cast<float*>(ws);
        } else {
            nsplit = 1;
        }
    }
    const double flops = 2.0 * double(d.M) * d.N * d.K * double(batch);
    const double bytes = 4.0 * double(batch) * (double(d.M) * d.K + double(d.N) * d.K + double(d.M) * d.N);
    ProfScope prof(LAMP_K_GEMM, flops, bytes, s);
    const dim3 grid((unsigned)tiles, (unsigned)batch, (unsigned)nsplit);
    // only K itself can leave a partial k-tile (the split size is a multiple of GK), and only a row operand needs help with it
    const bool tail = (d.K % GK) != 0 && (!ta || !tb);
#define LAMP_GEN_LAUNCH(GM_, WM_, WN_, TAIL_)                                                                               \
    do {                                                                                                                     \
        if (ta && tb)                                                                                                        \
            hipLaunchKernelGGL((gemm_gen_kernel<GM_, 64, WM_, WN_, true, true, false>), grid, dim3(256), 0, s, p, tiles_n);   \
        else if (ta)                                                                                                         \
            hipLaunchKernelGGL((gemm_gen_kernel<GM_, 64, WM_, WN_, true, false, TAIL_>), grid, dim3(256), 0, s, p, tiles_n);  \
        else if (tb)                                                                                                         \
            hipLaunchKernelGGL((gemm_gen_kernel<GM_, 64, WM_, WN_, false, true, TAIL_>), grid, dim3(256), 0, s, p, tiles_n);  \
        else                                                                                                                 \
            hipLaunchKernelGGL((gemm_gen_kernel<GM_, 64, WM_, WN_, false, false, TAIL_>), grid, dim3(256), 0, s, p, tiles_n); \
    } while (0)
    if (gm == 32 && tail)
        LAMP_GEN_LAUNCH(32, 1, 4, true);
    else if (gm == 32)
        LAMP_GEN_LAUNCH(32, 1, 4, false);
    else if (tail)
        LAMP_GEN_LAUNCH(64, 2, 2, true);
    else
        LAMP_GEN_LAUNCH(64, 2, 2, false);
#undef LAMP_GEN_LAUNCH
    if (int e = int(hipGetLastError())) return e;
    if (p.part) {
        const int64_t total = int64_t(d.M) * d.N;
        const unsigned g = unsigned(total / 256 + 1 < 2048 ? total / 256 + 1 : 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(g), dim3(256), 0, s, p.part, nsplit, d.M, d.N, d.alpha, d.relu_mask,
                           d.ld_mask, d.accumulate, d.C, d.ldc);
        return int(hipGetLastError());
    }
    return 0;
}

int launch_gemm_group(const lamp_gemm_desc* descs, int n, hipStream_t s) {
    if (n <= 0) return n == 0 ? 0 : LAMP_E_DIMS;
    if (!descs) return LAMP_E_NULL;
    for (int first = 0; first < n; first += GROUP_MAX) {
        const int cnt = n - first < GROUP_MAX ? n - first : GROUP_MAX;
        GroupParams g;
        g.n = cnt;
        g.share_begin[0] = 0;
        bool ta0 = false, tb0 = false, tail = false;
        double flops = 0, bytes = 0;
        for (int i = 0; i < cnt; ++i) {
            const lamp_gemm_desc& d = descs[first + i];
            if (d.M <= 0 || d.N <= 0 || d.K <= 0) return LAMP_E_DIMS;
            if (!d.A || !d.B || !d.C) return LAMP_E_NULL;
            if (d.batch0 != 1 || d.batch1 != 1 || d.relu_mask) return LAMP_E_UNSUPPORTED;
            const bool ta = d.a_row_stride == 1 && d.a_col_stride != 1, tb = d.b_row_stride == 1 && d.b_col_stride != 1;
            if ((!ta && d.a_col_stride != 1) || (!tb && d.b_col_stride != 1)) return LAMP_E_UNSUPPORTED;
            if (i == 0) ta0 = ta, tb0 = tb;
            if (ta != ta0 || tb != tb0) return LAMP_E_UNSUPPORTED;   // one operand form per call
            const int64_t lda = ta ? d.a_col_stride : d.a_row_stride, ldb = tb ? d.b_col_stride : d.b_row_stride;
            if (lda < 1 || ldb < 1 || d.ldc < d.N) return LAMP_E_DIMS;
            const int64_t span_a = ta ? int64_t(d.K) * lda : int64_t(64) * lda + d.K;
            const int64_t span_b = tb ? int64_t(d.K) * ldb : int64_t(64) * ldb + d.K;
            if (span_a * 4 >= 0x7fffffffLL || span_b * 4 >= 0x7fffffffLL || int64_t(64) * d.ldc * 4 >= 0x7fffffffLL)
                return LAMP_E_UNSUPPORTED;
            tail = tail || ((d.K % GK) != 0 && (!ta || !tb));
            GroupProblem& q = g.prob[i];
            q.A = d.A;
            q.B = d.B;
            q.C = d.C;
            q.lda = int(lda);
            q.ldb = int(ldb);
            q.ldc = int(d.ldc);
            q.M = d.M;
            q.N = d.N;
            q.K = d.K;
            q.tiles_n = (d.N + 63) / 64;
            const int64_t tiles = int64_t((d.M + 63) / 64) * q.tiles_n;
            if (tiles > (1 << 24)) return LAMP_E_DIMS;
            q.tiles = int(tiles);
            q.alpha = d.alpha;
            q.accumulate = d.accumulate;
            g.share_begin[i + 1] = g.share_begin[i] + (q.tiles + 7) / 8;
            flops += 2.0 * double(d.M) * d.N * d.K;
            bytes += 4.0 * (double(d.M) * d.K + double(d.N) * d.K + double(d.M) * d.N);
        }
        for (int i = cnt; i < GROUP_MAX; ++i) g.share_begin[i + 1] = g.share_begin[cnt];
        ProfScope prof(LAMP_K_GEMM, flops, bytes, s);
        const dim3 grid(unsigned(g.share_begin[cnt]) * 8u);
#define LAMP_GROUP_LAUNCH(TAIL_)                                                                                   \
    do {                                                                                                            \
        if (ta0 && tb0)                                                                                             \
            hipLaunchKernelGGL((gemm_group_kernel<true, true, false>), grid, dim3(256), 0, s, g);                   \
        else if (ta0)                                                                                               \
            hipLaunchKernelGGL((gemm_group_kernel<true, false, TAIL_>), grid, dim3(256), 0, s, g);                  \
        else if (tb0)                                                                                               \
            hipLaunchKernelGGL((gemm_group_kernel<false, true, TAIL_>), grid, dim3(256), 0, s, g);                  \
        else                                                                                                        \
            hipLaunchKernelGGL((gemm_group_kernel<false, false, TAIL_>), grid, dim3(256), 0, s, g);                 \
    } while (0)
        if (tail)
            LAMP_GROUP_LAUNCH(true);
        else
            LAMP_GROUP_LAUNCH(false);
#undef LAMP_GROUP_LAUNCH
        if (int e = int(hipGetLastError())) return e;
    }
    return 0;
}

}  // namespace lamp
