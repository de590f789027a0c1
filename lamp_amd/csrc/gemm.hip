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
// MFMAs per operand instead of four 4-byte reads.  LDS rows are padded by 4 floats, which makes
// the b128 fragment reads and the b128 staging writes bank-conflict free (row stride 36 floats =
// 9 sixteen-byte slots, odd => the 16 rows of a lane group land on 16 distinct slots).
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
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int MF = 32>
struct GemmTile {
    static constexpr int NT = WAVES_M * WAVES_N * 64;
    static constexpr int WTM = BM / WAVES_M;
    static constexpr int WTN = BN / WAVES_N;
    static constexpr int MI = WTM / MF;
    static constexpr int NI = WTN / MF;
    static constexpr int LDS_STRIDE = BK + 4;
    static constexpr int A_LD = BM * BK / 4 / NT;  // float4 loads per thread per tile
    static constexpr int B_LD = BN * BK / 4 / NT;
    static constexpr size_t LDS_BYTES = size_t(2) * (BM + BN) * LDS_STRIDE * sizeof(float);
    static_assert(MF == 32 || MF == 16, "MFMA block edge");
    static_assert(WTM % MF == 0 && WTN % MF == 0, "wave tile must be a multiple of the MFMA block");
    static_assert(BK % (MF == 32 ? 8 : 16) == 0, "BK must cover whole fragment reads");
    static_assert((BM * BK / 4) % NT == 0 && (BN * BK / 4) % NT == 0, "staging must divide evenly");
    static_assert(BK % 8 == 0, "BK must be a multiple of 8");
};

// LNM (deferred LayerNorm, production tiles only; 0 = plain GEMM, identical code to before):
//   bit 0  A holds pre-LayerNorm rows: (mean, rstd) from their partial sums, applied in the epilogue with folded weights
//   bit 1  R holds pre-LayerNorm rows: the residual is LayerNorm(R), recomputed from R and its rows' statistics
//   bit 2  the epilogue also writes per row and 16-column group the partial (sum, sum of squares) of the output
// Nothing is added to the main loop.  Summation orders are fixed and independent of the tile configuration (16-column
// butterflies, then groups in ascending column order), so a sample's bits still do not
// depend on the batch it is in.
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool KTAIL, int MF, int LNM = 0>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_nt_kernel(GemmParams p, int tiles_n_seg,
                                                                          int tiles_n) {
    constexpr bool LN = (LNM & 1) != 0, LNR = (LNM & 2) != 0, PS = (LNM & 4) != 0;
    static_assert(LNM == 0 || (MF == 16 && BN == 64), "deferred LayerNorm: 16x16x4 tiles with 64 columns");
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
    float* ln_lds = smem + 2 * (BM + BN) * S;  // deferred LayerNorm: [BM][4] = (A mean, A rstd, R mean, R rstd)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & (MF - 1), hi = lane / MF;  // row-in-block, k-group

    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = tile / tiles_n;
    const int tn_all = tile - tm * tiles_n;
    const int seg = tn_all / tiles_n_seg;
    const int tn = tn_all - seg * tiles_n_seg;
    const int64_t m0 = int64_t(tm) * BM;
    const int n0 = tn * BN;
    const int rows_m = int(p.M - m0 < BM ? p.M - m0 : BM);  // valid rows / columns of this tile
    const int rows_n = p.N - n0 < BN ? p.N - n0 : BN;

    const int lda = int(p.lda), ldw = int(p.ldw);
    const __amdgpu_buffer_rsrc_t rsA =
        make_rsrc(p.A + m0 * p.lda, (uint64_t(rows_m - 1) * lda + p.K) * 4u);
    const __amdgpu_buffer_rsrc_t rsW =
        make_rsrc(p.W[seg] + int64_t(n0) * p.ldw, (uint64_t(rows_n - 1) * ldw + p.K) * 4u);

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
#pragma unroll
            for (int i = 0; i < T::MI; ++i)
#pragma unroll
                for (int j = 0; j < T::NI; ++j) {
                    if constexpr (MF == 32) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                    }
                }
        }
    };

    const int nk = (p.K + BK - 1) / BK;
    gload(0, ra0, rb0);
    lstore(0, ra0, rb0);
    if (nk > 1) gload(BK, ra0, rb0);      // tile 1 -> set 0
    if (nk > 2) gload(2 * BK, ra1, rb1);  // tile 2 -> set 1
    // Deferred LayerNorm, consumer side: the rows' 16-column partial sums are LOADED here, together with the first tiles
    // (16 lanes per row; lane l takes partials l, l + 16, ...), and reduced only after the main loop, so their latency
    // hides under the MFMAs.
    constexpr int LN_ROWS = (LN && LNR ? 2 : 1) * BM;   // A rows, then R rows
    constexpr int LN_IT = (LN || LNR) ? (LN_ROWS * 16 + T::NT - 1) / T::NT : 1;
    constexpr int LN_NP = 4;                             // partials per lane: up to 64 groups = d <= 1024
    float lpa[LN_IT][LN_NP], lpb[LN_IT][LN_NP];
    if constexpr (LN || LNR) {
        const int sub = tid & 15;
#pragma unroll
        for (int it = 0; it < LN_IT; ++it) {
            const int row = (tid >> 4) + it * (T::NT / 16);
            const bool for_r = LNR && (!LN || row >= BM);
            const int lr = row >= BM ? row - BM : row;
            const float* part = for_r ? p.r_part : p.a_part;
            const int np = for_r ? p.r_nparts : p.a_nparts;
            const bool row_ok = row < LN_ROWS && lr < rows_m;
            const float* q = part + (m0 + (row_ok ? lr : 0)) * int64_t(np) * 2;
#pragma unroll
            for (int u = 0; u < LN_NP; ++u) {
                const int t2 = sub + 16 * u;
                const bool ok = row_ok && t2 < np;
                lpa[it][u] = ok ? q[2 * t2] : 0.f;
                lpb[it][u] = ok ? q[2 * t2 + 1] : 0.f;
            }
        }
    }
    __syncthreads();

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

    if constexpr (LN || LNR) {
        // reduce the partials loaded before the main loop: ascending within a lane, then a DPP row sum (an order that
        // depends on neither this kernel's nor the producer's tile configuration); (mean, rstd) -> LDS for the epilogue
        const int sub = tid & 15;
#pragma unroll
        for (int it = 0; it < LN_IT; ++it) {
            const int row = (tid >> 4) + it * (T::NT / 16);
            const bool for_r = LNR && (!LN || row >= BM);
            const int lr = row >= BM ? row - BM : row;
            float a = ((lpa[it][0] + lpa[it][1]) + lpa[it][2]) + lpa[it][3];
            float b = ((lpb[it][0] + lpb[it][1]) + lpb[it][2]) + lpb[it][3];
            a = row16_sum(a);
            b = row16_sum(b);
            if (sub == 0 && row < LN_ROWS) {
                const float inv_n = 1.0f / float(for_r ? p.N : p.K);
                const float mean = a * inv_n;
                const float rstd = 1.0f / sqrtf(fmaxf(fmaf(-mean, mean, b * inv_n), 0.f) + p.ln_eps);
                ln_lds[4 * lr + (for_r ? 2 : 0)] = mean;
                ln_lds[4 * lr + (for_r ? 3 : 1)] = rstd;
            }
        }
        __syncthreads();
    }

    // Epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Stores to rows past M fall outside the descriptor and are dropped by the hardware; columns past
    // N are steered to an out-of-range offset.
    const int ldc = int(p.ldc), ldr = int(p.ldr);
    const __amdgpu_buffer_rsrc_t rsC =
        make_rsrc(p.C[seg] + m0 * p.ldc + n0, (uint64_t(rows_m - 1) * ldc + rows_n) * 4u);
    const bool has_r = p.R != nullptr;
    const __amdgpu_buffer_rsrc_t rsR =
        make_rsrc(has_r ? p.R + m0 * p.ldr + n0 : p.A, has_r ? (uint64_t(rows_m - 1) * ldr + rows_n) * 4u : 0);
    const float* bias = p.bias[seg];
    const __amdgpu_buffer_rsrc_t rsBias = make_rsrc(bias ? bias + n0 : p.A, bias ? uint64_t(rows_n) * 4u : 0);
    const float* lns = LN ? p.ln_s[seg] : nullptr;
    const __amdgpu_buffer_rsrc_t rsS = make_rsrc(lns ? lns + n0 : p.A, lns ? uint64_t(rows_n) * 4u : 0);
    const bool r_ln = LNR && has_r;
    const __amdgpu_buffer_rsrc_t rsRG = make_rsrc(r_ln ? p.r_gamma + n0 : p.A, r_ln ? uint64_t(rows_n) * 4u : 0);
    const __amdgpu_buffer_rsrc_t rsRB = make_rsrc(r_ln ? p.r_beta + n0 : p.A, r_ln ? uint64_t(rows_n) * 4u : 0);
    // C/D layouts: 32x32 block: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5), r < 16;
    //              16x16 block: col = lane&15, row = 4*(lane>>4) + r,            r < 4.
    auto blk_row = [&](int r) { return MF == 32 ? (r & 3) + 8 * (r >> 2) : r; };
    const int lrow0 = wm * T::WTM + 4 * hi;
    const int lcol0 = wn * T::WTN + l31;
#pragma unroll
    for (int j = 0; j < T::NI; ++j) {
        const int lcol = lcol0 + j * MF;
        const bool col_ok = lcol < rows_n;
        const float bv = bload1(rsBias, col_ok ? unsigned(lcol) * 4u : OOB);
        float sv = 0.f, rg = 1.f, rb = 0.f;
        if constexpr (LN) sv = bload1(rsS, col_ok ? unsigned(lcol) * 4u : OOB);
        if constexpr (LNR) {
            rg = bload1(rsRG, (r_ln && col_ok) ? unsigned(lcol) * 4u : OOB);
            rb = bload1(rsRB, (r_ln && col_ok) ? unsigned(lcol) * 4u : OOB);
        }
#pragma unroll
        for (int i = 0; i < T::MI; ++i) {
            float res[NACC];
            if (has_r) {
#pragma unroll
                for (int r = 0; r < NACC; ++r) {
                    const int lrow = lrow0 + i * MF + blk_row(r);
                    res[r] = bload1(rsR, col_ok ? unsigned(lrow * ldr + lcol) * 4u : OOB);
                    if constexpr (LNR)  // LayerNorm(z_prev) recomputed from z_prev and its row statistics (explicit fma)
                        res[r] = fmaf((res[r] - ln_lds[4 * lrow + 2]) * ln_lds[4 * lrow + 3], rg, rb);
                }
            }
#pragma unroll
            for (int r = 0; r < NACC; ++r) {
                const int lrow = lrow0 + i * MF + blk_row(r);
                float v = acc[i][j][r];
                if constexpr (LN) {
                    if (lns)  // rstd * (acc - mean * s) + bias'
                        v = fmaf(ln_lds[4 * lrow + 1], fmaf(-ln_lds[4 * lrow], sv, v), bv);
                    else
                        v += bv;
                } else {
                    v += bv;
                }
                if (p.relu) v = fmaxf(v, 0.f);
                if (has_r) v += res[r];
                bstore1(rsC, col_ok ? unsigned(lrow * ldc + lcol) * 4u : OOB, v);
                if constexpr (PS) {
                    // 16-lane butterfly = the 16 columns of one group of this row; lane 0 of the group stores the
                    // partial (sum, sum of squares) [row][group][2].  Columns past N contribute exact zeros.
                    float a = col_ok ? v : 0.f, q = a * a;
                    a = row16_sum(a);
                    q = row16_sum(q);
                    if (l31 == 0 && lrow < rows_m) {
                        const int g = tn * 4 + ((wn * T::WTN + j * MF) >> 4);
                        float* out = p.part_out + ((m0 + lrow) * int64_t(tiles_n_seg) * 4 + g) * 2;
                        out[0] = a;
                        out[1] = q;
                    }
                }
            }
        }
    }
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool KTAIL, int MF, int LNM = 0>
static int launch_cfg2(const GemmParams& p, hipStream_t s) {
    using T = GemmTile<BM, BN, BK, WAVES_M, WAVES_N, MF>;
    auto kern = gemm_nt_kernel<BM, BN, BK, WAVES_M, WAVES_N, KTAIL, MF, LNM>;
    constexpr size_t LDS = T::LDS_BYTES + (LNM ? size_t(BM) * 4 * sizeof(float) : 0);
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(LDS));
        if (e != hipSuccess) return int(e);
        attr_done[dev] = true;
    }
    // 32-bit in-tile byte offsets
    const int64_t ldmax = p.lda > p.ldw ? (p.lda > p.ldc ? p.lda : p.ldc) : (p.ldw > p.ldc ? p.ldw : p.ldc);
    if (ldmax * (BM > BN ? BM : BN) * 4 >= 0x7fffffffLL || (p.R && p.ldr * BM * 4 >= 0x7fffffffLL))
        return LAMP_E_UNSUPPORTED;
    const int64_t tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n_seg = (p.N + BN - 1) / BN;
    const int tiles_n = tiles_n_seg * p.nseg;
    const int64_t nwg = tiles_m * tiles_n;
    if (nwg > 0x7fffffffLL) return LAMP_E_DIMS;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(T::NT), LDS, s, p, tiles_n_seg, tiles_n);
    return int(hipGetLastError());
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int MF = 32>
static int launch_cfg(const GemmParams& p, hipStream_t s) {
    if (p.K % BK) return launch_cfg2<BM, BN, BK, WAVES_M, WAVES_N, true, MF>(p, s);
    return launch_cfg2<BM, BN, BK, WAVES_M, WAVES_N, false, MF>(p, s);
}
// production tiles only: the deferred-LayerNorm variants (the callers' d_model is a multiple of BK: no K tail)
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
static int launch_cfg_ln(const GemmParams& p, int lnm, hipStream_t s) {
    if (p.K % BK) return LAMP_E_UNSUPPORTED;
    switch (lnm) {
        case 1: return launch_cfg2<BM, BN, BK, WAVES_M, WAVES_N, false, 16, 1>(p, s);
        case 2: return launch_cfg2<BM, BN, BK, WAVES_M, WAVES_N, false, 16, 2>(p, s);
        case 4: return launch_cfg2<BM, BN, BK, WAVES_M, WAVES_N, false, 16, 4>(p, s);
        case 6: return launch_cfg2<BM, BN, BK, WAVES_M, WAVES_N, false, 16, 6>(p, s);
        default: return LAMP_E_UNSUPPORTED;  // A pre-norm never coincides with a residual or partials on this path
    }
}

// Debug/tuning hook (not part of the ABI header): force a tile configuration.  0 = heuristic.
static int g_force_tile = 0;
extern "C" void lamp_debug_force_gemm_tile(int cfg) { g_force_tile = cfg; }

int launch_gemm(const GemmParams& p, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.nseg < 1 || p.nseg > GEMM_MAX_SEG) return LAMP_E_DIMS;
    if ((p.K & 3) || (p.lda & 3) || (p.ldw & 3)) return LAMP_E_ALIGN;
    if (!p.A) return LAMP_E_NULL;
    if (!aligned16(p.A)) return LAMP_E_ALIGN;
    for (int i = 0; i < p.nseg; ++i) {
        if (!p.W[i] || !p.C[i]) return LAMP_E_NULL;
        if (!aligned16(p.W[i])) return LAMP_E_ALIGN;
    }
    const double flops = 2.0 * double(p.M) * p.N * p.nseg * p.K;
    const double bytes = 4.0 * (double(p.M) * p.K + double(p.N) * p.nseg * p.K +
                                double(p.M) * p.N * p.nseg * (p.R ? 2 : 1));
    ProfScope prof(LAMP_K_GEMM, flops, bytes, s);
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
    // Tile choice, from tools/bench_kernels.py on MI355X (profiles/r01_gemm_tiles.txt).  Every configuration the
    // heuristic may pick is built on the 16x16x4 MFMA, whose k-accumulation order (16c + {j, 4+j, 8+j, 12+j} for
    // j = 0..3 in every 16-chunk) does not depend on BM/BN/BK -- so the choice, which depends on M (i.e. on the
    // batch size), never changes a result bit: samples stay bit-identical across batch sizes and shards.
    // Why 16x16x4 and not 32x32x2 (same peak rate): the unit of serial work is one wave's accumulator chain over K.
    // A 32x32 block at K = 512 is an 8.2 us chain; the M = B*L decoder shapes have 1440 of them for 1024 SIMDs, so
    // some SIMD runs two in sequence (16.4 us) however the blocks are grouped into workgroups -- measured 21.5 us for
    // every 32x32x2 tile from 32x32 (1 wave) to 128x64.  16x16 blocks are 2 us chains: 5760 / 1024 -> 12.3 us,
    // measured 17.5 us.  On the large shapes 128x64x16 with 64x32 wave tiles (4x2 blocks) reaches 135-144 TFLOP/s,
    // also ahead of the best 32x32x2 tile (128x128x32: 128-139).  The 32x32x2 tiles remain as forced configs 1-8.
    auto tiles = [&](int bm, int bn) { return ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * p.nseg; };
    const int64_t t64 = tiles(64, 64);
    int lnm = (p.r_part ? 2 : 0) | (p.part_out ? 4 : 0);
    for (int i = 0; i < p.nseg; ++i) lnm |= p.ln_s[i] ? 1 : 0;
    if (lnm) {  // same menu, deferred-LayerNorm instantiations
        if ((lnm & 1) && (!p.a_part || p.a_nparts < 1)) return LAMP_E_NULL;
        if ((lnm & 2) && (!p.R || !p.r_gamma || !p.r_beta || p.r_nparts < 1)) return LAMP_E_NULL;
        if (((lnm & 1) && p.a_nparts > 64) || ((lnm & 2) && p.r_nparts > 64)) return LAMP_E_UNSUPPORTED;  // d <= 1024
        if ((lnm & 4) && p.nseg != 1) return LAMP_E_UNSUPPORTED;
        if (tiles(128, 64) >= 2048 && p.K >= 512) return launch_cfg_ln<128, 64, 16, 2, 2>(p, lnm, s);
        if (t64 >= 2048) return launch_cfg_ln<64, 64, 32, 2, 2>(p, lnm, s);
        if (t64 >= 1200) return launch_cfg_ln<64, 64, 16, 2, 2>(p, lnm, s);
        return launch_cfg_ln<32, 64, 32, 1, 4>(p, lnm, s);
    }
    if (tiles(128, 64) >= 2048 && p.K >= 512) return launch_cfg<128, 64, 16, 2, 2, 16>(p, s);  // short K: fewer, deeper steps
    if (t64 >= 2048) return launch_cfg<64, 64, 32, 2, 2, 16>(p, s);
    if (t64 >= 1200) return launch_cfg<64, 64, 16, 2, 2, 16>(p, s);
    return launch_cfg<32, 64, 32, 1, 4, 16>(p, s);  // 4 waves of 32x16 (2 blocks each)
}

}  // namespace lamp
