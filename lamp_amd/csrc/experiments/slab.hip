// Skinny GEMM over row SLABS (round 5): C_s = act(A . W_s^T + b_s) + R for M up to ~10 k rows, one slab of rows per CU.
//
// The encoder of the headline batch is 9664 token rows: as 64 x 64 tiles that is 1208 tiles for 256 CUs (4.72 per CU: some CU
// runs five, one generation deep, prologues and epilogues in phase -- 0.72 of the roof, profiles/r04_gemm_trace.txt).  Here a
// workgroup owns a SLAB of 4 g consecutive rows (g <= 10 row groups: 9664 rows = 242 slabs of 40), keeps it in LDS, and
// multiplies it with the WHOLE weight matrix, streamed once per CU in the packed format 1 of lamp_pack_weight straight into
// MFMA operand registers.  The instruction is v_mfma_f32_4x4x1 (sixteen 4 x 4 blocks, one k each): a wave owns 64 output
// columns, the B operand is one weight column per lane, the A operand of row group g is ONE register per 16 k -- lane (block b,
// row i) holds A[4 g + i][k0 + b] and cbsz / abid broadcast block b to all sixteen blocks -- so a chunk of 16 k costs four
// 1-KiB loads and g four-byte LDS reads for 16 g MFMAs.  Everything that made chain_rows4_kernel (chain.hip) dense applies:
// one operation per MFMA gap, branch-free stream bookkeeping, the wait for the next chunk behind the last MFMA of this one.
//
// Same bits as gemm.hip: tools/probes/mfma_korder.hip shows that a 16x16x4 step is the sequential fmaf chain over its four k
// and that 4x4x1 instructions in the same k order give the same results; the order here is the library's (per 16 k: for j < 4:
// for q < 4: k = 4 q + j), the epilogue is acc + bias, relu, + residual.  So the choice between this kernel and the tile kernel
// (made from the row count) never changes a result bit.
//
// Ragged batches: the live row count comes from device memory (m_dev); the kernel spreads THOSE rows over the grid (rows per
// workgroup = the fewest groups of four that cover them) and dispatches on its own group count, so a half-empty batch costs
// half the MFMAs -- down to the floor of streaming W once per CU.
#include "../lamp_asm.h"

namespace lamp {

namespace {
constexpr int SLAB_GMAX = 10;   // row groups of four per workgroup: 40 rows x K = 512 floats = 80 KiB of LDS
constexpr int SLAB_WAVES = 8;   // x 64 columns = one pass of 512
}  // namespace

struct SlabParams {
    const float* A;
    const float* A_dense;   // as GemmParams::A_dense
    int64_t lda, M;
    const int* m_dev;
    int K, N, nseg;         // N per segment: a multiple of 512; K a multiple of 64
    const float* Wq[GEMM_MAX_SEG];
    const float* bias[GEMM_MAX_SEG];
    float* C[GEMM_MAX_SEG];
    int64_t ldc;
    const float* R;
    int64_t ldr;
    int relu;
};

// The product of one workgroup's slab (G live row groups, already in LDS) with every segment's weights.
template <int G>
__device__ __forceinline__ void slab_body(const SlabParams& p, const unsigned lds0, const int64_t row0, const int rows_m, const int wave,
                                          const int lane) {
    constexpr int DEPTH = 4, NMF = 16 * G;
    const int l3 = lane & 3;
    const int nc = p.K / 16, npass = p.N / 512, total = npass * nc;
    const unsigned row_bytes = unsigned(p.K) * 4u;
    const int bq = lane >> 4, be = (lane >> 2) & 3;
    auto a_lane = [&](int gg) { return lds0 + unsigned(4 * gg + l3) * row_bytes + unsigned((bq ^ l3) << 4) + unsigned(be) * 4u; };
    const unsigned f_voff = unsigned(lane) * 16u;
    const unsigned w_bytes = unsigned(p.N) * unsigned(p.K) * 4u;
    const unsigned pass_jump = 7u * unsigned(nc) * 4096u + 4096u;
    for (int seg = 0; seg < p.nseg; ++seg) {
        u32x4 rsW = raw_rsrc(p.Wq[seg], w_bytes);
        unsigned s_off = unsigned(wave) * unsigned(nc) * 4096u;
        int pt = 0, p_kt = 0, p_wrap = 0;
        auto adv_a = [&]() { ++pt; ++p_kt; p_wrap = p_kt == nc ? 1 : 0; };
        auto adv_b = [&]() { s_off += p_wrap ? pass_jump : 4096u; p_kt = p_wrap ? 0 : p_kt; };
        auto adv_c = [&]() { rsW[2] = pt < total ? w_bytes : 0u; };
        auto wload = [&](f32x4& dst, auto Q) { dst = buffer_read16_untracked_off<decltype(Q)::value * 1024>(rsW, f_voff, s_off); };
        f32x4 F[DEPTH][4];
        float fa[2][G];
        unsigned a_cur[G];
#pragma unroll
        for (int gg = 0; gg < G; ++gg) a_cur[gg] = a_lane(gg);
        // element (row r = 4 g + i, k = 16 ch + b): quad 4 ch + (b >> 2) of the row, slot quad ^ (r & 15) =
        // 16 (ch >> 2) + 4 ((ch & 3) ^ (g & 3)) + ((b >> 2) ^ i)
        auto aread = [&](auto SET, auto GG, auto J) {
            constexpr int set = decltype(SET)::value, gg = decltype(GG)::value, j = decltype(J)::value;
            fa[set][gg] = lds_read4_off<64 * (j ^ (gg & 3))>(a_cur[gg]);
        };
        static_for<0, DEPTH - 1>([&](auto J) {
            constexpr int j = decltype(J)::value;
            sgpr_guard(rsW, s_off);
            static_for<0, 4>([&](auto Q) { wload(F[j][decltype(Q)::value], Q); });
            adv_a(); adv_b(); adv_c();
        });
        static_for<0, G>([&](auto GG) { aread(std::integral_constant<int, 0>{}, GG, std::integral_constant<int, 0>{}); });
        wait_vmcnt<(DEPTH - 2) * 4>();
        wait_lgkmcnt<0>();
        const float* bias = p.bias[seg];
        float* C = p.C[seg];
        for (int pass = 0; pass < npass; ++pass) {
            const int col = pass * 512 + wave * 64 + lane;
            f32x4 acc[G];
#pragma unroll
            for (int gg = 0; gg < G; ++gg) acc[gg] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int kt0 = 0; kt0 < nc; kt0 += DEPTH) {
                const unsigned q_next = unsigned(kt0 + DEPTH == nc ? 0 : (kt0 + DEPTH) >> 2) * 256u;
                static_for<0, DEPTH>([&](auto J) {
                    constexpr int j = decltype(J)::value, set = j & 1, setn = (j + 1) & 1;
                    constexpr int jl = (j + DEPTH - 1) % DEPTH, jn = (j + 1) % DEPTH;
                    static_for<0, NMF>([&](auto I) {
                        // MFMA i of the chunk: component jj of quad q, row group gg -- k = 16 ch + 4 q + jj in the library's order
                        constexpr int i = decltype(I)::value, jj = i / (4 * G), q = (i / G) % 4, gg = i % G;
                        acc[gg] = __builtin_amdgcn_mfma_f32_4x4x1f32(fa[set][gg], F[j][q][jj], acc[gg], 4, 4 * q + jj, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        // one operation per gap.  Even gaps: the four loads of chunk t + 3, the stream bookkeeping in three parts,
                        // then (third chunk of a group: all its reads are behind us) the address registers; odd gaps: the G reads of
                        // chunk t + 1; last gap: the wait for chunk t + 1
                        if constexpr (i < 8 && i % 2 == 0) {
                            wload(F[jl][i / 2], std::integral_constant<int, i / 2>{});
                        } else if constexpr (i % 2 == 1 && i < 2 * G && i != NMF - 1) {
                            aread(std::integral_constant<int, setn>{}, std::integral_constant<int, i / 2>{}, std::integral_constant<int, jn>{});
                        } else if constexpr (i == 8) {
                            adv_a();
                        } else if constexpr (i == 10) {
                            adv_b();
                        } else if constexpr (i == 12) {
                            adv_c();
                        } else if constexpr (i >= 14 && i % 2 == 0 && (i - 14) / 2 < G && j == DEPTH - 2) {
                            a_cur[(i - 14) / 2] = a_lane((i - 14) / 2) + q_next;
                        } else if constexpr (i == NMF - 1) {
                            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((DEPTH - 2) * 4) : "memory");
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            }
            // the requests past the end of the stream (last pass) and the reads behind the last chunk have no consumer: their
            // registers -- the chunks in flight are sets 0 .. DEPTH - 2, the reads went to set 0 -- stay live until they have landed
            // (keep_alive, lamp_asm.h)
            if (pass + 1 == npass) wait_vmcnt<0>();
            static_for<0, DEPTH - 1>([&](auto J) { static_for<0, 4>([&](auto Q) { keep_alive(F[decltype(J)::value][decltype(Q)::value]); }); });
            static_for<0, G>([&](auto GG) { keep_alive(fa[0][decltype(GG)::value]); });
            // ---- epilogue of the pass: register i of acc[gg] = row 4 gg + i, this lane's column.  All global accesses are untracked
            // buffer instructions (a compiler-tracked store made hipcc drain the memory queue before EVERY row: 58 us instead of
            // 38 at 40 rows): rows past the slab's end fall outside the descriptors -- loads read 0, stores are dropped ----
            const unsigned c_voff = unsigned(col) * 4u, c_row = unsigned(p.ldc) * 4u, r_row = unsigned(p.ldr) * 4u;
            const u32x4 rsC = raw_rsrc(C + row0 * p.ldc, unsigned((uint64_t(rows_m - 1) * uint64_t(p.ldc) + uint64_t(p.N)) * 4u));
            const u32x4 rsR = raw_rsrc(p.R ? p.R + row0 * p.ldr : C, p.R ? unsigned((uint64_t(rows_m - 1) * uint64_t(p.ldr) + uint64_t(p.N)) * 4u) : 0u);
            const u32x4 rsB = raw_rsrc(bias ? bias : C, bias ? unsigned(p.N) * 4u : 0u);
            sgpr_guard(rsC);
            sgpr_guard(rsR);
            sgpr_guard(rsB);
            float b = buffer_read4_untracked(rsB, c_voff);
            float rv[G][4];
#pragma unroll
            for (int gg = 0; gg < G; ++gg)
#pragma unroll
                for (int i = 0; i < 4; ++i) rv[gg][i] = buffer_read4_untracked(rsR, c_voff + unsigned(4 * gg + i) * r_row);
            wait_vmcnt<0>();
            settle(b);
#pragma unroll
            for (int gg = 0; gg < G; ++gg)
#pragma unroll
                for (int i = 0; i < 4; ++i) settle(rv[gg][i]);
#pragma unroll
            for (int gg = 0; gg < G; ++gg)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = acc[gg][i] + b;          // an absent bias / residual read 0 through an empty descriptor: x + 0 keeps x's bits
                    if (p.relu) v = fmaxf(v, 0.f);     // (except -0 + 0 = +0, which compares equal and feeds nothing sign-sensitive)
                    if (p.R) v = v + rv[gg][i];
                    buffer_write4_untracked(rsC, c_voff + unsigned(4 * gg + i) * c_row, v);
                }
        }
        wait_vmcnt<0>();   // the empty requests past the end of the stream still write their registers
    }
}

template <int GMAX>
__global__ __launch_bounds__(512, 2) void slab_gemm_kernel(SlabParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int64_t M = p.M;
    const float* A = p.A;
    if (p.m_dev) {
        const int live = p.m_dev[0];
        M = live < M ? live : M;
        if (p.A_dense && p.m_dev[0] == p.m_dev[1]) A = p.A_dense;
    }
    // rows per workgroup: the fewest groups of four that cover the live rows with this grid
    const int groups = int((M + 3) >> 2);
    const int gpw = (groups + int(gridDim.x) - 1) / int(gridDim.x);
    const int64_t row0 = int64_t(blockIdx.x) * gpw * 4;
    if (row0 >= M || gpw < 1 || gpw > GMAX) return;
    const int rows_m = int(M - row0 < gpw * 4 ? M - row0 : gpw * 4);
    const int g_live = (rows_m + 3) >> 2;
    const unsigned lds0 = unsigned(reinterpret_cast<uintptr_t>((lds_ptr)smem));
    // the slab -> LDS: quad q of row r in slot q ^ (r & 15) (XOR on the source side of the LDS-DMA); rows past the slab's end read
    // as zeros (descriptor range check)
    {
        const __amdgpu_buffer_rsrc_t rs = rsrc_u(A + row0 * p.lda, (uint64_t(rows_m - 1) * uint64_t(p.lda) + uint64_t(p.K)) * 4u);
        const int ppr = p.K / 256, n = 4 * g_live * ppr;
        for (int k = wave; k < n; k += SLAB_WAVES) {
            const int r = k / ppr, part = k - r * ppr;
            const unsigned voff = unsigned(r) * unsigned(p.lda) * 4u + unsigned(part * 64 + (lane ^ (r & 15))) * 16u;
            lds_dma16(rs, smem + r * p.K + part * 256, r < rows_m ? voff : OOB, 0);
        }
    }
    wait_vmcnt<0>();
    wg_barrier();
    switch (g_live) {
#if !defined(SLAB_ONLY_G) || SLAB_ONLY_G == 1
        case 1: slab_body<1>(p, lds0, row0, rows_m, wave, lane); break;
#endif
#if !defined(SLAB_ONLY_G) || SLAB_ONLY_G == 2
        case 2: slab_body<2>(p, lds0, row0, rows_m, wave, lane); break;
#endif
#if !defined(SLAB_ONLY_G) || SLAB_ONLY_G == 3
        case 3: slab_body<3>(p, lds0, row0, rows_m, wave, lane); break;
#endif
#if !defined(SLAB_ONLY_G) || SLAB_ONLY_G == 4
        case 4: slab_body<4>(p, lds0, row0, rows_m, wave, lane); break;
#endif
#if !defined(SLAB_ONLY_G) || SLAB_ONLY_G == 5
        case 5: slab_body<5>(p, lds0, row0, rows_m, wave, lane); break;
#endif
#if !defined(SLAB_ONLY_G) || SLAB_ONLY_G == 6
        case 6: slab_body<6>(p, lds0, row0, rows_m, wave, lane); break;
#endif
#if !defined(SLAB_ONLY_G) || SLAB_ONLY_G == 7
        case 7: slab_body<7>(p, lds0, row0, rows_m, wave, lane); break;
#endif
#if !defined(SLAB_ONLY_G) || SLAB_ONLY_G == 8
        case 8: slab_body<8>(p, lds0, row0, rows_m, wave, lane); break;
#endif
#if !defined(SLAB_ONLY_G) || SLAB_ONLY_G == 9
        case 9: slab_body<9>(p, lds0, row0, rows_m, wave, lane); break;
#endif
#if !defined(SLAB_ONLY_G) || SLAB_ONLY_G == 10
        case 10: slab_body<10>(p, lds0, row0, rows_m, wave, lane); break;
#endif
        default: break;
    }
}

#ifdef LAMP_TUNING
static int g_slab_mode = -1;   // -1 = heuristic, 0 = never, 1 = whenever the shape allows
extern "C" __attribute__((visibility("default"))) void lamp_debug_force_slab(int mode) { g_slab_mode = mode; }
#endif

// Shapes this kernel takes (and, the heuristic part, is the faster route for): format-1 packs of every segment, widths in whole
// 512-column passes and 64-k groups, the slab of 4 g rows x K in LDS, at most one slab per CU.
bool slab_applies(int64_t M, int N, int K, int nseg, const float* const* Wq) {
    if (!Wq || M <= 0 || nseg < 1 || nseg > GEMM_MAX_SEG || N % 512 || K % 256 || K > 1024) return false;
    for (int i = 0; i < nseg; ++i)
        if (!Wq[i] || !aligned16(Wq[i])) return false;
    const int64_t groups = (M + 3) / 4;
    const int64_t gpw = (groups + 255) / 256;
    if (gpw > SLAB_GMAX || size_t(gpw) * 4 * size_t(K) * 4 > size_t(160) * 1024) return false;
#ifdef LAMP_TUNING
    if (g_slab_mode == 0) return false;
    if (g_slab_mode == 1) return true;
#endif
    // from 5 row groups per CU on (M > 4096) the tile kernel is past its one-generation shapes and this kernel's MFMAs per
    // weight byte are high enough to run near the matrix rate; below, the decoder's chain / tile launches stay
    return gpw >= 5;
}

int launch_slab_gemm(const GemmParams& g, const float* const* Wq, hipStream_t s) {
    if (!slab_applies(g.M, g.N, g.K, g.nseg, Wq)) return LAMP_E_UNSUPPORTED;
    if (!g.A || (g.A_dense && !g.m_dev)) return LAMP_E_NULL;
    if (!aligned16(g.A) || (g.A_dense && !aligned16(g.A_dense)) || (g.lda & 3)) return LAMP_E_ALIGN;
    SlabParams p{};
    p.A = g.A; p.A_dense = g.A_dense; p.lda = g.lda; p.M = g.M; p.m_dev = g.m_dev;
    p.K = g.K; p.N = g.N; p.nseg = g.nseg; p.ldc = g.ldc; p.R = g.R; p.ldr = g.ldr; p.relu = g.relu;
    for (int i = 0; i < g.nseg; ++i) {
        if (!g.C[i]) return LAMP_E_NULL;
        p.Wq[i] = Wq[i]; p.bias[i] = g.bias[i]; p.C[i] = g.C[i];
    }
    const int64_t groups = (g.M + 3) / 4;
    const int gpw = int((groups + 255) / 256);
    const unsigned grid = unsigned((groups + gpw - 1) / gpw);
    const size_t lds = size_t(gpw) * 4 * size_t(g.K) * 4;
    const double flops = 2.0 * double(g.M) * g.N * g.nseg * g.K;
    const double bytes = 4.0 * (double(g.M) * g.K + double(g.N) * g.nseg * g.K + double(g.M) * g.N * g.nseg * (g.R ? 2 : 1));
    ProfScope prof(LAMP_K_GEMM, flops, bytes, s);
    auto kern = slab_gemm_kernel<SLAB_GMAX>;
    static AttrOnce once;
    if (int e = once.set(reinterpret_cast<const void*>(kern), 160 * 1024)) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SLAB_WAVES * 64), lds, s, p);
    return int(hipGetLastError());
}

#ifdef LAMP_TUNING
// Tuning build: the slab kernel on its own (tools/bench_kernels.py slab, tests): up to two segments sharing A.
extern "C" __attribute__((visibility("default"))) int lamp_debug_slab_gemm(
    const float* A, long long M, int K, long long lda, const float* Wq0, const float* Wq1, int N, const float* bias, const float* R,
    long long ldr, int relu, float* C0, float* C1, long long ldc, const int* m_dev, void* stream) {
    GemmParams g{};
    g.A = A; g.lda = lda; g.M = M; g.K = K; g.N = N; g.nseg = Wq1 ? 2 : 1; g.ldc = ldc; g.R = R; g.ldr = ldr; g.relu = relu; g.m_dev = m_dev;
    g.bias[0] = bias; g.bias[1] = bias; g.C[0] = C0; g.C[1] = C1;
    const float* Wq[2] = {Wq0, Wq1};
    const int keep = g_slab_mode;
    g_slab_mode = 1;
    const int e = launch_slab_gemm(g, Wq, hipStream_t(stream));
    g_slab_mode = keep;
    return e;
}
#endif

}  // namespace lamp
