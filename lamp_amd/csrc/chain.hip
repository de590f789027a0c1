// The row-local part of a graph-decoder block as ONE launch (round 4):
//
//     attention output A, state Y  ->  Y = LN(A . Wfc^T + Y)  ->  H = relu(Y . W1^T + b1)  ->  Y = LN(H . W2^T + b2 + Y)
//     (lamp/SubLayers.py:110-119 after the attention, then lamp/SubLayers.py:133-142 -- the tail of MultiHeadAttention.forward and
//     PositionwiseFeedForward.forward, i.e. lamp/Layers.py:35-36 / :40-45 minus the attention products)
//
// instead of five (gemm, layernorm, gemm, gemm, layernorm).  Every step is row-wise, so a workgroup that owns a PANEL of 16 rows
// needs nothing from any other workgroup: no grid barrier, no counters, no hand-off through memory -- the panel's activations
// never leave the CU's LDS between the steps.  At batch 32 the decoder has M = B * L = 2880 rows = 180 panels for 256 CUs: the
// five launches it replaces are latency-bound there (17 us per GEMM for 12 us of matrix work, 5 us per LayerNorm for 1 us of
// memory traffic), and four such chains are half of a decoder layer pair's launches.
//
// Results are BIT-IDENTICAL to the separate launches, so the choice between the two routes (made from the row count, i.e.
// from the batch size) never changes a sample's bits:
//   * GEMM: the same v_mfma_f32_16x16x4_f32 fragments as gemm.hip (lane l: row l & 15, k = 16 c + 4 (l >> 4) + j fed to step
//     j), hence the same k-ordered fmaf chain per output element; epilogue (acc + bias, relu, + residual) in the same order.
//   * LayerNorm: ln_row_stats / ln_row_apply of lamp_kernels.h (one wave per row, the lane -> column assignment and the
//     reduction tree of layernorm_kernel, contraction off).
//
// Layout in LDS (160 KiB, one workgroup of 16 waves per CU):
//   X  [16][d]              the state rows (residual source, A operand of W1, LayerNorm in place)
//   H  [16][max(h d_v, d_ff)]  the attention output rows, then the FFN hidden rows (A operand of fc and W2)
//        both row-major, unpadded, the 16-byte quad q of row r stored in slot q ^ r: the A fragments (16 rows x one quad
//        per lane group) are then conflict-free b128 reads, and the LDS-DMA that fills them applies the XOR on its source side
//   ring: per WAVE one slot (NSLOT; two in one of the tuning geometries) of a [WCOLS W rows][32 k] stage -- production geometry:
//        32 rows, 4 KiB -- written by the wave itself from the registers its W stream arrives in (the image and swizzle of
//        gemm.hip's DMA = 2).  A wave multiplies the panel with ITS OWN output columns' weights, so the k loop has no barrier at
//        all: the sixteen waves drift apart and cover each other's waits.  (One slot is enough: a wave's LDS queue is in order, the
//        next stage's writes follow this stage's fragment reads without a wait.)
// A GEMM step walks N in passes of WAVES x WCOLS columns (16 x 32 = 512: one pass per segment at d = 512); a wave's W stream runs
// on across pass and segment boundaries (the next pass's first stages are requested while the last of this one are multiplied).
// W is streamed once per panel: 180 x 1 MiB per GEMM out of the L2s (all panels are at the same step at the same time),
// ~60-75 GB/s per CU -- whole 128-byte lines per row (BK = 32): 64-byte pieces would halve the L1 rate
// (profiles/r02_rejected_experiments.txt #9).  Why the stream goes through registers and not by LDS-DMA: DEPTH stages must be in
// flight to ride the L2 latency, and with 64 KiB of X and H resident the LDS has room for 1.5 stages per wave, the registers
// for two to four (profiles/r04_rejected_experiments.txt #4, v1).
//
// All LDS traffic of this kernel is inline assembly (explicit lgkmcnt waits): the row loads at the start are LDS-DMA, and
// hipcc orders every ds_read / ds_write it can see behind ALL earlier LDS-DMA with a full vmcnt(0) drain (gemm.hip, DMA = 1).
#include <type_traits>

#include "lamp_kernels.h"

namespace lamp {

namespace {
constexpr int ROWS = 16;          // rows of a panel = the MFMA block edge
constexpr int BK = 32;            // k per stage of the W stream: whole 128-byte lines per W row

typedef __attribute__((address_space(3))) void* lds_ptr;
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, float* dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)dst, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ f32x4 lds_read16(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// A 16-byte global load the compiler does not track: issued at the start of a pass, consumed after its k loop -- loads retire
// in order and every k step waits until at most one W stage is outstanding, so the value has landed long before.  (An
// ordinary load makes hipcc wait vmcnt(0) at the use, which drains the W stages requested ahead for the NEXT pass.)
__device__ __forceinline__ f32x4 global_read16_untracked(const float* ptr) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
// The W stream's loads, equally invisible to the compiler: its own wait-count bookkeeping is exact inside straight-line code
// but gives up at a loop's back edge -- an unrolled group of k steps then starts by draining EVERY stage in flight
// (s_waitcnt vmcnt(3) .. vmcnt(0) before the first ds_write), i.e. the prefetch depth collapses once per group.  With the
// loads in inline assembly the only vector-memory waits in the k loop are the counted ones written below.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 raw_rsrc(const float* base, unsigned bytes) {
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    return u32x4{unsigned(__builtin_amdgcn_readfirstlane(unsigned(b))), unsigned(__builtin_amdgcn_readfirstlane(unsigned(b >> 32) & 0xffffu)),
                 unsigned(__builtin_amdgcn_readfirstlane(bytes)), 0x00020000u};
}
__device__ __forceinline__ f32x4 buffer_read16_untracked(u32x4 rs, unsigned voff, unsigned soff) {
    f32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rs), "s"(soff) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write16(unsigned addr, f32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);   // hipcc moves register-only instructions (MFMAs) across an asm wait otherwise
}
__device__ __forceinline__ void wg_barrier() {   // LDS writes of this wave done, then the workgroup barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ const float* uniform_ptr(const float* q) {
    const uint64_t b = reinterpret_cast<uint64_t>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(b)), hi = __builtin_amdgcn_readfirstlane(unsigned(b >> 32));
    return reinterpret_cast<const float*>((uint64_t(hi) << 32) | lo);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_u(const float* base, uint64_t bytes) {
    const unsigned n = bytes >= 0x7fffffffull ? 0x7fffffffu : unsigned(bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(base)), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
}
}  // namespace

// One GEMM step of the chain: dst = act(src . W_s^T + bias_s) (+ what dst held), s < nseg; dst in LDS and / or global memory.
struct ChainGemm {
    const float* W[3];
    const float* bias[3];
    float* C[3];       // global destination per segment, nullable
    int nseg, N, K;    // N per segment (a multiple of 256), K = row length of src (a multiple of 64)
    int64_t ldw, ldc;
    int relu;
    int src;           // LDS buffer holding the A rows: 0 = X, 1 = H
    int dst;           // LDS buffer written (0 / 1), or -1
    int add_dst;       // residual: the element dst already holds is added (after bias / relu), in place
};
struct ChainLN {
    const float* g;
    const float* b;
    float eps;
    const float* res;  // nullable: residual rows in global memory, row % r_mod (layer 0: the shared label table)
    int r_mod;
    float* y;          // nullable: global copy of the normalised rows
    const float* w_out;   // nullable: fused read-out, logits[row] = <LN(row), w_out[row % n_labels]>
    int n_labels;
    float* logits;
};
struct ChainParams {
    int64_t M;
    int d;             // state width = row length of X
    int hw;            // row length of H
    const float* in_x; // nullable: rows loaded into X first (the residual), leading dimension d
    const float* in_h; // rows loaded into H first (the attention output), leading dimension ld_h, width k_h
    int64_t ld_h;
    int k_h;
    ChainGemm fc;      // src = H, dst = X (+ X when in_x)
    ChainLN ln1;
    ChainGemm w1;      // src = X, dst = H, relu
    ChainGemm w2;      // src = H, dst = X + X
    ChainLN ln2;
    int has_ffn;       // 0: stop after ln1
    unsigned long long* trace;   // tuning build: per-workgroup stamps
};

// Geometry: WAVES waves, each owning WCOLS output columns per pass (NB = WCOLS / 16 blocks of 16 x 16), DEPTH register sets of
// the W stream, NSLOT LDS slots per wave.  A step = one [WCOLS][32 k] stage = 8 NB MFMAs.
template <int WAVES, int WCOLS, int DEPTH, int NSLOT>
struct ChainGeom {
    static constexpr int NB = WCOLS / 16;
    static constexpr int NMF = 8 * NB;                       // MFMAs per step
    static constexpr int PASS_COLS = WAVES * WCOLS;
    static constexpr int STAGE_FLOATS = WCOLS * BK;
    static constexpr int NLD = STAGE_FLOATS / 256;           // 1 KiB loads (= stage writes) per stage
    static constexpr int NRD = 2 * (1 + NB);                 // fragment reads per stage
    static constexpr int NOPS = NLD + NRD + NLD + 1;         // intake operations per step, one per MFMA gap
    static constexpr int RING_FLOATS = WAVES * NSLOT * STAGE_FLOATS;
    static constexpr int RPW = ROWS / WAVES;                 // LayerNorm rows per wave
    static_assert(NOPS <= NMF && ROWS % WAVES == 0 && (DEPTH == 2 || DEPTH == 4) && NSLOT >= 0 && NSLOT <= 2, "geometry");
};

template <int NV, int WAVES, int WCOLS, int DEPTH, int NSLOT>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void chain_kernel(ChainParams p) {
    using G = ChainGeom<WAVES, WCOLS, DEPTH, NSLOT>;
    constexpr int NB = G::NB, NLD = G::NLD, STAGE_FLOATS = G::STAGE_FLOATS, PASS_COLS = G::PASS_COLS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, hi = lane >> 4;
    const int64_t row0 = int64_t(blockIdx.x) * ROWS;
    const int rows_m = int(p.M - row0 < ROWS ? p.M - row0 : ROWS);
    const unsigned lds0 = unsigned(reinterpret_cast<uintptr_t>((lds_ptr)smem));
    // byte addresses / float offsets of the three LDS regions
    const int x_floats = ROWS * p.d, h_floats = ROWS * p.hw;
    const unsigned ring_b = lds0 + unsigned(x_floats + h_floats + wave * (NSLOT * STAGE_FLOATS)) * 4u;
    auto buf_b = [&](int which) { return lds0 + (which ? unsigned(x_floats) * 4u : 0u); };
    auto buf_f = [&](int which) { return smem + (which ? x_floats : 0); };
    auto buf_w = [&](int which) { return which ? p.hw : p.d; };   // row length (floats)

    // ---- rows -> LDS (LDS-DMA, XOR on the source side): piece = 64 quads of one row ----
    auto load_rows = [&](const float* src, int64_t ld, int width, int which) {
        const __amdgpu_buffer_rsrc_t rs = rsrc_u(src + row0 * ld, (uint64_t(rows_m - 1) * uint64_t(ld) + uint64_t(width)) * 4u);
        const int ppr = width / 256;                 // pieces per row
        const int n = ROWS * ppr;
        float* dst = buf_f(which);
        const int rl = buf_w(which);
        for (int pc = wave; pc < n; pc += WAVES) {
            const int r = pc / ppr, part = pc - r * ppr;
            const unsigned voff = unsigned(r) * unsigned(ld) * 4u + unsigned(part * 64 + (lane ^ r)) * 16u;
            lds_dma16(rs, dst + r * rl + part * 256, r < rows_m ? voff : OOB, 0);
        }
    };
    if (p.in_x) load_rows(p.in_x, p.d, p.d, 0);
    load_rows(p.in_h, p.ld_h, p.k_h, 1);
    wait_vmcnt<0>();
    wg_barrier();

    // ---- per-lane constants of the fragment reads ----
    // A fragment of k-step kt, chunk c: row l15, quad 8 kt + 4 c + hi, stored in slot quad ^ l15 (the XOR touches the low four bits)
    unsigned a_off[2][2];   // [kt & 1][c] -> byte offset inside the 16-quad group of the row
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int c = 0; c < 2; ++c) a_off[o][c] = unsigned(((o * 8 + c * 4 + hi) ^ l15) << 4);
    // W fragment: block j (16 W rows), chunk c: stage row 16 j + l15, quad 4 c + hi in slot (quad ^ ((row >> 1) & 7))
    unsigned w_off[2];      // [c] -> byte offset inside a stage for block 0; block j adds j * 16 rows
#pragma unroll
    for (int c = 0; c < 2; ++c) w_off[c] = unsigned(l15 * BK * 4 + (((c * 4 + hi) ^ ((l15 >> 1) & 7)) << 4));
    // source of a stage: load i covers W rows 8 i .. 8 i + 7, lane -> (row 8 i + lane / 8, slot lane % 8)
    const int d_row = lane >> 3, d_slot = lane & 7;

    auto gemm = [&](const ChainGemm& g) {
        const int nk = g.K / BK, npass = g.N / PASS_COLS;
        const int total = g.nseg * npass * nk;                // stages of this wave's W stream
        const unsigned src_b = buf_b(g.src) + unsigned(l15) * unsigned(buf_w(g.src)) * 4u;
        const int ldw = int(g.ldw);
        // The W stream of this wave: stage t = (seg, pass, kt) = the [WCOLS W rows][32 k] tile of its output columns, rows
        // starting at W[seg] + (pass * PASS_COLS + wave * WCOLS) * ldw.  A stage travels
        //   global -> registers (DEPTH stages in flight: what keeps the stream coming at L2 latency under load -- with the
        //   stages in flight limited to two LDS slots, LDS-DMA straight into the ring, the chain ran 70 us for 41 us of
        //   matrix work) -> this wave's LDS slot (lane-linear image, XOR on the source side) -> MFMA fragments,
        // software-pipelined by one step (see the step below).  The LDS queue of a wave is in order, so a slot's write
        // follows its previous reads without a wait (one slot per wave is enough).  The loop body has no branches besides
        // the pass wrap of the producer: stages past the end of the stream are requested through an empty descriptor
        // (zeros, no memory access), which keeps the wait counts the same on every step.
        int pt = 0, p_seg = 0, p_pass = 0, p_kt = 0;          // producer position
        const unsigned w_bytes = unsigned((uint64_t(WCOLS - 1) * uint64_t(ldw) + uint64_t(g.K)) * 4u);
        u32x4 rsW = raw_rsrc(g.W[0] + int64_t(wave * WCOLS) * g.ldw, w_bytes);
        unsigned d_voff[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int r = 8 * i + d_row;
            d_voff[i] = unsigned(r * ldw + ((d_slot ^ ((r >> 1) & 7)) << 2)) * 4u;
        }
        f32x4 R[DEPTH][NLD];
        unsigned p_so = 0;
        // NSLOT == 0 ("W direct"): no LDS pass for W at all.  A wave is the ONLY consumer of its output columns' weights, and a
        // lane's 16-byte load of four consecutive k of W row (col0 + 16 j + l15) IS its MFMA fragment: the stream goes global ->
        // fragment registers, DEPTH stages deep (stage t + DEPTH - 1 is requested, into the set stage t - 1 was multiplied from,
        // in the gaps between the MFMAs of stage t).  Per instruction the lanes touch 16 rows x 64 bytes; the other half of each
        // 128-byte line is the next chunk's load, issued right behind it.
        constexpr bool WDIR = NSLOT == 0;
        constexpr int NL = 2 * NB;                 // fragment loads per stage
        f32x4 F[WDIR ? DEPTH : 1][2][NB];
        unsigned f_voff[2][NB];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < NB; ++j) f_voff[c][j] = unsigned((16 * j + l15) * ldw + c * 16 + hi * 4) * 4u;
        auto part_write = [&](f32x4 (&regs)[NLD], int set, int i) {   // registers -> LDS slot
            const unsigned st_b = ring_b + unsigned((NSLOT == 2 ? set : 0) * STAGE_FLOATS) * 4u;
#if !(defined(CHAIN_ABL) && (CHAIN_ABL & 4))
            lds_write16(st_b + unsigned(i * 1024 + lane * 16), regs[i]);
#endif
        };
        auto part_load = [&](f32x4 (&regs)[NLD], int i) {   // stage pt -> the registers just written out
            regs[i] = buffer_read16_untracked(rsW, d_voff[i], p_so);
        };
        auto part_book = [&]() {   // producer position
            ++pt;
            if (++p_kt == nk) {   // once per pass
                p_kt = 0;
                if (++p_pass == npass) {
                    p_pass = 0;
                    ++p_seg;
                }
                if (pt < total) rsW = raw_rsrc(g.W[p_seg] + int64_t(p_pass * PASS_COLS + wave * WCOLS) * g.ldw, w_bytes);
            }
            p_so = unsigned(p_kt) * (BK * 4u);
#if defined(CHAIN_ABL) && (CHAIN_ABL & 1)   // timing experiment: no W traffic
            rsW[2] = 0u;
#else
            if (pt >= total) rsW[2] = 0u;   // past the end of the stream: empty descriptor
#endif
        };
        f32x4 fa[2][2], fw[2][2][NB];   // [fragment set][chunk]([block])
        auto part_read = [&](int set, int kt, int r) {   // one fragment read of the stage in the slot -> fragment set
            const int c = r / (1 + NB), item = r % (1 + NB);
            const unsigned st_b = ring_b + unsigned((NSLOT == 2 ? set : 0) * STAGE_FLOATS) * 4u;
            if (item == 0) fa[set][c] = lds_read16(src_b + unsigned(kt >> 1) * 256u + a_off[set][c]);   // kt & 1 == stage & 1
            else fw[set][c][item - 1] = lds_read16(st_b + w_off[c] + unsigned((item - 1) * 16 * BK * 4));
        };
        if constexpr (WDIR) {
            // prologue: stages 0 .. DEPTH - 2 requested; A fragments of stage 0
            static_for<0, DEPTH - 1>([&](auto J) {
                constexpr int j = decltype(J)::value;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int jb = 0; jb < NB; ++jb) F[j][c][jb] = buffer_read16_untracked(rsW, f_voff[c][jb], p_so);
                part_book();
            });
#pragma unroll
            for (int c = 0; c < 2; ++c) fa[0][c] = lds_read16(src_b + a_off[0][c]);
        } else {
        // prologue: DEPTH stages requested, stage 0 through LDS into fragment set 0
        static_for<0, DEPTH>([&](auto J) {
            constexpr int j = decltype(J)::value;
#pragma unroll
            for (int i = 0; i < NLD; ++i) part_load(R[j], i);
            part_book();
        });
        wait_vmcnt<(DEPTH - 1) * NLD>();
#pragma unroll
        for (int i = 0; i < NLD; ++i) part_write(R[0], 0, i);
#pragma unroll
        for (int i = 0; i < NLD; ++i) part_load(R[0], i);
        part_book();
        static_for<0, G::NRD>([&](auto Rr) { part_read(0, 0, decltype(Rr)::value); });
        }

        for (int seg = 0; seg < g.nseg; ++seg) {
            for (int pass = 0; pass < npass; ++pass) {
                const int col0 = pass * PASS_COLS + wave * WCOLS;   // this wave's first output column of the pass
                // epilogue operand requested before the k loop (its round trip hides under it)
                const float* bias = g.bias[seg];
                f32x4 bv[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (bias) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) bv[j] = global_read16_untracked(bias + col0 + j * 16 + 4 * hi);
                }
                f32x4 acc[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int kt0 = 0; kt0 < nk; kt0 += DEPTH) {
                    static_for<0, DEPTH>([&](auto J) {
                        constexpr int j = decltype(J)::value, set = j & 1;
                        constexpr int jn = (j + 1) % DEPTH, setn = jn & 1;
                        int ktn = kt0 + j + 1;                 // in-pass step of the next stage (the next pass's 0 at the end)
                        ktn = ktn == nk ? 0 : ktn;
                        // One step = the MFMAs of stage t (fragment set `set`, complete: requested during the previous step) with
                        // the intake of stage t + 1 spread over the gaps between them, one operation per gap -- a wave's own LDS /
                        // memory instructions then issue in the shadow of its own MFMAs.  sched_barrier pins the written order.
                        if constexpr (WDIR) {
                            // stage t (set j) has landed when at most the DEPTH - 2 younger stages are outstanding
                            wait_vmcnt<(DEPTH - 2) * NL>();
                            wait_lgkmcnt<0>();
                            constexpr int jl = (j + DEPTH - 1) % DEPTH;   // the set stage t - 1 was multiplied from
                            static_for<0, G::NMF>([&](auto I) {
                                constexpr int i = decltype(I)::value, c = i / (4 * NB), comp = (i % (4 * NB)) / NB, jb = i % NB;
                                acc[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(F[j][c][jb][comp], fa[set][c][comp], acc[jb], 0, 0, 0);
                                __builtin_amdgcn_sched_barrier(0);
                                if constexpr (i < NL) {                    // stage t + DEPTH - 1 -> set jl
                                    F[jl][i / NB][i % NB] = buffer_read16_untracked(rsW, f_voff[i / NB][i % NB], p_so);
                                } else if constexpr (i < NL + 2) {         // A fragments of stage t + 1
                                    fa[setn][i - NL] = lds_read16(src_b + unsigned(ktn >> 1) * 256u + a_off[setn][i - NL]);
                                } else if constexpr (i == NL + 2) {
                                    part_book();
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            });
                            return;
                        }
                        wait_lgkmcnt<0>();
                        static_for<0, G::NMF>([&](auto I) {
                            constexpr int i = decltype(I)::value, c = i / (4 * NB), comp = (i % (4 * NB)) / NB, jb = i % NB;
#if !(defined(CHAIN_ABL) && (CHAIN_ABL & 2))
                            acc[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[set][c][jb][comp], fa[set][c][comp], acc[jb], 0, 0, 0);
#else
                            asm volatile("" ::"v"(fw[set][c][jb]), "v"(fa[set][c]));
#endif
                            __builtin_amdgcn_sched_barrier(0);
                            if constexpr (i < NLD) {                       // registers of stage t + 1 -> LDS
                                if constexpr (i == 0) wait_vmcnt<(DEPTH - 1) * NLD>();   // the oldest stage in flight has landed
                                part_write(R[jn], setn, i);
                            } else if constexpr (i < NLD + G::NRD) {      // ... and back as fragments
                                part_read(setn, ktn, i - NLD);
                            } else if constexpr (i < 2 * NLD + G::NRD) {  // stage t + 1 + DEPTH into the registers written out
                                part_load(R[jn], i - NLD - G::NRD);
                            } else if constexpr (i == 2 * NLD + G::NRD) {
                                part_book();
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        });
                    });
                }
                // ---- epilogue of the pass: lane (row l15, hi) holds columns col0 + 16 j + 4 hi .. + 3 of its row ----
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int col = col0 + j * 16 + 4 * hi;
                    float4 v = make_float4(acc[j][0] + bv[j][0], acc[j][1] + bv[j][1], acc[j][2] + bv[j][2], acc[j][3] + bv[j][3]);
                    if (g.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                    if (g.dst >= 0) {
                        const unsigned at = buf_b(g.dst) + unsigned(l15) * unsigned(buf_w(g.dst)) * 4u + unsigned(((col >> 2) ^ l15) << 4);
                        if (g.add_dst) {
                            const f32x4 r = lds_read16(at);
                            wait_lgkmcnt<0>();
                            v = make_float4(v.x + r.x, v.y + r.y, v.z + r.z, v.w + r.w);
                        }
                        lds_write16(at, f32x4{v.x, v.y, v.z, v.w});
                    }
                    if (g.C[seg] && l15 < rows_m)
                        *reinterpret_cast<float4*>(g.C[seg] + (row0 + l15) * g.ldc + col) = v;
                }
            }
        }
        wait_vmcnt<0>();   // the empty requests past the end of the stream still write their registers
        wg_barrier();      // dst complete for every wave; src free to be overwritten by the next step
    };

    // LayerNorm of the panel in place (LDS), one wave per row as in layernorm_kernel: lane l holds the float4 columns l + 64 i.
    // A wave's rows are wave + q WAVES; everything they need from global memory (gamma, beta, residual rows, read-out rows) is
    // requested up front in one batch -- the loads' round trips overlap instead of following one another.
    auto layernorm = [&](const ChainLN& n, int which) {
        constexpr int RPW = G::RPW;
        const int nv = p.d / 4;
        const float4* g4 = reinterpret_cast<const float4*>(n.g);
        const float4* b4 = reinterpret_cast<const float4*>(n.b);
        float4 gg[NV], bb[NV], rr[RPW][NV], ww[RPW][NV];
        int cq[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            cq[i] = lane + i * 64 < nv ? lane + i * 64 : nv - 1;   // clamped: the load is unconditional, the value unused
            gg[i] = g4[cq[i]];
            bb[i] = b4[cq[i]];
        }
        int64_t rowv[RPW];
        bool livev[RPW];
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int r = wave + q * WAVES;
            livev[q] = r < rows_m;
            rowv[q] = row0 + (livev[q] ? r : 0);
        }
        if (n.res) {
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
                const float4* rp = reinterpret_cast<const float4*>(n.res + (n.r_mod > 0 ? rowv[q] % n.r_mod : rowv[q]) * p.d);
#pragma unroll
                for (int i = 0; i < NV; ++i) rr[q][i] = rp[cq[i]];
            }
        }
        if (n.w_out) {
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
                const float4* wp = reinterpret_cast<const float4*>(n.w_out + (rowv[q] % n.n_labels) * p.d);
#pragma unroll
                for (int i = 0; i < NV; ++i) ww[q][i] = wp[cq[i]];
            }
        }
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int r = wave + q * WAVES;
            const unsigned row_b = buf_b(which) + unsigned(r) * unsigned(p.d) * 4u;
            f32x4 raw[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) raw[i] = lds_read16(row_b + unsigned((cq[i] ^ r) << 4));
            wait_lgkmcnt<0>();
            float4 v[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const bool in = lane + i * 64 < nv;
                v[i] = in ? make_float4(raw[i].x, raw[i].y, raw[i].z, raw[i].w) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (n.res && in) {
                    v[i].x += rr[q][i].x; v[i].y += rr[q][i].y; v[i].z += rr[q][i].z; v[i].w += rr[q][i].w;
                }
            }
            float mean, rstd;
            ln_row_stats<NV>(v, lane, nv, p.d, n.eps, mean, rstd);
            float4* yr = (n.y && livev[q]) ? reinterpret_cast<float4*>(n.y + rowv[q] * p.d) : nullptr;
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (lane + i * 64 < nv) {
                    const float4 o = ln_row_apply(v[i], mean, rstd, gg[i], bb[i]);
                    lds_write16(row_b + unsigned((cq[i] ^ r) << 4), f32x4{o.x, o.y, o.z, o.w});
                    if (yr) yr[cq[i]] = o;
                    if (n.w_out) dot += dot4_nocontract(o, ww[q][i]);
                }
            }
            if (n.w_out) {
                dot = wave64_sum(dot);
                if (lane == 0 && livev[q]) n.logits[rowv[q]] = dot;
            }
        }
        wg_barrier();
    };

#ifdef LAMP_TUNING
    unsigned long long t[6] = {};
    const unsigned long long c_begin = __builtin_readcyclecounter();
    if (p.trace) t[0] = wall_clock64();
#endif
    gemm(p.fc);
#ifdef LAMP_TUNING
    if (p.trace) t[1] = wall_clock64();
#endif
    layernorm(p.ln1, 0);
#ifdef LAMP_TUNING
    if (p.trace) t[2] = wall_clock64();
#endif
    if (p.has_ffn) {
        gemm(p.w1);
#ifdef LAMP_TUNING
        if (p.trace) t[3] = wall_clock64();
#endif
        gemm(p.w2);
#ifdef LAMP_TUNING
        if (p.trace) t[4] = wall_clock64();
#endif
        layernorm(p.ln2, 0);
    }
#ifdef LAMP_TUNING
    if (p.trace && tid == 0) {
        unsigned long long* o = p.trace + size_t(blockIdx.x) * 8;
        t[5] = wall_clock64();
        for (int i = 0; i < 6; ++i) o[i] = t[i];
        o[6] = __builtin_readcyclecounter() - c_begin;   // shader cycles from t[0] to t[5]: the clock the chain ran at
        o[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
}

#ifdef LAMP_TUNING
static unsigned long long* g_chain_trace = nullptr;
extern "C" __attribute__((visibility("default"))) void lamp_debug_set_chain_trace(unsigned long long* buf) { g_chain_trace = buf; }
static int g_chain_mode = -1;   // -1 = heuristic, 0 = never, 1 = whenever the shape allows
extern "C" __attribute__((visibility("default"))) void lamp_debug_force_chain(int mode) { g_chain_mode = mode; }
static int g_chain_geom = 0;    // 0 = production geometry, else index into the table of launch_chain
extern "C" __attribute__((visibility("default"))) void lamp_debug_chain_geometry(int idx) { g_chain_geom = idx; }
#endif

// Geometries (waves, columns per wave and pass, register sets, LDS slots per wave).  Production = the first one: SIXTEEN waves
// (a 1024-thread workgroup, four waves per SIMD) of 32 columns each -- 57.4 us against 59.8-60.1 for eight waves x 64 columns
// (profiles/r04_chain.txt): twice the waves to cover each other's LDS / memory waits for twice the A-fragment reads.
// (Four waves x 128 columns -- one wave per SIMD, 430 registers -- was tried and is gone: the allocator parks part of the
// W stream's registers in AGPRs and copies them right behind the untracked loads, before the data has landed: wrong
// results, and 92 us.  The inline-assembly loads are only safe while their destination registers stay put.)
#define LAMP_CHAIN_GEOMS(X) X(0, 16, 32, 2, 1) X(1, 8, 32, 4, 2) X(2, 8, 32, 2, 2) X(3, 8, 32, 4, 1) X(4, 8, 64, 2, 1) \
    X(5, 16, 32, 2, 0) X(6, 8, 64, 4, 0)
struct ChainGeomInfo {
    int waves, pass_cols, ring_floats;
};
static ChainGeomInfo chain_geom(int idx) {
    switch (idx) {
#define X(I, W, C, D, S) \
    case I: return ChainGeomInfo{W, ChainGeom<W, C, D, S>::PASS_COLS, ChainGeom<W, C, D, S>::RING_FLOATS};
        LAMP_CHAIN_GEOMS(X)
#undef X
        default: return ChainGeomInfo{0, 0, 0};
    }
}
static int chain_geom_index() {
#ifdef LAMP_TUNING
    return g_chain_geom;
#else
    return 0;
#endif
}

// Shapes the fused chain takes: widths that tile the passes and the swizzles, everything resident in 160 KiB of LDS, and
// (the heuristic part -- results do not depend on it) a row count for which it is the faster route: no more panels than
// CUs (beyond that the separate launches, which spread a GEMM's tiles over all CUs, win), and enough of them that the
// separate launches are no longer at their latency floor.
bool chain_applies(int64_t M, int d, int k_h, int dff, bool has_ffn) {
#ifdef LAMP_NO_CHAIN   // A/B builds (tools/build_variant.sh with EXTRA=-DLAMP_NO_CHAIN=1): always the separate launches
    return false;
#endif
    const ChainGeomInfo gi = chain_geom(chain_geom_index());
    const int hw = has_ffn ? (k_h > dff ? k_h : dff) : k_h;
    // N of every GEMM = whole passes; K of every GEMM = whole groups of DEPTH <= 4 stages (128 k) and whole 1 KiB row pieces
    if (M <= 0 || gi.waves == 0 || d % gi.pass_cols || k_h % 256 || (has_ffn && (dff % gi.pass_cols || dff % 256)) || d % 256 || d > 512)
        return false;
    if (size_t(ROWS) * size_t(d + hw) * 4 + size_t(gi.ring_floats) * 4 > size_t(160) * 1024) return false;
#ifdef LAMP_TUNING
    if (g_chain_mode == 0) return false;
    if (g_chain_mode == 1) return true;
#endif
    // Measured window (profiles/r04_chain.txt, d = d_ff = 512): the chain takes 57-58 us whatever the row count (62 at one
    // panel per CU); the five launches take 37 us at 720 rows, 52 at 1920-2048, then -- the 32 x 64 tiles of a 512-column GEMM
    // no longer fit two per CU -- 62 from 2112 rows on, 65 at 2880, 77-82 at 3360-4096.
    const int64_t panels = (M + ROWS - 1) / ROWS;
    return panels > 128 && panels <= 256;
}

template <int NV, int WAVES, int WCOLS, int DEPTH, int NSLOT>
static int launch_chain_geom(const ChainParams& p, size_t lds, unsigned grid, hipStream_t s) {
    auto kern = chain_kernel<NV, WAVES, WCOLS, DEPTH, NSLOT>;
    static AttrOnce once;
    if (int e = once.set(reinterpret_cast<const void*>(kern), 160 * 1024)) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, s, p);
    return int(hipGetLastError());
}

// out = LN1(A . Wfc^T + R), then (ffn given) out = LN2(relu(out . W1^T + b1) . W2^T + b2 + out); the final rows go to `y`
// (nullable when the read-out w_out is given).  R: `res` rows, or -- r_mod > 0 -- res[row % r_mod] added by LN1 (layer 0).
int launch_chain(const float* A, int64_t lda, int k_h, const float* res, int64_t r_mod, int64_t M, int d, const float* w_fc,
                 const float* ln1_g, const float* ln1_b, const lamp_ffn_weights* ffn, int dff, float* y, const float* w_out,
                 int n_labels, float* logits, hipStream_t s) {
    if (!A || !w_fc || !ln1_g || !ln1_b || (!y && !w_out)) return LAMP_E_NULL;
    if (ffn && (!ffn->w1 || !ffn->b1 || !ffn->w2 || !ffn->b2 || !ffn->ln_g || !ffn->ln_b)) return LAMP_E_NULL;
    if (!chain_applies(M, d, k_h, dff, ffn != nullptr)) return LAMP_E_UNSUPPORTED;
    ChainParams p{};
    p.M = M; p.d = d; p.hw = ffn ? (k_h > dff ? k_h : dff) : k_h;
    p.in_x = (res && r_mod == 0) ? res : nullptr;
    p.in_h = A; p.ld_h = lda; p.k_h = k_h;
    p.fc = ChainGemm{{w_fc, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, 1, d, k_h, k_h, 0, 0, 1, 0, p.in_x ? 1 : 0};
    const bool last1 = ffn == nullptr;
    p.ln1 = ChainLN{ln1_g, ln1_b, 1e-5f, r_mod > 0 ? res : nullptr, int(r_mod), last1 ? y : nullptr, last1 ? w_out : nullptr, n_labels, logits};
    p.has_ffn = ffn ? 1 : 0;
    if (ffn) {
        p.w1 = ChainGemm{{ffn->w1, nullptr, nullptr}, {ffn->b1, nullptr, nullptr}, {nullptr, nullptr, nullptr}, 1, dff, d, d, 0, 1, 0, 1, 0};
        p.w2 = ChainGemm{{ffn->w2, nullptr, nullptr}, {ffn->b2, nullptr, nullptr}, {nullptr, nullptr, nullptr}, 1, d, dff, dff, 0, 0, 1, 0, 1};
        p.ln2 = ChainLN{ffn->ln_g, ffn->ln_b, 1e-5f, nullptr, 0, y, w_out, n_labels, logits};
    }
    p.trace = nullptr;
#ifdef LAMP_TUNING
    p.trace = g_chain_trace;
#endif
    const int gidx = chain_geom_index();
    const size_t lds = size_t(ROWS) * size_t(p.d + p.hw) * 4 + size_t(chain_geom(gidx).ring_floats) * 4;
    const double fl = 2.0 * double(M) * (double(d) * k_h + (ffn ? 2.0 * double(d) * dff : 0.0));
    const double by = 4.0 * (double(M) * (k_h + 2.0 * d) + double(d) * k_h + (ffn ? 2.0 * double(d) * dff : 0.0));
    ProfScope prof(LAMP_K_GEMM, fl, by, s);
    const unsigned grid = unsigned((M + ROWS - 1) / ROWS);
    const int nv = (d / 4 + 63) / 64;   // d in {256, 512}: 1 or 2 float4 per lane in the LayerNorm
    switch (gidx) {
#ifdef LAMP_TUNING
#define X(I, W, C, D, S) \
    case I: return nv <= 1 ? launch_chain_geom<1, W, C, D, S>(p, lds, grid, s) : launch_chain_geom<2, W, C, D, S>(p, lds, grid, s);
        LAMP_CHAIN_GEOMS(X)
#undef X
#else
        case 0: return nv <= 1 ? launch_chain_geom<1, 16, 32, 2, 1>(p, lds, grid, s) : launch_chain_geom<2, 16, 32, 2, 1>(p, lds, grid, s);
#endif
        default: return LAMP_E_UNSUPPORTED;
    }
}

#ifdef LAMP_TUNING
// Tuning build: the chain on its own (tools/bench_kernels.py chain).
extern "C" __attribute__((visibility("default"))) int lamp_debug_launch_chain(
    const float* A, long long lda, int k_h, const float* res, long long r_mod, long long M, int d, const float* w_fc,
    const float* ln1_g, const float* ln1_b, const float* w1, const float* b1, const float* w2, const float* b2,
    const float* ln2_g, const float* ln2_b, int dff, float* y, void* stream) {
    lamp_ffn_weights f{};
    f.w1 = w1; f.b1 = b1; f.w2 = w2; f.b2 = b2; f.ln_g = ln2_g; f.ln_b = ln2_b;
    const int keep = g_chain_mode;
    g_chain_mode = 1;
    const int e = launch_chain(A, lda, k_h, res, r_mod, M, d, w_fc, ln1_g, ln1_b, w1 ? &f : nullptr, dff, y, nullptr, 0, nullptr,
                               hipStream_t(stream));
    g_chain_mode = keep;
    return e;
}
#endif

}  // namespace lamp
