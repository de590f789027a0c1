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
#include "lamp_asm.h"

namespace lamp {

namespace {
constexpr int ROWS = 16;          // rows of a panel = the MFMA block edge
constexpr int BK = 32;            // k per stage of the W stream: whole 128-byte lines per W row

}  // namespace

// One GEMM step of the chain: dst = act(src . W_s^T + bias_s) (+ what dst held), s < nseg; dst in LDS and / or global memory.
struct ChainGemm {
    const float* W[3];
    const float* Wp[3];  // the same matrices in the fragment-major layout of lamp_pack_weight (format 0; packed geometries), else unused
    const float* Wq[3];  // ... in format 1 (64 columns x 16 k chunks: chain_rows4_kernel), else unused
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
    // chain_rows4_kernel, when the LayerNorm operand rows do not fit LDS beside the panel (d_ff = 1024: bibtex).  res_alias: the
    // modulo-residual rows live in the unused upper half of the H rows while the fc step runs (row stride hw, first column k_h:
    // needs hw >= k_h + d).  wout_late: the read-out rows are loaded into H after the W2 step has consumed it, not in the prologue.
    int res_alias, wout_late;
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
    static constexpr bool PACKED = NSLOT >= 3;               // W arrives fragment-major (lamp_pack_weight): no LDS pass, whole lines
    static constexpr int RING_FLOATS = PACKED ? 0 : WAVES * NSLOT * STAGE_FLOATS;   // (NSLOT == 4: chain_packed_kernel below)
    static constexpr int RPW = ROWS / WAVES;                 // LayerNorm rows per wave
    static_assert(NOPS <= NMF && ROWS % WAVES == 0 && (DEPTH == 2 || DEPTH == 4) && NSLOT >= 0 && NSLOT <= 4, "geometry");
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
    const unsigned ring_b = lds0 + unsigned(x_floats + h_floats + wave * ((NSLOT == 3 ? 0 : NSLOT) * STAGE_FLOATS)) * 4u;
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

#ifdef LAMP_TUNING
    unsigned long long gt[3][4] = {};   // per GEMM step: shader-clock stamps at entry, after the prologue, after the k loop, at exit
    int gt_i = 0;
#define CHAIN_STAMP(k) do { if (p.trace) gt[gt_i][k] = __builtin_readcyclecounter(); } while (0)
#else
#define CHAIN_STAMP(k) do {} while (0)
#endif
    auto gemm = [&](const ChainGemm& g) {
        CHAIN_STAMP(0);
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
        // NSLOT == 3 ("W packed"): the weights were rearranged once per weight version into the order this loop consumes them
        // (lamp_pack_weight: per 16 output columns and 32 k, the two fragment chunks lane by lane), so a fragment load is ONE
        // contiguous KiB -- eight whole lines per instruction, like the lane-linear stage loads of the LDS route, and no LDS
        // pass at all.  Same fragments, same k order: same bits.
        constexpr bool WPK = G::PACKED;
        const unsigned w_bytes = WPK ? unsigned(NB) * unsigned(nk) * 2048u
                                     : unsigned((uint64_t(WCOLS - 1) * uint64_t(ldw) + uint64_t(g.K)) * 4u);
        auto w_base = [&](int seg_, int pass_) -> const float* {
            const int c0 = pass_ * PASS_COLS + wave * WCOLS;
            if constexpr (WPK) return g.Wp[seg_] + int64_t(c0 / 16) * int64_t(nk) * 512;
            else return g.W[seg_] + int64_t(c0) * g.ldw;
        };
        constexpr unsigned KT_BYTES = WPK ? 2048u : BK * 4u;   // soffset step per k stage
        u32x4 rsW = raw_rsrc(w_base(0, 0), w_bytes);
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
        constexpr bool WDIR = NSLOT == 0 || WPK;
        constexpr int NL = 2 * NB;                 // fragment loads per stage
        f32x4 F[WDIR ? DEPTH : 1][2][NB];
        unsigned f_voff[2][NB];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                f_voff[c][j] = WPK ? unsigned((j * nk * 2 + c) * 256 + lane * 4) * 4u : unsigned((16 * j + l15) * ldw + c * 16 + hi * 4) * 4u;
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
                if (pt < total) rsW = raw_rsrc(w_base(p_seg, p_pass), w_bytes);
            }
            p_so = unsigned(p_kt) * KT_BYTES;
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
                sgpr_guard(rsW, p_so);
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
            sgpr_guard(rsW, p_so);
#pragma unroll
            for (int i = 0; i < NLD; ++i) part_load(R[j], i);
            part_book();
        });
        wait_vmcnt<(DEPTH - 1) * NLD>();
#pragma unroll
        for (int i = 0; i < NLD; ++i) part_write(R[0], 0, i);
        sgpr_guard(rsW, p_so);
#pragma unroll
        for (int i = 0; i < NLD; ++i) part_load(R[0], i);
        part_book();
        static_for<0, G::NRD>([&](auto Rr) { part_read(0, 0, decltype(Rr)::value); });
        }

        CHAIN_STAMP(1);
        for (int seg = 0; seg < g.nseg; ++seg) {
            for (int pass = 0; pass < npass; ++pass) {
                const int col0 = pass * PASS_COLS + wave * WCOLS;   // this wave's first output column of the pass
                // epilogue operand requested before the k loop (its round trip hides under it)
                const float* bias = g.bias[seg];
                // unconditional (a load under `if` merges with the zero through a register copy placed right behind the load,
                // before its data has landed -- see settle() below): without a bias the load reads the weights and is not used
                f32x4 bv[NB];
                const float* bias_src = bias ? bias + col0 : g.W[seg];
#pragma unroll
                for (int j = 0; j < NB; ++j) bv[j] = global_read16_untracked(bias_src + j * 16 + 4 * hi);
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
                            constexpr int jl = (j + DEPTH - 1) % DEPTH;   // the set stage t - 1 was multiplied from
#if defined(CHAIN_SPREAD) && CHAIN_SPREAD
                            // The NL loads of stage t + DEPTH - 1 one every NMF / NL gaps instead of in the first NL: sixteen waves
                            // that all reach their loads together fill the texture addresser's queue (one KiB-sized load
                            // occupies it ~16 cycles) and sit behind it with their MFMAs unissued.
                            constexpr int LSTEP = G::NMF / NL;
                            // DEPTH 2: a load is consumed one step after its request, so the step-wide wait becomes one counted
                            // wait in front of the first MFMA that reads each load: the NL - 1 - li younger loads of its stage
                            // plus the ones this step has issued so far may stay in flight -- always NL - 1
                            if constexpr (DEPTH > 2) wait_vmcnt<(DEPTH - 2) * NL>();
                            wait_lgkmcnt<0>();
                            static_for<0, G::NMF>([&](auto I) {
                                constexpr int i = decltype(I)::value, c = i / (4 * NB), comp = (i % (4 * NB)) / NB, jb = i % NB;
                                if constexpr (DEPTH == 2 && comp == 0) {
                                    wait_vmcnt<NL - 1>();
                                    __builtin_amdgcn_sched_barrier(0);
                                }
#if !(defined(CHAIN_ABL) && (CHAIN_ABL & 2))
                                acc[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(F[j][c][jb][comp], fa[set][c][comp], acc[jb], 0, 0, 0);
#else
                                asm volatile("" ::"v"(F[j][c][jb]), "v"(fa[set][c]));
#endif
                                __builtin_amdgcn_sched_barrier(0);
                                if constexpr (i % LSTEP == 0) {            // stage t + DEPTH - 1 -> set jl
                                    constexpr int li = i / LSTEP;
                                    F[jl][li / NB][li % NB] = buffer_read16_untracked(rsW, f_voff[li / NB][li % NB], p_so);
                                } else if constexpr (i == 1 || i == 2) {   // A fragments of stage t + 1
                                    fa[setn][i - 1] = lds_read16(src_b + unsigned(ktn >> 1) * 256u + a_off[setn][i - 1]);
                                } else if constexpr (i == (NL - 1) * LSTEP + 1) {
                                    part_book();
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            });
                            return;
#endif
                            // stage t (set j) has landed when at most the DEPTH - 2 younger stages are outstanding
                            wait_vmcnt<(DEPTH - 2) * NL>();
                            wait_lgkmcnt<0>();
                            static_for<0, G::NMF>([&](auto I) {
                                constexpr int i = decltype(I)::value, c = i / (4 * NB), comp = (i % (4 * NB)) / NB, jb = i % NB;
#if !(defined(CHAIN_ABL) && (CHAIN_ABL & 2))
                                acc[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(F[j][c][jb][comp], fa[set][c][comp], acc[jb], 0, 0, 0);
#else
                                asm volatile("" ::"v"(F[j][c][jb]), "v"(fa[set][c]));
#endif
                                __builtin_amdgcn_sched_barrier(0);
                                if constexpr (i < NL) {                    // stage t + DEPTH - 1 -> set jl
#if !(defined(CHAIN_ABL) && (CHAIN_ABL & 16))
                                    F[jl][i / NB][i % NB] = buffer_read16_untracked(rsW, f_voff[i / NB][i % NB], p_so);
#endif
                                } else if constexpr (i < NL + 2) {         // A fragments of stage t + 1
#if !(defined(CHAIN_ABL) && (CHAIN_ABL & 8))
                                    fa[setn][i - NL] = lds_read16(src_b + unsigned(ktn >> 1) * 256u + a_off[setn][i - NL]);
#endif
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
                // requests past the end of the stream / reads behind the last stage: their registers stay live until they
                // have landed (keep_alive, lamp_asm.h)
                if (seg + 1 == g.nseg && pass + 1 == npass) wait_vmcnt<0>();
                if constexpr (WDIR) {
                    static_for<0, DEPTH - 1>([&](auto J) {   // the stages in flight behind the last step: sets 0 .. DEPTH - 2
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int jb = 0; jb < NB; ++jb) keep_alive(F[decltype(J)::value][c][jb]);
                    });
                } else {
                    static_for<0, DEPTH>([&](auto J) {
#pragma unroll
                        for (int i = 0; i < NLD; ++i) keep_alive(R[decltype(J)::value][i]);
                    });
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int jb = 0; jb < NB; ++jb) keep_alive(fw[0][c][jb]);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) keep_alive(fa[0][c]);
                CHAIN_STAMP(2);
                // ---- epilogue of the pass: lane (row l15, hi) holds columns col0 + 16 j + 4 hi .. + 3 of its row ----
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int col = col0 + j * 16 + 4 * hi;
                    const f32x4 b = bias ? bv[j] : f32x4{0.f, 0.f, 0.f, 0.f};
                    float4 v = make_float4(acc[j][0] + b[0], acc[j][1] + b[1], acc[j][2] + b[2], acc[j][3] + b[3]);
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
        CHAIN_STAMP(3);
#ifdef LAMP_TUNING
        ++gt_i;
#endif
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
        unsigned long long* o = p.trace + size_t(blockIdx.x) * 24;
        t[5] = wall_clock64();
        for (int i = 0; i < 6; ++i) o[i] = t[i];
        o[6] = __builtin_readcyclecounter() - c_begin;   // shader cycles from t[0] to t[5]: the clock the chain ran at
        o[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        for (int gi = 0; gi < 3; ++gi)   // wave 0's shader-clock stamps inside the three GEMM steps, relative to the kernel's start
            for (int k = 0; k < 4; ++k) o[8 + gi * 4 + k] = gt[gi][k] - c_begin;
    }
#endif
}


// =====================================================================================================================
// Round 5: the chain on PACKED weights (lamp_pack_weight; geometries with NSLOT == 4).
//
// What the per-wave stamps of the kernel above showed (profiles/r05_chain.txt): the four waves of a SIMD do NOT interleave their
// MFMAs -- the issue arbiter stays with the oldest ready wave, so wave 0 runs its whole k loop nearly alone (8.9 k cycles for
// 8.2 k cycles of MFMA issue), then the next one, and so on: a GEMM step costs the SUM over a SIMD's waves of each wave's own
// k-loop time, and every cycle a wave's own instruction stream leaves between two of its MFMAs is lost four times per step.
// With MFMAs only -- no loads, no LDS traffic -- the old loop still took 15.3 us per step against 13.7 us of issue: its
// per-stage bookkeeping (two taken branches, a dozen scalar instructions in one gap, address VALU with hazard nops) did not fit
// the 28 free issue cycles behind an MFMA.  Hence this kernel:
//   * W arrives fragment-major (one contiguous KiB per load instruction, straight into MFMA operand registers): no LDS pass
//     for W, no swizzle, 4 loads + 2 A-fragment reads per 16 MFMAs;
//   * the k loop is branch-free inside a group of DEPTH stages: stream bookkeeping is a handful of scalar selects split over
//     three gaps, the A-fragment addresses use the ds_read immediate offset (four address registers bumped once per group), the
//     counted wait for the NEXT stage sits behind the LAST MFMA of this one;
//   * bias, LayerNorm gain / shift, the modulo residual rows and the read-out rows are brought into LDS by LDS-DMA at kernel
//     start, together with the panel: the epilogues and the LayerNorms read LDS (~100 cycles) instead of waiting for global
//     loads (~1000 cycles) behind a workgroup barrier.
// Same fragments, same k order, same epilogue order, same LayerNorm functions: bit-identical to the kernel above.
template <int NV, int WAVES, int WCOLS, int DEPTH>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void chain_packed_kernel(ChainParams p) {
    constexpr int NB = WCOLS / 16, NL = 2 * NB, NMF = 8 * NB, PASS_COLS = WAVES * WCOLS, RPW = ROWS / WAVES;
    static_assert(NMF >= 16 && (DEPTH == 2 || DEPTH == 4) && ROWS % WAVES == 0 && 2 * NL + 5 < NMF, "geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, hi = lane >> 4;
    const int64_t row0 = int64_t(blockIdx.x) * ROWS;
    const int rows_m = int(p.M - row0 < ROWS ? p.M - row0 : ROWS);
    const unsigned lds0 = unsigned(reinterpret_cast<uintptr_t>((lds_ptr)smem));
    const int d = p.d, dff = p.has_ffn ? p.w1.N : 0;
    const int x_floats = ROWS * d, h_floats = ROWS * p.hw;
    // LDS: X | H | constants: g1, be1, g2, be2, b2 (d floats each), b1 (dff) | residual rows [16][d] | read-out rows [16][d]
    const int c_floats = x_floats + h_floats;
    const int c_g1 = c_floats, c_be1 = c_g1 + d, c_g2 = c_be1 + d, c_be2 = c_g2 + d, c_b2 = c_be2 + d, c_b1 = c_b2 + d;
    const int c_res = c_b1 + dff, c_wout = c_res + (p.ln1.res ? ROWS * d : 0);
    auto buf_b = [&](int which) { return lds0 + (which ? unsigned(x_floats) * 4u : 0u); };
    auto buf_f = [&](int which) { return smem + (which ? x_floats : 0); };
    auto buf_w = [&](int which) { return which ? p.hw : d; };

    // ---- everything this panel needs besides W -> LDS, one batch of LDS-DMA pieces (64 lanes x 16 bytes each) dealt round-robin
    // to the waves: the panel rows (XOR-swizzled on the source side, as above), then the vectors and the per-row operands of the
    // LayerNorms (lane-linear) ----
    int pc = 0;   // pieces dealt so far: the next item starts with the wave after the last one served
    auto first = [&]() { return (wave - pc) & (WAVES - 1); };
    auto load_rows = [&](const float* src, int64_t ld, int width, int which) {
        const __amdgpu_buffer_rsrc_t rs = rsrc_u(src + row0 * ld, (uint64_t(rows_m - 1) * uint64_t(ld) + uint64_t(width)) * 4u);
        const int ppr = width / 256, n = ROWS * ppr;
        float* dst = buf_f(which);
        const int rl = buf_w(which);
        for (int k = first(); k < n; k += WAVES) {
            const int r = k / ppr, part = k - r * ppr;
            const unsigned voff = unsigned(r) * unsigned(ld) * 4u + unsigned(part * 64 + (lane ^ r)) * 16u;
            lds_dma16(rs, dst + r * rl + part * 256, r < rows_m ? voff : OOB, 0);
        }
        pc += n;
    };
    auto load_vec = [&](const float* src, int n_floats, int at) {
        const __amdgpu_buffer_rsrc_t rs = rsrc_u(src, uint64_t(n_floats) * 4u);
        const int n = n_floats / 256;
        for (int k = first(); k < n; k += WAVES) lds_dma16(rs, smem + at + k * 256, unsigned(k * 64 + lane) * 16u, 0);
        pc += n;
    };
    auto load_mod_rows = [&](const float* table, int mod, int at) {   // row r <- table[(row0 + r) % mod]
        const __amdgpu_buffer_rsrc_t rs = rsrc_u(table, uint64_t(mod) * uint64_t(d) * 4u);
        const int ppr = d / 256, n = ROWS * ppr;
        for (int k = first(); k < n; k += WAVES) {
            const int r = k / ppr, part = k - r * ppr;
            const unsigned src_row = unsigned(row0 + r) % unsigned(mod);
            lds_dma16(rs, smem + at + r * d + part * 256, r < rows_m ? (src_row * unsigned(d) + unsigned(part * 64 + lane) * 4u) * 4u : OOB, 0);
        }
        pc += n;
    };
    if (p.in_x) load_rows(p.in_x, d, d, 0);
    load_rows(p.in_h, p.ld_h, p.k_h, 1);
    load_vec(p.ln1.g, d, c_g1);
    load_vec(p.ln1.b, d, c_be1);
    if (p.ln1.res) load_mod_rows(p.ln1.res, p.ln1.r_mod, c_res);
    const ChainLN& lnl = p.has_ffn ? p.ln2 : p.ln1;   // the LayerNorm that may carry the read-out
    if (p.has_ffn) {
        load_vec(p.ln2.g, d, c_g2);
        load_vec(p.ln2.b, d, c_be2);
        load_vec(p.w2.bias[0], d, c_b2);
        load_vec(p.w1.bias[0], dff, c_b1);
    }
    if (lnl.w_out) load_mod_rows(lnl.w_out, lnl.n_labels, c_wout);
    wait_vmcnt<0>();
    wg_barrier();

    unsigned a_off[2][2];   // A fragment of k-step kt, chunk c: row l15, quad 8 kt + 4 c + hi in slot quad ^ l15: [kt & 1][c]
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int c = 0; c < 2; ++c) a_off[o][c] = unsigned(((o * 8 + c * 4 + hi) ^ l15) << 4);

#ifdef LAMP_TUNING
    unsigned long long gt[3][4] = {};
    int gt_i = 0;
#endif
    // One GEMM step: dst = act(src . W^T + bias) (+ dst); bias_at: float offset of the bias vector in LDS, or -1.
    auto gemm = [&](const ChainGemm& g, int bias_at) {
        CHAIN_STAMP(0);
        const int nk = g.K / BK, npass = g.N / PASS_COLS, total = npass * nk;
        const unsigned w_bytes = unsigned(g.N) * unsigned(g.K) * 4u;
        u32x4 rsW = raw_rsrc(g.Wp[0], w_bytes);
        unsigned f_voff[2][NB];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < NB; ++j) f_voff[c][j] = unsigned((j * nk * 2 + c) * 256 + lane * 4) * 4u;
        // stream position: stage pt = (pass, kt) lives at byte ((pass * PASS_COLS + wave * WCOLS) / 16 * nk + kt) * 2048 of the
        // packed matrix (+ the lane's f_voff)
        const unsigned pass_jump = unsigned(PASS_COLS / 16 - 1) * unsigned(nk) * 2048u + 2048u;
        unsigned s_off = unsigned(wave * (WCOLS / 16)) * unsigned(nk) * 2048u;
        int pt = 0, p_kt = 0, p_wrap = 0;
        auto adv_a = [&]() { ++pt; ++p_kt; p_wrap = p_kt == nk ? 1 : 0; };
        auto adv_b = [&]() { s_off += p_wrap ? pass_jump : 2048u; p_kt = p_wrap ? 0 : p_kt; };
        auto adv_c = [&]() {
#if defined(CHAIN_ABL) && (CHAIN_ABL & 1)
            rsW[2] = 0u;
#else
            rsW[2] = pt < total ? w_bytes : 0u;   // past the end of the stream: empty descriptor (zeros, no memory access)
#endif
        };
        f32x4 F[DEPTH][2][NB], fa[2][2];
        unsigned a_cur[2][2];
        const unsigned src_b = buf_b(g.src) + unsigned(l15) * unsigned(buf_w(g.src)) * 4u;
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int c = 0; c < 2; ++c) a_cur[o][c] = src_b + a_off[o][c];
        // prologue: stages 0 .. DEPTH - 2 requested, A fragments of stage 0
        static_for<0, DEPTH - 1>([&](auto J) {
            constexpr int j = decltype(J)::value;
            sgpr_guard(rsW, s_off);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int jb = 0; jb < NB; ++jb) F[j][c][jb] = buffer_read16_untracked(rsW, f_voff[c][jb], s_off);
            adv_a(); adv_b(); adv_c();
        });
#pragma unroll
        for (int c = 0; c < 2; ++c) fa[0][c] = lds_read16(a_cur[0][c]);
        wait_vmcnt<(DEPTH - 2) * NL>();
        wait_lgkmcnt<0>();
        CHAIN_STAMP(1);
        for (int pass = 0; pass < npass; ++pass) {
            const int col0 = pass * PASS_COLS + wave * WCOLS;
            f32x4 acc[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int kt0 = 0; kt0 < nk; kt0 += DEPTH) {
                // quad offset of the NEXT group's first stage inside a row (the next pass starts at 0 again)
                const unsigned q_next = unsigned(kt0 + DEPTH == nk ? 0 : (kt0 + DEPTH) >> 1) * 256u;
                static_for<0, DEPTH>([&](auto J) {
                    constexpr int j = decltype(J)::value, set = j & 1, setn = (j + 1) & 1;
                    constexpr int jl = (j + DEPTH - 1) % DEPTH;                        // the register set stage t - 1 was multiplied from
                    constexpr int a_imm = (j + 1 == DEPTH) ? 0 : ((j + 1) >> 1) * 256;  // next stage's quad pair, relative to a_cur
                    static_for<0, NMF>([&](auto I) {
                        constexpr int i = decltype(I)::value, c = i / (4 * NB), comp = (i % (4 * NB)) / NB, jb = i % NB;
#if !(defined(CHAIN_ABL) && (CHAIN_ABL & 2))
                        acc[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(F[j][c][jb][comp], fa[set][c][comp], acc[jb], 0, 0, 0);
#else
                        asm volatile("" ::"v"(F[j][c][jb]), "v"(fa[set][c]));
#endif
                        __builtin_amdgcn_sched_barrier(0);
                        // one operation in the shadow of each MFMA (28 free issue cycles): even gaps the loads of stage
                        // t + DEPTH - 1, gaps 1 and 3 the A fragments of stage t + 1, then the bookkeeping in three parts
                        if constexpr (i % 2 == 0 && i / 2 < NL) {
                            constexpr int li = i / 2;
#if !(defined(CHAIN_ABL) && (CHAIN_ABL & 16))
                            F[jl][li / NB][li % NB] = buffer_read16_untracked(rsW, f_voff[li / NB][li % NB], s_off);
#endif
                        } else if constexpr (i == 1 || i == 3) {
#if !(defined(CHAIN_ABL) && (CHAIN_ABL & 8))
                            fa[setn][i / 2] = lds_read16_off<a_imm>(a_cur[setn][i / 2]);
#endif
                        } else if constexpr (i == 2 * NL + 1) {
                            adv_a();
                        } else if constexpr (i == 2 * NL + 2) {
                            // the address registers of the parity whose last read of this group is behind us move on to the next group
                            if constexpr (j == DEPTH - 2) {
                                a_cur[0][0] = src_b + a_off[0][0] + q_next;
                                a_cur[0][1] = src_b + a_off[0][1] + q_next;
                            } else if constexpr (j == DEPTH - 1) {
                                a_cur[1][0] = src_b + a_off[1][0] + q_next;
                                a_cur[1][1] = src_b + a_off[1][1] + q_next;
                            }
                        } else if constexpr (i == 2 * NL + 3) {
                            adv_b();
                        } else if constexpr (i == 2 * NL + 5) {
                            adv_c();
                        } else if constexpr (i == NMF - 1) {
                            // the next stage's operands: its W registers (all but the DEPTH - 2 younger stages landed) and its
                            // A fragments -- waited for behind this stage's last MFMA
                            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((DEPTH - 2) * NL) : "memory");
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            }
            // requests past the end of the stream / reads behind the last stage: live until landed (keep_alive, lamp_asm.h)
            if (pass + 1 == npass) wait_vmcnt<0>();
            static_for<0, DEPTH - 1>([&](auto J) {   // the stages in flight behind the last step: sets 0 .. DEPTH - 2
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int jb = 0; jb < NB; ++jb) keep_alive(F[decltype(J)::value][c][jb]);
            });
#pragma unroll
            for (int c = 0; c < 2; ++c) keep_alive(fa[0][c]);
            CHAIN_STAMP(2);
            // ---- epilogue of the pass: lane (row l15, hi) holds columns col0 + 16 j + 4 hi .. + 3 of its row ----
            f32x4 bv[NB], rv[NB];
            unsigned at[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int col = col0 + j * 16 + 4 * hi;
                at[j] = buf_b(g.dst) + unsigned(l15) * unsigned(buf_w(g.dst)) * 4u + unsigned(((col >> 2) ^ l15) << 4);
                // unconditional reads (see settle()): without a bias the read goes to the destination slot and is not used
                bv[j] = lds_read16(bias_at >= 0 ? lds0 + unsigned(bias_at + col) * 4u : at[j]);
                rv[j] = lds_read16(at[j]);
            }
            wait_lgkmcnt<0>();
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                settle(bv[j]);
                settle(rv[j]);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const f32x4 b = bias_at >= 0 ? bv[j] : f32x4{0.f, 0.f, 0.f, 0.f};
                float4 v = make_float4(acc[j][0] + b[0], acc[j][1] + b[1], acc[j][2] + b[2], acc[j][3] + b[3]);
                if (g.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                if (g.add_dst) v = make_float4(v.x + rv[j][0], v.y + rv[j][1], v.z + rv[j][2], v.w + rv[j][3]);
                lds_write16(at[j], f32x4{v.x, v.y, v.z, v.w});
            }
        }
        wait_vmcnt<0>();   // the empty requests past the end of the stream still write their registers
        wg_barrier();      // dst complete for every wave; src free to be overwritten by the next step
        CHAIN_STAMP(3);
#ifdef LAMP_TUNING
        ++gt_i;
#endif
    };

    // LayerNorm of the panel in place, one wave per row as in layernorm_kernel (lane l: float4 columns l + 64 i); gain, shift,
    // residual and read-out rows come from LDS.
    auto layernorm = [&](const ChainLN& n, int g_at, int be_at) {
        const int nv = d / 4;
        int cq[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) cq[i] = lane + i * 64 < nv ? lane + i * 64 : nv - 1;
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int r = wave + q * WAVES;
            const bool live = r < rows_m;
            const int64_t row = row0 + (live ? r : 0);
            const unsigned row_b = lds0 + unsigned(r) * unsigned(d) * 4u;
            f32x4 raw[NV], gg[NV], bb[NV], rr[NV], ww[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                // all five reads unconditional (see settle()): an absent operand reads the row itself and is not used
                const unsigned self = row_b + unsigned((cq[i] ^ r) << 4);
                raw[i] = lds_read16(self);
                gg[i] = lds_read16(lds0 + unsigned(g_at) * 4u + unsigned(cq[i]) * 16u);
                bb[i] = lds_read16(lds0 + unsigned(be_at) * 4u + unsigned(cq[i]) * 16u);
                rr[i] = lds_read16(n.res ? lds0 + unsigned(c_res + r * d) * 4u + unsigned(cq[i]) * 16u : self);
                ww[i] = lds_read16(n.w_out ? lds0 + unsigned(c_wout + r * d) * 4u + unsigned(cq[i]) * 16u : self);
            }
            wait_lgkmcnt<0>();
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                settle(raw[i]); settle(gg[i]); settle(bb[i]); settle(rr[i]); settle(ww[i]);
            }
            float4 v[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const bool in = lane + i * 64 < nv;
                v[i] = in ? make_float4(raw[i].x, raw[i].y, raw[i].z, raw[i].w) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (n.res && in) {
                    v[i].x += rr[i].x; v[i].y += rr[i].y; v[i].z += rr[i].z; v[i].w += rr[i].w;
                }
            }
            float mean, rstd;
            ln_row_stats<NV>(v, lane, nv, d, n.eps, mean, rstd);
            float4* yr = (n.y && live) ? reinterpret_cast<float4*>(n.y + row * d) : nullptr;
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (lane + i * 64 < nv) {
                    const float4 o = ln_row_apply(v[i], mean, rstd, make_float4(gg[i].x, gg[i].y, gg[i].z, gg[i].w),
                                                  make_float4(bb[i].x, bb[i].y, bb[i].z, bb[i].w));
                    lds_write16(row_b + unsigned((cq[i] ^ r) << 4), f32x4{o.x, o.y, o.z, o.w});
                    if (yr) yr[cq[i]] = o;
                    if (n.w_out) dot += dot4_nocontract(o, make_float4(ww[i].x, ww[i].y, ww[i].z, ww[i].w));
                }
            }
            if (n.w_out) {
                dot = wave64_sum(dot);
                if (lane == 0 && live) n.logits[row] = dot;
            }
        }
        wg_barrier();
    };

#ifdef LAMP_TUNING
    unsigned long long t[6] = {};
    const unsigned long long c_begin = __builtin_readcyclecounter();
    if (p.trace) t[0] = wall_clock64();
#endif
    gemm(p.fc, -1);
#ifdef LAMP_TUNING
    if (p.trace) t[1] = wall_clock64();
#endif
    layernorm(p.ln1, c_g1, c_be1);
#ifdef LAMP_TUNING
    if (p.trace) t[2] = wall_clock64();
#endif
    if (p.has_ffn) {
        gemm(p.w1, c_b1);
#ifdef LAMP_TUNING
        if (p.trace) t[3] = wall_clock64();
#endif
        gemm(p.w2, c_b2);
#ifdef LAMP_TUNING
        if (p.trace) t[4] = wall_clock64();
#endif
        layernorm(p.ln2, c_g2, c_be2);
    }
#ifdef LAMP_TUNING
    if (p.trace && tid == 0) {
        unsigned long long* o = p.trace + size_t(blockIdx.x) * 24;
        t[5] = wall_clock64();
        for (int i = 0; i < 6; ++i) o[i] = t[i];
        o[6] = __builtin_readcyclecounter() - c_begin;
        o[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        for (int gi = 0; gi < 3; ++gi)
            for (int k = 0; k < 4; ++k) o[8 + gi * 4 + k] = gt[gi][k] - c_begin;
    }
#endif
}


// =====================================================================================================================
// Round 5: panels of FOUR, EIGHT or TWELVE rows on v_mfma_f32_4x4x1_16b_f32.
//
// At batch 32 the decoder has 2880 rows = 180 sixteen-row panels for 256 CUs: 76 CUs idle under every chain launch, and a
// panel cannot be cut below the 16-row edge of the 16x16x4 instruction.  The 4x4x1 instruction (sixteen independent 4 x 4
// blocks, one k per instruction, same 64 FLOP per cycle and SIMD) has a row granularity of FOUR: a wave multiplies G row
// groups (R = 4 G rows) with 64 weight columns -- block h of an instruction = row group g x column group h, the A operand
// the same four rows in all sixteen blocks, the B operand one weight column per lane -- so 2880 rows become 240 panels of
// twelve rows, each with 3/4 of the work.  Why the bits do not move: tools/probes/mfma_korder.hip (profiles/r05_mfma_korder.txt)
// shows on hardware that one 16x16x4 step IS the sequential fmaf chain over its four k (it equals the host's fmaf chain bit
// for bit), and that four 4x4x1 instructions in the same k order give the same 256 results bit for bit.  The k order of
// every GEMM of this library is, per chunk of sixteen k, "for j < 4: for q < 4: k = 4 q + j" (gemm.hip's fragment trick: a
// lane's 16-byte read is four consecutive k, component j goes to step j) -- reproduced here instruction by instruction.
//
// Weights: format 1 of lamp_pack_weight -- per 64 columns and chunk of 16 k, quad q of every column lane by lane (one KiB
// per load instruction).  Eight waves x 64 columns = one pass of 512; everything else (LDS-resident operands, one operation
// per MFMA gap, stream bookkeeping, LayerNorm) as in chain_packed_kernel.
template <int NV, int G>
__global__ __launch_bounds__(512, 2) void chain_rows4_kernel(ChainParams p) {
    constexpr int WAVES = 8, R = 4 * G, DEPTH = 4, NMF = 16 * G, RPW = (R + WAVES - 1) / WAVES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l3 = lane & 3;
    const int64_t row0 = int64_t(blockIdx.x) * R;
    const int rows_m = int(p.M - row0 < R ? p.M - row0 : R);
    const unsigned lds0 = unsigned(reinterpret_cast<uintptr_t>((lds_ptr)smem));
    const int d = p.d, dff = p.has_ffn ? p.w1.N : 0;
    const int x_floats = R * d, h_floats = R * p.hw;
    const int c_floats = x_floats + h_floats;
    const int c_g1 = c_floats, c_be1 = c_g1 + d, c_g2 = c_be1 + d, c_be2 = c_g2 + d, c_b2 = c_be2 + d, c_b1 = c_b2 + d;
    const int c_res = p.res_alias ? x_floats + p.k_h : c_b1 + dff, res_ld = p.res_alias ? p.hw : d;
    const int c_wout = p.wout_late ? x_floats : (p.res_alias ? c_b1 + dff : c_res + (p.ln1.res ? R * d : 0));
    auto buf_b = [&](int which) { return lds0 + (which ? unsigned(x_floats) * 4u : 0u); };
    auto buf_f = [&](int which) { return smem + (which ? x_floats : 0); };
    auto buf_w = [&](int which) { return which ? p.hw : d; };

    int pc = 0;
    auto first = [&]() { return (wave - pc) & (WAVES - 1); };
    auto load_rows = [&](const float* src, int64_t ld, int width, int which) {   // quad q of row r in slot q ^ r (r < 16)
        const __amdgpu_buffer_rsrc_t rs = rsrc_u(src + row0 * ld, (uint64_t(rows_m - 1) * uint64_t(ld) + uint64_t(width)) * 4u);
        const int ppr = width / 256, n = R * ppr;
        float* dst = buf_f(which);
        const int rl = buf_w(which);
        for (int k = first(); k < n; k += WAVES) {
            const int r = k / ppr, part = k - r * ppr;
            const unsigned voff = unsigned(r) * unsigned(ld) * 4u + unsigned(part * 64 + (lane ^ (r & 15))) * 16u;
            lds_dma16(rs, dst + r * rl + part * 256, r < rows_m ? voff : OOB, 0);
        }
        pc += n;
    };
    auto load_vec = [&](const float* src, int n_floats, int at) {
        const __amdgpu_buffer_rsrc_t rs = rsrc_u(src, uint64_t(n_floats) * 4u);
        const int n = n_floats / 256;
        for (int k = first(); k < n; k += WAVES) lds_dma16(rs, smem + at + k * 256, unsigned(k * 64 + lane) * 16u, 0);
        pc += n;
    };
    auto load_mod_rows = [&](const float* table, int mod, int at, int ld) {
        const __amdgpu_buffer_rsrc_t rs = rsrc_u(table, uint64_t(mod) * uint64_t(d) * 4u);
        const int ppr = d / 256, n = R * ppr;
        for (int k = first(); k < n; k += WAVES) {
            const int r = k / ppr, part = k - r * ppr;
            const unsigned src_row = unsigned(row0 + r) % unsigned(mod);
            lds_dma16(rs, smem + at + r * ld + part * 256, r < rows_m ? (src_row * unsigned(d) + unsigned(part * 64 + lane) * 4u) * 4u : OOB, 0);
        }
        pc += n;
    };
    if (p.in_x) load_rows(p.in_x, d, d, 0);
    load_rows(p.in_h, p.ld_h, p.k_h, 1);
    load_vec(p.ln1.g, d, c_g1);
    load_vec(p.ln1.b, d, c_be1);
    if (p.ln1.res) load_mod_rows(p.ln1.res, p.ln1.r_mod, c_res, res_ld);
    const ChainLN& lnl = p.has_ffn ? p.ln2 : p.ln1;
    if (p.has_ffn) {
        load_vec(p.ln2.g, d, c_g2);
        load_vec(p.ln2.b, d, c_be2);
        load_vec(p.w2.bias[0], d, c_b2);
        load_vec(p.w1.bias[0], dff, c_b1);
    }
    if (lnl.w_out && !p.wout_late) load_mod_rows(lnl.w_out, lnl.n_labels, c_wout, d);
    wait_vmcnt<0>();
    wg_barrier();

#ifdef LAMP_TUNING
    unsigned long long gt[3][4] = {};
    int gt_i = 0;
#endif
    auto gemm = [&](const ChainGemm& g, int bias_at) {
        CHAIN_STAMP(0);
        const int nc = g.K / 16, npass = g.N / 512, total = npass * nc;   // stage = one chunk of 16 k x this wave's 64 columns: 4 KiB
        const unsigned w_bytes = unsigned(g.N) * unsigned(g.K) * 4u;
        u32x4 rsW = raw_rsrc(g.Wq[0], w_bytes);
        const unsigned f_voff = unsigned(lane) * 16u;   // quad q of the stage: + q KiB (instruction offset)
        const unsigned pass_jump = 7u * unsigned(nc) * 4096u + 4096u;
        unsigned s_off = unsigned(wave) * unsigned(nc) * 4096u;
        int pt = 0, p_kt = 0, p_wrap = 0;
        auto adv_a = [&]() { ++pt; ++p_kt; p_wrap = p_kt == nc ? 1 : 0; };
        auto adv_b = [&]() { s_off += p_wrap ? pass_jump : 4096u; p_kt = p_wrap ? 0 : p_kt; };
        auto adv_c = [&]() {
#if defined(CHAIN_ABL) && (CHAIN_ABL & 1)
            rsW[2] = 0u;
#else
            rsW[2] = pt < total ? w_bytes : 0u;
#endif
        };
        auto wload = [&](f32x4& dst, auto Q) { dst = buffer_read16_untracked_off<decltype(Q)::value * 1024>(rsW, f_voff, s_off); };
        f32x4 F[DEPTH][4];
        float fa[2][G];
        // A operand of chunk ch, row group g: ONE register -- lane (block b = l >> 2, i = l & 3) holds X[4 g + i][16 ch + b], and the
        // MFMA for k = 16 ch + b broadcasts block b to all sixteen blocks (cbsz = 4, abid = b).  Element (row r = 4 g + i, k) sits
        // in quad 4 ch + (b >> 2) of its row, slot quad ^ (r & 15) = 16 (ch >> 2) + 4 ((ch & 3) ^ (g & 3)) + ((b >> 2) ^ i): one address
        // register per row group + the immediate 64 ((ch & 3) ^ g); the registers move on by 256 bytes per group of four chunks.
        // A b32 read: the four rows of a block group land in four different 16-byte slots of one 64-byte window: conflict-free.
        unsigned a_cur[G];
        const unsigned src_b = buf_b(g.src), row_bytes = unsigned(buf_w(g.src)) * 4u;
        const int bq = lane >> 4, be = (lane >> 2) & 3;   // the block's k = 4 bq + be inside the chunk
        auto a_lane = [&](int gg) { return src_b + unsigned(4 * gg + l3) * row_bytes + unsigned((bq ^ l3) << 4) + unsigned(be) * 4u; };
#pragma unroll
        for (int gg = 0; gg < G; ++gg) a_cur[gg] = a_lane(gg);
        auto aread = [&](auto SET, auto GG, auto J) {
            constexpr int set = decltype(SET)::value, gg = decltype(GG)::value, j = decltype(J)::value;
            fa[set][gg] = lds_read4_off<64 * (j ^ (gg & 3))>(a_cur[gg]);
        };
        // prologue
        static_for<0, DEPTH - 1>([&](auto J) {
            constexpr int j = decltype(J)::value;
            sgpr_guard(rsW, s_off);
            static_for<0, 4>([&](auto Q) { wload(F[j][decltype(Q)::value], Q); });
            adv_a(); adv_b(); adv_c();
        });
        static_for<0, G>([&](auto GG) { aread(std::integral_constant<int, 0>{}, GG, std::integral_constant<int, 0>{}); });
        wait_vmcnt<(DEPTH - 2) * 4>();
        wait_lgkmcnt<0>();
        CHAIN_STAMP(1);
        for (int pass = 0; pass < npass; ++pass) {
            const int col = pass * 512 + wave * 64 + lane;   // this lane's output column
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
#if !(defined(CHAIN_ABL) && (CHAIN_ABL & 2))
                        acc[gg] = __builtin_amdgcn_mfma_f32_4x4x1f32(fa[set][gg], F[j][q][jj], acc[gg], 4, 4 * q + jj, 0);
#endif
                        __builtin_amdgcn_sched_barrier(0);
                        // one operation per gap: the four loads of stage t + 3, the G reads of stage t + 1, the bookkeeping, the
                        // address registers (third step of a group: all its reads are behind us), the wait for stage t + 1
                        if constexpr (i < 8 && i % 2 == 0) {
#if !(defined(CHAIN_ABL) && (CHAIN_ABL & 16))
                            wload(F[jl][i / 2], std::integral_constant<int, i / 2>{});
#endif
                        } else if constexpr (i < 2 * G && i % 2 == 1 && i != NMF - 1) {
#if !(defined(CHAIN_ABL) && (CHAIN_ABL & 8))
                            aread(std::integral_constant<int, setn>{}, std::integral_constant<int, i / 2>{}, std::integral_constant<int, jn>{});
#endif
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
            // requests past the end of the stream / reads behind the last chunk: live until landed (keep_alive, lamp_asm.h)
            if (pass + 1 == npass) wait_vmcnt<0>();
            static_for<0, DEPTH - 1>([&](auto J) { static_for<0, 4>([&](auto Q) { keep_alive(F[decltype(J)::value][decltype(Q)::value]); }); });
            static_for<0, G>([&](auto GG) { keep_alive(fa[0][decltype(GG)::value]); });
            CHAIN_STAMP(2);
            // ---- epilogue of the pass: register i of acc[gg] = row 4 gg + i, this lane's column ----
            const unsigned drow = unsigned(buf_w(g.dst)) * 4u;
            float rv[G][4];
            unsigned at[G][4];
#pragma unroll
            for (int gg = 0; gg < G; ++gg)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * gg + i;
                    at[gg][i] = buf_b(g.dst) + unsigned(r) * drow + unsigned(((col >> 2) ^ (r & 15)) << 4) + unsigned(col & 3) * 4u;
                    rv[gg][i] = lds_read4(at[gg][i]);   // unconditional (see settle()); used only with add_dst
                }
            float bias_v = lds_read4(bias_at >= 0 ? lds0 + unsigned(bias_at + col) * 4u : at[0][0]);
            wait_lgkmcnt<0>();
            settle(bias_v);
#pragma unroll
            for (int gg = 0; gg < G; ++gg)
#pragma unroll
                for (int i = 0; i < 4; ++i) settle(rv[gg][i]);
            const float bias = bias_at >= 0 ? bias_v : 0.f;
#pragma unroll
            for (int gg = 0; gg < G; ++gg)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = acc[gg][i] + bias;
                    if (g.relu) v = fmaxf(v, 0.f);
                    if (g.add_dst) v = v + rv[gg][i];
                    lds_write4(at[gg][i], v);
                }
        }
        wait_vmcnt<0>();
        wg_barrier();
        CHAIN_STAMP(3);
#ifdef LAMP_TUNING
        ++gt_i;
#endif
    };

    auto layernorm = [&](const ChainLN& n, int g_at, int be_at) {
        const int nv = d / 4;
        int cq[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) cq[i] = lane + i * 64 < nv ? lane + i * 64 : nv - 1;
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int r = wave + q * WAVES;
            if (r < R) {
                const bool live = r < rows_m;
                const int64_t row = row0 + (live ? r : 0);
                const unsigned row_b = lds0 + unsigned(r) * unsigned(d) * 4u;
                f32x4 raw[NV], gg[NV], bb[NV], rr[NV], ww[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const unsigned self = row_b + unsigned((cq[i] ^ (r & 15)) << 4);
                    raw[i] = lds_read16(self);
                    gg[i] = lds_read16(lds0 + unsigned(g_at) * 4u + unsigned(cq[i]) * 16u);
                    bb[i] = lds_read16(lds0 + unsigned(be_at) * 4u + unsigned(cq[i]) * 16u);
                    rr[i] = lds_read16(n.res ? lds0 + unsigned(c_res + r * res_ld) * 4u + unsigned(cq[i]) * 16u : self);
                    ww[i] = lds_read16(n.w_out ? lds0 + unsigned(c_wout + r * d) * 4u + unsigned(cq[i]) * 16u : self);
                }
                wait_lgkmcnt<0>();
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    settle(raw[i]); settle(gg[i]); settle(bb[i]); settle(rr[i]); settle(ww[i]);
                }
                float4 v[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const bool in = lane + i * 64 < nv;
                    v[i] = in ? make_float4(raw[i].x, raw[i].y, raw[i].z, raw[i].w) : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (n.res && in) {
                        v[i].x += rr[i].x; v[i].y += rr[i].y; v[i].z += rr[i].z; v[i].w += rr[i].w;
                    }
                }
                float mean, rstd;
                ln_row_stats<NV>(v, lane, nv, d, n.eps, mean, rstd);
                float4* yr = (n.y && live) ? reinterpret_cast<float4*>(n.y + row * d) : nullptr;
                float dot = 0.f;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    if (lane + i * 64 < nv) {
                        const float4 o = ln_row_apply(v[i], mean, rstd, make_float4(gg[i].x, gg[i].y, gg[i].z, gg[i].w),
                                                      make_float4(bb[i].x, bb[i].y, bb[i].z, bb[i].w));
                        lds_write16(row_b + unsigned((cq[i] ^ (r & 15)) << 4), f32x4{o.x, o.y, o.z, o.w});
                        if (yr) yr[cq[i]] = o;
                        if (n.w_out) dot += dot4_nocontract(o, make_float4(ww[i].x, ww[i].y, ww[i].z, ww[i].w));
                    }
                }
                if (n.w_out) {
                    dot = wave64_sum(dot);
                    if (lane == 0 && live) n.logits[row] = dot;
                }
            }
        }
        wg_barrier();
    };

#ifdef LAMP_TUNING
    unsigned long long t[6] = {};
    const unsigned long long c_begin = __builtin_readcyclecounter();
    if (p.trace) t[0] = wall_clock64();
#endif
    gemm(p.fc, -1);
#ifdef LAMP_TUNING
    if (p.trace) t[1] = wall_clock64();
#endif
    layernorm(p.ln1, c_g1, c_be1);
#ifdef LAMP_TUNING
    if (p.trace) t[2] = wall_clock64();
#endif
    if (p.has_ffn) {
        gemm(p.w1, c_b1);
#ifdef LAMP_TUNING
        if (p.trace) t[3] = wall_clock64();
#endif
        gemm(p.w2, c_b2);
#ifdef LAMP_TUNING
        if (p.trace) t[4] = wall_clock64();
#endif
        if (p.wout_late) {   // H is dead from here on (the W2 step ended with a barrier): the read-out rows move in
            load_mod_rows(p.ln2.w_out, p.ln2.n_labels, c_wout, d);
            wait_vmcnt<0>();
            wg_barrier();
        }
        layernorm(p.ln2, c_g2, c_be2);
    }
#ifdef LAMP_TUNING
    if (p.trace && tid == 0) {
        unsigned long long* o = p.trace + size_t(blockIdx.x) * 24;
        t[5] = wall_clock64();
        for (int i = 0; i < 6; ++i) o[i] = t[i];
        o[6] = __builtin_readcyclecounter() - c_begin;
        o[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        for (int gi = 0; gi < 3; ++gi)
            for (int k = 0; k < 4; ++k) o[8 + gi * 4 + k] = gt[gi][k] - c_begin;
    }
#endif
}

#ifdef LAMP_TUNING
static unsigned long long* g_chain_trace = nullptr;
extern "C" __attribute__((visibility("default"))) void lamp_debug_set_chain_trace(unsigned long long* buf) { g_chain_trace = buf; }
static int g_chain_mode = -1;   // -1 = heuristic, 0 = never, 1 = whenever the shape allows
extern "C" __attribute__((visibility("default"))) void lamp_debug_force_chain(int mode) { g_chain_mode = mode; }
static int g_chain_geom = -1;   // -1 = production choice (0, or the packed geometry when a weight pack is given), else an index into the table
extern "C" __attribute__((visibility("default"))) void lamp_debug_chain_geometry(int idx) { g_chain_geom = idx; }
#endif

// Geometries (waves, columns per wave and pass, register sets, LDS slots per wave).  Production = the first one: SIXTEEN waves
// (a 1024-thread workgroup, four waves per SIMD) of 32 columns each -- 57.4 us against 59.8-60.1 for eight waves x 64 columns
// (profiles/r04_chain.txt): twice the waves to cover each other's LDS / memory waits for twice the A-fragment reads.
// (Four waves x 128 columns -- one wave per SIMD, 430 registers -- was tried and is gone: the allocator parks part of the
// W stream's registers in AGPRs and copies them right behind the untracked loads, before the data has landed: wrong
// results, and 92 us.  The inline-assembly loads are only safe while their destination registers stay put.)
#define LAMP_CHAIN_GEOMS(X) X(0, 16, 32, 2, 1) X(1, 8, 32, 4, 2) X(2, 8, 32, 2, 2) X(3, 8, 32, 4, 1) X(4, 8, 64, 2, 1) \
    X(5, 16, 32, 2, 0) X(6, 8, 64, 4, 0) X(7, 16, 32, 2, 3) X(8, 16, 32, 4, 3) X(9, 8, 64, 2, 3) X(10, 8, 64, 4, 3) \
    X(11, 16, 32, 2, 4) X(12, 16, 32, 4, 4) X(13, 8, 64, 2, 4) X(14, 8, 64, 4, 4)
#ifndef LAMP_CHAIN_PACKED_GEOM
#define LAMP_CHAIN_PACKED_GEOM 12   // the geometry a caller-provided weight pack selects in the product build
#endif
static bool chain_geom_packed(int idx) { return idx >= 7 && idx <= 20; }   // 15-20: chain_rows4_kernel with 1-6 row groups (format-1 packs)
static bool chain_geom_lds_operands(int idx) { return idx >= 11 && idx <= 14; }   // chain_packed_kernel: LayerNorm / bias operands in LDS
#if LAMP_CHAIN_PACKED_GEOM == 11
#define LAMP_CHAIN_PACKED_CASE(X) X(11, 16, 32, 2, 4)
#elif LAMP_CHAIN_PACKED_GEOM == 12
#define LAMP_CHAIN_PACKED_CASE(X) X(12, 16, 32, 4, 4)
#elif LAMP_CHAIN_PACKED_GEOM == 13
#define LAMP_CHAIN_PACKED_CASE(X) X(13, 8, 64, 2, 4)
#else
#define LAMP_CHAIN_PACKED_CASE(X) X(14, 8, 64, 4, 4)
#endif
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
static int chain_geom_index(bool have_pack = false) {
#ifdef LAMP_TUNING
    if (g_chain_geom >= 0) return (chain_geom_packed(g_chain_geom) && !have_pack) ? 0 : g_chain_geom;
#endif
    return have_pack ? LAMP_CHAIN_PACKED_GEOM : 0;
}

// Shapes the fused chain takes: widths that tile the passes and the swizzles, everything resident in 160 KiB of LDS, and
// (the heuristic part -- results do not depend on it) a row count for which it is the faster route: no more panels than
// CUs (beyond that the separate launches, which spread a GEMM's tiles over all CUs, win), and enough of them that the
// separate launches are no longer at their latency floor.
// Which panel height serves M rows from format-1 packs: G row groups of four (chain_rows4_kernel), the fewest that still give
// every panel its own CU; 0 = not this kernel (no packs, widths that do not tile 512-column passes, more than 3072 rows, LDS).
static int rows4_groups(int64_t M, int d, int k_h, int dff, bool has_ffn, const lamp_chain_pack* pk, bool res_mod, bool w_out,
                        int* res_alias = nullptr, int* wout_late = nullptr) {
    if (res_alias) *res_alias = 0;
    if (wout_late) *wout_late = 0;
    if (!pk || !pk->fc4 || (has_ffn && (!pk->w14 || !pk->w24))) return 0;
    int G = int((M + 4 * 256 - 1) / (4 * 256));
    bool forced = false;
#ifdef LAMP_TUNING
    if (g_chain_geom >= 15 && g_chain_geom <= 20) {
        G = g_chain_geom - 14;
        forced = true;
    } else if (g_chain_geom >= 0) {
        return 0;
    }
#endif
    // one to three row groups up to 3072 rows; 3073-4096 rows belong to the sixteen-row panels of chain_packed_kernel (same
    // MFMA count, the 16x16x4 instruction at a higher clock); five and six row groups carry the chain to 6144 rows
    if (G < 1 || G > 6 || (G == 4 && !forced) || M > int64_t(4 * G) * 65535 || d % 512 || k_h % 64 || (has_ffn && dff % 512)) return 0;
    const int hw = has_ffn ? (k_h > dff ? k_h : dff) : k_h;
    const size_t base = size_t(4 * G) * size_t(d + hw) + size_t(5) * d + size_t(has_ffn ? dff : 0), rows = size_t(4 * G) * d;
    const size_t limit = size_t(160) * 1024 / 4;
    if (base + (res_mod ? rows : 0) + (w_out ? rows : 0) <= limit) return G;
    // The operand rows do not fit beside the panel (20-row panels with d_ff = 1024): both have a dead region of H to live in --
    // the modulo-residual rows the columns past k_h while the fc step runs, the read-out rows all of H after the W2 step.
    const bool ra = res_mod && has_ffn && hw >= k_h + d, wl = w_out && has_ffn && size_t(4 * G) * hw >= rows;
    if (base + (res_mod && !ra ? rows : 0) + (w_out && !wl ? rows : 0) > limit) return 0;
    if (res_alias) *res_alias = ra;
    if (wout_late) *wout_late = wl;
    return G;
}

bool chain_applies(int64_t M, int d, int k_h, int dff, bool has_ffn, const lamp_chain_pack* pk, bool res_mod, bool w_out) {
#ifdef LAMP_NO_CHAIN   // A/B builds (tools/build_variant.sh with EXTRA=-DLAMP_NO_CHAIN=1): always the separate launches
    (void)M; (void)d; (void)k_h; (void)dff; (void)has_ffn; (void)pk; (void)res_mod; (void)w_out;
    return false;
#else
    const ChainGeomInfo gi = chain_geom(chain_geom_index());
    const int hw = has_ffn ? (k_h > dff ? k_h : dff) : k_h;
    // N of every GEMM = whole passes; K of every GEMM = whole groups of DEPTH <= 4 stages (128 k) and whole 1 KiB row pieces.
    // (The product geometries tile 512 columns per pass: d_model 512 only; the 256-wide tuning geometries take d_model 256.)
    if (M <= 0 || gi.waves == 0 || d % gi.pass_cols || k_h % 256 || (has_ffn && (dff % gi.pass_cols || dff % 256)) || d % 256 || d > 512)
        return false;
    if (size_t(ROWS) * size_t(d + hw) * 4 + size_t(gi.ring_floats) * 4 > size_t(160) * 1024) return false;
#ifdef LAMP_TUNING
    if (g_chain_mode == 0) return false;
    if (g_chain_mode == 1) return true;
#endif
    const int64_t panels = (M + ROWS - 1) / ROWS;
    // With packed weights (round 5; profiles/r05_chain.txt, d = d_ff = 512): panels of 4 / 8 / 12 rows up to 3072 rows -- 30-31 us
    // at 720 rows (five launches: 36), 34 at 1440 (49), 38 at 2048 (50), 46 at 2400 (63), 50 at 2880-3072 (64-65) -- and sixteen-row
    // panels up to 4096 rows (54-56 us against 77-82).  Below ~500 rows the five launches sit at their latency floor (~30 us).
    if (pk && pk->fc && (!has_ffn || (pk->w1 && pk->w2))) {
        // (res_mod / w_out: this sub-chain's LayerNorm operands that have to sit in LDS beside the panel -- the modulo residual rows
        // of layer 0's first block, the read-out rows of the last block)
        if (rows4_groups(M, d, k_h, dff, has_ffn, pk, res_mod, w_out) > 0) return M >= 512;
        return panels > 128 && panels <= 256;
    }
    // From the native layouts (round 4; profiles/r04_chain.txt): the chain takes 57-58 us whatever the row count (62 at one
    // panel per CU); the five launches take 37 us at 720 rows, 52 at 1920-2048, then -- the 32 x 64 tiles of a 512-column GEMM
    // no longer fit two per CU -- 62 from 2112 rows on, 65 at 2880, 77-82 at 3360-4096.
    return panels > 128 && panels <= 256;
#endif
}

template <int NV, int WAVES, int WCOLS, int DEPTH, int NSLOT>
static int launch_chain_geom(const ChainParams& p, size_t lds, unsigned grid, hipStream_t s) {
    void (*kern)(ChainParams);
    if constexpr (NSLOT == 4) kern = chain_packed_kernel<NV, WAVES, WCOLS, DEPTH>;
    else kern = chain_kernel<NV, WAVES, WCOLS, DEPTH, NSLOT>;
    static AttrOnce once;
    if (int e = once.set(reinterpret_cast<const void*>(kern), 160 * 1024)) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, s, p);
    return int(hipGetLastError());
}

// out = LN1(A . Wfc^T + R), then (ffn given) out = LN2(relu(out . W1^T + b1) . W2^T + b2 + out); the final rows go to `y`
// (nullable when the read-out w_out is given).  R: `res` rows, or -- r_mod > 0 -- res[row % r_mod] added by LN1 (layer 0).
// `pk` (nullable): fragment-major copies of w_fc / w1 / w2 (lamp_pack_weight) -- selects the W-packed geometry, same bits.
int launch_chain(const float* A, int64_t lda, int k_h, const float* res, int64_t r_mod, int64_t M, int d, const float* w_fc,
                 const float* ln1_g, const float* ln1_b, const lamp_ffn_weights* ffn, int dff, float* y, const float* w_out,
                 int n_labels, float* logits, hipStream_t s, const lamp_chain_pack* pk) {
    if (!A || !w_fc || !ln1_g || !ln1_b || (!y && !w_out)) return LAMP_E_NULL;
    if (ffn && (!ffn->w1 || !ffn->b1 || !ffn->w2 || !ffn->b2 || !ffn->ln_g || !ffn->ln_b)) return LAMP_E_NULL;
    if (!chain_applies(M, d, k_h, dff, ffn != nullptr, pk, res != nullptr && r_mod > 0, w_out != nullptr)) return LAMP_E_UNSUPPORTED;
    // every pointer below is read or written with 16-byte accesses (LDS-DMA rows, dwordx4 W stream, float4 LayerNorm operands)
    if (!aligned16(A) || !aligned16(res) || !aligned16(w_fc) || !aligned16(ln1_g) || !aligned16(ln1_b) || !aligned16(y) ||
        !aligned16(w_out) || (lda & 3))
        return LAMP_E_ALIGN;
    if (ffn && (!aligned16(ffn->w1) || !aligned16(ffn->b1) || !aligned16(ffn->w2) || !aligned16(ffn->b2) || !aligned16(ffn->ln_g) ||
                !aligned16(ffn->ln_b)))
        return LAMP_E_ALIGN;
    const bool have_pack = pk && pk->fc && (!ffn || (pk->w1 && pk->w2));
    const bool have_pack4 = pk && pk->fc4 && (!ffn || (pk->w14 && pk->w24));
    if (have_pack && (!aligned16(pk->fc) || !aligned16(pk->w1) || !aligned16(pk->w2))) return LAMP_E_ALIGN;
    if (have_pack4 && (!aligned16(pk->fc4) || !aligned16(pk->w14) || !aligned16(pk->w24))) return LAMP_E_ALIGN;
    ChainParams p{};
    p.M = M; p.d = d; p.hw = ffn ? (k_h > dff ? k_h : dff) : k_h;
    p.in_x = (res && r_mod == 0) ? res : nullptr;
    p.in_h = A; p.ld_h = lda; p.k_h = k_h;
    auto step = [&](const float* W, const float* Wp, const float* Wq, const float* bias, int N, int K, int relu, int src, int dst, int add_dst) {
        ChainGemm g{};
        g.W[0] = W; g.Wp[0] = have_pack ? Wp : nullptr; g.Wq[0] = have_pack4 ? Wq : nullptr; g.bias[0] = bias;
        g.nseg = 1; g.N = N; g.K = K; g.ldw = K; g.ldc = 0; g.relu = relu; g.src = src; g.dst = dst; g.add_dst = add_dst;
        return g;
    };
    p.fc = step(w_fc, pk ? pk->fc : nullptr, pk ? pk->fc4 : nullptr, nullptr, d, k_h, 0, 1, 0, p.in_x ? 1 : 0);
    const bool last1 = ffn == nullptr;
    p.ln1 = ChainLN{ln1_g, ln1_b, 1e-5f, r_mod > 0 ? res : nullptr, int(r_mod), last1 ? y : nullptr, last1 ? w_out : nullptr, n_labels, logits};
    p.has_ffn = ffn ? 1 : 0;
    if (ffn) {
        p.w1 = step(ffn->w1, pk ? pk->w1 : nullptr, pk ? pk->w14 : nullptr, ffn->b1, dff, d, 1, 0, 1, 0);
        p.w2 = step(ffn->w2, pk ? pk->w2 : nullptr, pk ? pk->w24 : nullptr, ffn->b2, d, dff, 0, 1, 0, 1);
        p.ln2 = ChainLN{ffn->ln_g, ffn->ln_b, 1e-5f, nullptr, 0, y, w_out, n_labels, logits};
    }
    p.trace = nullptr;
#ifdef LAMP_TUNING
    p.trace = g_chain_trace;
#endif
    const double fl = 2.0 * double(M) * (double(d) * k_h + (ffn ? 2.0 * double(d) * dff : 0.0));
    const double by = 4.0 * (double(M) * (k_h + 2.0 * d) + double(d) * k_h + (ffn ? 2.0 * double(d) * dff : 0.0));
    const int nv = (d / 4 + 63) / 64;   // d in {256, 512}: 1 or 2 float4 per lane in the LayerNorm
    // Panels of 4 G rows (chain_rows4_kernel) when they spread the rows over more CUs than sixteen-row panels would
    {
        const ChainLN& lnl = ffn ? p.ln2 : p.ln1;
        const int G = rows4_groups(M, d, k_h, dff, ffn != nullptr, have_pack4 ? pk : nullptr, p.ln1.res != nullptr, lnl.w_out != nullptr,
                                   &p.res_alias, &p.wout_late);
        const size_t lds4 = (size_t(4 * G) * size_t(p.d + p.hw) + size_t(5) * d + size_t(ffn ? dff : 0) +
                             (p.ln1.res && !p.res_alias ? size_t(4 * G) * d : 0) + (lnl.w_out && !p.wout_late ? size_t(4 * G) * d : 0)) * 4;
        if (G > 0) {
            ProfScope prof(LAMP_K_GEMM, fl, by, s);
            const unsigned grid4 = unsigned((M + 4 * G - 1) / (4 * G));
            void (*kern)(ChainParams) = nullptr;
            switch (nv * 10 + G) {
                case 11: kern = chain_rows4_kernel<1, 1>; break;
                case 12: kern = chain_rows4_kernel<1, 2>; break;
                case 13: kern = chain_rows4_kernel<1, 3>; break;
                case 14: kern = chain_rows4_kernel<1, 4>; break;
                case 15: kern = chain_rows4_kernel<1, 5>; break;
                case 16: kern = chain_rows4_kernel<1, 6>; break;
                case 21: kern = chain_rows4_kernel<2, 1>; break;
                case 22: kern = chain_rows4_kernel<2, 2>; break;
                case 23: kern = chain_rows4_kernel<2, 3>; break;
                case 24: kern = chain_rows4_kernel<2, 4>; break;
                case 25: kern = chain_rows4_kernel<2, 5>; break;
                case 26: kern = chain_rows4_kernel<2, 6>; break;
                default: return LAMP_E_UNSUPPORTED;
            }
            static AttrOnce once[12];
            if (int e = once[(nv - 1) * 6 + G - 1].set(reinterpret_cast<const void*>(kern), 160 * 1024)) return e;
            hipLaunchKernelGGL(kern, dim3(grid4), dim3(512), lds4, s, p);
            return int(hipGetLastError());
        }
#ifdef LAMP_TUNING
        if (g_chain_geom >= 15) return LAMP_E_UNSUPPORTED;   // a forced 4x4x1 geometry that does not fit this call
#endif
    }
    int gidx = chain_geom_index(have_pack);
    size_t lds = size_t(ROWS) * size_t(p.d + p.hw) * 4 + size_t(chain_geom(gidx).ring_floats) * 4;
    if (chain_geom_lds_operands(gidx)) {
        // + the LayerNorm / bias vectors, the modulo residual rows and the read-out rows (chain_packed_kernel)
        const ChainLN& lnl = ffn ? p.ln2 : p.ln1;
        const size_t with_ops = lds + (size_t(5) * d + size_t(ffn ? dff : 0) + (p.ln1.res ? size_t(ROWS) * d : 0) + (lnl.w_out ? size_t(ROWS) * d : 0)) * 4;
        if (with_ops <= size_t(160) * 1024) lds = with_ops;
        else gidx = 8;   // does not fit beside the panel: the packed stream with the operands from global memory
    }
    ProfScope prof(LAMP_K_GEMM, fl, by, s);
    const unsigned grid = unsigned((M + ROWS - 1) / ROWS);
#define X(I, W, C, D, S) \
    case I: return nv <= 1 ? launch_chain_geom<1, W, C, D, S>(p, lds, grid, s) : launch_chain_geom<2, W, C, D, S>(p, lds, grid, s);
    switch (gidx) {
#ifdef LAMP_TUNING
        LAMP_CHAIN_GEOMS(X)
#else
        X(0, 16, 32, 2, 1)
        X(8, 16, 32, 4, 3)
        LAMP_CHAIN_PACKED_CASE(X)
#endif
        default: return LAMP_E_UNSUPPORTED;
    }
#undef X
}

// W [N, K] (leading dimension ldw) -> the order a chain kernel streams it in, one contiguous KiB per load instruction:
//   format 0 (16x16x4 fragments): per block of 16 output columns cb and 32-deep k stage kt, chunk c in {0, 1}, lane l: the four
//     consecutive k that lane feeds to the four MFMAs of the chunk --
//     packed[(((cb * nk + kt) * 2 + c) * 64 + l) * 4 + j] = W[16 cb + (l & 15)][32 kt + 16 c + 4 (l >> 4) + j];
//   format 1 (4x4x1 panels): per block of 64 output columns cb and chunk of 16 k ch, quad q, lane l = column --
//     packed[(((cb * nc + ch) * 4 + q) * 64 + l) * 4 + j] = W[64 cb + l][16 ch + 4 q + j].
template <int FORMAT>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ W, int N, int K, int64_t ldw, float* __restrict__ out) {
    const int64_t q = int64_t(blockIdx.x) * 256 + threadIdx.x;   // one float4 of the output each
    if (q >= int64_t(N) * K / 4) return;
    const int l = int(q & 63);
    float4 v;
    if constexpr (FORMAT == 0) {
        const int nk = K / BK, c = int((q >> 6) & 1);
        const int64_t st = q >> 7;
        const int kt = int(st % nk), cb = int(st / nk);
        v = *reinterpret_cast<const float4*>(W + int64_t(16 * cb + (l & 15)) * ldw + 32 * kt + 16 * c + 4 * (l >> 4));
    } else {
        const int nc = K / 16, qq = int((q >> 6) & 3);
        const int64_t st = q >> 8;
        const int ch = int(st % nc), cb = int(st / nc);
        v = *reinterpret_cast<const float4*>(W + int64_t(64 * cb + l) * ldw + 16 * ch + 4 * qq);
    }
    reinterpret_cast<float4*>(out)[q] = v;
}
int launch_pack_weight(const float* W, int N, int K, int64_t ldw, int format, float* out, hipStream_t s) {
    if (!W || !out) return LAMP_E_NULL;
    if (N <= 0 || K <= 0 || ldw < K) return LAMP_E_DIMS;
    if (format != 0 && format != 1) return LAMP_E_UNSUPPORTED;
    if (format == 0 ? (N % 16 || K % BK) : (N % 64 || K % 16)) return LAMP_E_UNSUPPORTED;
    if (!aligned16(W) || !aligned16(out) || (ldw & 3)) return LAMP_E_ALIGN;
    const int64_t n4 = int64_t(N) * K / 4;
    if (format == 0) hipLaunchKernelGGL(pack_weight_kernel<0>, dim3(unsigned((n4 + 255) / 256)), dim3(256), 0, s, W, N, K, ldw, out);
    else hipLaunchKernelGGL(pack_weight_kernel<1>, dim3(unsigned((n4 + 255) / 256)), dim3(256), 0, s, W, N, K, ldw, out);
    return int(hipGetLastError());
}

#ifdef LAMP_TUNING
// Tuning build: the chain on its own (tools/bench_kernels.py chain).  pk_*: optional packed copies of the three matrices.
extern "C" __attribute__((visibility("default"))) int lamp_debug_launch_chain(
    const float* A, long long lda, int k_h, const float* res, long long r_mod, long long M, int d, const float* w_fc,
    const float* ln1_g, const float* ln1_b, const float* w1, const float* b1, const float* w2, const float* b2,
    const float* ln2_g, const float* ln2_b, int dff, float* y, void* stream, const float* pk_fc, const float* pk_w1,
    const float* pk_w2, const float* pk_fc4, const float* pk_w14, const float* pk_w24) {
    lamp_ffn_weights f{};
    f.w1 = w1; f.b1 = b1; f.w2 = w2; f.b2 = b2; f.ln_g = ln2_g; f.ln_b = ln2_b;
    const lamp_chain_pack pk{pk_fc, pk_w1, pk_w2, pk_fc4, pk_w14, pk_w24};
    const int keep = g_chain_mode;
    g_chain_mode = 1;
    const int e = launch_chain(A, lda, k_h, res, r_mod, M, d, w_fc, ln1_g, ln1_b, w1 ? &f : nullptr, dff, y, nullptr, 0, nullptr,
                               hipStream_t(stream), (pk_fc || pk_fc4) ? &pk : nullptr);
    g_chain_mode = keep;
    return e;
}
#endif

}  // namespace lamp
