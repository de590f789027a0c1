// Fused masked attention for LONG key sequences: the same arithmetic as attention.hip's attn_kernel<128, 1, 0, MK> -- same MFMA
// (v_mfma_f32_32x32x2_f32), same operand order, same lazy online softmax, hence the same bits -- with the DATA PATHS turned
// round (profiles/r05_attn_ablation.txt: in attn_kernel the K / V fragment loads of four unsynchronised waves cost 9 % of the
// kernel, every wave pulling its own copy of the tile through a 32 KiB L1 in 32-byte pieces):
//   * the 32-query block's Q fragments live in REGISTERS (64 per lane, loaded and pre-scaled once) instead of LDS;
//   * the workgroup's four waves (128 queries of one (sample, head)) share ONE copy of each 32-key K / V tile in LDS, fetched
//     by LDS-DMA (buffer_load_dwordx4 ... lds: no registers, 512-byte rows, each wave requests a quarter of the tile), double
//     buffered: tile t+1 is requested at the top of tile t's step and has the whole step (~8 k cycles) to land; one workgroup
//     barrier per tile;
//   * K rows are stored XOR-swizzled in 16-byte chunks (chunk q of row r at position q ^ (r & 15)) -- the swizzle is applied on
//     the GLOBAL side of the DMA (each lane fetches the chunk that belongs at its linear LDS position), so the A-fragment read
//     of lane (key = l & 31, hi) -- 16 bytes of ITS key row -- is conflict-free; V rows are linear (a fragment read is 512
//     contiguous bytes of one row);
//   * two workgroups per CU (64 KiB of LDS and <= 256 registers each): the partner's MFMAs cover a wave's softmax and its wait
//     at the barrier.
// Key tiles past a sample's keys and head dimensions past d_k / d_v come out of the buffer descriptors' range check as zeros
// (LDS-DMA writes the zeros, tools/probes/lds_dma_oob.hip).
// A shared mask's sparsity hint (AttnParams::tiles, one list per 32-query block) becomes the UNION of the four blocks' lists: a
// tile that is fully blocked for one of the blocks contributes exp2(-inf) = 0 to it exactly, as in the kernel without the hint.
// Reference: lamp/SubLayers.py:27-43 (ScaledDotProductAttention.forward).
#include "lamp_asm.h"

namespace lamp {
namespace {

constexpr int TILE_FLOATS = 32 * 128;                 // one K or V tile
constexpr int BUF_FLOATS = 2 * TILE_FLOATS;           // K | V
constexpr int MAX_LIST_TILES = 2048;                  // union tile list in LDS (hinted masks): lk <= 65536
constexpr size_t LDS_BYTES = size_t(2) * BUF_FLOATS * 4;
constexpr size_t LDS_BYTES_LIST = LDS_BYTES + (MAX_LIST_TILES / 32) * 4 + MAX_LIST_TILES * 4;

// One KiB of a tile: lane l's 16 bytes land at lds_addr + 16 l.  m0 carries the LDS address; the instruction behind s_mov m0
// needs one wait state.
// The tile's base goes in the SCALAR offset: the descriptor's range check covers voffset + soffset (tools/probes/lds_dma_oob.hip),
// so the per-lane offsets are loop invariants and a request costs no vector instruction.
__device__ __forceinline__ void dma16(u32x4 rs, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
                 : "memory");   // m0: nothing else in this kernel uses it (tests/test_kernel_resources.py checks the ISA)
}

template <int MK>
__global__ __launch_bounds__(256, 2) void attn_tile_kernel(AttnParams p) {
    static_assert(MK == LAMP_MASK_NONE || MK == LAMP_MASK_BITS_U32, "the plan's masks");
    constexpr int DKC = 16, DVB = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef TILE_TRACE
    const unsigned long long t_entry = wall_clock64();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int nqb = (p.lq + 127) / 128;
    const int item = xcd_remap(blockIdx.x, gridDim.x);
    const int qblk = item % nqb, bh = item / nqb;
    const int h = bh % p.H, b = bh / p.H;
    const int q0 = (qblk * 4 + wave) * 32;
    const int qi = q0 + l31;
    const bool wave_active = q0 < p.lq;
    const int qc = qi < p.lq ? qi : p.lq - 1;

    const int q_r = int(p.lay.q_r), k_r = int(p.lay.k_r), v_r = int(p.lay.v_r);
    const int lk_b = __builtin_amdgcn_readfirstlane(p.kv_len ? p.kv_len[b] : p.lk);
    const int row0 = __builtin_amdgcn_readfirstlane(p.kv_len ? p.kv_off[b] : 0);
    const int64_t k_row0 = p.kv_len ? int64_t(row0) * k_r : int64_t(b) * p.lay.k_b;
    const int64_t v_row0 = p.kv_len ? int64_t(row0) * v_r : int64_t(b) * p.lay.v_b;
    const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(p.Q + int64_t(b) * p.lay.q_b + int64_t(h) * p.lay.q_h,
                                                 (uint64_t(p.lq - 1) * q_r + p.dk) * 4u);
    const u32x4 rsK = raw_rsrc(p.K + k_row0 + int64_t(h) * p.lay.k_h, lk_b > 0 ? unsigned((uint64_t(lk_b - 1) * k_r + p.dk) * 4u) : 0u);
    const u32x4 rsV = raw_rsrc(p.V + v_row0 + int64_t(h) * p.lay.v_h, lk_b > 0 ? unsigned((uint64_t(lk_b - 1) * v_r + p.dv) * 4u) : 0u);
    const u32x4 rsM = MK == LAMP_MASK_BITS_U32
                          ? raw_rsrc(reinterpret_cast<const float*>(static_cast<const unsigned*>(p.mask) + int64_t(b) * p.m_sb),
                                     unsigned((uint64_t(p.lq - 1) * uint64_t(p.m_sq) + (p.lk + 31) / 32) * 4u))
                          : raw_rsrc(p.K, 0u);

    f32x4 qf[DKC];   // this lane's Q fragments, loaded behind the first tile's requests (below)

    // ---- the key tiles this workgroup visits ----
    const int nt = (lk_b + 31) / 32;
    int n_act = nt;
    const int* lst = nullptr;
    if (p.tiles) {
        unsigned* bm = reinterpret_cast<unsigned*>(smem + 2 * BUF_FLOATS);
        int* out = reinterpret_cast<int*>(bm + MAX_LIST_TILES / 32);
        if (tid < MAX_LIST_TILES / 32) bm[tid] = 0u;
        __syncthreads();
        if (wave_active) {
            const int* tl = p.tiles + int64_t(q0 >> 5) * p.tiles_stride;
            const int n = tl[0];
            for (int i = lane; i < n; i += 64) {
                const int t = tl[1 + i];
                if (t < nt) atomicOr(&bm[t >> 5], 1u << (t & 31));
            }
        }
        __syncthreads();
        int total = 0;
        for (int w = 0; w < (nt + 31) / 32; ++w) total += __builtin_popcount(bm[w]);
        for (int j = tid; j < nt; j += 256) {
            const int w = j >> 5;
            const unsigned word = bm[w];
            if ((word >> (j & 31)) & 1u) {
                int pos = __builtin_popcount(word & ((1u << (j & 31)) - 1u));
                for (int u = 0; u < w; ++u) pos += __builtin_popcount(bm[u]);
                out[pos] = j;
            }
        }
        __syncthreads();
        n_act = __builtin_amdgcn_readfirstlane(total);
        lst = out;
    }
    auto tile_at = [&](int idx) { return idx < n_act ? (lst ? __builtin_amdgcn_readfirstlane(lst[idx]) : idx) : nt; };

    // ---- DMA: wave w requests rows 8w .. 8w+7 of the K and of the V tile, two rows (1 KiB) per instruction ----
    // Per-lane offsets inside tile 0, with the head-dimension check folded in once (OOB + a tile's base stays out of range:
    // bases are < 2^31); the tile's base is the instruction's scalar offset.
    const int rr = lane >> 5, pp = lane & 31;   // row within the pair, linear 16-byte position
    unsigned kvo[4], vvo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * wave + 2 * i + rr;
        const int q = pp ^ (r & 15);            // the chunk that belongs at position pp of row r
        kvo[i] = 4 * q < p.dk ? unsigned(r * k_r + 4 * q) * 4u : OOB;
        vvo[i] = 4 * pp < p.dv ? unsigned(r * v_r + 4 * pp) * 4u : OOB;
    }
    const unsigned k_tile = 32u * unsigned(k_r) * 4u, v_tile = 32u * unsigned(v_r) * 4u;   // bytes from one tile to the next
    auto dma_k = [&](int i, int kt, int buf) {
        dma16(rsK, unsigned(buf) * (BUF_FLOATS * 4u) + unsigned(wave) * 4096u + 1024u * i, kvo[i], unsigned(kt) * k_tile);
    };
    auto dma_v = [&](int i, int kt, int buf) {
        dma16(rsV, unsigned(buf) * (BUF_FLOATS * 4u) + TILE_FLOATS * 4u + unsigned(wave) * 4096u + 1024u * i, vvo[i], unsigned(kt) * v_tile);
    };
    // this row's 32 mask bits of a tile: an UNTRACKED load (lamp_asm.h) -- a load hipcc counts would make it wait, in the middle
    // of QK^T, until all but one of the loads behind it have landed: the next tile's DMA
    const unsigned mrow = unsigned(int64_t(qc) * p.m_sq) * 4u;
    auto load_mask = [&](int kt) -> float {
        if constexpr (MK == LAMP_MASK_BITS_U32) return buffer_read4_untracked(rsM, mrow + unsigned(kt) * 4u);
        return 0.f;
    };
    // the value of the lane 32 away without LDS: v_permlane32_swap on (x, x) leaves (low half, low half) and (high half, high half)
    auto other_half_max = [](float x) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        float m;
        asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(r[0]), "v"(r[1]));
        return m;
    };
    auto other_half_sum = [](float x) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    };

    // Fragment addresses.  K: row l31, chunk (2c + hi) ^ (l31 & 15) = (ky ^ 32 (c & 7)) + 256 (c >> 3): EIGHT per-lane addresses,
    // everything else (c >> 3, the buffer) an immediate offset -- no vector instruction per read.  V: row 4hi + key(r), 16 bytes
    // at 4 l31: one address, the row an immediate.
    typedef __attribute__((address_space(3))) const f32x4* lds_f4;
    const unsigned ky = unsigned(l31) * 512u + unsigned((l31 & 15) ^ hi) * 16u;
    unsigned ka[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ka[j] = ky ^ (32u * j);
        asm volatile("" : "+v"(ka[j]));   // keep the eight addresses: recomputing one costs as much as an MFMA pass
    }
    unsigned va = TILE_FLOATS * 4u + unsigned(hi) * 2048u + unsigned(l31) * 16u;
    asm volatile("" : "+v"(va));
    auto read_k = [&](int buf, int c) { return *(lds_f4)(uintptr_t)(ka[c & 7] + unsigned(buf) * (BUF_FLOATS * 4u) + unsigned(c >> 3) * 256u); };
    auto read_v = [&](int buf, int r) { return *(lds_f4)(uintptr_t)(va + unsigned(buf) * (BUF_FLOATS * 4u) + unsigned((r & 3) + 8 * (r >> 2)) * 512u); };

    f32x16 o[DVB];
#pragma unroll
    for (int e = 0; e < DVB; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[e][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    constexpr float RESCALE_THR = 32.0f;   // as attn_kernel
    constexpr int AHEAD = 3;               // fragment reads in flight in front of the MFMAs that use them

    // tuning build: shader-clock cycles and 100 MHz wall clock around wave 0's key loop -> the clock the loop really runs at
#ifdef TILE_TRACE
    const unsigned long long t_c0 = __builtin_readcyclecounter(), t_w0 = wall_clock64();
    unsigned long long ph[4] = {0, 0, 0, 0};
#endif

    // ONE TILE STEP (PAR = the tile's buffer, a compile-time constant: the loop below is unrolled by two).  Every vector
    // instruction here costs matrix-pipe time (profiles/r05_mfma_chain.txt): on gfx950 the fp32 MFMA and the vector ALU are one
    // resource, ~4.6 cycles per instruction on top of the 128 x 64 of the MFMAs -- so the step is written for instruction COUNT:
    //   * blocked keys enter as the ACCUMULATOR'S initial value (-inf where blocked, 0 elsewhere: -inf + anything finite = -inf,
    //     0 + x = what the constant-zero accumulator gives): two instructions per score (bit-field extract, shift into
    //     0xff800000), none after the product; keys past the sample's last one are merged into the mask word (one scalar OR);
    //     without a mask only the sample's last, partial tile takes this path at all;
    //   * fragment addresses are loop invariants plus immediates; the DMA offsets one add per request;
    //   * score - max as eight packed subtractions; lane exchanges through v_permlane32_swap.
    auto step = [&](auto par, int idx, int kt, int kn, float mword, float& mnext) {
        constexpr int PAR = decltype(par)::value;
#ifdef TILE_TRACE
        const unsigned long long ts0 = __builtin_readcyclecounter();
        if (p.trace && tid == 0 && idx < 256)   // when this workgroup started each of its first 256 tiles
            p.trace[size_t(gridDim.x) * 8 + size_t(blockIdx.x) * 256 + idx] = wall_clock64();
#endif
        mnext = load_mask(kn);
        {
            const int valid = lk_b - kt * 32;                                     // keys of this tile (>= 1)
            const unsigned tail = valid >= 32 ? 0u : ~0u << valid;                // scalar
            f32x16 s;
            auto qk = [&](auto biased) {
                if constexpr (decltype(biased)::value) {
                    const int mw = int((__float_as_uint(mword) | tail) >> (4 * hi));
#pragma unroll
                    for (int r = 0; r < 16; ++r) {   // 0 or 0xff800000 = -inf: bit-field extract (0 / -1), AND -- in assembly, or hipcc makes it three
                        int t;
                        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(t) : "v"(mw), "n"((r & 3) + 8 * (r >> 2)));
                        s[r] = __int_as_float(t & int(0xff800000u));
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = 0.f;
                }
                f32x4 kk[AHEAD + 1];
#pragma unroll
                for (int c = 0; c < AHEAD; ++c) kk[c] = read_k(PAR, c);
#pragma unroll
                for (int c = 0; c < DKC; ++c) {
                    if (c + AHEAD < DKC) kk[(c + AHEAD) % (AHEAD + 1)] = read_k(PAR, c + AHEAD);
                    if ((c & 1) == 0) {   // the next tile's requests, one per two fragments
                        if (c < 8) dma_k(c >> 1, kn, PAR ^ 1);
                        else dma_v((c - 8) >> 1, kn, PAR ^ 1);
                    }
                    const f32x4 k4 = kk[c % (AHEAD + 1)];
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4[0], qf[c][0], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4[1], qf[c][1], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4[2], qf[c][2], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(k4[3], qf[c][3], s, 0, 0, 0);
                }
            };
            if (MK == LAMP_MASK_BITS_U32 || tail != 0u) qk(std::true_type{});
            else qk(std::false_type{});
#ifdef TILE_TRACE
            ph[0] += __builtin_readcyclecounter() - ts0;
#endif
            f32x4 vv[AHEAD + 1];   // the first V fragments fly under the softmax
#pragma unroll
            for (int r = 0; r < AHEAD; ++r) vv[r] = read_v(PAR, r);
#if defined(TILE_ABL) && (TILE_ABL & 1)   // timing experiments (EXTRA=-DTILE_ABL=.. tools/build_variant.sh): 1 = no softmax arithmetic
            l_run += 1.0f;
#else
            // row maximum: seven v_max3 and one v_max, in assembly (fmaxf() brings a canonicalising v_max x, x per operand chain)
            float tmax;
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(s[0]), "v"(s[1]), "v"(s[2]));
#pragma unroll
            for (int r = 3; r < 15; r += 2) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(tmax) : "v"(s[r]), "v"(s[r + 1]));
            asm("v_max_f32 %0, %0, %1" : "+v"(tmax) : "v"(s[15]));
            tmax = other_half_max(tmax);
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
            const f32x2 mm = {m_use, m_use};
            f32x2 ps = {0.f, 0.f};   // even and odd registers summed apart (v_pk_add_f32), as attn_kernel does
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 d;   // (s[r], s[r+1]) - (m, m) as ONE packed add (hipcc splits it into two)
                asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(f32x2{s[r], s[r + 1]}), "v"(mm));
                d[0] = __builtin_amdgcn_exp2f(d[0]);
                d[1] = __builtin_amdgcn_exp2f(d[1]);
                asm("v_pk_add_f32 %0, %0, %1" : "+v"(ps) : "v"(d));
                s[r] = d[0];
                s[r + 1] = d[1];
            }
            l_run += other_half_sum(ps[0] + ps[1]);
#endif
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r + AHEAD < 16) vv[(r + AHEAD) % (AHEAD + 1)] = read_v(PAR, r + AHEAD);
                const f32x4 v4 = vv[r % (AHEAD + 1)];
#pragma unroll
                for (int e = 0; e < DVB; ++e) o[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(v4[e], s[r], o[e], 0, 0, 0);
            }
        }
#ifdef TILE_TRACE
        const unsigned long long ts2 = __builtin_readcyclecounter();
#endif
        wait_vmcnt<0>();       // this wave's quarter of tile kn has landed ...
        settle(mnext);
#ifdef TILE_TRACE
        const unsigned long long ts3 = __builtin_readcyclecounter();
#endif
#if !(defined(TILE_ABL) && (TILE_ABL & 2))
        __syncthreads();       // ... and everybody's; everybody is done reading tile kt
#endif
#ifdef TILE_TRACE   // where a step's cycles go (QK^T | softmax + PV | wait for the DMA | barrier)
        const unsigned long long ts4 = __builtin_readcyclecounter();
        ph[1] += ts2 - ts0;
        ph[2] += ts3 - ts2;
        ph[3] += ts4 - ts3;
#endif
    };
    // a wave whose 32 queries lie past the last one: its share of the requests and the barriers, nothing else
    auto idle_step = [&](int kn, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_k(i, kn, buf);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_v(i, kn, buf);
        wait_vmcnt<0>();
#if !(defined(TILE_ABL) && (TILE_ABL & 2))
        __syncthreads();
#endif
    };

    if (n_act > 0) {
        int kt = tile_at(0);
        sgpr_guard(rsK);
        sgpr_guard(rsV);
        sgpr_guard(rsM);
        float mw0 = load_mask(kt), mw1 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_k(i, kt, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_v(i, kt, 0);
        // ---- this lane's Q fragments: Q[query][8c + 4hi .. +3], pre-scaled exactly as attn_kernel's LDS copy; requested BEHIND
        // the first tile so that the two memory round trips of a workgroup's start overlap ----
#pragma unroll
        for (int c = 0; c < DKC; ++c) {
            const int col = 8 * c + 4 * hi;
            const float4 v = bload4(rsQ, (qi < p.lq && col < p.dk) ? unsigned(qi * q_r + col) * 4u : OOB, 0);
            qf[c] = f32x4{v.x * p.scale_log2e, v.y * p.scale_log2e, v.z * p.scale_log2e, v.w * p.scale_log2e};
        }
        wait_vmcnt<0>();
        settle(mw0);
        __syncthreads();
        if (wave_active) {
            int idx = 0;
            for (; idx + 1 < n_act; idx += 2) {   // two tiles per trip: the buffer of a step is a compile-time constant
                const int k1 = tile_at(idx + 1);
                step(std::integral_constant<int, 0>{}, idx, kt, k1, mw0, mw1);
                kt = tile_at(idx + 2);
                step(std::integral_constant<int, 1>{}, idx + 1, k1, kt, mw1, mw0);
            }
            if (idx < n_act) step(std::integral_constant<int, 0>{}, idx, kt, nt, mw0, mw1);   // an odd last tile
        } else {
            for (int idx = 0; idx < n_act; ++idx) idle_step(tile_at(idx + 1), (idx + 1) & 1);
        }
    }

#ifdef TILE_TRACE
    if (p.trace && tid == 0) {
        unsigned long long* t = p.trace + size_t(blockIdx.x) * 8;
        t[0] = t_c0; t[1] = __builtin_readcyclecounter(); t[2] = t_w0; t[3] = wall_clock64(); t[4] = unsigned(n_act);
        t[5] = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_ID: bits 8-11 CU, 12 SH, 13-15 SE (placement)
        t[6] = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // XCC_ID
        t[7] = t_entry;
        unsigned long long* u = p.trace + size_t(gridDim.x) * (8 + 256) + size_t(blockIdx.x) * 4;
        u[0] = ph[0]; u[1] = ph[1] - ph[0]; u[2] = ph[2]; u[3] = ph[3];
    }
#endif
    const float inv_l = 1.0f / l_run;
#pragma unroll
    for (int e = 0; e < DVB; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[e][r] *= inv_l;

    // ---- store: lane (query, hi), register r, block e  <->  O[query][4 * key(r, hi) + e] ----
#if defined(TILE_ABL) && (TILE_ABL & 8)   // 8 = no output stores
    if (wave_active && qi < p.lq && item == 0x7fffffff) {
#else
    if (wave_active && qi < p.lq) {
#endif
        float* Orow = p.O + int64_t(b) * p.lay.o_b + int64_t(h) * p.lay.o_h + int64_t(qi) * p.lay.o_r;
        const bool vec = ((p.lay.o_b | p.lay.o_h | p.lay.o_r) & 3) == 0 && (reinterpret_cast<uintptr_t>(p.O) & 15u) == 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = DVB * ((r & 3) + 8 * (r >> 2) + 4 * hi);
            if (col >= p.dv) continue;
            if (vec) {
                *reinterpret_cast<float4*>(Orow + col) = make_float4(o[0][r], o[1][r], o[2][r], o[3][r]);
            } else {
#pragma unroll
                for (int e = 0; e < DVB; ++e) Orow[col + e] = o[e][r];
            }
        }
    }
}

#ifdef LAMP_TUNING
unsigned long long* g_tile_trace = nullptr;   // 8 words per workgroup of the NEXT launch(es)
#else
constexpr unsigned long long* g_tile_trace = nullptr;
#endif

template <int MK>
int launch_tile_mk(const AttnParams& p0, hipStream_t s) {
    AttnParams p = p0;
    p.trace = g_tile_trace;
    auto kern = attn_tile_kernel<MK>;
    const size_t lds = p.tiles ? LDS_BYTES_LIST : LDS_BYTES;
    static AttrOnce once;
    if (int e = once.set(reinterpret_cast<const void*>(kern), LDS_BYTES_LIST)) return e;
    const int64_t nwg = int64_t((p.lq + 127) / 128) * p.H * p.B;
    if (nwg > 0x7fffffffLL) return LAMP_E_DIMS;
    hipLaunchKernelGGL(kern, dim3(unsigned(nwg)), dim3(256), lds, s, p);
    return int(hipGetLastError());
}

}  // namespace

#ifdef LAMP_TUNING
extern "C" __attribute__((visibility("default"))) void lamp_debug_set_attn_tile_trace(unsigned long long* buf) { g_tile_trace = buf; }
#endif

// Shape-only rule (a sample's bits must not depend on its batch): the plan's mask kinds, no map output, head dimensions of the
// 128-wide instantiation, more than 256 queries (below: attention_small.hip) and at least 8 key tiles -- under that the
// prologue (Q fragments, first tile, tile list) is not amortised.
bool attn_tile_applies(const AttnParams& p) {
    if (p.P || p.lse || !p.V || !p.O) return false;
    if (p.mask_kind != LAMP_MASK_NONE && p.mask_kind != LAMP_MASK_BITS_U32) return false;
    const int dmax = p.dk > p.dv ? p.dk : p.dv;
    if (dmax <= 64 || dmax > 128) return false;
    if (p.lq <= 256 || p.lk < 256) return false;
    if (p.tiles && p.lk > 32 * MAX_LIST_TILES) return false;
    return true;
}

int launch_attn_tile(const AttnParams& p, hipStream_t s) {
    return p.mask_kind == LAMP_MASK_BITS_U32 ? launch_tile_mk<LAMP_MASK_BITS_U32>(p, s) : launch_tile_mk<LAMP_MASK_NONE>(p, s);
}

}  // namespace lamp
