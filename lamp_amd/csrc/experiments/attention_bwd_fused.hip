// NOT BUILT, NOT SHIPPED: the record of a round-6 experiment, kept as source because it was parity-green when
// wired into lamp_mha_bwd (28 / 28 of tests/test_gpu_training.py incl. the fp64-oracle gradients and a fused-vs-five-launch test with
// dropout) and NOT faster: profiles/r06_rejected_experiments.txt #8 has the measurements and the reason (per-wave latency, not
// operand traffic or integer overhead).  To revive: add it to lamp_amd/build.py SOURCES and call launch_attn_bwd_fused from
// lamp_mha_bwd where d_k = d_v in {64, 128} in place of the four lamp_gemm launches + softmax_bwd.
//
// Backward of softmax attention for training (reference: lamp/Modules.py:26-46 run backwards by autograd):
//     dPd = dO V^T,  dS = inv_t * P o (drop'(dPd) - rowsum(P o drop'(dPd))),  dQ = dS K,  dK = dS^T Q,  dV = Pd^T dO
// in TWO launches instead of four batched lamp_gemm launches + softmax_bwd (round 6):
//   attn_bwd_rows_kernel   one workgroup per (head, sample, 16 query rows): its four waves share the keys (wave w takes key
//                          blocks w, w + 4, ...): dPd^T block on the matrix pipe, dS in registers (stored once, [H*B, lq, lk],
//                          for the second launch), dQ accumulated over the wave's keys, the four partial dQ summed through LDS
//                          in a fixed order.  The row sum is dO_i . O_i (O = Pd V, what the forward wrote), so no pass over a
//                          whole score row is needed before dS.
//   attn_bwd_cols_kernel   one wave per (head, sample, 16 keys, product): dK = dS^T Q or dV = Pd^T dO over all query blocks.
// The dropped map Pd is never read: the counter-based mask is recomputed from the element index (drop1, lamp_kernels.h) --
// the same values lamp_dropout stored in the forward.
// fp32 on v_mfma_f32_16x16x4_f32, operands straight from global memory in MFMA operand order: both products of a wave are
// chained through REGISTERS -- the accumulator layout of the first (rows 4g + r, column l % 16) is the A-operand layout of
// the second under the k-permutation k <-> 4g + r, so dS never moves through LDS; the output columns are permuted
// (lane l % 16 owns columns 4 (l % 16) .. + 3 of a 64-column group) so that B operands are float4 loads and results float4 stores.
// Summation order differs from the five-launch route (LAMP_ATTN_BWD_UNFUSED=1 keeps that route: bit-identical to the
// per-launch Python route, tests/test_gpu_training.py); both are checked against fp64 autograd of the oracle.
#include "lamp_kernels.h"

namespace lamp {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

struct AttnBwdParams {
    const float *q, *k, *v;   // [B * lq | lk, hd]: head h at columns h * D
    const float *dO, *O;      // [B * lq, hd]: gradient of / the concatenated head outputs
    const float* P;           // [H * B, lq, lk] softmax probabilities (before dropout)
    float* dS;                // [H * B, lq, lk]
    float *dq, *dk, *dv;
    int B, H, lq, lk, hd;
    float inv_t;
    DropoutSpec drop;
};

__device__ __forceinline__ f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ f4 zero4() { return f4{0.f, 0.f, 0.f, 0.f}; }
#define LAMP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Loads are never predicated: out-of-range rows / keys read a clamped (valid) address and the value is replaced by 0 afterwards
// (a select, not a branch), and the operands of block n + 1 are requested before the MFMAs of block n (two register sets, the
// loop unrolled by two so that no set is ever copied).

// Address arithmetic is 32-bit offsets from wave-uniform bases, advanced by ADDS from block to block (the first version's
// 64-bit row * stride products were ~60 quarter-rate integer multiplies per block -- as many cycles as the block's MFMAs, and
// vector instructions are not hidden under fp32 MFMAs on this chip).

// operands of one key block of the rows kernel
template <int D>
struct RowsOps {
    f4 va[D / 16];        // V[key l16][16 s + 4 g ..]: A operand of dPd^T
    float pv[4];          // P[row][key 4 g + r]
};

template <int D, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_rows_kernel(AttnBwdParams p) {
    constexpr int NS = D / 16;   // k-steps of 16 head columns in the first product
    constexpr int NG = D / 64;   // 64-column groups of the second product's output
    __shared__ __attribute__((aligned(16))) float part[3][16][D];   // partial dQ of waves 1..3
    const int lane = threadIdx.x & 63, l16 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nib = (p.lq + 15) >> 4, nkb = (p.lk + 15) >> 4;
    const int ib = blockIdx.x % nib, hb = blockIdx.x / nib, h = hb / p.B, b = hb % p.B;
    const int i_n = ib * 16 + l16;   // the query row this lane carries as column index n of dPd^T / row index m of dS
    const bool i_ok = i_n < p.lq;
    const int i_c = i_ok ? i_n : p.lq - 1;
    const int64_t qrow = int64_t(b) * p.lq + i_c;
    const float* dOrow = p.dO + qrow * p.hd + h * D + 4 * g;
    const float* Orow = p.O + qrow * p.hd + h * D + 4 * g;
    f4 dOB[NS];
    float Di = 0.f;   // rowsum(P o dP) of row i_n = dO_i . O_i
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const f4 t = ld4(dOrow + 16 * s);
        dOB[s] = i_ok ? t : zero4();
        const f4 o = ld4(Orow + 16 * s);
        Di += dOB[s].x * o.x + dOB[s].y * o.y + dOB[s].z * o.z + dOB[s].w * o.w;
    }
    Di += __shfl_xor(Di, 16, 64);
    Di += __shfl_xor(Di, 32, 64);
    f4 acc_dq[NG * 4];
#pragma unroll
    for (int n = 0; n < NG * 4; ++n) acc_dq[n] = zero4();
    const int64_t prow = (int64_t(hb) * p.lq + i_c) * p.lk;   // element index of P[hb][row][0]: the dropout counter's base
    const float* Prow = p.P + prow;
    float* dSrow = p.dS + prow;
    const float* vbase = p.v + int64_t(b) * p.lk * p.hd + h * D;    // wave-uniform
    const float* kbase = p.k + int64_t(b) * p.lk * p.hd + h * D;
    const int lk1 = p.lk - 1;
    const unsigned row_max = unsigned(lk1) * unsigned(p.hd);        // offset of the last key's row
    const unsigned step = 64u * unsigned(p.hd);                     // four key blocks on
    // this wave's first block; every offset below moves by `step` (rows) / 64 (keys) per block and is clamped on use
    unsigned v_off = unsigned(wave * 16 + l16) * unsigned(p.hd) + 4 * g;      // V[key l16] row, this lane's 4 columns of a k-step
    unsigned k_off = unsigned(wave * 16 + 4 * g) * unsigned(p.hd) + 4 * l16;  // K[key 4 g] row, this lane's 4 columns of a group
    int j0 = wave * 16 + 4 * g;                                               // first of this lane's four keys

    auto load = [&](RowsOps<D>& o, unsigned voff, int j) {
        const float* vrow = vbase + min(voff, row_max + 4 * g);
#pragma unroll
        for (int s = 0; s < NS; ++s) o.va[s] = ld4(vrow + 16 * s);
#pragma unroll
        for (int r = 0; r < 4; ++r) o.pv[r] = Prow[min(j + r, lk1)];
    };
    auto compute = [&](const RowsOps<D>& o, RowsOps<D>& nxt) {
        // the second product's B operand K[key 4 g + r][64 n + 4 l16 ..]: requested now, needed after the first product's MFMAs
        f4 kq[4][NG];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* krow = kbase + min(k_off + unsigned(r) * unsigned(p.hd), row_max + 4 * l16);
#pragma unroll
            for (int n = 0; n < NG; ++n) kq[r][n] = ld4(krow + 64 * n);
        }
        // (after the K rows: loads return in order, and the second product must not wait for the prefetch.  Past the last block the
        //  clamps make it a re-read of the last key: a load under `if` would make hipcc's wait-count pass merge "issued" with "not
        //  issued" at the join and wait for vmcnt(0) in front of the MFMAs, which serialises load and compute)
        load(nxt, v_off + step, j0 + 64);
        // (a key past lk reads key lk - 1's row: its dPd column is finite garbage that P = 0 below turns into dS = 0, and 0 times
        //  the clamped K row adds nothing -- non-finite rows propagate exactly as they do through the real key lk - 1)
        f4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const f4 va = o.va[s];
            acc0 = LAMP_MFMA(va.x, dOB[s].x, acc0);
            acc1 = LAMP_MFMA(va.y, dOB[s].y, acc1);
            acc0 = LAMP_MFMA(va.z, dOB[s].z, acc0);
            acc1 = LAMP_MFMA(va.w, dOB[s].w, acc1);
        }
        const f4 acc = acc0 + acc1;   // acc[r] = dPd[i_n][j0 + r]
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = i_ok && j0 + r < p.lk;
            const float pv = ok ? o.pv[r] : 0.f;
            const float pd = DROP ? drop1(pv, prow + (j0 + r), p.drop) : pv;
            ds[r] = (pd * acc[r] - pv * Di) * p.inv_t;
            if (ok) dSrow[j0 + r] = ds[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int n = 0; n < NG; ++n) {
                acc_dq[n * 4 + 0] = LAMP_MFMA(ds[r], kq[r][n].x, acc_dq[n * 4 + 0]);
                acc_dq[n * 4 + 1] = LAMP_MFMA(ds[r], kq[r][n].y, acc_dq[n * 4 + 1]);
                acc_dq[n * 4 + 2] = LAMP_MFMA(ds[r], kq[r][n].z, acc_dq[n * 4 + 2]);
                acc_dq[n * 4 + 3] = LAMP_MFMA(ds[r], kq[r][n].w, acc_dq[n * 4 + 3]);
            }
        }
        v_off += step;
        k_off += step;
        j0 += 64;
    };
    RowsOps<D> A, Bf;
    load(A, v_off, j0);
    for (int kb = wave; kb < nkb; kb += 8) {
        compute(A, Bf);
        if (kb + 4 >= nkb) break;
        compute(Bf, A);
    }
    // acc_dq[n * 4 + t][r] = dQ[ib * 16 + 4g + r][64 n + 4 l16 + t]: waves 1..3 hand theirs to wave 0's order of summation
    if (wave > 0) {
#pragma unroll
        for (int n = 0; n < NG; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<f4*>(&part[wave - 1][4 * g + r][64 * n + 4 * l16]) =
                    f4{acc_dq[n * 4 + 0][r], acc_dq[n * 4 + 1][r], acc_dq[n * 4 + 2][r], acc_dq[n * 4 + 3][r]};
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int n = 0; n < NG; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ib * 16 + 4 * g + r;
                f4 o = f4{acc_dq[n * 4 + 0][r], acc_dq[n * 4 + 1][r], acc_dq[n * 4 + 2][r], acc_dq[n * 4 + 3][r]};
#pragma unroll
                for (int w = 0; w < 3; ++w) o += *reinterpret_cast<const f4*>(&part[w][4 * g + r][64 * n + 4 * l16]);
                if (i < p.lq) *reinterpret_cast<f4*>(p.dq + (int64_t(b) * p.lq + i) * p.hd + h * D + 64 * n + 4 * l16) = o;
            }
    }
}

// operands of one query block of the cols kernel
template <int D, int KB>
struct ColsOps {
    float a[KB][4];      // dS (dK waves) or P (dV waves) [row 4 g + r][key 16 u + l16]
    f4 bv[4][D / 64];    // Q (dK waves) or dO (dV waves) [row 4 g + r][64 n + 4 l16 ..]
};

// One wave per (head, sample, 16 KB keys, product): even tasks dK = dS^T Q, odd tasks dV = Pd^T dO.  One operand stream per
// wave; every B operand register feeds KB MFMAs (the kernel is bound by operand traffic from L2, not by the matrix pipe:
// KB = 1 measured 33 us for the reuters enc-dec attention whose MFMAs need 13).
template <int D, bool DROP, int KB>
__global__ __launch_bounds__(256, 2) void attn_bwd_cols_kernel(AttnBwdParams p) {
    constexpr int NG = D / 64;
    const int lane = threadIdx.x & 63, l16 = lane & 15, g = lane >> 4;
    const int nib = (p.lq + 15) >> 4, nkb = (p.lk + 16 * KB - 1) / (16 * KB);
    const int task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (task >= 2 * p.H * p.B * nkb) return;
    const bool is_dv = task & 1;
    const int kb = (task >> 1) % nkb, hb = (task >> 1) / nkb, h = hb / p.B, b = hb % p.B;
    const int j_n = kb * 16 * KB + l16;   // the first key this lane carries as column index of the [i][j] blocks of dS / Pd
    f4 acc[KB][NG * 4];
#pragma unroll
    for (int u = 0; u < KB; ++u)
#pragma unroll
        for (int n = 0; n < NG * 4; ++n) acc[u][n] = zero4();
    // wave-uniform bases + 32-bit offsets advanced by adds
    const float* bbase = (is_dv ? p.dO : p.q) + int64_t(b) * p.lq * p.hd + h * D;
    const float* abase = (is_dv ? p.P : p.dS) + int64_t(hb) * p.lq * p.lk;
    int64_t e_row = (int64_t(hb) * p.lq + 4 * g) * p.lk + j_n;    // dropout counter of [hb][4 g][j_n], advanced with the blocks
    const int lq1 = p.lq - 1, lk1 = p.lk - 1;
    const unsigned a_last = unsigned(lq1) * unsigned(p.lk), b_max = unsigned(lq1) * unsigned(p.hd) + 4 * l16;
    const unsigned a_step = 16u * unsigned(p.lk), b_step = 16u * unsigned(p.hd);
    unsigned a_row = unsigned(4 * g) * unsigned(p.lk), b_off = unsigned(4 * g) * unsigned(p.hd) + 4 * l16;
    int i0 = 4 * g;

    auto load = [&](ColsOps<D, KB>& o, unsigned arow, unsigned boff) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned ar = min(arow + unsigned(r) * unsigned(p.lk), a_last);
#pragma unroll
            for (int u = 0; u < KB; ++u) o.a[u][r] = abase[ar + min(j_n + 16 * u, lk1)];
            const float* brow = bbase + min(boff + unsigned(r) * unsigned(p.hd), b_max);
#pragma unroll
            for (int n = 0; n < NG; ++n) o.bv[r][n] = ld4(brow + 64 * n);
        }
    };
    auto compute = [&](const ColsOps<D, KB>& o, ColsOps<D, KB>& nxt) {
        load(nxt, a_row + a_step, b_off + b_step);   // (unconditional: see the rows kernel)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a[KB];
#pragma unroll
            for (int u = 0; u < KB; ++u) {
                a[u] = i0 + r < p.lq && j_n + 16 * u < p.lk ? o.a[u][r] : 0.f;   // (a row past lq is row lq - 1 again, times 0)
                if (DROP && is_dv) a[u] = drop1(a[u], e_row + (r * p.lk + 16 * u), p.drop);
            }
#pragma unroll
            for (int n = 0; n < NG; ++n)
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    acc[u][n * 4 + 0] = LAMP_MFMA(a[u], o.bv[r][n].x, acc[u][n * 4 + 0]);
                    acc[u][n * 4 + 1] = LAMP_MFMA(a[u], o.bv[r][n].y, acc[u][n * 4 + 1]);
                    acc[u][n * 4 + 2] = LAMP_MFMA(a[u], o.bv[r][n].z, acc[u][n * 4 + 2]);
                    acc[u][n * 4 + 3] = LAMP_MFMA(a[u], o.bv[r][n].w, acc[u][n * 4 + 3]);
                }
        }
        a_row += a_step;
        b_off += b_step;
        e_row += a_step;
        i0 += 16;
    };
    ColsOps<D, KB> A, Bf;
    load(A, a_row, b_off);
    for (int ib = 0; ib < nib; ib += 2) {
        compute(A, Bf);
        if (ib + 1 >= nib) break;
        compute(Bf, A);
    }
    float* out = is_dv ? p.dv : p.dk;
#pragma unroll
    for (int u = 0; u < KB; ++u)
#pragma unroll
        for (int n = 0; n < NG; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = (kb * KB + u) * 16 + 4 * g + r;
                if (j >= p.lk) continue;
                *reinterpret_cast<f4*>(out + (int64_t(b) * p.lk + j) * p.hd + h * D + 64 * n + 4 * l16) =
                    f4{acc[u][n * 4 + 0][r], acc[u][n * 4 + 1][r], acc[u][n * 4 + 2][r], acc[u][n * 4 + 3][r]};
            }
}

constexpr int COLS_KB = 2;

template <int D, bool DROP>
int launch_both(const AttnBwdParams& p, hipStream_t s) {
    const int64_t nib = (p.lq + 15) / 16, nkb = (p.lk + 16 * COLS_KB - 1) / (16 * COLS_KB);
    const int64_t rows_wg = int64_t(p.H) * p.B * nib, cols_wg = (2 * int64_t(p.H) * p.B * nkb + 3) / 4;
    if (rows_wg > 0x7fffffffLL || cols_wg > 0x7fffffffLL) return LAMP_E_DIMS;
    hipLaunchKernelGGL((attn_bwd_rows_kernel<D, DROP>), dim3(unsigned(rows_wg)), dim3(256), 0, s, p);
    hipLaunchKernelGGL((attn_bwd_cols_kernel<D, DROP, COLS_KB>), dim3(unsigned(cols_wg)), dim3(256), 0, s, p);
    return int(hipGetLastError());
}

}  // namespace

bool attn_bwd_fused_applies(int dk, int dv, int hdk, int hdv) {
    return dk == dv && hdk == hdv && (dk == 64 || dk == 128);
}

int launch_attn_bwd_fused(const float* q, const float* k, const float* v, const float* dO, const float* O, const float* P,
                          float* dS, float* dq, float* dk, float* dv, int B, int H, int lq, int lk, int D, float inv_t,
                          const DropoutSpec* drop, hipStream_t s) {
    if (!q || !k || !v || !dO || !O || !P || !dS || !dq || !dk || !dv) return LAMP_E_NULL;
    if (B <= 0 || H <= 0 || lq <= 0 || lk <= 0) return LAMP_E_DIMS;
    if (D != 64 && D != 128) return LAMP_E_UNSUPPORTED;
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(dO) || !aligned16(O) || !aligned16(dq) || !aligned16(dk) ||
        !aligned16(dv))
        return LAMP_E_ALIGN;
    const bool dr = drop && drop->threshold > 0;
    const AttnBwdParams p{q, k, v, dO, O, P, dS, dq, dk, dv, B, H, lq, lk, H * D, inv_t,
                          dr ? *drop : DropoutSpec{0u, 1.f, 0u}};
    if (D == 128) return dr ? launch_both<128, true>(p, s) : launch_both<128, false>(p, s);
    return dr ? launch_both<64, true>(p, s) : launch_both<64, false>(p, s);
}

}  // namespace lamp
