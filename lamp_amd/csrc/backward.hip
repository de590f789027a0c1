// Bandwidth-bound pieces of the backward pass: LayerNorm backward, column sums (bias and
// label-table gradients), dropout (counter-based, the mask is recomputed -- never stored), softmax backward, label
// read-out backward, embedding scatter-add.  The matrix products of the backward pass are lamp_gemm (gemm_gen.hip).
// Reductions over rows are two-stage with a fixed summation order: results do not depend on scheduling (the only
// atomics are in the embedding scatter-add, like torch's own embedding backward).
#include "lamp_kernels.h"

namespace lamp {

namespace {

__device__ __forceinline__ float wsum(float v) { return wave64_sum(v); }

constexpr int LNB_MAX_WG = 256;

// dz = d/dz LayerNorm(z) . dy, z = dropout(x) + res;  dz_drop = dropout(dz) (the gradient of x; DROP only);
// partial[wg] = [sum_rows dy * zhat | sum_rows dy | sum_rows (gradient of x)]  over this WG's rows (third part: NP == 3)
template <int NV, bool DROP, int NP>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                            int64_t r_mod, int64_t M, int d,
                                                            const float* __restrict__ g, float eps,
                                                            const float* __restrict__ dy, float* __restrict__ dz,
                                                            float* __restrict__ dz_drop, float* __restrict__ partial,
                                                            DropoutSpec drop) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [NP][d]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = d / 4;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4 gg[NV], ag[NV], ab[NV], ax[NP == 3 ? NV : 1];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        gg[i] = c < nv ? g4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (NP == 3) ax[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float inv_d = 1.0f / float(d);
    for (int64_t row = int64_t(blockIdx.x) * 4 + wave; row < M; row += int64_t(gridDim.x) * 4) {
        const float4* xr = reinterpret_cast<const float4*>(x + row * d);
        const float4* rr = res ? reinterpret_cast<const float4*>(res + (r_mod > 0 ? row % r_mod : row) * d) : nullptr;
        const float4* dr = reinterpret_cast<const float4*>(dy + row * d);
        float4 v[NV], t[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            v[i] = c < nv ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (DROP) v[i] = drop4(v[i], row * d + 4 * c, drop);
            if (rr && c < nv) {
                const float4 w = rr[c];
                v[i].x += w.x; v[i].y += w.y; v[i].z += w.z; v[i].w += w.w;
            }
            t[i] = c < nv ? dr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mean = wsum(s) * inv_d;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
            }
        }
        const float rstd = 1.0f / sqrtf(wsum(ss) * inv_d + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;  // zhat
            ag[i].x += t[i].x * v[i].x; ag[i].y += t[i].y * v[i].y; ag[i].z += t[i].z * v[i].z; ag[i].w += t[i].w * v[i].w;
            ab[i].x += t[i].x; ab[i].y += t[i].y; ab[i].z += t[i].z; ab[i].w += t[i].w;
            t[i].x *= gg[i].x; t[i].y *= gg[i].y; t[i].z *= gg[i].z; t[i].w *= gg[i].w;  // dy * gamma
            s1 += (t[i].x + t[i].y) + (t[i].z + t[i].w);
            s2 += (t[i].x * v[i].x + t[i].y * v[i].y) + (t[i].z * v[i].z + t[i].w * v[i].w);
        }
        s1 = wsum(s1) * inv_d;
        s2 = wsum(s2) * inv_d;
        float4* zr = reinterpret_cast<float4*>(dz + row * d);
        float4* zd = DROP ? reinterpret_cast<float4*>(dz_drop + row * d) : nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                float4 o = make_float4(rstd * (t[i].x - s1 - v[i].x * s2), rstd * (t[i].y - s1 - v[i].y * s2),
                                       rstd * (t[i].z - s1 - v[i].z * s2), rstd * (t[i].w - s1 - v[i].w * s2));
                zr[c] = o;
                if constexpr (DROP) {
                    o = drop4(o, row * d + 4 * c, drop);
                    zd[c] = o;
                }
                if constexpr (NP == 3) {
                    ax[i].x += o.x; ax[i].y += o.y; ax[i].z += o.z; ax[i].w += o.w;
                }
            }
        }
    }
    // combine the four waves' column sums in a fixed order (wave 0, 1, 2, 3) through one [NP][d] LDS block
    float4* r4 = reinterpret_cast<float4*>(red);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane + i * 64;
                if (c < nv) {
                    if (w == 0) {
                        r4[0 * nv + c] = ag[i];
                        r4[1 * nv + c] = ab[i];
                        if constexpr (NP == 3) r4[2 * nv + c] = ax[i];
                    } else {
                        float4 a = r4[0 * nv + c], b = r4[1 * nv + c];
                        a.x += ag[i].x; a.y += ag[i].y; a.z += ag[i].z; a.w += ag[i].w;
                        b.x += ab[i].x; b.y += ab[i].y; b.z += ab[i].z; b.w += ab[i].w;
                        r4[0 * nv + c] = a;
                        r4[1 * nv + c] = b;
                        if constexpr (NP == 3) {
                            float4 x3 = r4[2 * nv + c];
                            x3.x += ax[i].x; x3.y += ax[i].y; x3.z += ax[i].z; x3.w += ax[i].w;
                            r4[2 * nv + c] = x3;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    float* out = partial + int64_t(blockIdx.x) * NP * d;
    for (int e = threadIdx.x; e < NP * d; e += 256) out[e] = red[e];
}

// partial[gy][col] = sum over this row chunk of x[row][col]; eight independent chains per thread keep loads in flight
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int64_t M, int64_t N, int64_t ldx,
                                                             int64_t rows_per_chunk, float* __restrict__ partial) {
    const int64_t col = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (col >= N) return;
    const int64_t r0 = int64_t(blockIdx.y) * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < M ? r0 + rows_per_chunk : M;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int64_t r = r0;
    for (; r + 8 <= r1; r += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += x[(r + u) * ldx + col];
    }
    for (int u = 0; r < r1; ++r, ++u) s[u] += x[r * ldx + col];
    partial[int64_t(blockIdx.y) * N + col] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

// out[c] = sum_p partial[p][c] for c in [0, n_total); segment c / n_seg goes to out0 / out1 / out2.  A workgroup
// owns 32 columns; its 8 thread slices each add every 8th partial, and the slices are combined in a fixed order.
__device__ __forceinline__ void reduce_partials_block(const float* __restrict__ partial, int P, int64_t n_total,
                                                      int64_t n_seg, float* __restrict__ out0, float* __restrict__ out1,
                                                      float* __restrict__ out2, int64_t block) {
    __shared__ float red[8][32];
    const int cx = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int64_t c = block * 32 + cx;
    float s0 = 0.f, s1 = 0.f;
    if (c < n_total) {
        int p = slice;
        for (; p + 8 < P; p += 16) {
            s0 += partial[int64_t(p) * n_total + c];
            s1 += partial[int64_t(p + 8) * n_total + c];
        }
        if (p < P) s0 += partial[int64_t(p) * n_total + c];
    }
    red[slice][cx] = s0 + s1;
    __syncthreads();
    if (slice == 0 && c < n_total) {
        const float s = ((red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx])) +
                        ((red[4][cx] + red[5][cx]) + (red[6][cx] + red[7][cx]));
        const int64_t seg = c / n_seg;
        float* o = seg == 0 ? out0 : (seg == 1 ? out1 : out2);
        o[c - seg * n_seg] = s;
    }
}
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int P, int64_t n_total,
                                                              int64_t n_seg, float* __restrict__ out0,
                                                              float* __restrict__ out1, float* __restrict__ out2) {
    reduce_partials_block(partial, P, n_total, n_seg, out0, out1, out2, blockIdx.x);
}
// Up to REDUCE_GROUP_MAX such reductions in one launch (lamp_reduce_partials_grouped): the second stages of a whole backward
// pass's LayerNorm-parameter and bias gradients.  Same summation order per output as the single launch.
constexpr int REDUCE_GROUP_MAX = 32;
struct ReduceGroup {
    lamp_reduce_job job[REDUCE_GROUP_MAX];
    int block_begin[REDUCE_GROUP_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void reduce_group_kernel(ReduceGroup g) {
    int idx = 0;
    for (int i = 1; i < g.n; ++i)
        if (int(blockIdx.x) >= g.block_begin[i]) idx = i;
    const lamp_reduce_job& j = g.job[idx];
    reduce_partials_block(j.partial, j.n_partials, j.n_total, j.n_seg, j.out[0], j.out[1], j.out[2],
                          int(blockIdx.x) - g.block_begin[idx]);
}

// Counter-based dropout (mix32 / DropoutSpec in lamp_kernels.h).  The same call with the same seed applied to the
// gradient is the backward pass.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, int64_t n, unsigned threshold,
                                                      float scale, unsigned seed, float* __restrict__ y) {
    for (int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x; e < n; e += int64_t(gridDim.x) * 256) {
        const unsigned h = mix32(unsigned(e), unsigned(uint64_t(e) >> 32), seed);
        y[e] = h >= threshold ? x[e] * scale : 0.f;
    }
}

// dS = scale * P * (dP - sum_k P * dP) per row (softmax backward; P = 0 on blocked entries keeps them at 0)
// DROP: dP is the gradient of the DROPPED probabilities (lamp/SubLayers.py:40); its dropout backward -- lamp_dropout's mask for
// element index row * lk + c -- is applied on load (the same values a separate lamp_dropout launch would have stored).
template <bool DROP>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ P, const float* __restrict__ dP,
                                                          int64_t rows, int lk, float scale, float* __restrict__ dS,
                                                          DropoutSpec ds) {
    const int lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = P + row * lk;
    const float* g = dP + row * lk;
    auto grad = [&](int c) { return DROP ? drop1(g[c], row * lk + c, ds) : g[c]; };
    float s = 0.f;
    for (int c = lane; c < lk; c += 64) s += p[c] * grad(c);
    s = wsum(s);
    float* o = dS + row * lk;
    for (int c = lane; c < lk; c += 64) o[c] = scale * p[c] * (grad(c) - s);
}

// Read-out backward (lamp/Models.py:124-126): dy[b,i,:] = dl[b,i] * w[i,:];  dw[i,:] = sum_b dl[b,i] * y[b,i,:]
__global__ __launch_bounds__(256) void diag_bwd_kernel(const float* __restrict__ y, const float* __restrict__ w,
                                                       const float* __restrict__ dl, int B, int L, int d,
                                                       float* __restrict__ dy, float* __restrict__ dw) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= L) return;
    const float4* wr = reinterpret_cast<const float4*>(w + int64_t(i) * d);
    for (int c = lane; c < d / 4; c += 64) {
        const float4 ww = wr[c];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b = 0; b < B; ++b) {
            const int64_t row = int64_t(b) * L + i;
            const float g = dl[row];
            const float4 yy = reinterpret_cast<const float4*>(y + row * d)[c];
            acc.x += g * yy.x; acc.y += g * yy.y; acc.z += g * yy.z; acc.w += g * yy.w;
            reinterpret_cast<float4*>(dy + row * d)[c] = make_float4(g * ww.x, g * ww.y, g * ww.z, g * ww.w);
        }
        reinterpret_cast<float4*>(dw + int64_t(i) * d)[c] = acc;
    }
}

// d_emb[seq[t], :] += dout[t, :]   (skipping pad_idx, whose row nn.Embedding(padding_idx=...) never updates)
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ seq, int64_t n_tok,
                                                        const float* __restrict__ dout, int d, int n_vocab,
                                                        int64_t pad_idx, float* __restrict__ d_emb) {
    const int lane = threadIdx.x & 63;
    const int64_t t = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (t >= n_tok) return;
    const int64_t tok = seq[t];
    if (tok == pad_idx || tok < 0 || tok >= n_vocab) return;
    const float* src = dout + t * d;
    float* dst = d_emb + tok * d;
    for (int c = lane; c < d; c += 64) atomicAdd(dst + c, src[c]);
}

inline int lnb_grid(int64_t M) {
    const int64_t n = (M + 3) / 4;
    return int(n < LNB_MAX_WG ? n : LNB_MAX_WG);
}
inline int colsum_chunks(int64_t M) {
    int64_t c = (M + 63) / 64;
    return int(c < 1 ? 1 : (c > 512 ? 512 : c));
}

}  // namespace

size_t layernorm_bwd_workspace_bytes(int64_t M, int d) { return size_t(lnb_grid(M)) * 3 * d * sizeof(float); }

int launch_layernorm_bwd(const float* x, const float* res, int64_t r_mod, int64_t M, int d, const float* g, float eps,
                         const DropoutSpec* drop, const float* dy, float* dz, float* dz_drop, float* dgamma, float* dbeta,
                         float* dbias, void* ws, size_t ws_bytes, hipStream_t s, lamp_reduce_job* job_out) {
    if (M <= 0 || d <= 0) return LAMP_E_DIMS;
    if ((d & 3) || d > 4096) return LAMP_E_UNSUPPORTED;
    const bool dr = drop && drop->threshold > 0;
    if (!x || !g || !dy || !dz || !dgamma || !dbeta || !ws || (dr && !dz_drop)) return LAMP_E_NULL;
    if (!aligned16(x) || !aligned16(dy) || !aligned16(dz) || !aligned16(g) || (res && !aligned16(res)) || !aligned16(ws) ||
        (dr && !aligned16(dz_drop)))
        return LAMP_E_ALIGN;
    if (ws_bytes < layernorm_bwd_workspace_bytes(M, d)) return LAMP_E_WORKSPACE;
    const int grid = lnb_grid(M);
    float* partial = static_cast<float*>(ws);
    const int np = dbias ? 3 : 2;
    const size_t lds = size_t(np) * d * sizeof(float);
    const int nv = (d / 4 + 63) / 64;
    const DropoutSpec ds = dr ? *drop : DropoutSpec{0u, 1.f, 0u};
    ProfScope prof(LAMP_K_LAYERNORM, 0.0, 12.0 * double(M) * d, s);
#define LAMP_LNB(NV_, DROP_, NP_)                                                                                       \
    hipLaunchKernelGGL((layernorm_bwd_kernel<NV_, DROP_, NP_>), dim3(grid), dim3(256), lds, s, x, res, r_mod, M, d, g, eps, \
                       dy, dz, dz_drop, partial, ds)
#define LAMP_LNB_NV(NV_)                    \
    do {                                    \
        if (dr && np == 3)                  \
            LAMP_LNB(NV_, true, 3);         \
        else if (dr)                        \
            LAMP_LNB(NV_, true, 2);         \
        else if (np == 3)                   \
            LAMP_LNB(NV_, false, 3);        \
        else                                \
            LAMP_LNB(NV_, false, 2);        \
    } while (0)
    if (nv <= 1)
        LAMP_LNB_NV(1);
    else if (nv <= 2)
        LAMP_LNB_NV(2);
    else if (nv <= 4)
        LAMP_LNB_NV(4);
    else if (nv <= 8)
        LAMP_LNB_NV(8);
    else
        LAMP_LNB_NV(16);
#undef LAMP_LNB_NV
#undef LAMP_LNB
    if (int e = int(hipGetLastError())) return e;
    if (job_out) {   // the caller reduces later (lamp_reduce_partials_grouped); ws stays live until then
        *job_out = lamp_reduce_job{partial, int64_t(np) * d, int64_t(d), {dgamma, dbeta, dbias}, grid, 0};
        return 0;
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((np * d + 31) / 32), dim3(256), 0, s, partial, grid, int64_t(np) * d,
                       int64_t(d), dgamma, dbeta, dbias);
    return int(hipGetLastError());
}

size_t colsum_workspace_bytes(int64_t M, int64_t N) { return size_t(colsum_chunks(M)) * N * sizeof(float); }

int launch_colsum(const float* x, int64_t M, int64_t N, int64_t ldx, float* out, void* ws, size_t ws_bytes, hipStream_t s,
                  lamp_reduce_job* job_out) {
    if (M <= 0 || N <= 0 || ldx < N) return LAMP_E_DIMS;
    if (!x || !out || !ws) return LAMP_E_NULL;
    if (ws_bytes < colsum_workspace_bytes(M, N)) return LAMP_E_WORKSPACE;
    const int chunks = colsum_chunks(M);
    const int64_t rows_per_chunk = (M + chunks - 1) / chunks;
    const int64_t gx = (N + 255) / 256;
    if (gx > 0x7fffffffLL) return LAMP_E_DIMS;
    float* partial = static_cast<float*>(ws);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)gx, chunks), dim3(256), 0, s, x, M, N, ldx, rows_per_chunk, partial);
    if (int e = int(hipGetLastError())) return e;
    const int64_t gr = (N + 31) / 32;
    if (gr > 0x7fffffffLL) return LAMP_E_DIMS;
    if (job_out) {
        *job_out = lamp_reduce_job{partial, N, N, {out, out, out}, chunks, 0};
        return 0;
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)gr), dim3(256), 0, s, partial, chunks, N, N, out, out, out);
    return int(hipGetLastError());
}

int launch_reduce_group(const lamp_reduce_job* jobs, int n, hipStream_t s) {
    if (n < 0) return LAMP_E_DIMS;
    if (n && !jobs) return LAMP_E_NULL;
    for (int first = 0; first < n; first += REDUCE_GROUP_MAX) {
        const int cnt = n - first < REDUCE_GROUP_MAX ? n - first : REDUCE_GROUP_MAX;
        ReduceGroup g;
        g.n = cnt;
        g.block_begin[0] = 0;
        for (int i = 0; i < cnt; ++i) {
            const lamp_reduce_job& j = jobs[first + i];
            if (j.n_partials <= 0 || j.n_total <= 0 || j.n_seg <= 0 || j.n_total > 3 * j.n_seg || j.n_total > (1 << 26))
                return LAMP_E_DIMS;
            if (!j.partial || !j.out[0] || (j.n_total > j.n_seg && !j.out[1]) || (j.n_total > 2 * j.n_seg && !j.out[2]))
                return LAMP_E_NULL;
            g.job[i] = j;
            g.block_begin[i + 1] = g.block_begin[i] + int((j.n_total + 31) / 32);
        }
        for (int i = cnt; i < REDUCE_GROUP_MAX; ++i) g.block_begin[i + 1] = g.block_begin[cnt];
        hipLaunchKernelGGL(reduce_group_kernel, dim3(unsigned(g.block_begin[cnt])), dim3(256), 0, s, g);
        if (int e = int(hipGetLastError())) return e;
    }
    return 0;
}

int launch_dropout(const float* x, int64_t n, float p, uint32_t seed, float* y, hipStream_t s) {
    if (n <= 0) return LAMP_E_DIMS;
    if (!(p >= 0.f) || !(p < 1.f)) return LAMP_E_UNSUPPORTED;
    if (!x || !y) return LAMP_E_NULL;
    const DropoutSpec ds = make_dropout(p, seed);
    const int64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)(g < 8192 ? g : 8192)), dim3(256), 0, s, x, n, ds.threshold, ds.scale,
                       ds.seed, y);
    return int(hipGetLastError());
}

int launch_softmax_bwd(const float* P, const float* dP, int64_t rows, int lk, float scale, float* dS, hipStream_t s,
                       const DropoutSpec* drop) {
    if (rows <= 0 || lk <= 0) return LAMP_E_DIMS;
    if (!P || !dP || !dS) return LAMP_E_NULL;
    const int64_t g = (rows + 3) / 4;
    if (g > 0x7fffffffLL) return LAMP_E_DIMS;
    if (drop && drop->threshold > 0)
        hipLaunchKernelGGL(softmax_bwd_kernel<true>, dim3((unsigned)g), dim3(256), 0, s, P, dP, rows, lk, scale, dS, *drop);
    else
        hipLaunchKernelGGL(softmax_bwd_kernel<false>, dim3((unsigned)g), dim3(256), 0, s, P, dP, rows, lk, scale, dS,
                           DropoutSpec{0u, 1.f, 0u});
    return int(hipGetLastError());
}

int launch_diag_bwd(const float* y, const float* w, const float* dl, int B, int L, int d, float* dy, float* dw,
                    hipStream_t s) {
    if (B <= 0 || L <= 0 || d <= 0) return LAMP_E_DIMS;
    if (d & 3) return LAMP_E_UNSUPPORTED;
    if (!y || !w || !dl || !dy || !dw) return LAMP_E_NULL;
    if (!aligned16(y) || !aligned16(w) || !aligned16(dy) || !aligned16(dw)) return LAMP_E_ALIGN;
    hipLaunchKernelGGL(diag_bwd_kernel, dim3((L + 3) / 4), dim3(256), 0, s, y, w, dl, B, L, d, dy, dw);
    return int(hipGetLastError());
}

int launch_embed_bwd(const int64_t* seq, int64_t n_tok, const float* dout, int d, int n_vocab, int64_t pad_idx,
                     float* d_emb, hipStream_t s) {
    if (n_tok <= 0 || d <= 0 || n_vocab <= 0) return LAMP_E_DIMS;
    if (!seq || !dout || !d_emb) return LAMP_E_NULL;
    const int64_t g = (n_tok + 3) / 4;
    if (g > 0x7fffffffLL) return LAMP_E_DIMS;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)g), dim3(256), 0, s, seq, n_tok, dout, d, n_vocab, pad_idx, d_emb);
    return int(hipGetLastError());
}

}  // namespace lamp
