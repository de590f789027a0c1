// PROBE (not part of the product, not built by lamp_amd/build.py): how fast is an fp32-in / fp32-out GEMM  C = A . W^T  on gfx950 when
// every fp32 product is emulated by six bf16 MFMAs -- each operand split into three bf16 pieces (hi + mid + lo = the 24-bit
// mantissa exactly), the three smallest cross terms dropped, fp32 accumulation -- and how large is its error?  DESIGN.md
// section 10 raises this as a question for a ruling (it is NOT the fp32 fmaf chain the product computes); this file supplies
// the measured side.  W is split once (weights), A is split while it is staged into LDS (activations).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/gemm_bf16x6.hip -o /tmp/gemm_bf16x6 -ldl && /tmp/gemm_bf16x6 [lamp_amd/liblamp_hip.so]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r = x - (float)h;
    m = (__bf16)r;
    l = (__bf16)(r - (float)m);
}

// W [N, K] fp32 -> three bf16 planes [N, K]
__global__ void split_planes(const float* __restrict__ w, __bf16* __restrict__ h, __bf16* __restrict__ m, __bf16* __restrict__ l, size_t n) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) split3(w[i], h[i], m[i], l[i]);
}

constexpr int BM = 128, BN = 128, BK = 32, LS = BK + 8;   // LDS row stride in bf16 (80 bytes: conflict-free b128 fragment reads)

// PRODUCTS: 6 = hi.hi + hi.mid + mid.hi + mid.mid + hi.lo + lo.hi;  3 = hi.hi + hi.mid + mid.hi;  9 = all
template <int PRODUCTS>
__global__ __launch_bounds__(256) void gemm_split_kernel(const float* __restrict__ A, const __bf16* __restrict__ Wh,
                                                         const __bf16* __restrict__ Wm, const __bf16* __restrict__ Wl,
                                                         float* __restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) __bf16 As[3][BM][LS];
    __shared__ __attribute__((aligned(16))) __bf16 Ws[3][BN][LS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, kg = lane >> 5;
    const int tiles_n = N / BN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const float* Ab = A + size_t(tm) * BM * K;
    const __bf16* Wb[3] = {Wh + size_t(tn) * BN * K, Wm + size_t(tn) * BN * K, Wl + size_t(tn) * BN * K};

    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[4];     // A tile 128 x 32 floats = 1024 float4, 4 per thread
    bf16x8 rw[3][2];  // W planes 128 x 32 bf16 = 512 chunks of 8, 2 per thread and plane
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, c4 = idx & 7;
            ra[i] = *reinterpret_cast<const float4*>(Ab + size_t(row) * K + k0 + c4 * 4);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + i * 256, row = idx >> 2, c8 = idx & 3;
                rw[p][i] = *reinterpret_cast<const bf16x8*>(Wb[p] + size_t(row) * K + k0 + c8 * 8);
            }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, c4 = idx & 7;
            __bf16 hh[4], mm[4], ll[4];
            split3(ra[i].x, hh[0], mm[0], ll[0]);
            split3(ra[i].y, hh[1], mm[1], ll[1]);
            split3(ra[i].z, hh[2], mm[2], ll[2]);
            split3(ra[i].w, hh[3], mm[3], ll[3]);
            const bf16x4 h = {hh[0], hh[1], hh[2], hh[3]}, m = {mm[0], mm[1], mm[2], mm[3]}, l = {ll[0], ll[1], ll[2], ll[3]};
            *reinterpret_cast<bf16x4*>(&As[0][row][c4 * 4]) = h;
            *reinterpret_cast<bf16x4*>(&As[1][row][c4 * 4]) = m;
            *reinterpret_cast<bf16x4*>(&As[2][row][c4 * 4]) = l;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + i * 256, row = idx >> 2, c8 = idx & 3;
                *reinterpret_cast<bf16x8*>(&Ws[p][row][c8 * 8]) = rw[p][i];
            }
    };

    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();   // the previous step's fragment reads are done
        lstore();
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);   // in flight under the MFMAs
#pragma unroll
        for (int c = 0; c < BK / 16; ++c) {
            bf16x8 a[3][2], b[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[p][i] = *reinterpret_cast<const bf16x8*>(&As[p][wm * 64 + i * 32 + l31][c * 16 + kg * 8]);
                    b[p][i] = *reinterpret_cast<const bf16x8*>(&Ws[p][wn * 64 + i * 32 + l31][c * 16 + kg * 8]);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // smallest terms first
                    if (PRODUCTS >= 9) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[2][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[1][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[2][j], acc[i][j], 0, 0, 0);
                    }
                    if (PRODUCTS >= 6) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
                }
        }
    }
    // C/D layout of the 32x32 block: column (W row) = lane & 31, row (A row) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) {
                const int row = tm * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                const int col = tn * BN + wn * 64 + j * 32 + l31;
                C[size_t(row) * N + col] = acc[i][j][r];
            }
}

typedef int (*linear_fn)(const float*, long long, int, long long, const float*, int, long long, const float*, const float*, long long,
                         int, float*, long long, void*);

template <int P>
float time_split(const float* A, const __bf16* h, const __bf16* m, const __bf16* l, float* C, int M, int N, int K, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = (M / BM) * (N / BN);
    for (int i = 0; i < 3; ++i) gemm_split_kernel<P><<<grid, 256>>>(A, h, m, l, C, M, N, K);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) gemm_split_kernel<P><<<grid, 256>>>(A, h, m, l, C, M, N, K);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
    linear_fn lamp_linear = nullptr;
    if (argc > 1) {
        void* lib = dlopen(argv[1], RTLD_NOW);
        if (lib) lamp_linear = (linear_fn)dlsym(lib, "lamp_linear_fwd");
        if (!lamp_linear) printf("(could not load lamp_linear_fwd from %s: %s)\n", argv[1], dlerror());
    }
    const int shapes[][3] = {{9728, 512, 512}, {9728, 2048, 512}, {2944, 512, 512}, {31488, 2048, 1024}, {4096, 4096, 4096}};
    printf("%-22s %28s %22s %22s %22s\n", "M x N x K", "product fp32 MFMA (us / TF)", "bf16x3 (us / TF / err)", "bf16x6 (us / TF / err)", "bf16x9 (us / TF / err)");
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<float> hA(size_t(M) * K), hW(size_t(N) * K);
        srand(1);
        auto rnd = [] { float s = 0; for (int i = 0; i < 4; ++i) s += float(rand()) / RAND_MAX - 0.5f; return s * 1.7f; };   // ~N(0, 1)
        for (auto& v : hA) v = rnd();
        for (auto& v : hW) v = rnd() / sqrtf(float(K));
        float *A, *W, *C, *Cref;
        __bf16 *h, *m, *l;
        CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&W, hW.size() * 4)); CK(hipMalloc(&C, size_t(M) * N * 4)); CK(hipMalloc(&Cref, size_t(M) * N * 4));
        CK(hipMalloc(&h, hW.size() * 2)); CK(hipMalloc(&m, hW.size() * 2)); CK(hipMalloc(&l, hW.size() * 2));
        CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
        split_planes<<<1024, 256>>>(W, h, m, l, hW.size());
        const int iters = double(M) * N * K > 1e11 ? 5 : 20;
        const double flop = 2.0 * M * N * K;
        // fp64 reference on 48 rows
        const int nr = 48;
        std::vector<double> ref(size_t(nr) * N), scale(size_t(nr) * N);
        for (int r = 0; r < nr; ++r) {
            const int row = int((long long)r * 7919 % M);
            for (int n = 0; n < N; ++n) {
                double s = 0, sa = 0;
                for (int k = 0; k < K; ++k) { const double p = double(hA[size_t(row) * K + k]) * hW[size_t(n) * K + k]; s += p; sa += fabs(p); }
                ref[size_t(r) * N + n] = s; scale[size_t(r) * N + n] = sa;
            }
        }
        std::vector<float> hC(size_t(M) * N);
        auto err = [&](float* dC) {
            CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0;
            for (int r = 0; r < nr; ++r) {
                const int row = int((long long)r * 7919 % M);
                for (int n = 0; n < N; ++n) worst = fmax(worst, fabs(hC[size_t(row) * N + n] - ref[size_t(r) * N + n]) / scale[size_t(r) * N + n]);
            }
            return worst;
        };
        float us_ref = 0; double e_ref = 0;
        if (lamp_linear) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int i = 0; i < 3; ++i) lamp_linear(A, M, K, K, W, N, K, nullptr, nullptr, N, 0, Cref, N, nullptr);
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) lamp_linear(A, M, K, K, W, N, K, nullptr, nullptr, N, 0, Cref, N, nullptr);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            us_ref = ms * 1e3f / iters;
            e_ref = err(Cref);
        }
        const float t3 = time_split<3>(A, h, m, l, C, M, N, K, iters); const double e3 = err(C);
        const float t6 = time_split<6>(A, h, m, l, C, M, N, K, iters); const double e6 = err(C);
        const float t9 = time_split<9>(A, h, m, l, C, M, N, K, iters); const double e9 = err(C);
        char name[64];
        snprintf(name, sizeof name, "%d x %d x %d", M, N, K);
        printf("%-22s %9.1f / %6.1f / %.1e   %8.1f / %6.1f / %.1e %8.1f / %6.1f / %.1e %8.1f / %6.1f / %.1e\n", name, us_ref,
               us_ref > 0 ? flop / us_ref / 1e6 : 0.0, e_ref, t3, flop / t3 / 1e6, e3, t6, flop / t6 / 1e6, e6, t9, flop / t9 / 1e6, e9);
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C)); CK(hipFree(Cref)); CK(hipFree(h)); CK(hipFree(m)); CK(hipFree(l));
    }
    return 0;
}
