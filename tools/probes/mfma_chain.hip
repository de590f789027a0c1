// Probe: what does a DEPENDENT chain of v_mfma_f32_32x32x2_f32 cost on gfx950?
//
// The QK^T product of the attention kernels accumulates 64 MFMAs into ONE 32x32 block (the next instruction's C operand is the
// previous one's result); PV rotates over four blocks.  Each wave runs ITERS x 64 MFMAs over NACC accumulators (1 = fully
// dependent, 2, 4 = the rotation of tools/probes/mfma_issue.hip) with 1 or 2 waves per SIMD on all CUs; reported: shader cycles
// per MFMA and SIMD (64 = the pipe never waits).  Pattern "attn": 64 dependent, then 64 over four accumulators, as one
// attention tile step issues them.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/probes/mfma_chain.hip -o tools/probes/mfma_chain && tools/probes/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MIX: other instructions of the SAME wave between its MFMAs -- 1: one v_add per MFMA; 2: four v_add per four MFMAs; 3: sixteen
// v_add per four MFMAs; 4: one ds_read_b128 per four MFMAs; 5: one ds_read_b128 + s_waitcnt lgkmcnt(0) per four MFMAs (the
// fragment is used at once); 6: one v_exp_f32 per MFMA
template <int NACC, int MIX>
__global__ __launch_bounds__(256) void mixed(const float* __restrict__ ab, float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    float a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = ab[(j * 2 + 0) * 256 + threadIdx.x];
        b[j] = ab[(j * 2 + 1) * 256 + threadIdx.x];
    }
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = a[0];
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = a[i & 3];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 frag = {0, 0, 0, 0};
    const unsigned addr = (threadIdx.x & 63) * 16;
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            acc[r % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r & 3], b[(r >> 2) & 3], acc[r % NACC], 0, 0, 0);
            if (MIX == 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[r & 15]));
            if (MIX == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(x[r & 15]));
            if ((r & 3) == 3) {
                if (MIX == 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[i]));
                }
                if (MIX == 3) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[i]));
                }
                if (MIX == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(frag) : "v"(addr));
                if (MIX == 5) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(frag) : "v"(addr));
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)");
    float s = frag[0];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[m][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}

template <int NACC, int MIX>
static void run_mixed(const float* ab, float* out, unsigned long long* cyc, int wg_per_cu) {
    const int grid = 256 * wg_per_cu;
    const int iters = int(3e-3 * 2.4e9 / (64.0 * 64 * wg_per_cu));
    mixed<NACC, MIX><<<grid, 256>>>(ab, out, cyc, iters / 8);
    mixed<NACC, MIX><<<grid, 256>>>(ab, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double c = 0;
    for (auto v : h) c += double(v);
    c /= h.size();
    const char* names[7] = {"MFMAs only", "+ 1 v_add per MFMA", "+ 4 v_add per 4 MFMAs", "+ 16 v_add per 4 MFMAs", "+ 1 ds_read_b128 per 4 MFMAs",
                            "+ 1 ds_read_b128 and wait per 4", "+ 1 v_exp per MFMA"};
    printf("%d accumulator(s) %-34s %d wave/SIMD   %6.2f cycles per MFMA and wave\n", NACC, names[MIX], wg_per_cu, c / (double(iters) * 64));
}

template <int NACC>   // 1, 2, 4; 0 = the attention pattern
__global__ __launch_bounds__(256) void chain(const float* __restrict__ ab, float* out, unsigned long long* cyc, int iters) {
    float a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = ab[(j * 2 + 0) * 256 + threadIdx.x];
        b[j] = ab[(j * 2 + 1) * 256 + threadIdx.x];
    }
    f32x16 acc[5];
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (NACC == 0) {
#pragma unroll
            for (int r = 0; r < 64; ++r) acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r & 3], b[(r >> 2) & 3], acc[4], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 64; ++r) acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(r >> 2) & 3], acc[4][r & 15], acc[r & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 64; ++r)
                acc[r % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r & 3], b[(r >> 2) & 3], acc[r % NACC], 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[m][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
}

template <int NACC>
static void run(const float* ab, float* out, unsigned long long* cyc, int wg_per_cu) {
    const int grid = 256 * wg_per_cu, per_iter = NACC == 0 ? 128 : 64;
    const int iters = int(5e-3 * 2.4e9 / (64.0 * per_iter * wg_per_cu));
    chain<NACC><<<grid, 256>>>(ab, out, cyc, iters / 8);
    chain<NACC><<<grid, 256>>>(ab, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double c = 0;
    for (auto v : h) c += double(v);
    c /= h.size();
    const double cpm = c / (double(iters) * per_iter) / wg_per_cu;
    printf("%-28s %d wave/SIMD   %6.2f cycles/MFMA/SIMD   (pipe busy %5.1f %%)\n",
           NACC == 0 ? "64 dependent + 64 over four" : NACC == 1 ? "one accumulator (dependent)" : NACC == 2 ? "two accumulators" : "four accumulators",
           wg_per_cu, cpm, 6400.0 / cpm);
}

int main() {
    float *ab, *out;
    unsigned long long* cyc;
    std::vector<float> h(8 * 256);
    srand(1234);
    for (auto& v : h) v = float(rand()) / RAND_MAX * 2.f - 1.f;
    hipMalloc(&ab, h.size() * 4);
    hipMemcpy(ab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 512 * 256 * 4);
    hipMalloc(&cyc, 512 * 4 * 8);
    printf("# v_mfma_f32_32x32x2_f32: cycles per instruction and SIMD by accumulator rotation (64 = 16 passes, the pipe never idle)\n");
    for (int w : {1, 2}) {
        run<1>(ab, out, cyc, w);
        run<2>(ab, out, cyc, w);
        run<4>(ab, out, cyc, w);
        run<0>(ab, out, cyc, w);
    }
    printf("# the same wave's other instructions between its MFMAs (cycles per MFMA as ONE wave sees them: 64 = hidden; with two waves per SIMD 128 = hidden)\n");
    for (int w : {1, 2}) {
        run_mixed<1, 0>(ab, out, cyc, w);
        run_mixed<1, 1>(ab, out, cyc, w);
        run_mixed<1, 2>(ab, out, cyc, w);
        run_mixed<1, 3>(ab, out, cyc, w);
        run_mixed<1, 4>(ab, out, cyc, w);
        run_mixed<1, 5>(ab, out, cyc, w);
        run_mixed<1, 6>(ab, out, cyc, w);
        run_mixed<4, 0>(ab, out, cyc, w);
        run_mixed<4, 1>(ab, out, cyc, w);
        run_mixed<4, 3>(ab, out, cyc, w);
        run_mixed<4, 5>(ab, out, cyc, w);
    }
    return 0;
}
