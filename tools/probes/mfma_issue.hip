// Probe: fp32-MFMA issue rate and the clock it is sustained at, on gfx950.
//
// Each wave runs ITERS x [16 (16x16x4) or 4 (32x32x2) independent-accumulator MFMAs, fully unrolled, one backward branch per
// 64 / 32 MFMAs] -- no LDS, no memory traffic inside the loop -- with 1 or 2 waves per SIMD on all 256 CUs, for >= 10 ms.
// Reported per arm: TFLOP/s between HIP events, and the EFFECTIVE shader clock during the run, measured in the kernel itself
// as (s_memtime ticks) / (s_memrealtime ticks x 10 ns): s_memtime counts shader-clock cycles, s_memrealtime is the constant
// 100 MHz reference, so the ratio is the clock the power manager actually granted -- no profiler involved.  Arms differ in the
// operand DATA (zeros / small integers / full-range random): the chip clocks to its power budget, and MFMA power depends on
// the bits that toggle (MI355X_MICROARCH.md "DVFS give-back").
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/probes/mfma_issue.hip -o tools/probes/mfma_issue && tools/probes/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Stamp {
    unsigned long long cyc0, cyc1, rt0, rt1;
};

template <int SHAPE>   // 16: 16x16x4, 32: 32x32x2
__global__ __launch_bounds__(256) void mfma_issue(const float* __restrict__ ab, float* out, Stamp* stamps, int iters) {
    // operands from memory (one A and one B value per lane and slot) so that the data pattern is the caller's choice
    float a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = ab[(j * 2 + 0) * 256 + threadIdx.x];
        b[j] = ab[(j * 2 + 1) * 256 + threadIdx.x];
    }
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    float s = 0;
    if constexpr (SHAPE == 16) {
        f32x4 acc[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) acc[m] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < 16; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(m + r) & 3], b[m & 3], acc[m], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    } else {
        f32x16 acc[4];   // 64 registers: two such waves fit one SIMD
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(m + r) & 3], b[(m + (r >> 2)) & 3], acc[m], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[m][e];
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c0, c1, r0, r1};
}

template <int SHAPE>
static void run(const char* data_name, const float* ab, float* out, Stamp* stamps, int wg_per_cu, double target_ms) {
    const int grid = 256 * wg_per_cu;
    const int per_iter = SHAPE == 16 ? 64 : 32;                       // MFMAs per loop iteration and wave
    const double flop_per_mfma = SHAPE == 16 ? 2.0 * 16 * 16 * 4 : 2.0 * 32 * 32 * 2;
    const double cyc_per_mfma = SHAPE == 16 ? 32.0 : 64.0;
    // iterations for ~target_ms at 2.4 GHz with wg_per_cu waves per SIMD
    const int iters = int(target_ms * 1e-3 * 2.4e9 / (cyc_per_mfma * per_iter * wg_per_cu));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    mfma_issue<SHAPE><<<grid, 256>>>(ab, out, stamps, iters / 8);   // warm-up (clock ramp)
    hipDeviceSynchronize();
    std::vector<double> tf, ghz, cpm;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        mfma_issue<SHAPE><<<grid, 256>>>(ab, out, stamps, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<Stamp> h(grid * 4);
        hipMemcpy(h.data(), stamps, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
        double clk = 0, cyc = 0;
        for (auto& st : h) {
            clk += double(st.cyc1 - st.cyc0) / (double(st.rt1 - st.rt0) * 10e-9) / 1e9;
            cyc += double(st.cyc1 - st.cyc0);
        }
        clk /= h.size();
        cyc /= h.size();
        const double waves = double(grid) * 4;
        tf.push_back(waves * iters * per_iter * flop_per_mfma / (ms * 1e-3) / 1e12);
        ghz.push_back(clk);
        cpm.push_back(cyc / (double(iters) * per_iter) / wg_per_cu);   // shader cycles per MFMA and SIMD
    }
    std::sort(tf.begin(), tf.end());
    std::sort(ghz.begin(), ghz.end());
    std::sort(cpm.begin(), cpm.end());
    printf("%-8s %-10s %d wave/SIMD  %7.2f ms/launch  %6.1f TFLOP/s  eff. clock %5.3f GHz  %5.2f cycles/MFMA/SIMD  (= %5.1f %% of 64 FLOP/clk/SIMD at that clock)\n",
           SHAPE == 16 ? "16x16x4" : "32x32x2", data_name, wg_per_cu, target_ms, tf[1], ghz[1], cpm[1], 100.0 * cyc_per_mfma / cpm[1]);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double target_ms = argc > 1 ? atof(argv[1]) : 20.0;
    float *ab[3], *out;
    Stamp* stamps;
    hipMalloc(&out, 512 * 256 * sizeof(float));
    hipMalloc(&stamps, 512 * 4 * sizeof(Stamp));
    const char* names[3] = {"zeros", "small-int", "random"};
    for (int d = 0; d < 3; ++d) {
        std::vector<float> h(8 * 256);
        srand(1234);
        for (auto& v : h) v = d == 0 ? 0.f : d == 1 ? float(rand() % 5 - 2) : (float(rand()) / RAND_MAX * 2.f - 1.f);
        hipMalloc(&ab[d], h.size() * sizeof(float));
        hipMemcpy(ab[d], h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    }
    printf("# fp32 MFMA issue micro-benchmark: 256 CUs x 4 SIMDs, no LDS / memory traffic in the loop; peak at 2.4 GHz = 157.3 TFLOP/s\n");
    for (int d = 0; d < 3; ++d)
        for (int w : {1, 2}) {
            run<16>(names[d], ab[d], out, stamps, w, target_ms);
            run<32>(names[d], ab[d], out, stamps, w, target_ms);
        }
    return 0;
}
