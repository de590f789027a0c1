// Probe: do v_pk_fma_f32 (vector ALU) instructions execute in the shadow of v_mfma_f32_16x16x4_f32 (matrix pipe) on gfx950?
// Each wave runs ITERS x [4 independent MFMAs + 4 V independent packed FMAs]; time vs V tells whether the vector work is free.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/coissue.hip -o gpurun_out/coissue && gpurun_out/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(256) void coissue(float* out, int iters, float seed) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x2 v[8];
    for (int j = 0; j < 8; ++j) v[j] = f32x2{seed + j, seed - j};
    const float a = seed * threadIdx.x, b = seed + threadIdx.x;
    const f32x2 c = {1.0001f, 0.9999f}, d = {seed, -seed};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < V; ++j) v[(m * V + j) & 7] = __builtin_elementwise_fma(v[(m * V + j) & 7], c, d);
        }
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    for (int j = 0; j < 8; ++j) s += v[j][0] + v[j][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
void run(float* out, int wg_per_cu) {
    const int iters = 20000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    coissue<V><<<grid, 256>>>(out, 200, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    coissue<V><<<grid, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = double(grid) * 4, mfma_flop = waves * iters * 4 * 2048.0, valu_flop = waves * iters * 4.0 * V * 256.0;
    printf("V=%d pk_fma per MFMA, %d WG/CU: %8.3f ms   MFMA %6.1f TFLOP/s   VALU %6.1f TFLOP/s   total %6.1f\n", V, wg_per_cu, ms,
           mfma_flop / ms / 1e9, valu_flop / ms / 1e9, (mfma_flop + valu_flop) / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2}) {
        run<0>(out, w); run<1>(out, w); run<2>(out, w); run<4>(out, w); run<6>(out, w); run<8>(out, w);
    }
    return 0;
}
