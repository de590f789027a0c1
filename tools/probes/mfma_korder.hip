// Probe: is one v_mfma_f32_16x16x4_f32 the same arithmetic as four v_mfma_f32_4x4x1_16b_f32 (one k each) in k order?
//
// The GEMMs of this library are bit-identical across tile shapes because every output element is one k-ordered chain of the
// 16x16x4 instruction.  A kernel built on the 4x4x1 (16 blocks) instruction -- whose row granularity is 4 instead of 16 --
// could only join them if the hardware's accumulation INSIDE a 16x16x4 step is the sequential fma chain over its four k.
// One wave computes C = A . B^T for A, B [16 x K] both ways and compares bits, for the four orders the 4-deep step could use.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_korder.hip -o tools/probes/mfma_korder && tools/probes/mfma_korder
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// C[16][16] = A[16][K] . B[16][K]^T with 16x16x4: lane l supplies A[l & 15][4 s + (l >> 4)], B likewise; D: lane l holds
// C[4 (l >> 4) + r][l & 15], r = 0..3
__global__ void ref16(const float* A, const float* B, float* C, int K) {
    const int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < K / 4; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * K + 4 * s + (l >> 4)], B[(l & 15) * K + 4 * s + (l >> 4)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}
// the same products with 4x4x1 (16 blocks per instruction): block b = l >> 2 pairs row group g and column group h; here one
// instruction covers row group g (all 16 blocks the same 4 rows) x 4 column groups x ... 16 blocks = 16 column groups of 4,
// but C has only 4 column groups: blocks (g', h) = (b >> 2, b & 3) cover all 4 x 4 groups at once.
// lane l: block b = l >> 2, i = j = l & 3.  A operand: A[4 (b >> 2) + i][k]; B operand: B[4 (b & 3) + j][k];
// D register r: C[4 (b >> 2) + r][4 (b & 3) + (l & 3)].   order: 0 = k ascending, 1 = descending within each group of 4,
// 2 = pairs (0,1,2,3 -> 0,2,1,3), 3 = (1,0,3,2)
__global__ void alt4(const float* A, const float* B, float* C, int K, int order) {
    const int l = threadIdx.x, b = l >> 2;
    f32x4 acc = {0, 0, 0, 0};
    const int perm[4][4] = {{0, 1, 2, 3}, {3, 2, 1, 0}, {0, 2, 1, 3}, {1, 0, 3, 2}};
    for (int s = 0; s < K / 4; ++s)
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * s + perm[order][q];
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[(4 * (b >> 2) + (l & 3)) * K + k], B[(4 * (b & 3) + (l & 3)) * K + k], acc, 0, 0, 0);
        }
    for (int r = 0; r < 4; ++r) C[(4 * (b >> 2) + r) * 16 + 4 * (b & 3) + (l & 3)] = acc[r];
}

int main() {
    const int K = 512;
    std::vector<float> hA(16 * K), hB(16 * K);
    srand(1);
    for (auto& v : hA) v = (rand() / float(RAND_MAX) - 0.5f) * 4.f;
    for (auto& v : hB) v = (rand() / float(RAND_MAX) - 0.5f) * 4.f;
    float *A, *B, *C0, *C1;
    hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C0, 1024); hipMalloc(&C1, 1024);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(ref16, dim3(1), dim3(64), 0, 0, A, B, C0, K);
    std::vector<float> r0(256), r1(256);
    hipMemcpy(r0.data(), C0, 1024, hipMemcpyDeviceToHost);
    // fp64 reference and a host fma chain in ascending k
    int host_chain_equal = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float acc = 0.f;
            for (int k = 0; k < K; ++k) acc = __builtin_fmaf(hA[i * K + k], hB[j * K + k], acc);
            host_chain_equal += memcmp(&acc, &r0[i * 16 + j], 4) == 0;
        }
    printf("16x16x4 vs host fmaf chain (k ascending): %d / 256 elements bit-equal\n", host_chain_equal);
    const char* names[4] = {"k ascending", "k descending in fours", "0,2,1,3", "1,0,3,2"};
    for (int o = 0; o < 4; ++o) {
        hipLaunchKernelGGL(alt4, dim3(1), dim3(64), 0, 0, A, B, C1, K, o);
        hipMemcpy(r1.data(), C1, 1024, hipMemcpyDeviceToHost);
        int eq = 0;
        double maxd = 0;
        for (int i = 0; i < 256; ++i) {
            eq += memcmp(&r0[i], &r1[i], 4) == 0;
            maxd = std::max(maxd, double(fabsf(r0[i] - r1[i])));
        }
        printf("4x4x1 chain, order %-22s: %3d / 256 elements bit-equal to 16x16x4 (max |diff| %.3g)\n", names[o], eq, maxd);
    }
    return 0;
}
