// Probe: do 16-byte buffer / global loads need 16-byte alignment on gfx950, or only dword alignment?  gemm_gen.hip takes its
// 16-byte staging loads only for 16-byte aligned operands (vec_ok) and falls back to four scalar loads otherwise -- e.g. the
// attention-backward products over probability maps with 302 keys (1208-byte rows: 8-byte aligned).  One wave loads a float4 at
// byte offset 16 * lane + 4 * shift for shift = 0..3 through (a) a raw buffer load, (b) a plain global load, and the host
// compares with the four floats that live there.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/unaligned_b128.hip -o tools/probes/unaligned_b128 && tools/probes/unaligned_b128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* src, unsigned bytes, int shift, float* out_buf, float* out_glb) {
    const int l = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, int(bytes), 0x00020000);
    const unsigned off = unsigned(l) * 16u + 4u * unsigned(shift);
    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    f32x4 b;
    const float* p = src + 4 * l + shift;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(b) : "v"(p) : "memory");
    for (int i = 0; i < 4; ++i) {
        out_buf[4 * l + i] = a[i];
        out_glb[4 * l + i] = b[i];
    }
}

int main() {
    const int n = 64 * 4 + 8;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = float(i + 1);
    float *src, *o1, *o2;
    hipMalloc(&src, n * 4); hipMalloc(&o1, 256 * 4); hipMalloc(&o2, 256 * 4);
    hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (int shift = 0; shift < 4; ++shift) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, unsigned(n * 4), shift, o1, o2);
        std::vector<float> r1(256), r2(256);
        hipMemcpy(r1.data(), o1, 256 * 4, hipMemcpyDeviceToHost);
        hipMemcpy(r2.data(), o2, 256 * 4, hipMemcpyDeviceToHost);
        int e1 = 0, e2 = 0;
        for (int i = 0; i < 256; ++i) {
            e1 += r1[i] != h[i + shift];
            e2 += r2[i] != h[i + shift];
        }
        printf("byte offset 16*lane + %2d: buffer_load_dwordx4 %s (%d wrong), global_load_dwordx4 %s (%d wrong)\n", 4 * shift,
               e1 ? "WRONG" : "ok", e1, e2 ? "WRONG" : "ok", e2);
        bad += e1 + e2;
    }
    printf(bad ? "16-byte loads need more than dword alignment here\n" : "16-byte loads work at any dword alignment\n");
    return 0;
}
