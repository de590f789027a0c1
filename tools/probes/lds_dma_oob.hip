// Probe: what does an LDS-DMA load (buffer_load_dwordx4 ... lds) leave in LDS for a lane whose offset fails the buffer
// descriptor's range check?  attention_tile.hip relies on ZEROS being written (key rows past a sample's last key, head
// dimensions past d_k): stale LDS contents there would be multiplied by probabilities that are exactly 0 -- harmless unless
// the stale value is Inf / NaN.  One wave: LDS poisoned with NaN, one DMA instruction whose lanes are in range / just past the
// end / at the 0x80000000 sentinel, LDS read back.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_dma_oob.hip -o tools/probes/lds_dma_oob && tools/probes/lds_dma_oob
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

// the same with part of the offset in the instruction's SCALAR offset: does the range check see voffset + soffset?
__global__ void probe_soffset(const float* src, unsigned bytes, float* out, unsigned soff) {
    __shared__ __attribute__((aligned(16))) float buf[256];
    const int l = threadIdx.x;
    for (int i = 0; i < 4; ++i) buf[4 * l + i] = __builtin_nanf("");
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, int(bytes), 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)buf, 16, unsigned(l) * 16u, soff, 0, 0);   // lane l: bytes soff + 16 l ..
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[4 * l + i] = buf[4 * l + i];
}

__global__ void probe(const float* src, unsigned bytes, float* out) {
    __shared__ __attribute__((aligned(16))) float buf[256];
    const int l = threadIdx.x;
    for (int i = 0; i < 4; ++i) buf[4 * l + i] = __builtin_nanf("");
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, int(bytes), 0x00020000);
    // lanes 0-15 in range, 16-31 straddle / pass the end of the descriptor, 32-47 the sentinel offset, 48-63 in range again
    unsigned off = unsigned(l) * 16u;
    if (l >= 16 && l < 32) off = bytes - 8u + unsigned(l - 16) * 16u;
    if (l >= 32 && l < 48) off = 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)buf, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[4 * l + i] = buf[4 * l + i];
}

int main() {
    const int n = 1024;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = float(i + 1);
    float *src, *out;
    hipMalloc(&src, n * 4);
    hipMalloc(&out, 256 * 4);
    hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, unsigned(n * 4), out);
    std::vector<float> r(256);
    hipMemcpy(r.data(), out, 256 * 4, hipMemcpyDeviceToHost);
    int bad_in = 0, nonzero_oob = 0, nan_oob = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const float v = r[4 * l + i];
            const bool in_range = l < 16 || l >= 48;
            if (in_range) bad_in += v != float(4 * l + i + 1);
            else if (l >= 32 || l > 16 || i >= 2) {   // lane 16 holds the last 8 valid bytes in its first two floats
                nan_oob += std::isnan(v);
                nonzero_oob += !(v == 0.0f);
            }
        }
    printf("in-range lanes wrong: %d; out-of-range values not zero: %d (NaN left in place: %d)\n", bad_in, nonzero_oob, nan_oob);
    printf("lane 16 (straddles the end): %g %g %g %g\n", r[64], r[65], r[66], r[67]);
    printf("%s\n", (bad_in == 0 && nonzero_oob == 0) ? "LDS-DMA writes ZEROS for range-checked lanes" : "LDS-DMA does NOT zero-fill");
    // scalar offset: descriptor of 2048 bytes (the allocation is 4096), soffset 1536: lanes 0-31 in range, lanes 32-63 past num_records
    hipLaunchKernelGGL(probe_soffset, dim3(1), dim3(64), 0, 0, src, 2048u, out, 1536u);
    hipMemcpy(r.data(), out, 256 * 4, hipMemcpyDeviceToHost);
    int ok_in = 0, zero_out = 0, data_out = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const float v = r[4 * l + i], want = float(1536 / 4 + 4 * l + i + 1);
            if (l < 32) ok_in += v == want;
            else { zero_out += v == 0.0f; data_out += v == want; }
        }
    printf("scalar offset: in-range values right %d / 128; past num_records: zeros %d / 128, memory contents %d / 128 -> the range check %s soffset\n",
           ok_in, zero_out, data_out, zero_out == 128 ? "INCLUDES" : "does NOT include");
    return (bad_in == 0 && nonzero_oob == 0) ? 0 : 1;
}
