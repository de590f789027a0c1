// Inline-assembly building blocks shared by the hand-scheduled kernels (chain.hip, slab.hip): LDS-DMA, untracked loads with
// hand-counted waits, LDS accesses the compiler must not order behind LDS-DMA, buffer descriptors in scalar registers.
#pragma once
#include <type_traits>

#include "lamp_kernels.h"

namespace lamp {
namespace {
typedef __attribute__((address_space(3))) void* lds_ptr;
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, float* dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)dst, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ f32x4 lds_read16(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// A 16-byte global load the compiler does not track: issued at the start of a pass, consumed after its k loop -- loads retire
// in order and every k step waits until at most one W stage is outstanding, so the value has landed long before.  (An
// ordinary load makes hipcc wait vmcnt(0) at the use, which drains the W stages requested ahead for the NEXT pass.)
__device__ __forceinline__ f32x4 global_read16_untracked(const float* ptr) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
// The W stream's loads, equally invisible to the compiler: its own wait-count bookkeeping is exact inside straight-line code
// but gives up at a loop's back edge -- an unrolled group of k steps then starts by draining EVERY stage in flight
// (s_waitcnt vmcnt(3) .. vmcnt(0) before the first ds_write), i.e. the prefetch depth collapses once per group.  With the
// loads in inline assembly the only vector-memory waits in the k loop are the counted ones written below.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 raw_rsrc(const float* base, unsigned bytes) {
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    return u32x4{unsigned(__builtin_amdgcn_readfirstlane(unsigned(b))), unsigned(__builtin_amdgcn_readfirstlane(unsigned(b >> 32) & 0xffffu)),
                 unsigned(__builtin_amdgcn_readfirstlane(bytes)), 0x00020000u};
}
__device__ __forceinline__ f32x4 buffer_read16_untracked(u32x4 rs, unsigned voff, unsigned soff) {
    f32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rs), "s"(soff) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write16(unsigned addr, f32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);   // hipcc moves register-only instructions (MFMAs) across an asm wait otherwise
}
__device__ __forceinline__ void wg_barrier() {   // LDS writes of this wave done, then the workgroup barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ const float* uniform_ptr(const float* q) {
    const uint64_t b = reinterpret_cast<uint64_t>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(b)), hi = __builtin_amdgcn_readfirstlane(unsigned(b >> 32));
    return reinterpret_cast<const float*>((uint64_t(hi) << 32) | lo);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_u(const float* base, uint64_t bytes) {
    const unsigned n = bytes >= 0x7fffffffull ? 0x7fffffffu : unsigned(bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(base)), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
}

// The values of inline-assembly loads become usable at the counted wait that covers them.  The compiler sees them "defined" at
// the load itself, so two rules keep it from touching the registers early: (1) no control-flow merge between an untracked load
// and its wait -- a load under `if` makes the merge a register COPY placed right behind the load, before the data has landed
// (found in round 5: LayerNorm operands read under `if (n.res)` came out stale, never twice the same); loads are unconditional,
// from a harmless address when the operand is absent, and the CHOICE happens after the wait; (2) settle() right behind the
// wait: an empty asm that "rewrites" the registers, so that every use is ordered behind it.
__device__ __forceinline__ void settle(f32x4& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void settle(float& a) { asm volatile("" : "+v"(a)); }
// (3) a register whose untracked load is still in flight must stay LIVE until the wait that covers it, even if its value is
// never used -- the requests a stream issues past its end (empty descriptor, zeros) and the fragment reads behind the last
// stage: a register the compiler considers dead is handed to the next temporary, and the landing load then overwrites THAT
// (round 5, slab.hip: one wave of a workgroup in ten wrote garbage, never the same one).  keep_alive() after the wait is a use.
// (4) a scalar register written by the VECTOR unit (v_readfirstlane of a descriptor word, v_readlane of a spilled scalar) must not
// be read by a vector-memory instruction within five wait states.  hipcc pads that hazard for the instructions it knows; an
// inline-assembly load is opaque to its hazard recognizer (round 5, slab.hip: scalar-register pressure made it reload the
// stream's scalar offset with v_readlane right in front of a prologue load -- the load went out with the OLD offset, and one
// 1-KiB quad of one wave in ten came back as zeros).  sgpr_guard() in front of the loads that follow freshly built scalars;
// tools/check_untracked_loads.py checks every inline-assembly memory instruction for the distance.
// (the scalars are operands of the nop: the compiler has to have them in their registers BEFORE it)
__device__ __forceinline__ void sgpr_guard(u32x4 rs, unsigned soff) { asm volatile("s_nop 4" ::"s"(rs), "s"(soff) : "memory"); }
__device__ __forceinline__ void sgpr_guard(u32x4 rs) { asm volatile("s_nop 4" ::"s"(rs) : "memory"); }
__device__ __forceinline__ void keep_alive(const f32x4& a) { asm volatile("" ::"v"(a)); }
__device__ __forceinline__ void keep_alive(float a) { asm volatile("" ::"v"(a)); }
template <int OFF>
__device__ __forceinline__ f32x4 lds_read16_off(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}


template <int OFF>
__device__ __forceinline__ f32x4 buffer_read16_untracked_off(u32x4 rs, unsigned voff, unsigned soff) {
    f32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(v) : "v"(voff), "s"(rs), "s"(soff), "n"(OFF) : "memory");
    return v;
}
// four-byte buffer accesses the compiler does not track (epilogues next to a stream in flight): range-checked by the
// descriptor (voffset past num_records: loads return 0, stores are dropped)
__device__ __forceinline__ float buffer_read4_untracked(u32x4 rs, unsigned voff) {
    float v;
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rs) : "memory");
    return v;
}
__device__ __forceinline__ void buffer_write4_untracked(u32x4 rs, unsigned voff, float v) {
    asm volatile("buffer_store_dword %0, %1, %2, 0 offen" ::"v"(v), "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ float lds_read4(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
template <int OFF>
__device__ __forceinline__ float lds_read4_off(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ void lds_write4(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

}  // namespace
}  // namespace lamp
