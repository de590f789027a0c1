// Internal (non-ABI) declarations shared by the HIP translation units of liblamp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lamp_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace lamp {

constexpr int GEMM_MAX_SEG = 4;

// C_s[M, N] = act(A[M,K] . W_s[N,K]^T + bias_s) + R   for s in [0, nseg): up to four weight
// matrices sharing one A operand are served by ONE launch (Q/K/V projections read in their native
// nn.Linear layouts, no concatenated repack -- SURVEY.md section 8b "Ownership").
struct GemmParams {
    const float* A;
    int64_t lda;
    int64_t M;
    int K;
    int N;  // per segment
    int nseg;
    const float* W[GEMM_MAX_SEG];
    int64_t ldw;
    const float* bias[GEMM_MAX_SEG];
    float* C[GEMM_MAX_SEG];
    int64_t ldc;
    const float* R;  // residual, same for every segment (only meaningful with nseg == 1)
    int64_t ldr;
    int64_t r_mod;   // > 0: residual row = m % r_mod (one [r_mod, N] block shared by every sample)
    int relu;
};

struct AttnParams {
    const float* Q;
    const float* K;
    const float* V;
    float* O;
    float* P;  // nullable: (H*B, lq, lk) probabilities, index h*B + b.  With P given, V and O may
               // both be NULL: probabilities only (the reference's dead encoder self-attention).
    int B, H, lq, lk, dk, dv;
    lamp_attn_layout lay;
    float scale_log2e;  // inv_temperature * log2(e)
    int mask_kind;
    const void* mask;
    int64_t m_sb, m_sq;
};

int launch_gemm(const GemmParams& p, hipStream_t s);
int launch_attn(const AttnParams& p, hipStream_t s);
// y = LayerNorm(x + residual[row % r_mod or row])   (residual nullable; r_mod 0 = per-row residual)
int launch_layernorm(const float* x, int64_t M, int d, const float* g, const float* b, float eps,
                     const float* residual, int64_t r_mod, float* y, hipStream_t s);
int launch_embed(const int64_t* seq, const int64_t* pos, int64_t n_tok, const float* emb, int n_vocab,
                 const float* pos_table, int n_position, int d, float* out, hipStream_t s);
int launch_diag(const float* y, const float* w, int B, int L, int d, float* logits, hipStream_t s);

// ---- profiling (lamp_prof_* in the ABI) ----
struct ProfScope {
    int idx;
    hipStream_t s;
    ProfScope(int kernel_class, double flops, double bytes, hipStream_t s);
    ~ProfScope();
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace lamp
