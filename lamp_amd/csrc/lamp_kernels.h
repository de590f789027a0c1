// Internal (non-ABI) declarations shared by the HIP translation units of liblamp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
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
    int relu;
    // Residual GATHERED in the epilogue instead of read from R (R must be NULL): row r adds rg_emb[rg_tok[r]][col] (+ rg_pos_table[
    // rg_pos[r]][col]) -- the embedded input row of the encoder's first layer (lamp/Encoders.py:66,75), which then is never
    // materialised: the gather kernel writes the two row -> index maps (8 bytes per row) instead of the row (4 d bytes).  The sum
    // is the single fp32 add the gather kernel would have done: same bits as the residual read.  N must be the tables' width.
    const int* rg_tok;            // nullable: [M] token index per row (already range-checked by the writer)
    const int* rg_pos;            // [M] position index per row (unused without a position table)
    const float* rg_emb;          // [n_vocab, N]
    const float* rg_pos_table;    // [n_position, N] or NULL
    const int* m_dev;   // nullable: the live row count in DEVICE memory (<= M, which then only sizes the launch) -- the
                        // packed token rows of a ragged batch, counted on the device (pointwise.hip: seq_plan_kernel)
    const float* A_dense;  // nullable, with m_dev = SeqPlan::rows: read A_dense instead of A when m_dev[0] == m_dev[1], i.e. no
                           // position of the batch was skipped and the padded encoder output IS the packed matrix (the
                           // last encoder LayerNorm then skips its packed copy)
    int vec_epilogue;   // set by launch_gemm: bias / residual / C can move as 16-byte accesses
    int walk_gn;        // set by the launcher: > 0 = W-resident tile walk with this many column panels per group (gemm.hip)
    unsigned long long* trace;  // tuning build only (per-workgroup timeline, see gemm.hip); NULL in production
};

struct AttnParams {
    const float* Q;
    const float* K;
    const float* V;
    float* O;
    float* P;  // nullable: (H*B, lq, lk) probabilities, index h*B + b.  With P given, V and O may
               // both be NULL: probabilities only (the reference's dead encoder self-attention).
    float* lse;  // nullable; with P: single-pass write-out -- scores into P, row log2-sum-exp here [H][B][lq] (this call's samples),
                 // then normalised in place by a second launch (training forward)
    float* scratch;  // (H*B, lq, lk) floats: score scratch of the general path (d_k or d_v > 128) when P is not given
    int B, H, lq, lk, dk, dv;
    int P_batch, P_b0;  // P is indexed (h * P_batch + P_b0 + b): the maps of a micro-batch inside a larger batch
    lamp_attn_layout lay;
    float scale_log2e;  // inv_temperature * log2(e)
    int mask_kind;
    const void* mask;
    int64_t m_sb, m_sq;
    const int* tiles;      // optional per-32-query-block active key-tile lists (shared masks), or nullptr
    int64_t tiles_stride;
    int sparse_rows;       // lamp_mask.flags & LAMP_MASK_SPARSE_ROWS: the shared mask's rows allow few keys, unstructured
    int64_t allowed_pairs; // unblocked entries of that mask (0 = unknown): the FLOPs attention_sparse.hip executes
    // Ragged keys (lamp_forward's enc-dec attention; SeqPlan): sample b has kv_len[b] <= lk keys -- every key past it is a PAD
    // token, i.e. exactly masked -- and its K / V rows start at row kv_off[b] of the K / V matrices (packed token rows;
    // lay.k_b / lay.v_b are then unused).  Both in device memory, both or neither.  The key loop stops at kv_len[b] and
    // the key split is chosen from kv_len[b], so a sample's bits do not depend on the padded length of its batch.
    const int* kv_len;
    const int* kv_off;
    unsigned long long* trace;  // tuning build only: per-workgroup timeline of attention_small.hip; NULL in production
};

int launch_gemm(const GemmParams& p, hipStream_t s);
// whether launch_gemm takes a gathered residual (GemmParams::rg_tok) for this product: the 16-byte epilogue of whole k-tiles of
// every tile of the menu (shape / alignment rule only: the caller decides BEFORE it leaves the embedded rows unwritten)
bool gemm_gathered_residual_ok(int N, int K, int64_t ldc, const float* bias, const float* C, const float* emb, const float* pos_table);
// slab.hip (tuning build only: measured slower than the tile kernel, profiles/r05_rejected_experiments.txt): the same product
// from format-1 weight packs, one slab of rows per CU, bit-identical to launch_gemm
bool slab_applies(int64_t M, int N, int K, int nseg, const float* const* Wq);
int launch_slab_gemm(const GemmParams& p, const float* const* Wq, hipStream_t s);
int launch_attn(const AttnParams& p, hipStream_t s);
int launch_attn_general(const AttnParams& p, hipStream_t s);  // any d_k / d_v: scores through memory
bool attn_small_applies(const AttnParams& p, bool any_lq = false);  // attention_small.hip: lq <= 256 or lk <= 64 (shape-only rule)
int launch_attn_small(const AttnParams& p, int force_ksplit, hipStream_t s);
// attention_tile.hip: long key sequences, K / V tiles shared by a workgroup through LDS-DMA; the bits of attn_kernel<128, 1, 0, MK>
bool attn_tile_applies(const AttnParams& p);
int launch_attn_tile(const AttnParams& p, hipStream_t s);
// attention_sparse.hip: only the allowed (query, key) pairs of a sparse, unstructured shared mask (exact; not the dense kernels' bits)
bool attn_sparse_applies(const AttnParams& p);
int launch_attn_sparse(const AttnParams& p, hipStream_t s);
size_t gemm_gen_workspace_bytes(int M, int N, int K, int batch);
int launch_gemm_gen(const lamp_gemm_desc& d, void* ws, size_t ws_bytes, hipStream_t s);
int launch_gemm_group(const lamp_gemm_desc* descs, int n, hipStream_t s);
// Counter-based dropout (lamp_dropout): element e of a site is kept iff mix32(e, seed) >= threshold, kept values are
// multiplied by scale = 1 / (1 - p).  threshold == 0: dropout off.
struct DropoutSpec {
    unsigned threshold;
    float scale;
    unsigned seed;
};
inline DropoutSpec make_dropout(float p, uint32_t seed) {
    const double t = double(p) * 4294967296.0;
    return DropoutSpec{t >= 4294967295.0 ? 4294967295u : unsigned(t), 1.0f / (1.0f - p), seed};
}
__device__ __forceinline__ unsigned mix32(unsigned lo, unsigned hi, unsigned seed) {
    unsigned h = lo ^ (hi * 0x9E3779B9u) ^ seed;
    h ^= h >> 16; h *= 0x7feb352du;
    h ^= h >> 15; h *= 0x846ca68bu;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ float drop1(float v, int64_t e, const DropoutSpec& d) {
    return mix32(unsigned(e), unsigned(uint64_t(e) >> 32), d.seed) >= d.threshold ? v * d.scale : 0.f;
}
__device__ __forceinline__ float4 drop4(float4 v, int64_t e, const DropoutSpec& d) {
    return make_float4(drop1(v.x, e, d), drop1(v.y, e + 1, d), drop1(v.z, e + 2, d), drop1(v.w, e + 3, d));
}

// Ragged token batches (lamp_forward): per-sample extents counted ON THE DEVICE by seq_plan_kernel (pointwise.hip), so
// that the encoder runs on the packed non-PAD rows and the enc-dec attention stops at each sample's last real key --
// without a host round trip.  All pointers are device int32 arrays in the caller's workspace.
struct SeqPlan {
    int* klen;  // [nb]     keys of sample b: 1 + its last non-PAD token position
    int* plen;  // [nb]     rows of sample b in the packed matrix (>= klen; T in the padded layout)
    int* off;   // [nb + 1] first packed row of sample b; off[nb] = n_tok
    int* rows;  // [2]      n_tok, and n_tok + 1 when the shared PAD row (row n_tok) is live
    unsigned* padbits;  // [nb][words] bit-packed key mask: bit j of word j / 32 set = position j holds a PAD token
    int words;          // (T + 31) / 32
    unsigned long long* granules;   // nullable: the 2 nb + 2 hand-off words of embed_plan_kernel, zeroed again by the last encoder
                                    // LayerNorm of the forward (a replayed HIP graph repeats the launch's epoch: no stale match)
};
int launch_seq_plan(const int64_t* seq, const int64_t* pos, int nb, int T, int64_t seq_stride, bool packed,
                    const SeqPlan& sp, hipStream_t s);
// The first encoder layer's W1 folded into the embedding tables (lamp_model::enc0_emb_w1 / enc0_pos_w1): the gather kernels then
// also write that layer's hidden rows, hid[row] = relu(e1[tok] (+ p1[pos])), beside the embedded rows.  hid == nullptr: off.
struct EmbedFold {
    const float* e1;   // [n_vocab, dff]     emb . W1^T  (+ b1 when the model has no position table)
    const float* p1;   // [n_position, dff]  pos_table . W1^T + b1, or nullptr without a position table
    int dff;
    float* hid;        // [rows, dff]: same row index as the embedded rows
    int* row_tok;      // nullable, with row_pos: instead of the embedded row (4 d bytes) the gather writes the row's token and
    int* row_pos;      //   position index (8 bytes) -- the only reader of that row, the first layer's second GEMM, adds it as a
                       //   gathered residual (GemmParams::rg_tok)
};
int launch_embed_packed(const int64_t* seq, const int64_t* pos, int nb, int T, const float* emb, int n_vocab,
                        const float* pos_table, int n_position, int d, const SeqPlan& sp, float* out, hipStream_t s,
                        const EmbedFold* fold = nullptr);
// the two above (packed layout) as ONE launch; granules: 2 * nb + 2 unsigned 64-bit words of workspace.  Their content on entry
// must not carry THIS launch's epoch tag: the tag is a host counter (unique per launch), and the last encoder LayerNorm of the
// forward zeroes the granules again -- which is what makes a REPLAYED HIP graph (same tag every replay) safe.  Precondition of a
// replay therefore: the previous replay ran to that LayerNorm (an aborted replay, or a workspace restored from a copy taken
// between the gather and that LayerNorm, would leave granules of the same epoch behind; zero the 2 nb + 2 words first).
int launch_embed_plan(const int64_t* seq, const int64_t* pos, bool plan_uses_pos, int nb, int T, const float* emb, int n_vocab,
                      const float* pos_table, int n_position, int d, const SeqPlan& sp, unsigned long long* granules, float* out,
                      hipStream_t s, const EmbedFold* fold = nullptr);

// y = LayerNorm(dropout(x) + residual[row % r_mod or row])   (residual nullable; r_mod 0 = per-row residual; drop nullable)
// m_dev: live row count in device memory (M sizes the launch).  scatter (+ T, y_flat): the last LayerNorm of the packed
// encoder -- M = nb * T flat positions, y = the packed rows (in place), y_flat = the padded [nb, T, d] encoder output.
int launch_layernorm(const float* x, int64_t M, int d, const float* g, const float* b, float eps,
                     const float* residual, int64_t r_mod, float* y, hipStream_t s, const float* w_out = nullptr,
                     int n_labels = 0, float* logits = nullptr,  // w_out: fused read-out, y may then be NULL
                     const DropoutSpec* drop = nullptr, const int* m_dev = nullptr, const SeqPlan* scatter = nullptr,
                     int T = 0, float* y_flat = nullptr);
// chain.hip: the row-local tail of a decoder block -- attention output projection (+ residual) -> LayerNorm [-> FFN ->
// LayerNorm] -- as ONE launch over 16-row panels, bit-identical to the separate launches.  chain_applies: shape limits
// (LDS residency, tiling) and the row-count heuristic (at most one panel per CU).
bool chain_applies(int64_t M, int d, int k_h, int dff, bool has_ffn, const lamp_chain_pack* pk = nullptr, bool res_mod = false,
                   bool w_out = false);
int launch_chain(const float* A, int64_t lda, int k_h, const float* res, int64_t r_mod, int64_t M, int d, const float* w_fc,
                 const float* ln1_g, const float* ln1_b, const lamp_ffn_weights* ffn, int dff, float* y, const float* w_out,
                 int n_labels, float* logits, hipStream_t s, const lamp_chain_pack* pk = nullptr);
int launch_pack_weight(const float* W, int N, int K, int64_t ldw, int format, float* out, hipStream_t s);
size_t layernorm_bwd_workspace_bytes(int64_t M, int d);
int launch_layernorm_bwd(const float* x, const float* res, int64_t r_mod, int64_t M, int d, const float* g, float eps,
                         const DropoutSpec* drop, const float* dy, float* dz, float* dz_drop, float* dgamma, float* dbeta,
                         float* dbias, void* ws, size_t ws_bytes, hipStream_t s, lamp_reduce_job* job_out = nullptr);
size_t colsum_workspace_bytes(int64_t M, int64_t N);
int launch_colsum(const float* x, int64_t M, int64_t N, int64_t ldx, float* out, void* ws, size_t ws_bytes, hipStream_t s,
                  lamp_reduce_job* job_out = nullptr);
// job_out (both above): skip the second-stage launch and describe it instead -- the partials in ws then have to outlive the
// call until launch_reduce_group has run them (lamp_reduce_partials_grouped)
int launch_reduce_group(const lamp_reduce_job* jobs, int n, hipStream_t s);
int launch_dropout(const float* x, int64_t n, float p, uint32_t seed, float* y, hipStream_t s);
int launch_softmax_bwd(const float* P, const float* dP, int64_t rows, int lk, float scale, float* dS, hipStream_t s,
                       const DropoutSpec* drop = nullptr);   // drop: dP is the gradient of dropout(P), mask applied on load
int launch_diag_bwd(const float* y, const float* w, const float* dl, int B, int L, int d, float* dy, float* dw,
                    hipStream_t s);
int launch_embed_bwd(const int64_t* seq, int64_t n_tok, const float* dout, int d, int n_vocab, int64_t pad_idx,
                     float* d_emb, hipStream_t s);
int launch_embed(const int64_t* seq, const int64_t* pos, int64_t n_tok, const float* emb, int n_vocab,
                 const float* pos_table, int n_position, int d, float* out, hipStream_t s, const EmbedFold* fold = nullptr);
int launch_diag(const float* y, const float* w, int B, int L, int d, float* logits, hipStream_t s);
int launch_prior_graph(const int64_t* ids, const int64_t* offsets, int64_t n_samples, int L, float* adj,
                       uint8_t* blocked, hipStream_t s);
int launch_sigmoid_bce(const float* logits, const float* targets, int64_t n_rows, int L, float* probs,
                       float* row_loss, hipStream_t s);

// ---- profiling (lamp_prof_* in the ABI) ----
struct ProfScope {
    int idx;
    hipStream_t s;
    ProfScope(int kernel_class, double flops, double bytes, hipStream_t s);
    ~ProfScope();
};

// ---- buffer-descriptor helpers (device) ----
// Raw buffer loads return 0 and raw buffer stores are dropped when voffset >= num_records (the
// scalar soffset is NOT range checked), which replaces every bounds branch in the kernels.
constexpr unsigned OOB = 0x80000000u;  // an offset no descriptor of ours covers (num_records <= 0x7fffffff)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint64_t bytes) {
    const unsigned n = bytes >= 0x7fffffffull ? 0x7fffffffu : unsigned(bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    // NB: convert the WHOLE vector with one bit_cast.  Subscripting the builtin's result (v[0], v.x ...)
    // is miscompiled by hipcc 7.2 into a buffer_load_dword splatted over the four lanes.
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 bload2(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}
__device__ __forceinline__ void bstore1(__amdgpu_buffer_rsrc_t r, unsigned voff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, 0, 0);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bstore4(__amdgpu_buffer_rsrc_t r, unsigned voff, float4 v) {
    const f32x4 f = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f), r, voff, 0, 0);
}

__device__ __forceinline__ unsigned bload_u8(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_amdgcn_raw_buffer_load_b8(r, voff, 0, 0);
}
__device__ __forceinline__ unsigned long long bload_u64(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0));
}

// Sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15), result in every lane, as four pure-DPP steps (quad
// swaps, then half-row and row mirrors between groups that already hold equal values).  __shfl_xor compiles to
// ds_bpermute (LDS crossbar, ~100 cycles each); these are VALU-speed.  Same pairing as an xor butterfly.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);  // row_half_mirror
    v += dpp_move<0x140>(v);  // row_mirror
    return v;
}

// Sum over the 64 lanes of a wave, result in every lane: DPP row sums, then the four row totals through readlane
// (v_readlane -> SGPR).  Replaces the __shfl_xor butterfly (six ds_bpermute round trips through the LDS crossbar).
__device__ __forceinline__ float wave64_sum(float v) {
    v = row16_sum(v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

// LayerNorm arithmetic of ONE row held by one wave as NV float4 per lane (lane l: float4 columns l + 64 i; nv = d / 4 of them
// are real, the rest zeros).  Shared by layernorm_kernel (pointwise.hip) and the fused decoder chain (chain.hip), with
// floating-point contraction OFF: what is written is what is computed, in every kernel that inlines it -- the two routes
// give the same bits (hipcc's fma-contraction choices otherwise differ between instantiations of the same source).
template <int NV>
__device__ __forceinline__ void ln_row_stats(const float4 (&v)[NV], int lane, int nv, int d, float eps, float& mean,
                                             float& rstd) {
#pragma clang fp contract(off)
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    mean = wave64_sum(s) / float(d);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + i * 64 < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
            ss += (a * a + b * b) + (cc * cc + dd * dd);
        }
    }
    rstd = 1.0f / sqrtf(wave64_sum(ss) / float(d) + eps);
}
__device__ __forceinline__ float4 ln_row_apply(float4 v, float mean, float rstd, float4 gg, float4 bb) {
#pragma clang fp contract(off)
    return make_float4((v.x - mean) * rstd * gg.x + bb.x, (v.y - mean) * rstd * gg.y + bb.y,
                       (v.z - mean) * rstd * gg.z + bb.z, (v.w - mean) * rstd * gg.w + bb.w);
}
__device__ __forceinline__ float dot4_nocontract(float4 o, float4 w) {
#pragma clang fp contract(off)
    return (o.x * w.x + o.y * w.y) + (o.z * w.z + o.w * w.w);
}

// Workgroup id -> work-item id such that each XCD (workgroup b runs on XCD b % 8) gets a CONTIGUOUS range
// of work items: GEMM tiles sharing an A row-panel, or attention query blocks sharing one (sample, head)'s
// K/V, then hit the same 4 MiB L2.  Bijective for any count; placement affects speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// hipFuncSetAttribute once per (kernel instantiation, device), safe under DataParallel-style threads.
struct AttrOnce {
    std::mutex mu;
    bool done[64] = {};
    int set(const void* kern, size_t lds) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return int(e);
        if (dev < 0 || dev >= 64) return LAMP_E_UNSUPPORTED;
        std::lock_guard<std::mutex> lk(mu);
        if (done[dev]) return 0;
        e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return int(e);
        done[dev] = true;
        return 0;
    }
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace lamp
