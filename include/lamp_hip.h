/*
 * lamp_hip.h -- C ABI of liblamp_hip.so, the MI355X (gfx950) native library behind the
 * LaMP label-graph message-passing forward path.
 *
 * The reference (QData/LaMP) is pure Python on PyTorch: it owns no native code and no FFI.  The
 * "FFI for this path" is therefore the set of nn.Module.forward methods on the path; each entry
 * point below replaces one of them and cites it (paths relative to the reference root).  The
 * Python side (lamp_amd/_native.py) binds these with ctypes and raw data_ptr()s; INTEGRATION.md
 * shows the stub.
 *
 * Conventions
 *   - plain C99 types only; no torch / HIP types in signatures (a stream is a void* holding a
 *     hipStream_t; NULL = the default stream).
 *   - all tensors are fp32, row-major, device-resident; indices are int64; masks are uint8.
 *   - mask convention everywhere (as in the reference): NONZERO = BLOCKED.
 *   - the caller owns every byte: inputs, outputs, weights, workspace.  The library never
 *     allocates or frees device memory and keeps no pointer after return.
 *   - every call only enqueues work on `stream` (asynchronous w.r.t. the host) and is re-entrant and
 *     thread-safe for distinct streams / devices (one host thread per device, as nn.DataParallel runs
 *     its replicas, works); the device is the one current for the calling thread.
 *   - no library state that results depend on: the process-wide data are lazily initialised, immutable per-device
 *     kernel attributes, one atomic launch counter (the tag of lamp_forward's plan hand-off granules: it only has
 *     to differ from launch to launch) and -- for diagnostics, mutex-protected -- the lamp_prof_* event records.
 *   - return value: 0 = ok, > 0 = a hipError_t from a launch, < 0 = lamp_status below.  Nothing
 *     throws or aborts across the boundary.  NaN produced by fully masked attention rows is data,
 *     not an error (reference behaviour).
 */
#ifndef LAMP_HIP_H
#define LAMP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the entry points declared here are its ONLY dynamic symbols. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define LAMP_HIP_ABI_VERSION 5   /* 5: lamp_model grows (enc0_emb_w1 / enc0_pos_w1, label_csr); 4: + lamp_gemm_grouped, lamp_{ffn,mha}_train_fwd / _bwd */

typedef void* lamp_stream_t; /* hipStream_t */

enum lamp_status {
    LAMP_OK = 0,
    LAMP_E_DIMS = -1,        /* non-positive or inconsistent dimensions */
    LAMP_E_ALIGN = -2,       /* pointer / leading dimension not 16-byte aligned where required */
    LAMP_E_WORKSPACE = -3,   /* workspace too small */
    LAMP_E_UNSUPPORTED = -4, /* valid request this build has no kernel for (e.g. d_k not a multiple of 4) */
    LAMP_E_NULL = -5         /* required pointer is NULL */
};

/* Attention mask descriptor.  One code path covers the reference's three mask uses:
 *   key padding  (lamp/utils.py:26-34)      : U8, stride_q = 0, stride_b = lk      [B, lk]
 *                                             or KEY_TOKENS_I64 straight from src_seq
 *   label graph  (lamp/Decoders.py:109-116) : U8, stride_b = 0, stride_q = lk      [lq, lk]
 *   arbitrary    (module-level callers)     : U8, stride_b = lq*lk, stride_q = lk  [B, lq, lk]
 * The same mask is applied to every head (lamp/SubLayers.py:102 repeats it n_head times). */
enum lamp_mask_kind {
    LAMP_MASK_NONE = 0,
    LAMP_MASK_U8 = 1,            /* uint8, nonzero = blocked; element (b,q,k) at ptr[b*stride_b + q*stride_q + k] */
    LAMP_MASK_KEY_TOKENS_I64 = 2, /* int64 token ids [B, stride_b]; key k of sample b is blocked iff token == 0 (PAD) */
    LAMP_MASK_BITS_U32 = 3        /* bit-packed rows: uint32 words, element (b,q,k) = bit (k & 31) of word
                                     ptr[b*stride_b + q*stride_q + (k >> 5)] (strides in words), 1 = blocked; bits
                                     past lk are ignored.  One 4-byte load covers a whole 32-key tile of a row. */
};

/* lamp_mask.flags */
#define LAMP_MASK_SPARSE_ROWS 1  /* a SHARED bit-packed mask (kind BITS_U32, stride_b == 0) whose rows allow only a small fraction
                                    of the keys, with no block structure to skip (an unstructured label graph): the library may
                                    compute the allowed (query, key) pairs only (csrc/attention_sparse.hip) instead of visiting every
                                    key tile.  Set by the caller from the mask's density (lamp_amd/Decoders.py); exact either way. */

typedef struct lamp_mask {
    int32_t kind;
    int32_t flags;               /* LAMP_MASK_SPARSE_ROWS or 0 */
    const void* ptr;
    int64_t stride_b;
    int64_t stride_q;
    /* Optional sparsity hint for a SHARED mask (stride_b == 0): for every block of 32 query rows, the list of
     * 32-key tiles that contain at least one unblocked entry -- int32 rows of length tile_list_stride:
     * [count, tile_0, tile_1, ...] in ascending order.  Tiles not listed are skipped entirely (exact: their
     * probabilities are 0).  NULL = visit every tile.  Ignored when attention maps are written.  This is the
     * label graph's block structure (lamp/Decoders.py:109-113) handed to the kernel once per model. */
    const int32_t* tile_list;
    int64_t tile_list_stride;
    int64_t allowed_pairs;       /* with LAMP_MASK_SPARSE_ROWS: number of unblocked (query, key) entries of the shared mask (the
                                    work the pair kernel executes: lamp_prof_* counts 2 * allowed_pairs * (d_k + d_v) FLOP per
                                    (sample, head) for it); 0 = unknown */
} lamp_mask;

/* Element strides of the four attention operands, so that one kernel serves both the
 * reference's head-major (h*B, l, d_k) batches (lamp/SubLayers.py:96-98) and the fused
 * [B, l, h*d_k] projections written by lamp_mha_fwd without any permute copy. */
typedef struct lamp_attn_layout {
    int64_t q_b, q_h, q_r;
    int64_t k_b, k_h, k_r;
    int64_t v_b, v_h, v_r;
    int64_t o_b, o_h, o_r;
} lamp_attn_layout;

/* Weights of one MultiHeadAttention (lamp/SubLayers.py:46-74), in nn.Linear's native [out, in] layout. */
typedef struct lamp_mha_weights {
    const float* w_qs; /* [n_head*d_k, d_model] */
    const float* w_ks; /* [n_head*d_k, d_model] */
    const float* w_vs; /* [n_head*d_v, d_model] */
    const float* fc;   /* [d_model, n_head*d_v]; NULL when n_head == 1 (lamp/SubLayers.py:72-74) */
    const float* ln_g; /* [d_model] */
    const float* ln_b; /* [d_model] */
    int32_t n_head;
    int32_t present;   /* 0 = this attention block does not exist (no_dec_self_att) */
} lamp_mha_weights;

/* Weights of one PositionwiseFeedForward (lamp/SubLayers.py:125-131); Conv1d(k=1) weights
 * [out, in, 1] are read as [out, in]. */
typedef struct lamp_ffn_weights {
    const float* w1;   /* [d_inner, d_model] */
    const float* b1;   /* [d_inner] */
    const float* w2;   /* [d_model, d_inner] */
    const float* b2;   /* [d_model] */
    const float* ln_g; /* [d_model] */
    const float* ln_b; /* [d_model] */
} lamp_ffn_weights;

typedef struct lamp_enc_layer {  /* lamp/Layers.py:9-20 */
    lamp_mha_weights slf_attn;   /* dead compute unless attention maps are requested */
    lamp_ffn_weights pos_ffn;
} lamp_enc_layer;

typedef struct lamp_dec_layer {  /* lamp/Layers.py:22-48 */
    lamp_mha_weights enc_attn;
    lamp_ffn_weights pos_ffn1;
    lamp_mha_weights slf_attn;   /* .present == 0 under no_dec_self_att */
    lamp_ffn_weights pos_ffn2;
} lamp_dec_layer;

/* Optional, weights-only: the three weight matrices of one decoder sub-chain -- the attention's output projection `fc`
 * (lamp/SubLayers.py:110) and the feed-forward block's w_1 / w_2 that follows it (lamp/SubLayers.py:133-142) -- rearranged
 * by lamp_pack_weight into the order the fused chain launch streams them.  Like dec0_query below they depend on weights
 * only: a caller may build them once per weight version.  Results are bit-identical with and without them. */
typedef struct lamp_chain_pack {
    const float* fc;  /* lamp_pack_weight(fc [d_model, n_head*d_v], format 0) */
    const float* w1;  /* lamp_pack_weight(w_1 [d_inner, d_model], format 0) */
    const float* w2;  /* lamp_pack_weight(w_2 [d_model, d_inner], format 0) */
    const float* fc4; /* the same three in format 1 (panels of 4 / 8 / 12 rows: batches whose B * n_labels rows would */
    const float* w14; /* leave CUs without a sixteen-row panel); either trio may be NULL */
    const float* w24;
} lamp_chain_pack;

/* The whole graph-encoder / graph-decoder model (lamp/Models.py:18-94).  Host-side struct of
 * device pointers; enc_layers / dec_layers are host arrays. */
typedef struct lamp_model {
    int32_t n_src_vocab, n_position, n_labels;
    int32_t d_model, d_inner, d_k, d_v;
    int32_t n_layers_enc, n_layers_dec;
    int32_t label_mask_flags;   /* lamp_mask.flags of the label graph (LAMP_MASK_SPARSE_ROWS), 0 otherwise */
    const float* src_word_emb;  /* [n_src_vocab, d_model]  encoder.src_word_emb.weight */
    const float* position_enc;  /* [n_position, d_model] or NULL (no_enc_pos_embedding) */
    const float* tgt_word_emb;  /* [n_labels, d_model]     decoder.tgt_word_emb.weight */
    const float* w_out;         /* [n_labels, d_model]     tgt_word_proj.linear.weight */
    const uint8_t* label_mask;  /* [n_labels, n_labels] nonzero = blocked, or NULL ('none') */
    const uint32_t* label_mask_bits; /* optional bit-packed copy of label_mask (LAMP_MASK_BITS_U32 rows of
                                        ceil(n_labels/32) words); used instead of the byte mask when given */
    const int32_t* label_tiles; /* optional active-tile list of label_mask (see lamp_mask.tile_list), row stride
                                   ceil(n_labels/32) + 1; NULL = dense */
    const lamp_enc_layer* enc_layers;
    const lamp_dec_layer* dec_layers;
    /* Optional: decoder layer 0's enc-attention query, tgt_word_emb . w_qs^T  [n_labels, n_head*d_k].  It
     * depends on weights only (the decoder input is the label table for every sample, lamp/Decoders.py:
     * 132-134), so a caller may compute it once per weight version (lamp_linear_fwd) and
     * pass it here; NULL = lamp_forward projects it on every call. */
    const float* dec0_query;
    /* Optional: host array of 2 * n_layers_dec packs -- entry 2 i: layer i's (enc_attn.fc, pos_ffn1), entry 2 i + 1: its
     * (slf_attn.fc, pos_ffn2); entries with NULL members, or a NULL array, leave that sub-chain on the native layouts. */
    const lamp_chain_pack* chain_packs;
    /* Optional (both or neither; ABI 5): encoder layer 0's first FFN matrix folded into the embedding tables.  The token /
     * position gather of lamp/Encoders.py:66,75 is a one-hot product, so relu((Emb[tok] + Pos[p]) W1^T + b1) of
     * lamp/SubLayers.py:135 equals relu(enc0_emb_w1[tok] + enc0_pos_w1[p]) with the weights-only tables
     *   enc0_emb_w1 [n_src_vocab, d_inner] = src_word_emb . W1^T            (+ b1 when position_enc is NULL)
     *   enc0_pos_w1 [n_position, d_inner]  = position_enc . W1^T + b1       (NULL when position_enc is NULL)
     * of enc_layers[0].pos_ffn (a caller builds them once per weight version with lamp_linear_fwd).  lamp_forward then
     * gathers that layer's hidden rows beside the embedded rows and does not launch its first GEMM: a re-association, so
     * results differ from the unfolded route in the last bits (not bit-identical, well inside 1e-4).  NULL = unfolded. */
    const float* enc0_emb_w1;
    const float* enc0_pos_w1;
    int64_t label_mask_allowed; /* lamp_mask.allowed_pairs of the label graph (with LAMP_MASK_SPARSE_ROWS), else 0 */
} lamp_model;

/* Optional extra outputs of lamp_forward (return_attns / int_preds, lamp/Models.py:127-135).
 * Any pointer (or the whole struct) may be NULL.  Attention maps are (n_head*B, lq, lk) with
 * index head*B + b, exactly the reference's layout. */
typedef struct lamp_aux {
    float* const* enc_self_attn; /* n_layers_enc pointers, each (h*B, T, T) */
    float* const* dec_self_attn; /* n_layers_dec pointers, each (h2*B, L, L) */
    float* const* dec_enc_attn;  /* n_layers_dec pointers, each (h*B, L, T) */
    float* const* int_preds;     /* lamp/Models.py:128-133: one (B, L) per intermediate output but the last */
    int32_t n_int_preds;
    int32_t reserved;
} lamp_aux;

/* ---- library ------------------------------------------------------------------------------ */
int lamp_version(void);                 /* LAMP_HIP_ABI_VERSION the library was built with */
const char* lamp_strerror(int status);  /* text for lamp_status / hipError_t values */

/* ---- building blocks ---------------------------------------------------------------------- */

/* nn.Linear / XavierLinear / Conv1d(k=1) (lamp/SubLayers.py:7-13, 91-93, 110, 127-128):
 *   C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) + residual[M,N]      (bias, residual may be NULL)
 * lda/ldw/ldc/ldr are leading dimensions in elements; K, lda, ldw must be multiples of 4. */
int lamp_linear_fwd(const float* A, int64_t M, int32_t K, int64_t lda,
                    const float* W, int32_t N, int64_t ldw, const float* bias,
                    const float* residual, int64_t ldr, int32_t relu,
                    float* C, int64_t ldc, lamp_stream_t stream);

/* nn.LayerNorm over the last dim, biased variance, eps inside the sqrt (lamp/SubLayers.py:68,130).
 * y may alias x.  d must be a multiple of 4. */
int lamp_layernorm_fwd(const float* x, int64_t M, int32_t d, const float* gamma, const float* beta,
                       float eps, float* y, lamp_stream_t stream);

/* ScaledDotProductAttention.forward (lamp/SubLayers.py:27-43), eval mode:
 *   S = (Q K^T) * inv_temperature ; S[blocked] = -inf ; P = softmax_k(S) ; O = P V
 * for B samples x H heads.  `attn` (nullable) receives P as (H*B, lq, lk), index head*B + b.
 * A fully blocked row yields NaN in O and P, as in the reference.  d_k, d_v multiples of 4; up to 128 the fused
 * single-kernel path runs, beyond that a general three-launch path (S = QK^T, masked softmax, PV) that keeps its
 * scores in `attn` -- which must then be given (LAMP_E_WORKSPACE otherwise). */
int lamp_sdpa_fwd(const float* q, const float* k, const float* v, float* out, float* attn,
                  int32_t B, int32_t H, int32_t lq, int32_t lk, int32_t d_k, int32_t d_v,
                  float inv_temperature, const lamp_mask* mask, const lamp_attn_layout* layout,
                  lamp_stream_t stream);

/* Same outputs as lamp_sdpa_fwd with attn != NULL, at the speed of the map-less kernel: one pass over the keys writes
 * the scaled scores into `attn` and every row's log2-sum-exp into `lse` [(H*B) * lq] (caller-provided, also an
 * output), then `attn` is normalised in place by a second launch.  The maps agree with lamp_sdpa_fwd's exact
 * two-pass softmax to rounding; a fully blocked row is NaN in both.  The training forward uses this (the maps are
 * the backward pass's input); a mask's tile list is ignored (every tile is visited). */
int lamp_sdpa_fwd_fast_maps(const float* q, const float* k, const float* v, float* out, float* attn, float* lse,
                            int32_t B, int32_t H, int32_t lq, int32_t lk, int32_t d_k, int32_t d_v,
                            float inv_temperature, const lamp_mask* mask, const lamp_attn_layout* layout,
                            lamp_stream_t stream);

/* MultiHeadAttention.forward (lamp/SubLayers.py:77-121), eval mode:
 *   out = LayerNorm( concat_heads(SDPA(xq Wq^T, xkv Wk^T, xkv Wv^T)) Wfc^T + xq )
 * xq [B, lq, d_model], xkv [B, lk, d_model] (may alias xq), out [B, lq, d_model];
 * attn nullable (n_head*B, lq, lk).  Workspace: lamp_mha_workspace_bytes(). */
size_t lamp_mha_workspace_bytes(int32_t B, int32_t lq, int32_t lk, int32_t d_model, int32_t n_head,
                                int32_t d_k, int32_t d_v);
int lamp_mha_fwd(const float* xq, const float* xkv, int32_t B, int32_t lq, int32_t lk,
                 int32_t d_model, int32_t d_k, int32_t d_v, const lamp_mha_weights* w,
                 const lamp_mask* mask, float* out, float* attn,
                 void* workspace, size_t workspace_bytes, lamp_stream_t stream);

/* PositionwiseFeedForward.forward (lamp/SubLayers.py:133-142), eval mode:
 *   out = LayerNorm( relu(x W1^T + b1) W2^T + b2 + x ),  x/out [M, d_model] (out may alias x).
 * Workspace: M * d_inner floats. */
size_t lamp_ffn_workspace_bytes(int64_t M, int32_t d_model, int32_t d_inner);
int lamp_ffn_fwd(const float* x, int64_t M, int32_t d_model, int32_t d_inner,
                 const lamp_ffn_weights* w, float* out,
                 void* workspace, size_t workspace_bytes, lamp_stream_t stream);

/* GraphEncoder's input stage (lamp/Encoders.py:66,75): out[t,:] = emb[src_seq[t],:] (+ pos[src_pos[t],:]).
 * An index outside its table writes NaN into that row (PyTorch would raise; a kernel cannot). */
int lamp_embed_fwd(const int64_t* src_seq, const int64_t* src_pos, int64_t n_tokens,
                   const float* emb, int32_t n_vocab, const float* pos_table, int32_t n_position,
                   int32_t d_model, float* out, lamp_stream_t stream);

/* Label read-out (lamp/Models.py:124-126): logits[b,i] = <y[b,i,:], w_out[i,:]>, i.e. the diagonal
 * of y . w_out^T without forming the (B, L, L) product. */
int lamp_diag_logits_fwd(const float* y, const float* w_out, int32_t B, int32_t L, int32_t d_model,
                         float* logits, lamp_stream_t stream);

/* Post-processing of the evaluation loop (test.py:49-51), the step right after the hot path:
 *   probs[b,i]  = sigmoid(logits[b,i])                                   (probs nullable)
 *   row_loss[b] = sum_i  max(x,0) - x*z + log1p(exp(-|x|)),  x = logits[b,i], z = targets[b,i]
 * i.e. F.binary_cross_entropy_with_logits summed per row (row_loss and targets nullable together);
 * the 'mean' reduction is sum(row_loss) / (n_rows * L). */
int lamp_sigmoid_bce_fwd(const float* logits, const float* targets, int64_t n_rows, int32_t L,
                         float* probs, float* row_loss, lamp_stream_t stream);

/* Prior label graph, the input of label_mask='prior' (utils/data_loader.py:37-47), built on the device from
 * the train split's label sets in CSR form: label_ids[offsets[s] .. offsets[s+1]) are the 0-based label
 * indices of sample s (target-vocabulary ids minus the 4 special tokens, BOS/EOS stripped).
 *   adj     [L, L] float: 1 where i == j or labels i and j share a sample, else 0  (the reference's matrix)
 *   blocked [L, L] u8, nullable: the decoder self-attention mask derived from it (lamp/Decoders.py:105-113;
 *           no row of adj is empty because of the diagonal), 1 = blocked = (adj == 0).
 * Ids outside [0, L) are skipped (the reference would index out of range); validate on the host. */
int lamp_prior_graph_build(const int64_t* label_ids, const int64_t* offsets, int64_t n_samples, int32_t L,
                           float* adj, uint8_t* blocked, lamp_stream_t stream);

/* Weights-only repack for lamp_model.chain_packs: W [N, K] (leading dimension ldw; nn.Linear / Conv1d(k=1) layout) ->
 * packed [N * K] floats in the order a chain kernel streams them (exact copy of the values, no arithmetic):
 *   format 0: per block of 16 output rows and 32 k the 16x16x4 MFMA fragments lane by lane (N % 16 == 0, K % 32 == 0);
 *   format 1: per block of 64 output rows and 16 k, four k per row and load (N % 64 == 0, K % 16 == 0).
 * 16-byte aligned pointers. */
int lamp_pack_weight(const float* W, int32_t N, int32_t K, int64_t ldw, int32_t format, float* packed, lamp_stream_t stream);

/* ---- backward-pass building blocks (training through train.py:36-48) ------------------- */

/* General strided, batched GEMM:   C_z[m, n] (+)= alpha * sum_k A_z(m, k) * B_z(n, k),   z = (z0, z1),
 *   A_z(m, k) = A[z0*a_batch0 + z1*a_batch1 + m*a_row_stride + k*a_col_stride]   (B likewise with n),
 *   C_z[m, n] = C[z0*c_batch0 + z1*c_batch1 + m*ldc + n].
 * For each operand one of (row_stride, col_stride) must be 1, which covers every product of the backward pass
 * without a transpose copy: dX = dY.W, dW = dY^T.X, dP = dO.V^T, dV = P^T.dO, dQ = dS.K, dK = dS^T.Q.
 * relu_mask (batch 1 only, nullable): result elements whose mask value is not > 0 are zeroed (ReLU backward).
 * accumulate != 0 adds to C.  Workspace (nullable): lamp_gemm_workspace_bytes() lets deep-K / small-output
 * products (weight gradients) split K deterministically. */
typedef struct lamp_gemm_desc {
    const float* A;
    const float* B;
    float* C;
    int32_t M, N, K;
    int32_t batch0, batch1;
    int32_t accumulate;
    int64_t a_row_stride, a_col_stride, a_batch0, a_batch1;
    int64_t b_row_stride, b_col_stride, b_batch0, b_batch1;
    int64_t ldc, c_batch0, c_batch1;
    const float* relu_mask;
    int64_t ld_mask;
    float alpha;
    int32_t reserved;
} lamp_gemm_desc;
size_t lamp_gemm_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t batch);
int lamp_gemm(const lamp_gemm_desc* d, void* workspace, size_t workspace_bytes, lamp_stream_t stream);

/* n independent lamp_gemm products in ONE launch (descs: host array; batch0 = batch1 = 1, relu_mask NULL, the same operand
 * form -- which of A / B is k-contiguous -- in all of them, distinct C).  No K split, no workspace: every 64 x 64 output
 * tile accumulates k in ascending order, so the results are those of lamp_gemm without workspace, whatever the grouping.
 * Meant for the weight gradients of a whole backward pass (train.py:40 `loss.backward()`: dW = dY^T.X of every nn.Linear /
 * Conv1d(k=1) on the path, lamp/SubLayers.py:60-64,129-130), which are off the critical path of the data gradients and fill
 * the chip together where each alone would need a K split.  Order descs deepest K first. */
int lamp_gemm_grouped(const lamp_gemm_desc* descs, int32_t n, lamp_stream_t stream);

/* ---- training-mode sub-layers, ONE call each ------------------------------------------------------------------
 * The launches that lamp_amd/training.py's autograd functions would issue one Python round trip at a time (a reuters
 * training step is ~190 launches and was bound by the issuing thread, not the device), issued by the library.  Same
 * kernels, same order, same bits as the per-launch route.  All buffers are the caller's ([rows, cols] row-major, fp32):
 * "saved" ones must live until the matching backward call, gradients of weights may be NULL = the caller computes them
 * later from the buffers this call leaves behind (lamp_gemm_grouped). */

/* Second stage of a column reduction (LayerNorm-parameter and bias gradients of `loss.backward()`, train.py:40 -- what autograd
 * computes for nn.LayerNorm / Conv1d biases in lamp/SubLayers.py:64,129-131): out[c] = sum over n_partials rows of
 * partial[p * n_total + c], c < n_total, in a fixed order; columns [i * n_seg, (i + 1) * n_seg) go to out[i] (n_total <= 3 n_seg).
 * lamp_ffn_bwd / lamp_mha_bwd describe theirs in such jobs instead of launching them when given a `partials` buffer, so that
 * ONE lamp_reduce_partials_grouped launch can run the jobs of a whole backward pass: 17 tiny dependent launches per reuters
 * step become one.  A job's partials and outputs are the caller's buffers and must stay live until that launch. */
typedef struct lamp_reduce_job {
    const float* partial;
    int64_t n_total, n_seg;
    float* out[3];
    int32_t n_partials;
    int32_t reserved;
} lamp_reduce_job;
int lamp_reduce_partials_grouped(const lamp_reduce_job* jobs, int32_t n, lamp_stream_t stream);

/* PositionwiseFeedForward.forward in train() mode (lamp/SubLayers.py:133-142):
 *   h = relu(x W1^T + b1) [M, d_inner] (saved), o = h W2^T + b2 [M, d_model] (saved), y = LayerNorm(dropout(o) + x). */
int lamp_ffn_train_fwd(const float* x, int64_t M, int32_t d_model, int32_t d_inner, const lamp_ffn_weights* w,
                       float dropout_p, uint32_t seed, float* h, float* o, float* y, lamp_stream_t stream);
/* Its backward.  dx [M, d_model] = gradient of x; d_o [M, d_model] = gradient of o (required iff dropout_p > 0; without
 * dropout it IS dx before the last accumulation, and dW2 must then be requested here); dh [M, d_inner] = gradient of the
 * pre-ReLU hidden; dW1 [d_inner, d_model] = dh^T x and dW2 [d_model, d_inner] = d_o^T h nullable (deferred);
 * db1 [d_inner], db2, dgamma, dbeta [d_model].
 * partials (nullable, lamp_ffn_bwd_partials_bytes): when given, db1 / db2 / dgamma / dbeta are NOT final on return -- jobs[0..1]
 * describe the two reductions that finish them (see lamp_reduce_job). */
size_t lamp_ffn_bwd_workspace_bytes(int64_t M, int32_t d_model, int32_t d_inner);
size_t lamp_ffn_bwd_partials_bytes(int64_t M, int32_t d_model, int32_t d_inner);
int lamp_ffn_bwd(const float* x, const float* h, const float* o, const float* dy, int64_t M, int32_t d_model,
                 int32_t d_inner, const lamp_ffn_weights* w, float dropout_p, uint32_t seed, float* dx, float* d_o, float* dh,
                 float* dW1, float* dW2, float* db1, float* db2, float* dgamma, float* dbeta, void* workspace,
                 size_t workspace_bytes, void* partials, size_t partials_bytes, lamp_reduce_job* jobs, lamp_stream_t stream);

typedef struct lamp_mha_train_desc {
    int32_t B, lq, lk, d_model, n_head, d_k, d_v;   /* d_k, d_v <= 128 */
    float inv_temperature;                          /* 1 / sqrt(d_k) (lamp/SubLayers.py:62) */
    float p_attn, p_out;                            /* dropout of the probabilities (SubLayers.py:40) and of the output (:113) */
    uint32_t seed_attn, seed_out;
} lamp_mha_train_desc;
/* MultiHeadAttention.forward in train() mode (lamp/SubLayers.py:77-121) on xq [B, lq, d_model], xk / xv [B, lk, d_model]
 * (the same pointer for both in every layer of the reference).  Saved: q [B, lq, H*d_k], k, v [B, lk, H*d_*], a [B, lq, H*d_v]
 * (concatenated head outputs, computed from the DROPPED probabilities), P [H*B, lq, lk] (softmax, index head * B + b),
 * o [B*lq, d_model] (fc output; unused when w->fc is NULL).  Pd [H*B, lq, lk] = dropout(P), required iff p_attn > 0 -- what
 * the reference returns as the attention map.  lse: H*B*lq floats of scratch.  y = LayerNorm(dropout(o) + xq). */
int lamp_mha_train_fwd(const lamp_mha_train_desc* c, const lamp_mha_weights* w, const float* xq, const float* xk,
                       const float* xv, const lamp_mask* mask, float* q, float* k, float* v, float* a, float* P, float* Pd,
                       float* lse, float* o, float* y, lamp_stream_t stream);
/* Its backward.  Outputs: dxq [B*lq, d_model]; dxk [B*lk, d_model] (+ the value branch when dxv is NULL, else dxv gets it;
 * dxk == dxq is allowed when lq == lk -- self-attention, xq and xk one tensor: dxq then holds the SUM of the three branches);
 * dgamma, dbeta.  Left behind for deferred weight gradients: d_o [B*lq, d_model] (required iff p_out > 0), dq [B*lq, H*d_k],
 * dk [B*lk, H*d_k], dv [B*lk, H*d_v]; dwq / dwk / dwv [H*d_*, d_model], dfc [d_model, H*d_v] nullable (dfc required when
 * w->fc is set and p_out == 0).  Pd: the forward's dropout(P) (saved; required iff p_attn > 0).  Scratch: da [B*lq, H*d_v]
 * (iff w->fc), dP [H*B, lq, lk].  partials (nullable, lamp_mha_bwd_partials_bytes): when given, dgamma / dbeta are NOT final on
 * return -- job[0] describes the reduction that finishes them (see lamp_reduce_job). */
size_t lamp_mha_bwd_workspace_bytes(const lamp_mha_train_desc* c);
size_t lamp_mha_bwd_partials_bytes(const lamp_mha_train_desc* c);
int lamp_mha_bwd(const lamp_mha_train_desc* c, const lamp_mha_weights* w, const float* xq, const float* xk, const float* xv,
                 const float* q, const float* k, const float* v, const float* a, const float* P, const float* Pd,
                 const float* o, const float* dy, float* dxq, float* d_o, float* da, float* dP, float* dq, float* dk, float* dv,
                 float* dxk, float* dxv, float* dgamma, float* dbeta, float* dwq, float* dwk, float* dwv, float* dfc,
                 void* workspace, size_t workspace_bytes, void* partials, size_t partials_bytes, lamp_reduce_job* job,
                 lamp_stream_t stream);

/* y = LayerNorm(dropout(x) + residual[row % residual_rows]) (residual nullable; residual_rows 0 = one residual row
 * per x row): the dropout, add & norm that closes every sub-layer (lamp/SubLayers.py:113-115,138-140), with neither
 * the dropped tensor nor the sum ever stored.  dropout_p = 0: plain add & norm; otherwise lamp_dropout's counter-based
 * mask for `seed`, element index = row * d + column. */
int lamp_layernorm_residual_fwd(const float* x, const float* residual, int64_t residual_rows, int64_t M, int32_t d,
                                const float* gamma, const float* beta, float eps, float dropout_p, uint32_t seed,
                                float* y, lamp_stream_t stream);

/* Backward of the above; z = dropout(x) + residual is recomputed from the same inputs.  Given dy it returns
 *   dz [M, d]      gradient of z, i.e. of the residual branch,
 *   dx [M, d]      gradient of x = dropout(dz) with the same mask (written only when dropout_p > 0; with p = 0 the
 *                  gradient of x is dz itself and dx may be NULL),
 *   dgamma, dbeta  [d]: sum_rows dy * zhat, sum_rows dy,
 *   dbias [d]      (nullable): sum_rows of the gradient of x -- the bias gradient of the linear map that produced x.
 * d <= 4096.  Row sums are two-stage in a fixed order (deterministic). */
size_t lamp_layernorm_bwd_workspace_bytes(int64_t M, int32_t d);
int lamp_layernorm_bwd(const float* x, const float* residual, int64_t residual_rows, int64_t M, int32_t d,
                       const float* gamma, float eps, float dropout_p, uint32_t seed, const float* dy, float* dz,
                       float* dx, float* dgamma, float* dbeta, float* dbias, void* workspace, size_t workspace_bytes,
                       lamp_stream_t stream);

/* out[n] = sum_m x[m*ldx + n]: bias gradients, and the label-table gradient (sum over the batch). */
size_t lamp_colsum_workspace_bytes(int64_t M, int64_t N);
int lamp_colsum(const float* x, int64_t M, int64_t N, int64_t ldx, float* out, void* workspace,
                size_t workspace_bytes, lamp_stream_t stream);

/* nn.Dropout in training mode (lamp/SubLayers.py:40,113,138): y[e] = keep(e, seed) ? x[e] / (1 - p) : 0 with a
 * counter-based generator -- keep(e, seed) = mix32(e, seed) >= p * 2^32 -- so the mask is a pure function of
 * (element index, seed) and is never stored: calling it again on the gradient with the same seed IS the backward
 * pass.  y may alias x.  (The stream of random numbers necessarily differs from torch's Philox stream.) */
int lamp_dropout(const float* x, int64_t n, float p, uint32_t seed, float* y, lamp_stream_t stream);

/* Softmax backward per row of length lk: dS = scale * P * (dP - sum_k P * dP); dS may alias dP. */
int lamp_softmax_bwd(const float* P, const float* dP, int64_t rows, int32_t lk, float scale, float* dS,
                     lamp_stream_t stream);

/* Backward of lamp_diag_logits_fwd: dy[b,i,:] = dlogits[b,i] * w_out[i,:], dw[i,:] = sum_b dlogits[b,i] * y[b,i,:]. */
int lamp_diag_logits_bwd(const float* y, const float* w_out, const float* dlogits, int32_t B, int32_t L,
                         int32_t d_model, float* dy, float* dw, lamp_stream_t stream);

/* Backward of the embedding gather: d_emb[src_seq[t], :] += dout[t, :] (atomic adds; rows of pad_idx are skipped as
 * nn.Embedding(padding_idx) does; pass -1 for none).  d_emb must be initialised by the caller. */
int lamp_embed_bwd(const int64_t* src_seq, int64_t n_tokens, const float* dout, int32_t d_model, int32_t n_vocab,
                   int64_t pad_idx, float* d_emb, lamp_stream_t stream);

/* ---- the whole hot path ------------------------------------------------------------------- */

/* Bytes of workspace lamp_forward needs to process `micro_batch` samples of padded length T at a
 * time.  lamp_forward splits B into micro-batches of floor(workspace_bytes / bytes(1)) samples. */
size_t lamp_forward_workspace_bytes(const lamp_model* m, int32_t micro_batch, int32_t T,
                                    int32_t want_attn);

/* LAMP.forward (lamp/Models.py:110-137) for encoder='graph', decoder='graph', eval mode:
 *   src_seq, src_pos int64 [B, T]  ->  logits [B, n_labels], enc_output [B, T, d_model].
 * The encoder self-attention, whose output the reference discards (lamp/Layers.py:16-18), is
 * computed only when aux->enc_self_attn is given.
 * Ragged batches (utils/data_loader.py:261-279 pads to the longest document): PAD positions are not computed.  The call's
 * first kernel counts each sample's extents on the device; the encoder runs on the packed non-PAD rows plus ONE shared PAD
 * row (every PAD position of lamp/Encoders.py:64-79 holds the same row-wise result), and the enc-dec attention stops at
 * each sample's last non-PAD key (keys past it are exactly masked, lamp/utils.py:26-34).  enc_output is still the padded
 * [B, T, d_model] tensor the reference returns, PAD positions included, and a sample's outputs do not depend on T, on
 * B or on the micro-batch split, bit for bit. */
int lamp_forward(const lamp_model* m, const int64_t* src_seq, const int64_t* src_pos,
                 int32_t B, int32_t T, float* logits, float* enc_output, const lamp_aux* aux,
                 void* workspace, size_t workspace_bytes, lamp_stream_t stream);

/* ---- per-kernel timing (HIP events on the launch stream; used by bench.py's roofline) ------ */
enum lamp_kernel_class {
    LAMP_K_EMBED = 0, LAMP_K_GEMM = 1, LAMP_K_ATTN = 2, LAMP_K_LAYERNORM = 3, LAMP_K_DIAG = 4,
    LAMP_K_COUNT = 5
};
int lamp_prof_enable(int32_t on);  /* bracket every launch with a hipEvent pair while on */
int lamp_prof_reset(void);
/* Synchronises the recorded events and returns, per class: launches, summed milliseconds, and the
 * summed algorithmic FLOPs and bytes of those launches (as the launcher computed them). */
int lamp_prof_read(int32_t kernel_class, int64_t* launches, double* total_ms, double* flops,
                   double* bytes);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* LAMP_HIP_H */
