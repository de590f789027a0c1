// C-ABI entry points of liblamp_hip.so (see include/lamp_hip.h) and the whole-forward launcher.
//
// Host-side only: argument validation, workspace carving and the launch sequence.  No device
// memory is allocated here, no pointer is retained, nothing synchronises (except lamp_prof_read).
#include <mutex>
#include <vector>

#include "lamp_kernels.h"

namespace lamp {

// ------------------------------------------------------------------ profiling
namespace {
struct ProfRec {
    int cls;
    double flops, bytes;
    hipEvent_t e0, e1;
};
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof;          // records in use
std::vector<hipEvent_t> g_event_pool; // recycled events
constexpr size_t PROF_MAX = 1 << 16;

hipEvent_t take_event() {
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

ProfScope::ProfScope(int kernel_class, double flops, double bytes, hipStream_t stream) : idx(-1), s(stream) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on || g_prof.size() >= PROF_MAX) return;
    ProfRec r{kernel_class, flops, bytes, take_event(), take_event()};
    if (!r.e0 || !r.e1) return;
    (void)hipEventRecord(r.e0, s);
    idx = int(g_prof.size());
    g_prof.push_back(r);
}

ProfScope::~ProfScope() {
    if (idx < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (idx < int(g_prof.size())) (void)hipEventRecord(g_prof[idx].e1, s);
}

// ------------------------------------------------------------------ helpers
#define LAMP_CK(expr)            \
    do {                         \
        int _e = (expr);         \
        if (_e != 0) return _e;  \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Carver {
    char* base;
    size_t off, cap;
    bool ok;
    Carver(void* p, size_t bytes) : base(static_cast<char*>(p)), off(0), cap(bytes), ok(true) {}
    float* take(size_t n_floats) {
        const size_t b = align_up(n_floats * sizeof(float), 256);
        if (off + b > cap) {
            ok = false;
            return nullptr;
        }
        float* r = reinterpret_cast<float*>(base + off);
        off += b;
        return r;
    }
};

// The residual of a GEMM as a gather from the embedding tables (GemmParams::rg_tok): the encoder's first layer, whose input rows
// are then never written (EmbedFold::row_tok).
struct ResGather {
    const int* tok;
    const int* pos;
    const float* emb;
    const float* pos_table;
};

static int linear(const float* A, int64_t M, int K, int64_t lda, const float* const* W, int nseg, int N,
                  int64_t ldw, const float* const* bias, const float* R, int64_t ldr, int relu,
                  float* const* C, int64_t ldc, hipStream_t s, const int* m_dev = nullptr,
                  const float* A_dense = nullptr, const ResGather* rg = nullptr) {
    GemmParams p{};
    p.A = A; p.lda = lda; p.M = M; p.K = K; p.N = N; p.nseg = nseg; p.ldw = ldw; p.ldc = ldc;
    p.R = R; p.ldr = ldr; p.relu = relu; p.m_dev = m_dev; p.A_dense = A_dense;
    if (rg) {
        p.R = nullptr;
        p.rg_tok = rg->tok; p.rg_pos = rg->pos; p.rg_emb = rg->emb; p.rg_pos_table = rg->pos_table;
    }
    for (int i = 0; i < nseg; ++i) {
        p.W[i] = W[i];
        p.bias[i] = bias ? bias[i] : nullptr;
        p.C[i] = C[i];
    }
    return launch_gemm(p, s);
}

static int check_mask(const lamp_mask* m) {
    if (!m) return 0;
    if (m->kind != LAMP_MASK_NONE && m->kind != LAMP_MASK_U8 && m->kind != LAMP_MASK_KEY_TOKENS_I64 &&
        m->kind != LAMP_MASK_BITS_U32)
        return LAMP_E_UNSUPPORTED;
    if (m->kind != LAMP_MASK_NONE && !m->ptr) return LAMP_E_NULL;
    return 0;
}

// Scratch of one MultiHeadAttention call on B samples.
struct MhaScratch {
    float *Q, *K, *V, *A;
    float* S = nullptr;    // (h*B, lq, lk) score scratch, only for d_k or d_v > 128 (attention_general.hip)
    float* lse = nullptr;  // [h][B][lq] row log-sum-exp: lets requested attention maps come from the single-pass kernel
    int* plan_ints = nullptr;  // plan_int_count(B, lk) ints: the SeqPlan of a key-token mask when the caller did not bring one
};
static inline size_t plan_int_count(int64_t B, int T) { return size_t(3) * size_t(B) + 3 + size_t(B) * size_t((T + 31) / 32); }
static inline SeqPlan plan_from(int* ints, int64_t B, int T) {
    return SeqPlan{ints, ints + B, ints + 2 * B, ints + 3 * B + 1, reinterpret_cast<unsigned*>(ints + 3 * B + 3), (T + 31) / 32, nullptr};
}
static inline bool wide_heads(int dk, int dv) { return dk > 128 || dv > 128; }

// The position-wise feed-forward block that follows an attention block in a decoder layer (lamp/Layers.py:35-36, :40-45):
// handed to mha_core so that the attention's output projection, its LayerNorm and this block run as ONE launch
// (chain.hip) when the shape allows.  `done` tells the caller whether that happened.
struct FfnTail {
    const lamp_ffn_weights* ffn;
    int dff;
    const float* w_out;   // fused read-out of the last decoder block (nullable), as in ffn_core
    int n_labels;
    float* logits;
    bool done;
    const lamp_chain_pack* pk;   // nullable: weights-only packed copies of (fc, w1, w2) for the chain launch
};

// MultiHeadAttention.forward (lamp/SubLayers.py:77-121).  `xq_shared`: xq is ONE [lq, d] block used
// by every sample (decoder layer 0: the label embeddings) -- its projection is then
// computed once, and the residual is read modulo lq.  `out` may alias xq unless xq_shared.
// `out == nullptr` computes the attention map only (the reference's dead encoder self-attention).
static int mha_core(const float* xq, bool xq_shared, const float* xkv, int B, int lq, int lk, int d, int dk,
                    int dv, const lamp_mha_weights& w, const lamp_mask* mask, float* out, float* attn,
                    const MhaScratch& sc, hipStream_t s, bool kv_ready = false,
                    const float* q_ready = nullptr, int P_batch = 0, int P_b0 = 0, const SeqPlan* keys = nullptr,
                    bool keys_packed = false, const float* xkv_dense = nullptr, FfnTail* tail = nullptr) {
    // keys: per-sample key extents of a key-token mask (ragged batches); keys_packed: xkv holds the packed token rows
    // (keys->rows[0] of them, counted on the device) instead of [B, lk, d]
    const int h = w.n_head;
    if (h < 1 || dk < 1 || dv < 1) return LAMP_E_DIMS;
    if (!w.w_qs || !w.w_ks || !w.w_vs || (out && (!w.ln_g || !w.ln_b))) return LAMP_E_NULL;
    if (h == 1 && out && dv != d) return LAMP_E_DIMS;  // no fc: O is added to the residual directly
    if (h > 1 && out && !w.fc) return LAMP_E_NULL;
    const int hdk = h * dk, hdv = h * dv;
    const int64_t Mq = xq_shared ? lq : int64_t(B) * lq;
    const int64_t Mk = int64_t(B) * lk;
    const bool self = (xq == xkv) && !xq_shared && lq == lk;
    const bool need_v = out != nullptr;

    if (self && hdk == hdv && need_v) {
        float* C[3] = {sc.Q, sc.K, sc.V};
        const float* W[3] = {w.w_qs, w.w_ks, w.w_vs};
        LAMP_CK(linear(xq, Mq, d, d, W, 3, hdk, d, nullptr, nullptr, 0, 0, C, hdk, s));
    } else {
        if (!q_ready) {
            const float* W[1] = {w.w_qs};
            float* C[1] = {sc.Q};
            LAMP_CK(linear(xq, Mq, d, d, W, 1, hdk, d, nullptr, nullptr, 0, 0, C, hdk, s));
        }
        const int* mk_dev = keys_packed ? keys->rows : nullptr;
        const float* xd = keys_packed ? xkv_dense : nullptr;  // the padded rows, read instead when nothing was skipped
        if (kv_ready) {
            // sc.K / sc.V were projected earlier (every decoder layer's K/V in one launch, see forward_range)
        } else if (need_v && hdk == hdv) {
            const float* W[2] = {w.w_ks, w.w_vs};
            float* C[2] = {sc.K, sc.V};
            LAMP_CK(linear(xkv, Mk, d, d, W, 2, hdk, d, nullptr, nullptr, 0, 0, C, hdk, s, mk_dev, xd));
        } else {
            const float* Wk[1] = {w.w_ks};
            float* Ck[1] = {sc.K};
            LAMP_CK(linear(xkv, Mk, d, d, Wk, 1, hdk, d, nullptr, nullptr, 0, 0, Ck, hdk, s, mk_dev, xd));
            if (need_v) {
                const float* Wv[1] = {w.w_vs};
                float* Cv[1] = {sc.V};
                LAMP_CK(linear(xkv, Mk, d, d, Wv, 1, hdv, d, nullptr, nullptr, 0, 0, Cv, hdv, s, mk_dev, xd));
            }
        }
    }

    // A key-token mask without a plan (lamp_mha_fwd on its own): count each sample's keys here, padded layout, so that
    // the module-by-module route takes the same per-sample key split as lamp_forward -- same bits.
    SeqPlan local_plan{};
    if (!keys && mask && mask->kind == LAMP_MASK_KEY_TOKENS_I64 && sc.plan_ints && !wide_heads(dk, dv)) {
        local_plan = plan_from(sc.plan_ints, B, lk);
        LAMP_CK(launch_seq_plan(static_cast<const int64_t*>(mask->ptr), nullptr, B, lk, mask->stride_b, false, local_plan, s));
        keys = &local_plan;
    }

    AttnParams a{};
    a.Q = q_ready ? q_ready : sc.Q; a.K = sc.K; a.V = need_v ? sc.V : nullptr; a.O = need_v ? sc.A : nullptr; a.P = attn;
    a.scratch = sc.S;
    // maps + output: single-pass kernel (same O bits as without maps), scores normalised in place afterwards
    if (attn && need_v && sc.lse) a.lse = sc.lse;
    a.B = B; a.H = h; a.lq = lq; a.lk = lk; a.dk = dk; a.dv = dv;
    a.P_batch = P_batch > 0 ? P_batch : B; a.P_b0 = P_b0;
    a.lay.q_b = xq_shared ? 0 : int64_t(lq) * hdk; a.lay.q_h = dk; a.lay.q_r = hdk;
    a.lay.k_b = int64_t(lk) * hdk; a.lay.k_h = dk; a.lay.k_r = hdk;
    a.lay.v_b = int64_t(lk) * hdv; a.lay.v_h = dv; a.lay.v_r = hdv;
    a.lay.o_b = int64_t(lq) * hdv; a.lay.o_h = dv; a.lay.o_r = hdv;
    a.scale_log2e = float(1.4426950408889634 / sqrt(double(dk)));
    a.mask_kind = mask ? mask->kind : LAMP_MASK_NONE;
    a.mask = mask ? mask->ptr : nullptr;
    a.m_sb = mask ? mask->stride_b : 0;
    a.m_sq = mask ? mask->stride_q : 0;
    a.tiles = (mask && !attn) ? mask->tile_list : nullptr;
    a.tiles_stride = mask ? mask->tile_list_stride : 0;
    a.sparse_rows = mask && (mask->flags & LAMP_MASK_SPARSE_ROWS) != 0;
    a.allowed_pairs = mask ? mask->allowed_pairs : 0;
    if (keys && mask && mask->kind == LAMP_MASK_KEY_TOKENS_I64) {
        a.kv_len = keys->klen;
        a.kv_off = keys->off;
        if (!wide_heads(dk, dv)) {   // the plan's bit-packed copy of the same mask: one word per 32-key tile
            a.mask_kind = LAMP_MASK_BITS_U32;
            a.mask = keys->padbits;
            a.m_sb = keys->words;
            a.m_sq = 0;
        }
    }
    LAMP_CK(launch_attn(a, s));
    if (!out) return 0;

    const int64_t M = int64_t(B) * lq;
    const int64_t r_mod = xq_shared ? lq : 0;
    if (tail) tail->done = false;
    if (h > 1 && tail && tail->ffn && chain_applies(M, d, hdv, tail->dff, true, tail->pk, xq_shared, tail->w_out != nullptr)) {
        // fc (+ residual) -> LayerNorm -> W1 -> W2 (+ residual) -> LayerNorm in one launch over 16-row panels (same bits)
        tail->done = true;
        return launch_chain(sc.A, hdv, hdv, xq, r_mod, M, d, w.fc, w.ln_g, w.ln_b, tail->ffn, tail->dff,
                            tail->w_out ? nullptr : out, tail->w_out, tail->n_labels, tail->logits, s, tail->pk);
    }
    if (h > 1 && tail && tail->ffn && tail->pk && M > 6144 && (!tail->w_out || tail->n_labels == lq)) {
        // Just past one chain launch's reach (6145-12288 rows: batch 69-136 at 90 labels) the tail is still faster as TWO chain
        // launches over halves of the batch -- whole samples, so that row % lq of the shared residual / read-out rows stays the
        // group-local row -- when each half fills the CUs with 24-row panels: batch 128 = 2 x 5760 rows, +6 % whole-forward
        // (48.9 k against 46.1 k samples/s).  With smaller halves (batch 96: 2 x 4320 rows) it ties, and from ~190 samples on the
        // five launches have enough tiles per GEMM to win (profiles/r05_batch_sweep.txt): both keep the separate launches.
        const int per = (B + 1) / 2, last = B - per;
        if (int64_t(per) * lq <= 6144 && int64_t(last) * lq > 4608 &&
            chain_applies(int64_t(per) * lq, d, hdv, tail->dff, true, tail->pk, xq_shared, tail->w_out != nullptr) &&
            chain_applies(int64_t(last) * lq, d, hdv, tail->dff, true, tail->pk, xq_shared, tail->w_out != nullptr)) {
            for (int g = 0; g < 2; ++g) {
                const int64_t r0 = int64_t(g) * per * lq, rows = int64_t(g == 0 ? per : last) * lq;
                LAMP_CK(launch_chain(sc.A + r0 * hdv, hdv, hdv, xq_shared ? xq : xq + r0 * d, r_mod, rows, d, w.fc, w.ln_g, w.ln_b,
                                     tail->ffn, tail->dff, tail->w_out ? nullptr : out + r0 * d, tail->w_out, tail->n_labels,
                                     tail->logits ? tail->logits + r0 : nullptr, s, tail->pk));
            }
            tail->done = true;
            return 0;
        }
    }
    if (h > 1) {
        const float* W[1] = {w.fc};
        float* C[1] = {out};
        if (xq_shared) {
            // residual = the shared [lq, d] block: added (row modulo lq) by the LayerNorm kernel
            LAMP_CK(linear(sc.A, M, hdv, hdv, W, 1, d, hdv, nullptr, nullptr, 0, 0, C, d, s));
            return launch_layernorm(out, M, d, w.ln_g, w.ln_b, 1e-5f, xq, r_mod, out, s);
        }
        LAMP_CK(linear(sc.A, M, hdv, hdv, W, 1, d, hdv, nullptr, xq, d, 0, C, d, s));
        return launch_layernorm(out, M, d, w.ln_g, w.ln_b, 1e-5f, nullptr, 0, out, s);
    }
    return launch_layernorm(sc.A, M, d, w.ln_g, w.ln_b, 1e-5f, xq, r_mod, out, s);
}

// K and V projections of one attention block on their own (same launches mha_core would issue).
static int project_kv(const float* xkv, int64_t Mk, int d, int dk, int dv, const lamp_mha_weights& w, float* K,
                      float* V, hipStream_t s, const int* m_dev = nullptr, const float* A_dense = nullptr) {
    const int hdk = w.n_head * dk, hdv = w.n_head * dv;
    if (hdk == hdv) {
        const float* W[2] = {w.w_ks, w.w_vs};
        float* C[2] = {K, V};
        return linear(xkv, Mk, d, d, W, 2, hdk, d, nullptr, nullptr, 0, 0, C, hdk, s, m_dev, A_dense);
    }
    const float* Wk[1] = {w.w_ks};
    float* Ck[1] = {K};
    LAMP_CK(linear(xkv, Mk, d, d, Wk, 1, hdk, d, nullptr, nullptr, 0, 0, Ck, hdk, s, m_dev, A_dense));
    const float* Wv[1] = {w.w_vs};
    float* Cv[1] = {V};
    return linear(xkv, Mk, d, d, Wv, 1, hdv, d, nullptr, nullptr, 0, 0, Cv, hdv, s, m_dev, A_dense);
}

// K and V projections of the first n decoder layers' enc-attention from the same encoder output: ONE launch when the
// 2n weight matrices fit the GEMM's segment list (n <= 2) -- 4 x more tiles per launch than layer by layer.
static int project_kv_layers(const float* x, int64_t Me, int d, int dk, int dv, const lamp_dec_layer* layers, int n,
                             float* const* K, float* const* V, hipStream_t s, const int* m_dev = nullptr,
                             const float* A_dense = nullptr) {
    const int h = layers[0].enc_attn.n_head;
    bool uniform = 2 * n <= GEMM_MAX_SEG && h * dk == h * dv;
    for (int i = 1; i < n && uniform; ++i) uniform = layers[i].enc_attn.n_head == h;
    if (uniform) {
        const float* W[GEMM_MAX_SEG];
        float* C[GEMM_MAX_SEG];
        for (int i = 0; i < n; ++i) {
            if (!layers[i].enc_attn.w_ks || !layers[i].enc_attn.w_vs) return LAMP_E_NULL;
            W[2 * i] = layers[i].enc_attn.w_ks;
            W[2 * i + 1] = layers[i].enc_attn.w_vs;
            C[2 * i] = K[i];
            C[2 * i + 1] = V[i];
        }
        return linear(x, Me, d, d, W, 2 * n, h * dk, d, nullptr, nullptr, 0, 0, C, h * dk, s, m_dev, A_dense);
    }
    for (int i = 0; i < n; ++i) LAMP_CK(project_kv(x, Me, d, dk, dv, layers[i].enc_attn, K[i], V[i], s, m_dev, A_dense));
    return 0;
}

// PositionwiseFeedForward.forward (lamp/SubLayers.py:133-142); out may alias x.
// Packed encoder rows (ragged batches): `rows_dev` = the live row count in device memory (M is the upper bound the
// launches are sized for); with `scatter` this is the LAST encoder layer, whose LayerNorm also writes the padded
// [nb, T, d] encoder output `y_flat` (see layernorm_kernel<.., RG = 2>).
// `hidden_ready`: relu(x W1^T + b1) is already in `hidden` (the encoder's first layer with W1 folded into the embedding
// tables, pointwise.hip: gather_row) -- the first GEMM is not launched.
static int ffn_core(const float* x, int64_t M, int d, int dff, const lamp_ffn_weights& w, float* out,
                    float* hidden, hipStream_t s, const float* w_out = nullptr, int n_labels = 0,
                    float* logits = nullptr, const int* rows_dev = nullptr, const SeqPlan* scatter = nullptr,
                    int nb = 0, int T = 0, float* y_flat = nullptr, bool hidden_ready = false, const ResGather* rg = nullptr) {
    if (!w.w1 || !w.b1 || !w.w2 || !w.b2 || !w.ln_g || !w.ln_b) return LAMP_E_NULL;
    if (rg && !hidden_ready) return LAMP_E_UNSUPPORTED;   // x itself does not exist then: only the residual may ask for it
    if (!hidden_ready) {
        const float* W[1] = {w.w1};
        const float* b[1] = {w.b1};
        float* C[1] = {hidden};
        LAMP_CK(linear(x, M, d, d, W, 1, dff, d, b, nullptr, 0, 1, C, dff, s, rows_dev));
    }
    {
        const float* W[1] = {w.w2};
        const float* b[1] = {w.b2};
        float* C[1] = {out};
        LAMP_CK(linear(hidden, M, dff, dff, W, 1, d, dff, b, x, d, 0, C, d, s, rows_dev, nullptr, rg));
    }
    if (scatter)
        return launch_layernorm(out, int64_t(nb) * T, d, w.ln_g, w.ln_b, 1e-5f, nullptr, 0, out, s, nullptr, 0, nullptr,
                                nullptr, nullptr, scatter, T, y_flat);
    if (rows_dev)
        return launch_layernorm(out, M, d, w.ln_g, w.ln_b, 1e-5f, nullptr, 0, out, s, nullptr, 0, nullptr, nullptr, rows_dev);
    // with w_out: the final decoder LayerNorm also produces the logits and its output row is not stored
    return launch_layernorm(out, M, d, w.ln_g, w.ln_b, 1e-5f, nullptr, 0, w_out ? nullptr : out, s, w_out, n_labels,
                            logits);
}

static size_t mha_ws_floats(int64_t B, int64_t lq, int64_t lk, int hdk, int hdv, int64_t score_floats = 0) {
    // Q, K, V, A (+ the score scratch of wide heads) -- each rounded to 256 bytes by the carver
    auto r = [](size_t n) { return align_up(n * sizeof(float), 256) / sizeof(float); };
    return r(size_t(B) * lq * hdk) + r(size_t(B) * lk * hdk) + r(size_t(B) * lk * hdv) + r(size_t(B) * lq * hdv) +
           (score_floats ? r(size_t(score_floats)) : 0);
}

}  // namespace lamp

using namespace lamp;

namespace {
int sdpa_impl(const float* q, const float* k, const float* v, float* out, float* attn, float* lse, int32_t B, int32_t H,
              int32_t lq, int32_t lk, int32_t d_k, int32_t d_v, float inv_temperature, const lamp_mask* mask,
                  const lamp_attn_layout* layout, lamp_stream_t stream) {
    if (!layout) return LAMP_E_NULL;
    LAMP_CK(check_mask(mask));
    AttnParams a{};
    a.Q = q; a.K = k; a.V = v; a.O = out; a.P = attn; a.lse = lse;
    a.B = B; a.H = H; a.lq = lq; a.lk = lk; a.dk = d_k; a.dv = d_v;
    a.P_batch = B; a.P_b0 = 0;
    a.lay = *layout;
    a.scale_log2e = float(double(inv_temperature) * 1.4426950408889634);
    a.mask_kind = mask ? mask->kind : LAMP_MASK_NONE;
    a.mask = mask ? mask->ptr : nullptr;
    a.m_sb = mask ? mask->stride_b : 0;
    a.m_sq = mask ? mask->stride_q : 0;
    a.tiles = (mask && !attn) ? mask->tile_list : nullptr;
    a.tiles_stride = mask ? mask->tile_list_stride : 0;
    a.sparse_rows = mask && (mask->flags & LAMP_MASK_SPARSE_ROWS) != 0;
    a.allowed_pairs = mask ? mask->allowed_pairs : 0;
    return launch_attn(a, hipStream_t(stream));
}
}  // namespace

// ================================================================== C ABI
extern "C" {

int lamp_version(void) { return LAMP_HIP_ABI_VERSION; }

const char* lamp_strerror(int status) {
    switch (status) {
        case LAMP_OK: return "ok";
        case LAMP_E_DIMS: return "lamp: non-positive or inconsistent dimensions";
        case LAMP_E_ALIGN: return "lamp: pointer or leading dimension not 16-byte aligned";
        case LAMP_E_WORKSPACE: return "lamp: workspace too small";
        case LAMP_E_UNSUPPORTED: return "lamp: configuration not supported by this build";
        case LAMP_E_NULL: return "lamp: required pointer is NULL";
        default: break;
    }
    if (status > 0) return hipGetErrorString(hipError_t(status));
    return "lamp: unknown status";
}

int lamp_linear_fwd(const float* A, int64_t M, int32_t K, int64_t lda, const float* W, int32_t N, int64_t ldw,
                    const float* bias, const float* residual, int64_t ldr, int32_t relu, float* C, int64_t ldc,
                    lamp_stream_t stream) {
    const float* Ws[1] = {W};
    const float* bs[1] = {bias};
    float* Cs[1] = {C};
    if (lda < K || ldw < K || ldc < N || (residual && ldr < N)) return LAMP_E_DIMS;
    return linear(A, M, K, lda, Ws, 1, N, ldw, bs, residual, ldr, relu, Cs, ldc, hipStream_t(stream));
}

int lamp_layernorm_fwd(const float* x, int64_t M, int32_t d, const float* gamma, const float* beta, float eps,
                       float* y, lamp_stream_t stream) {
    return launch_layernorm(x, M, d, gamma, beta, eps, nullptr, 0, y, hipStream_t(stream));
}

int lamp_sdpa_fwd(const float* q, const float* k, const float* v, float* out, float* attn, int32_t B, int32_t H,
                  int32_t lq, int32_t lk, int32_t d_k, int32_t d_v, float inv_temperature, const lamp_mask* mask,
                  const lamp_attn_layout* layout, lamp_stream_t stream) {
    return sdpa_impl(q, k, v, out, attn, nullptr, B, H, lq, lk, d_k, d_v, inv_temperature, mask, layout, stream);
}

int lamp_sdpa_fwd_fast_maps(const float* q, const float* k, const float* v, float* out, float* attn, float* lse,
                            int32_t B, int32_t H, int32_t lq, int32_t lk, int32_t d_k, int32_t d_v,
                            float inv_temperature, const lamp_mask* mask, const lamp_attn_layout* layout,
                            lamp_stream_t stream) {
    if (!attn || !lse || !out || !v) return LAMP_E_NULL;
    return sdpa_impl(q, k, v, out, attn, lse, B, H, lq, lk, d_k, d_v, inv_temperature, mask, layout, stream);
}

size_t lamp_mha_workspace_bytes(int32_t B, int32_t lq, int32_t lk, int32_t d_model, int32_t n_head, int32_t d_k,
                                int32_t d_v) {
    (void)d_model;
    if (B <= 0 || lq <= 0 || lk <= 0 || n_head <= 0 || d_k <= 0 || d_v <= 0) return 0;
    return (mha_ws_floats(B, lq, lk, n_head * d_k, n_head * d_v,
                          wide_heads(d_k, d_v) ? int64_t(n_head) * B * lq * lk : 0) +
            align_up(size_t(n_head) * B * lq * sizeof(float), 256) / sizeof(float) +
            align_up(plan_int_count(B, lk) * sizeof(int), 256) / sizeof(float)) * sizeof(float);
}

int lamp_mha_fwd(const float* xq, const float* xkv, int32_t B, int32_t lq, int32_t lk, int32_t d_model,
                 int32_t d_k, int32_t d_v, const lamp_mha_weights* w, const lamp_mask* mask, float* out,
                 float* attn, void* workspace, size_t workspace_bytes, lamp_stream_t stream) {
    if (!xq || !xkv || !w || !out || !workspace) return LAMP_E_NULL;
    if (B <= 0 || lq <= 0 || lk <= 0 || d_model <= 0 || d_k <= 0 || d_v <= 0 || w->n_head <= 0) return LAMP_E_DIMS;
    if (d_model & 3) return LAMP_E_UNSUPPORTED;
    LAMP_CK(check_mask(mask));
    const int hdk = w->n_head * d_k, hdv = w->n_head * d_v;
    Carver c(workspace, workspace_bytes);
    MhaScratch sc;
    sc.Q = c.take(size_t(B) * lq * hdk);
    sc.K = c.take(size_t(B) * lk * hdk);
    sc.V = c.take(size_t(B) * lk * hdv);
    sc.A = c.take(size_t(B) * lq * hdv);
    if (wide_heads(d_k, d_v)) sc.S = c.take(size_t(w->n_head) * B * lq * lk);
    sc.lse = c.take(size_t(w->n_head) * B * lq);
    sc.plan_ints = reinterpret_cast<int*>(c.take(plan_int_count(B, lk)));
    if (!c.ok) return LAMP_E_WORKSPACE;
    return mha_core(xq, false, xkv, B, lq, lk, d_model, d_k, d_v, *w, mask, out, attn, sc, hipStream_t(stream));
}

size_t lamp_ffn_workspace_bytes(int64_t M, int32_t d_model, int32_t d_inner) {
    (void)d_model;
    if (M <= 0 || d_inner <= 0) return 0;
    return align_up(size_t(M) * d_inner * sizeof(float), 256);
}

int lamp_ffn_fwd(const float* x, int64_t M, int32_t d_model, int32_t d_inner, const lamp_ffn_weights* w,
                 float* out, void* workspace, size_t workspace_bytes, lamp_stream_t stream) {
    if (!x || !w || !out || !workspace) return LAMP_E_NULL;
    if (M <= 0 || d_model <= 0 || d_inner <= 0) return LAMP_E_DIMS;
    if ((d_model & 3) || (d_inner & 3)) return LAMP_E_UNSUPPORTED;
    if (workspace_bytes < size_t(M) * d_inner * sizeof(float)) return LAMP_E_WORKSPACE;
    return ffn_core(x, M, d_model, d_inner, *w, out, static_cast<float*>(workspace), hipStream_t(stream));
}

int lamp_embed_fwd(const int64_t* src_seq, const int64_t* src_pos, int64_t n_tokens, const float* emb,
                   int32_t n_vocab, const float* pos_table, int32_t n_position, int32_t d_model, float* out,
                   lamp_stream_t stream) {
    return launch_embed(src_seq, src_pos, n_tokens, emb, n_vocab, pos_table, n_position, d_model, out,
                        hipStream_t(stream));
}

int lamp_pack_weight(const float* W, int32_t N, int32_t K, int64_t ldw, int32_t format, float* packed, lamp_stream_t stream) {
    return launch_pack_weight(W, N, K, ldw, format, packed, hipStream_t(stream));
}

int lamp_diag_logits_fwd(const float* y, const float* w_out, int32_t B, int32_t L, int32_t d_model,
                         float* logits, lamp_stream_t stream) {
    return launch_diag(y, w_out, B, L, d_model, logits, hipStream_t(stream));
}

static_assert(sizeof(lamp_gemm_desc) == 160, "lamp_gemm_desc layout is part of the ABI");
static_assert(sizeof(lamp_mask) == 56, "lamp_mask layout is part of the ABI (lamp_amd/_native.py: Mask)");
static_assert(sizeof(lamp_model) == 152, "lamp_model layout is part of the ABI (lamp_amd/_native.py: Model)");

size_t lamp_gemm_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t batch) {
    return gemm_gen_workspace_bytes(M, N, K, batch);
}

int lamp_gemm(const lamp_gemm_desc* d, void* workspace, size_t workspace_bytes, lamp_stream_t stream) {
    if (!d) return LAMP_E_NULL;
    return launch_gemm_gen(*d, workspace, workspace_bytes, hipStream_t(stream));
}

int lamp_gemm_grouped(const lamp_gemm_desc* descs, int32_t n, lamp_stream_t stream) {
    return launch_gemm_group(descs, n, hipStream_t(stream));
}

int lamp_layernorm_residual_fwd(const float* x, const float* residual, int64_t residual_rows, int64_t M, int32_t d,
                                const float* gamma, const float* beta, float eps, float dropout_p, uint32_t seed,
                                float* y, lamp_stream_t stream) {
    if (!y) return LAMP_E_NULL;
    if (!(dropout_p >= 0.f) || !(dropout_p < 1.f)) return LAMP_E_UNSUPPORTED;
    const DropoutSpec ds = make_dropout(dropout_p, seed);
    return launch_layernorm(x, M, d, gamma, beta, eps, residual, residual_rows, y, hipStream_t(stream), nullptr, 0,
                            nullptr, dropout_p > 0.f ? &ds : nullptr);
}

size_t lamp_layernorm_bwd_workspace_bytes(int64_t M, int32_t d) { return layernorm_bwd_workspace_bytes(M, d); }

int lamp_layernorm_bwd(const float* x, const float* residual, int64_t residual_rows, int64_t M, int32_t d,
                       const float* gamma, float eps, float dropout_p, uint32_t seed, const float* dy, float* dz,
                       float* dx, float* dgamma, float* dbeta, float* dbias, void* workspace, size_t workspace_bytes,
                       lamp_stream_t stream) {
    if (!(dropout_p >= 0.f) || !(dropout_p < 1.f)) return LAMP_E_UNSUPPORTED;
    const DropoutSpec ds = make_dropout(dropout_p, seed);
    return launch_layernorm_bwd(x, residual, residual_rows, M, d, gamma, eps, dropout_p > 0.f ? &ds : nullptr, dy, dz, dx,
                                dgamma, dbeta, dbias, workspace, workspace_bytes, hipStream_t(stream));
}

size_t lamp_colsum_workspace_bytes(int64_t M, int64_t N) { return colsum_workspace_bytes(M, N); }

int lamp_colsum(const float* x, int64_t M, int64_t N, int64_t ldx, float* out, void* workspace, size_t workspace_bytes,
                lamp_stream_t stream) {
    return launch_colsum(x, M, N, ldx, out, workspace, workspace_bytes, hipStream_t(stream));
}

int lamp_dropout(const float* x, int64_t n, float p, uint32_t seed, float* y, lamp_stream_t stream) {
    return launch_dropout(x, n, p, seed, y, hipStream_t(stream));
}

int lamp_softmax_bwd(const float* P, const float* dP, int64_t rows, int32_t lk, float scale, float* dS,
                     lamp_stream_t stream) {
    return launch_softmax_bwd(P, dP, rows, lk, scale, dS, hipStream_t(stream));
}

int lamp_diag_logits_bwd(const float* y, const float* w_out, const float* dlogits, int32_t B, int32_t L, int32_t d_model,
                         float* dy, float* dw, lamp_stream_t stream) {
    return launch_diag_bwd(y, w_out, dlogits, B, L, d_model, dy, dw, hipStream_t(stream));
}

int lamp_embed_bwd(const int64_t* src_seq, int64_t n_tokens, const float* dout, int32_t d_model, int32_t n_vocab,
                   int64_t pad_idx, float* d_emb, lamp_stream_t stream) {
    return launch_embed_bwd(src_seq, n_tokens, dout, d_model, n_vocab, pad_idx, d_emb, hipStream_t(stream));
}

// ---- training-mode sub-layers, one call each (lamp_amd/training.py) -------------------------------------------------------
namespace {
// C_z[m, n] (+)= sum_k A_z(m, k) B_z(n, k) through gemm_gen; strides in elements, (row, col) per operand
struct Opd {
    const float* p;
    int64_t rs, cs, b0, b1;
};
int gg(const Opd& A, const Opd& B, float* C, int64_t ldc, int64_t cb0, int64_t cb1, int M, int N, int K, int nb0, int nb1,
       bool accumulate, const float* relu_mask, void* ws, size_t ws_bytes, hipStream_t s) {
    lamp_gemm_desc d{};
    d.A = A.p; d.B = B.p; d.C = C;
    d.M = M; d.N = N; d.K = K;
    d.batch0 = nb0; d.batch1 = nb1;
    d.accumulate = accumulate ? 1 : 0;
    d.a_row_stride = A.rs; d.a_col_stride = A.cs; d.a_batch0 = A.b0; d.a_batch1 = A.b1;
    d.b_row_stride = B.rs; d.b_col_stride = B.cs; d.b_batch0 = B.b0; d.b_batch1 = B.b1;
    d.ldc = ldc; d.c_batch0 = cb0; d.c_batch1 = cb1;
    d.relu_mask = relu_mask; d.ld_mask = N;
    d.alpha = 1.f;
    const size_t need = gemm_gen_workspace_bytes(M, N, K, nb0 * nb1);
    return launch_gemm_gen(d, need <= ws_bytes ? ws : nullptr, need <= ws_bytes ? ws_bytes : 0, s);
}
inline Opd rows(const float* p, int64_t ld) { return Opd{p, ld, 1, 0, 0}; }        // [m, k], k contiguous
inline Opd cols(const float* p, int64_t ld) { return Opd{p, 1, ld, 0, 0}; }        // stored [k, m]: the transposed read
inline size_t max3(size_t a, size_t b, size_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
}  // namespace

int lamp_ffn_train_fwd(const float* x, int64_t M, int32_t d_model, int32_t d_inner, const lamp_ffn_weights* w,
                       float dropout_p, uint32_t seed, float* h, float* o, float* y, lamp_stream_t stream) {
    if (!x || !w || !h || !o || !y) return LAMP_E_NULL;
    if (!w->w1 || !w->b1 || !w->w2 || !w->b2 || !w->ln_g || !w->ln_b) return LAMP_E_NULL;
    if (M <= 0 || d_model <= 0 || d_inner <= 0) return LAMP_E_DIMS;
    if (!(dropout_p >= 0.f) || !(dropout_p < 1.f)) return LAMP_E_UNSUPPORTED;
    hipStream_t s = hipStream_t(stream);
    {
        const float* W[1] = {w->w1};
        const float* b[1] = {w->b1};
        float* C[1] = {h};
        LAMP_CK(linear(x, M, d_model, d_model, W, 1, d_inner, d_model, b, nullptr, 0, 1, C, d_inner, s));
    }
    {
        const float* W[1] = {w->w2};
        const float* b[1] = {w->b2};
        float* C[1] = {o};
        LAMP_CK(linear(h, M, d_inner, d_inner, W, 1, d_model, d_inner, b, nullptr, 0, 0, C, d_model, s));
    }
    const DropoutSpec ds = make_dropout(dropout_p, seed);
    return launch_layernorm(o, M, d_model, w->ln_g, w->ln_b, 1e-5f, x, 0, y, s, nullptr, 0, nullptr,
                            dropout_p > 0.f ? &ds : nullptr);
}

size_t lamp_ffn_bwd_workspace_bytes(int64_t M, int32_t d_model, int32_t d_inner) {
    if (M <= 0 || M > 0x7fffffff || d_model <= 0 || d_inner <= 0) return 0;
    return max3(layernorm_bwd_workspace_bytes(M, d_model), colsum_workspace_bytes(M, d_inner),
                gemm_gen_workspace_bytes(d_model > d_inner ? d_model : d_inner, d_model > d_inner ? d_model : d_inner,
                                         int(M), 1));
}

size_t lamp_ffn_bwd_partials_bytes(int64_t M, int32_t d_model, int32_t d_inner) {
    if (M <= 0 || d_model <= 0 || d_inner <= 0) return 0;
    return align_up(layernorm_bwd_workspace_bytes(M, d_model), 256) + colsum_workspace_bytes(M, d_inner);
}

int lamp_reduce_partials_grouped(const lamp_reduce_job* jobs, int32_t n, lamp_stream_t stream) {
    return launch_reduce_group(jobs, n, hipStream_t(stream));
}

int lamp_ffn_bwd(const float* x, const float* h, const float* o, const float* dy, int64_t M, int32_t d_model,
                 int32_t d_inner, const lamp_ffn_weights* w, float dropout_p, uint32_t seed, float* dx, float* d_o, float* dh,
                 float* dW1, float* dW2, float* db1, float* db2, float* dgamma, float* dbeta, void* workspace,
                 size_t workspace_bytes, void* partials, size_t partials_bytes, lamp_reduce_job* jobs, lamp_stream_t stream) {
    if (partials && (!jobs || partials_bytes < lamp_ffn_bwd_partials_bytes(M, d_model, d_inner))) return LAMP_E_WORKSPACE;
    if (!x || !h || !o || !dy || !w || !dx || !dh || !db1 || !db2 || !dgamma || !dbeta) return LAMP_E_NULL;
    if (!w->w1 || !w->w2 || !w->ln_g) return LAMP_E_NULL;
    if (M <= 0 || M > 0x7fffffff || d_model <= 0 || d_inner <= 0) return LAMP_E_DIMS;
    if (!(dropout_p >= 0.f) || !(dropout_p < 1.f)) return LAMP_E_UNSUPPORTED;
    const bool drop = dropout_p > 0.f;
    if (drop && !d_o) return LAMP_E_NULL;
    if (!drop && !dW2) return LAMP_E_UNSUPPORTED;   // without dropout d_o IS dx, which is accumulated into: dW2 cannot wait
    if (workspace_bytes < lamp_ffn_bwd_workspace_bytes(M, d_model, d_inner) || (!workspace && workspace_bytes))
        return LAMP_E_WORKSPACE;
    hipStream_t s = hipStream_t(stream);
    const DropoutSpec ds = make_dropout(dropout_p, seed);
    const int Mi = int(M);
    const size_t ln_bytes = align_up(layernorm_bwd_workspace_bytes(M, d_model), 256);
    char* keep = static_cast<char*>(partials);
    if (keep)
        LAMP_CK(launch_layernorm_bwd(o, x, 0, M, d_model, w->ln_g, 1e-5f, drop ? &ds : nullptr, dy, dx, drop ? d_o : nullptr,
                                     dgamma, dbeta, db2, keep, ln_bytes, s, &jobs[0]));
    else
        LAMP_CK(launch_layernorm_bwd(o, x, 0, M, d_model, w->ln_g, 1e-5f, drop ? &ds : nullptr, dy, dx, drop ? d_o : nullptr,
                                     dgamma, dbeta, db2, workspace, workspace_bytes, s));
    const float* g_o = drop ? d_o : dx;
    if (dW2)   // dW2 = d_o^T h
        LAMP_CK(gg(cols(g_o, d_model), cols(h, d_inner), dW2, d_inner, 0, 0, d_model, d_inner, Mi, 1, 1, false, nullptr,
                   workspace, workspace_bytes, s));
    // dh = relu'(h) * (d_o W2)
    LAMP_CK(gg(rows(g_o, d_model), cols(w->w2, d_inner), dh, d_inner, 0, 0, Mi, d_inner, d_model, 1, 1, false, h, workspace,
               workspace_bytes, s));
    if (keep)
        LAMP_CK(launch_colsum(dh, M, d_inner, d_inner, db1, keep + ln_bytes, partials_bytes - ln_bytes, s, &jobs[1]));
    else
        LAMP_CK(launch_colsum(dh, M, d_inner, d_inner, db1, workspace, workspace_bytes, s));
    if (dW1)   // dW1 = dh^T x
        LAMP_CK(gg(cols(dh, d_inner), cols(x, d_model), dW1, d_model, 0, 0, d_inner, d_model, Mi, 1, 1, false, nullptr,
                   workspace, workspace_bytes, s));
    // dx = residual branch + dh W1
    return gg(rows(dh, d_inner), cols(w->w1, d_model), dx, d_model, 0, 0, Mi, d_model, d_inner, 1, 1, true, nullptr, workspace,
              workspace_bytes, s);
}

int lamp_mha_train_fwd(const lamp_mha_train_desc* c, const lamp_mha_weights* w, const float* xq, const float* xk,
                       const float* xv, const lamp_mask* mask, float* q, float* k, float* v, float* a, float* P, float* Pd,
                       float* lse, float* o, float* y, lamp_stream_t stream) {
    if (!c || !w || !xq || !xk || !xv || !q || !k || !v || !a || !P || !lse || !y) return LAMP_E_NULL;
    if (!w->w_qs || !w->w_ks || !w->w_vs || !w->ln_g || !w->ln_b) return LAMP_E_NULL;
    const int B = c->B, lq = c->lq, lk = c->lk, d = c->d_model, H = c->n_head, dk = c->d_k, dv = c->d_v;
    if (B <= 0 || lq <= 0 || lk <= 0 || d <= 0 || H <= 0 || dk <= 0 || dv <= 0) return LAMP_E_DIMS;
    if (H != w->n_head) return LAMP_E_DIMS;
    if (dk > 128 || dv > 128) return LAMP_E_UNSUPPORTED;   // wide heads keep their scores in the map buffer: per-launch route
    if (!(c->p_attn >= 0.f) || !(c->p_attn < 1.f) || !(c->p_out >= 0.f) || !(c->p_out < 1.f)) return LAMP_E_UNSUPPORTED;
    const bool has_fc = w->fc != nullptr;
    if (has_fc ? !o : (H * dv != d)) return has_fc ? LAMP_E_NULL : LAMP_E_DIMS;
    if (c->p_attn > 0.f && !Pd) return LAMP_E_NULL;
    hipStream_t s = hipStream_t(stream);
    const int hdk = H * dk, hdv = H * dv;
    const int64_t Mq = int64_t(B) * lq, Mk = int64_t(B) * lk;
    // the projections that share their input go out as segments of one launch (same bits as one launch each)
    const bool one_kv = xk == xv && hdk == hdv, one_qkv = one_kv && xq == xk && lq == lk;
    {
        const float* W[3] = {w->w_qs, w->w_ks, w->w_vs};
        float* C[3] = {q, k, v};
        LAMP_CK(linear(xq, Mq, d, d, W, one_qkv ? 3 : 1, hdk, d, nullptr, nullptr, 0, 0, C, hdk, s));
    }
    if (!one_qkv && one_kv) {
        const float* W[2] = {w->w_ks, w->w_vs};
        float* C[2] = {k, v};
        LAMP_CK(linear(xk, Mk, d, d, W, 2, hdk, d, nullptr, nullptr, 0, 0, C, hdk, s));
    } else if (!one_qkv) {
        const float* Wk[1] = {w->w_ks};
        float* Ck[1] = {k};
        LAMP_CK(linear(xk, Mk, d, d, Wk, 1, hdk, d, nullptr, nullptr, 0, 0, Ck, hdk, s));
        const float* Wv[1] = {w->w_vs};
        float* Cv[1] = {v};
        LAMP_CK(linear(xv, Mk, d, d, Wv, 1, hdv, d, nullptr, nullptr, 0, 0, Cv, hdv, s));
    }
    const lamp_attn_layout lay{int64_t(lq) * hdk, dk, hdk, int64_t(lk) * hdk, dk, hdk, int64_t(lk) * hdv, dv, hdv,
                               int64_t(lq) * hdv, dv, hdv};
    LAMP_CK(sdpa_impl(q, k, v, a, P, lse, B, H, lq, lk, dk, dv, c->inv_temperature, mask, &lay, stream));
    if (c->p_attn > 0.f) {   // the reference drops probabilities AFTER the softmax (lamp/SubLayers.py:40-41): a = dropout(P) V
        LAMP_CK(launch_dropout(P, int64_t(H) * B * lq * lk, c->p_attn, c->seed_attn, Pd, s));
        LAMP_CK(gg(Opd{Pd, lk, 1, int64_t(B) * lq * lk, int64_t(lq) * lk}, Opd{v, 1, hdv, dv, int64_t(lk) * hdv}, a, hdv, dv,
                   int64_t(lq) * hdv, lq, dv, lk, H, B, false, nullptr, nullptr, 0, s));
    }
    const DropoutSpec ds = make_dropout(c->p_out, c->seed_out);
    const float* pre = a;
    if (has_fc) {
        const float* W[1] = {w->fc};
        float* C[1] = {o};
        LAMP_CK(linear(a, Mq, hdv, hdv, W, 1, d, hdv, nullptr, nullptr, 0, 0, C, d, s));
        pre = o;
    }
    return launch_layernorm(pre, Mq, d, w->ln_g, w->ln_b, 1e-5f, xq, 0, y, s, nullptr, 0, nullptr,
                            c->p_out > 0.f ? &ds : nullptr);
}

size_t lamp_mha_bwd_workspace_bytes(const lamp_mha_train_desc* c) {
    if (!c || c->B <= 0 || c->lq <= 0 || c->lk <= 0 || c->d_model <= 0) return 0;
    const int64_t Mq = int64_t(c->B) * c->lq, Mk = int64_t(c->B) * c->lk;
    const int hd = c->n_head * (c->d_k > c->d_v ? c->d_k : c->d_v);
    const int big = hd > c->d_model ? hd : c->d_model;
    return max3(layernorm_bwd_workspace_bytes(Mq, c->d_model), gemm_gen_workspace_bytes(big, big, int(Mq), 1),
                gemm_gen_workspace_bytes(big, big, int(Mk), 1));
}

size_t lamp_mha_bwd_partials_bytes(const lamp_mha_train_desc* c) {
    if (!c || c->B <= 0 || c->lq <= 0 || c->d_model <= 0) return 0;
    return layernorm_bwd_workspace_bytes(int64_t(c->B) * c->lq, c->d_model);
}

int lamp_mha_bwd(const lamp_mha_train_desc* c, const lamp_mha_weights* w, const float* xq, const float* xk, const float* xv,
                 const float* q, const float* k, const float* v, const float* a, const float* P, const float* Pd,
                 const float* o, const float* dy, float* dxq, float* d_o, float* da, float* dP, float* dq, float* dk_,
                 float* dv_, float* dxk, float* dxv, float* dgamma, float* dbeta, float* dwq, float* dwk, float* dwv, float* dfc,
                 void* workspace, size_t workspace_bytes, void* partials, size_t partials_bytes, lamp_reduce_job* job,
                 lamp_stream_t stream) {
    if (partials && (!job || partials_bytes < lamp_mha_bwd_partials_bytes(c))) return LAMP_E_WORKSPACE;
    if (!c || !w || !xq || !xk || !xv || !q || !k || !v || !a || !P || !dy) return LAMP_E_NULL;
    if (!dxq || !dP || !dq || !dk_ || !dv_ || !dxk || !dgamma || !dbeta) return LAMP_E_NULL;
    if (!w->w_qs || !w->w_ks || !w->w_vs || !w->ln_g) return LAMP_E_NULL;
    const int B = c->B, lq = c->lq, lk = c->lk, d = c->d_model, H = c->n_head, dk = c->d_k, dv = c->d_v;
    if (B <= 0 || lq <= 0 || lk <= 0 || d <= 0 || H <= 0 || dk <= 0 || dv <= 0 || H != w->n_head) return LAMP_E_DIMS;
    const int64_t Mq64 = int64_t(B) * lq, Mk64 = int64_t(B) * lk;
    if (Mq64 > 0x7fffffff || Mk64 > 0x7fffffff || int64_t(H) * B > 65535) return LAMP_E_DIMS;
    if (!(c->p_attn >= 0.f) || !(c->p_attn < 1.f) || !(c->p_out >= 0.f) || !(c->p_out < 1.f)) return LAMP_E_UNSUPPORTED;
    const bool has_fc = w->fc != nullptr, drop_o = c->p_out > 0.f, drop_a = c->p_attn > 0.f;
    if (has_fc && (!o || !da)) return LAMP_E_NULL;
    if (drop_o && !d_o) return LAMP_E_NULL;
    if (drop_a && !Pd) return LAMP_E_NULL;
    if (has_fc && !drop_o && !dfc) return LAMP_E_UNSUPPORTED;   // d_o IS dxq then, which is accumulated into: dfc cannot wait
    if (workspace_bytes < lamp_mha_bwd_workspace_bytes(c) || (!workspace && workspace_bytes)) return LAMP_E_WORKSPACE;
    hipStream_t s = hipStream_t(stream);
    const int hdk = H * dk, hdv = H * dv, Mq = int(Mq64), Mk = int(Mk64);
    const DropoutSpec ds = make_dropout(c->p_out, c->seed_out);
    void* ws = workspace;
    const size_t wsb = workspace_bytes;
    // add & norm (+ output dropout): dxq <- the residual branch, d_o <- the gradient of the fc output
    if (partials)
        LAMP_CK(launch_layernorm_bwd(has_fc ? o : a, xq, 0, Mq64, d, w->ln_g, 1e-5f, drop_o ? &ds : nullptr, dy, dxq,
                                     drop_o ? d_o : nullptr, dgamma, dbeta, nullptr, partials, partials_bytes, s, job));
    else
        LAMP_CK(launch_layernorm_bwd(has_fc ? o : a, xq, 0, Mq64, d, w->ln_g, 1e-5f, drop_o ? &ds : nullptr, dy, dxq,
                                     drop_o ? d_o : nullptr, dgamma, dbeta, nullptr, ws, wsb, s));
    const float* g_o = drop_o ? d_o : dxq;
    const float* g_a = g_o;   // gradient of the concatenated head outputs [Mq, H*dv]
    if (has_fc) {
        if (dfc) LAMP_CK(gg(cols(g_o, d), cols(a, hdv), dfc, hdv, 0, 0, d, hdv, Mq, 1, 1, false, nullptr, ws, wsb, s));
        LAMP_CK(gg(rows(g_o, d), cols(w->fc, hdv), da, hdv, 0, 0, Mq, hdv, d, 1, 1, false, nullptr, ws, wsb, s));
        g_a = da;
    }   // (single head without output dropout: g_a IS dxq; every product that reads it is issued before (*) accumulates into it)
    // head views: index (h, b) -> batch0 = head, batch1 = sample
    const int64_t PB0 = int64_t(B) * lq * lk, PB1 = int64_t(lq) * lk;
    const float* Pu = drop_a ? Pd : P;
    const DropoutSpec da_spec = make_dropout(c->p_attn, c->seed_attn);
    // dV = Pd^T dA
    LAMP_CK(gg(Opd{Pu, 1, lk, PB0, PB1}, Opd{g_a, 1, hdv, dv, int64_t(lq) * hdv}, dv_, hdv, dv, int64_t(lk) * hdv, lk, dv, lq,
               H, B, false, nullptr, nullptr, 0, s));
    // dPd = dA V^T
    LAMP_CK(gg(Opd{g_a, hdv, 1, dv, int64_t(lq) * hdv}, Opd{v, hdv, 1, dv, int64_t(lk) * hdv}, dP, lk, PB0, PB1, lq, lk, dv, H,
               B, false, nullptr, nullptr, 0, s));
    // dS = softmax backward of dropout-backward(dPd), in place (the mask is applied on load)
    LAMP_CK(launch_softmax_bwd(P, dP, int64_t(H) * B * lq, lk, c->inv_temperature, dP, s, drop_a ? &da_spec : nullptr));
    // dQ = dS K, dK = dS^T Q
    LAMP_CK(gg(Opd{dP, lk, 1, PB0, PB1}, Opd{k, 1, hdk, dk, int64_t(lk) * hdk}, dq, hdk, dk, int64_t(lq) * hdk, lq, dk, lk, H, B,
               false, nullptr, nullptr, 0, s));
    LAMP_CK(gg(Opd{dP, 1, lk, PB0, PB1}, Opd{q, 1, hdk, dk, int64_t(lq) * hdk}, dk_, hdk, dk, int64_t(lk) * hdk, lk, dk, lq, H, B,
               false, nullptr, nullptr, 0, s));
    if (dwq) LAMP_CK(gg(cols(dq, hdk), cols(xq, d), dwq, d, 0, 0, hdk, d, Mq, 1, 1, false, nullptr, ws, wsb, s));
    if (dwk) LAMP_CK(gg(cols(dk_, hdk), cols(xk, d), dwk, d, 0, 0, hdk, d, Mk, 1, 1, false, nullptr, ws, wsb, s));
    if (dwv) LAMP_CK(gg(cols(dv_, hdv), cols(xv, d), dwv, d, 0, 0, hdv, d, Mk, 1, 1, false, nullptr, ws, wsb, s));
    // (*) data gradients of the three projections
    LAMP_CK(gg(rows(dq, hdk), cols(w->w_qs, d), dxq, d, 0, 0, Mq, d, hdk, 1, 1, true, nullptr, ws, wsb, s));
    // dxk == dxq (self-attention: query and key source are one tensor): its gradient is the sum, accumulated in place
    LAMP_CK(gg(rows(dk_, hdk), cols(w->w_ks, d), dxk, d, 0, 0, Mk, d, hdk, 1, 1, dxk == dxq, nullptr, ws, wsb, s));
    if (dxv) return gg(rows(dv_, hdv), cols(w->w_vs, d), dxv, d, 0, 0, Mk, d, hdv, 1, 1, false, nullptr, ws, wsb, s);
    return gg(rows(dv_, hdv), cols(w->w_vs, d), dxk, d, 0, 0, Mk, d, hdv, 1, 1, true, nullptr, ws, wsb, s);
}

int lamp_prior_graph_build(const int64_t* label_ids, const int64_t* offsets, int64_t n_samples, int32_t L, float* adj,
                           uint8_t* blocked, lamp_stream_t stream) {
    return launch_prior_graph(label_ids, offsets, n_samples, L, adj, blocked, hipStream_t(stream));
}

int lamp_sigmoid_bce_fwd(const float* logits, const float* targets, int64_t n_rows, int32_t L, float* probs,
                         float* row_loss, lamp_stream_t stream) {
    return launch_sigmoid_bce(logits, targets, n_rows, L, probs, row_loss, hipStream_t(stream));
}

// ------------------------------------------------------------------ whole forward
static int model_heads(const lamp_model* m, int* h_max) {
    int h = 1;
    for (int i = 0; i < m->n_layers_enc; ++i) h = h > m->enc_layers[i].slf_attn.n_head ? h : m->enc_layers[i].slf_attn.n_head;
    for (int i = 0; i < m->n_layers_dec; ++i) {
        const lamp_dec_layer& l = m->dec_layers[i];
        h = h > l.enc_attn.n_head ? h : l.enc_attn.n_head;
        if (l.slf_attn.present) h = h > l.slf_attn.n_head ? h : l.slf_attn.n_head;
    }
    *h_max = h;
    return 0;
}

struct FwdPlan {
    size_t fixed_floats;       // independent of the micro-batch
    size_t per_sample_floats;  // times micro-batch
    int R;                     // rows per sample of the widest activation
    int hdk, hdv;
    size_t side_kv_floats;     // per sample; K/V of every decoder layer's enc-attention, projected in one launch
    size_t score_floats;       // per sample; (h, Rq, R) score scratch of wide heads, else 0
    size_t lse_floats;         // per sample; (h, Rq)
};

static int make_plan(const lamp_model* m, int T, int want_attn, FwdPlan* pl) {
    if (!m) return LAMP_E_NULL;
    if (m->d_model <= 0 || m->d_inner <= 0 || m->d_k <= 0 || m->d_v <= 0 || m->n_labels <= 0 || T <= 0 ||
        m->n_layers_enc < 0 || m->n_layers_dec < 0)
        return LAMP_E_DIMS;
    if ((m->n_layers_enc && !m->enc_layers) || (m->n_layers_dec && !m->dec_layers)) return LAMP_E_NULL;
    int h = 1;
    model_heads(m, &h);
    const int L = m->n_labels;
    const int R = T > L ? T : L;
    pl->R = R;
    pl->hdk = h * m->d_k;
    pl->hdv = h * m->d_v;
    const int Rq = want_attn ? R : L;  // the Q / A buffers only see encoder rows when maps are wanted
    // + 64 floats of slack per region for the carver's 256-byte rounding
    pl->fixed_floats = 64 * 11;
    pl->per_sample_floats = size_t(R) * m->d_inner + size_t(Rq) * pl->hdk + size_t(R) * pl->hdk +
                            size_t(R) * pl->hdv + size_t(Rq) * pl->hdv + size_t(L) * m->d_model;
    // wide heads (d_k or d_v > 128): the scores of the largest attention of the forward go through this scratch
    pl->score_floats = wide_heads(m->d_k, m->d_v) ? size_t(h) * Rq * R : 0;
    pl->per_sample_floats += pl->score_floats;
    pl->lse_floats = size_t(h) * Rq;  // row log-sum-exp of an attention whose maps are requested
    pl->per_sample_floats += pl->lse_floats;
    // ragged batches: the packed token rows of the encoder ([n_tok + 1, d]: + the shared PAD row), one more row of the
    // FFN hidden buffer for it, and the SeqPlan's 3 mb + 3 ints
    pl->per_sample_floats += size_t(T) * m->d_model + 3 + size_t((T + 31) / 32);
    pl->fixed_floats += size_t(m->d_model) + size_t(m->d_inner) + 64 * 3 + 4;
    // the plan's hand-off granules inside the merged plan + gather launch: 2 mb + 2 eight-byte words
    pl->per_sample_floats += 4;
    pl->fixed_floats += 4 + 64;
    // the row -> (token, position) maps of the gathered residual (folded first encoder layer): 2 (mb T + 1) ints
    pl->per_sample_floats += 2 * size_t(T);
    pl->fixed_floats += 2 + 64;
    // K/V of all decoder layers' enc-attention, projected together right after the encoder when the batch fits
    pl->side_kv_floats = size_t(m->n_layers_dec) * T * (pl->hdk + pl->hdv);
    return 0;
}

size_t lamp_forward_workspace_bytes(const lamp_model* m, int32_t micro_batch, int32_t T, int32_t want_attn) {
    FwdPlan pl;
    if (micro_batch <= 0 || make_plan(m, T, want_attn, &pl) != 0) return 0;
    return (pl.fixed_floats + (pl.per_sample_floats + pl.side_kv_floats) * size_t(micro_batch)) * sizeof(float);
}

// The whole batch in micro-batches that fit `workspace`.  `kv_ahead`: the enc-attention K/V projections of EVERY
// decoder layer (they depend only on the encoder output) are issued as one multi-segment launch right after the
// encoder.  Samples are independent and no kernel variant depends on the batch size, so a sample's results are
// bit-identical for every micro-batch split.
constexpr int MAX_AHEAD_LAYERS = 16;
static int forward_range(const lamp_model* m, const FwdPlan& pl, const int64_t* src_seq, const int64_t* src_pos,
                         int32_t B, int32_t T, float* logits, float* enc_output, const lamp_aux* aux,
                         void* workspace, size_t workspace_bytes, hipStream_t s, bool kv_ahead) {
    const bool want_enc_attn = aux && aux->enc_self_attn;
    const int d = m->d_model, dff = m->d_inner, dk = m->d_k, dv = m->d_v, L = m->n_labels;
    const int n_ahead = kv_ahead ? m->n_layers_dec : 0;
    const size_t per_sample = pl.per_sample_floats + (n_ahead ? pl.side_kv_floats : 0);
    const size_t ws_floats = workspace_bytes / sizeof(float);
    if (ws_floats < pl.fixed_floats + per_sample) return LAMP_E_WORKSPACE;
    int64_t mb = int64_t((ws_floats - pl.fixed_floats) / per_sample);
    if (mb > B) mb = B;

    // Ragged batches.  Every micro-batch first counts its samples' extents on the device (SeqPlan).  `packed`: the encoder
    // runs on the packed non-PAD token rows (+ ONE shared PAD row: all PAD positions of lamp/Encoders.py:64-79 hold the
    // same row-wise result), its last LayerNorm scatters into the padded enc_output, and K / V are projected from the
    // packed rows only.  Row-wise kernels and an M-independent k-order make this bit-identical to computing every
    // padded position.  Not packed (the dead encoder self-attention's maps are wanted, or wide heads): padded layout as
    // before; the enc-dec attention still stops at each sample's last real key either way.
    const bool packed = !want_enc_attn && !wide_heads(dk, dv) && m->n_layers_enc > 0;
    const int Rq = want_enc_attn ? pl.R : L;
    float *H = nullptr, *Y = nullptr, *Xp = nullptr;
    int* plan_ints = nullptr;
    int* row_maps = nullptr;   // [2][mb T + 1]: token / position index of every (packed) encoder row, for the gathered residual
    unsigned long long* granules = nullptr;
    float* Kahead[MAX_AHEAD_LAYERS] = {};
    float* Vahead[MAX_AHEAD_LAYERS] = {};
    MhaScratch sc{};
    for (int attempt = 0; attempt < 2; ++attempt) {
        Carver c(workspace, workspace_bytes);
        H = c.take(size_t(mb) * pl.R * dff + dff);
        sc.Q = c.take(size_t(mb) * Rq * pl.hdk);
        sc.K = c.take(size_t(mb) * pl.R * pl.hdk);
        sc.V = c.take(size_t(mb) * pl.R * pl.hdv);
        sc.A = c.take(size_t(mb) * Rq * pl.hdv);
        sc.S = pl.score_floats ? c.take(size_t(mb) * pl.score_floats) : nullptr;
        sc.lse = c.take(size_t(mb) * pl.lse_floats);
        Y = c.take(size_t(mb) * L * d);
        Xp = c.take(size_t(mb) * T * d + d);
        plan_ints = reinterpret_cast<int*>(c.take(plan_int_count(mb, T)));
        granules = reinterpret_cast<unsigned long long*>(c.take(size_t(4) * mb + 4));
        row_maps = reinterpret_cast<int*>(c.take(2 * (size_t(mb) * T + 1)));
        for (int i = 0; i < n_ahead; ++i) {
            Kahead[i] = c.take(size_t(mb) * T * pl.hdk);
            Vahead[i] = c.take(size_t(mb) * T * pl.hdv);
        }
        if (c.ok) break;
        if (attempt == 1 || mb <= 1) return LAMP_E_WORKSPACE;
        --mb;  // rounding slack exhausted: one sample fewer
    }

    for (int64_t b0 = 0; b0 < B; b0 += mb) {
        const int nb = int(B - b0 < mb ? B - b0 : mb);
        const int64_t* seq = src_seq + b0 * T;
        const int64_t* pos = src_pos ? src_pos + b0 * T : nullptr;
        float* x = enc_output + b0 * int64_t(T) * d;  // the padded encoder output of this micro-batch
        const int64_t Me = int64_t(nb) * T;
        SeqPlan sp = plan_from(plan_ints, nb, T);
        sp.granules = packed ? granules : nullptr;
        // Packed layout: the plan rides in the first workgroups of the embedding gather's launch and hands its results to the
        // gather through 8-byte granules (pointwise.hip: embed_plan_kernel; round 3 had it as a launch of its own -- a
        // dependent launch costs 5-8 us on this chain however little it does -- after folding it into EVERY workgroup of
        // the gather had measured slower, 21.9 us against 6.6 + 10.5).  Padded layout: the plan kernel on its own.
        if (!packed) LAMP_CK(launch_seq_plan(seq, m->position_enc ? pos : nullptr, nb, T, T, packed, sp, s));

        // ---- GraphEncoder.forward (lamp/Encoders.py:64-110) ----
        lamp_mask pad_mask{LAMP_MASK_KEY_TOKENS_I64, 0, seq, T, 0, nullptr, 0};
        const float* xk = x;  // what the decoder's K / V projections read
        // Encoder layer 0's W1 folded into the embedding tables (lamp_model::enc0_emb_w1): the gather writes that layer's
        // hidden rows into H beside the embedded rows, and its first GEMM is not launched.
        // ... and the embedded rows themselves are not written either: their one reader, the residual of that layer's second
        // GEMM, gathers them from the tables through the row maps the gather kernel leaves instead (8 bytes per row).
        const bool folded = m->enc0_emb_w1 && m->n_layers_enc > 0;
        int* row_tok = row_maps, *row_pos = row_maps + (size_t(mb) * T + 1);
        // (padded layout = the dead self-attention's maps are wanted: layer 0's map reads the embedded rows, so they are written)
        const bool gather_res = folded && packed &&
                                gemm_gathered_residual_ok(d, dff, d, m->enc_layers[0].pos_ffn.b2, Xp, m->src_word_emb, m->position_enc);
        const EmbedFold fold{m->enc0_emb_w1, m->enc0_pos_w1, dff, folded ? H : nullptr, gather_res ? row_tok : nullptr,
                             gather_res ? row_pos : nullptr};
        const ResGather rg{row_tok, row_pos, m->src_word_emb, m->position_enc};
        if (packed) {
            LAMP_CK(launch_embed_plan(seq, pos, m->position_enc != nullptr, nb, T, m->src_word_emb, m->n_src_vocab,
                                      m->position_enc, m->n_position, d, sp, granules, Xp, s, &fold));
            for (int i = 0; i < m->n_layers_enc; ++i) {
                const bool last = i + 1 == m->n_layers_enc;
                LAMP_CK(ffn_core(Xp, Me + 1, d, dff, m->enc_layers[i].pos_ffn, Xp, H, s, nullptr, 0, nullptr, sp.rows + 1,
                                 last ? &sp : nullptr, nb, T, x, folded && i == 0, gather_res && i == 0 ? &rg : nullptr));  // lamp/Layers.py:18
            }
            xk = Xp;
        } else {
            LAMP_CK(launch_embed(seq, pos, Me, m->src_word_emb, m->n_src_vocab, m->position_enc, m->n_position, d, x, s, &fold));
            for (int i = 0; i < m->n_layers_enc; ++i) {
                const lamp_enc_layer& l = m->enc_layers[i];
                if (want_enc_attn && aux->enc_self_attn[i]) {
                    // lamp/Layers.py:16 -- only the attention map of this block is ever observable.  Maps are
                    // (h*B, T, T) over the WHOLE batch: this micro-batch fills rows h*B + b0 + b.
                    LAMP_CK(mha_core(x, false, x, nb, T, T, d, dk, dv, l.slf_attn, &pad_mask, nullptr,
                                     aux->enc_self_attn[i], sc, s, false, nullptr, B, int(b0), &sp));
                }
                LAMP_CK(ffn_core(x, Me, d, dff, l.pos_ffn, x, H, s, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, nullptr,
                                 folded && i == 0));  // lamp/Layers.py:18
            }
        }
        const int* kv_rows = packed ? sp.rows : nullptr;
        if (n_ahead > 0)
            LAMP_CK(project_kv_layers(xk, Me, d, dk, dv, m->dec_layers, n_ahead, Kahead, Vahead, s, kv_rows, packed ? x : nullptr));

        // ---- GraphDecoder.forward (lamp/Decoders.py:127-163) ----
        // the label graph: bit-packed rows when the caller provides them (one dword per 32-key tile), else bytes
        lamp_mask label_mask{LAMP_MASK_NONE, 0, nullptr, 0, 0, nullptr, 0};
        if (m->label_mask_bits)
            label_mask = lamp_mask{LAMP_MASK_BITS_U32, m->label_mask_flags, m->label_mask_bits, 0, (L + 31) / 32, m->label_tiles,
                                   (L + 31) / 32 + 1, m->label_mask_allowed};
        else if (m->label_mask)
            label_mask = lamp_mask{LAMP_MASK_U8, 0, m->label_mask, 0, L, m->label_tiles, (L + 31) / 32 + 1};
        const int64_t Md = int64_t(nb) * L;
        int n_int = 0;
        auto int_pred = [&](void) -> int {
            if (aux && aux->int_preds && n_int < aux->n_int_preds && aux->int_preds[n_int])
                LAMP_CK(launch_diag(Y, m->w_out, nb, L, d, aux->int_preds[n_int] + b0 * L, s));
            ++n_int;
            return 0;
        };
        for (int i = 0; i < m->n_layers_dec; ++i) {
            const lamp_dec_layer& l = m->dec_layers[i];
            const bool has_slf = l.slf_attn.present != 0;
            float* Penc = (aux && aux->dec_enc_attn) ? aux->dec_enc_attn[i] : nullptr;
            float* Pslf = (aux && aux->dec_self_attn) ? aux->dec_self_attn[i] : nullptr;
            MhaScratch sci = sc;
            const bool ahead = n_ahead > 0;
            if (ahead) {
                sci.K = Kahead[i];
                sci.V = Vahead[i];
            }
            // input->label messages (lamp/Layers.py:35); layer 0's query is the label table itself (its LayerNorm
            // kernel adds the shared residual)
            // the feed-forward block behind each attention block rides in the attention's tail launch when the shape
            // allows (chain.hip: same bits either way); pos_ffn2 follows the self-attention, or pos_ffn1 when there is none
            const bool last = i + 1 == m->n_layers_dec;
            const lamp_chain_pack* pk = m->chain_packs ? m->chain_packs + 2 * i : nullptr;
            FfnTail t1{&l.pos_ffn1, dff, nullptr, 0, nullptr, false, pk};
            FfnTail t2{&l.pos_ffn2, dff, last ? m->w_out : nullptr, L, last ? logits + b0 * L : nullptr, false, pk ? pk + 1 : nullptr};
            if (i == 0)
                LAMP_CK(mha_core(m->tgt_word_emb, true, xk, nb, L, T, d, dk, dv, l.enc_attn, &pad_mask, Y, Penc, sci, s,
                                 ahead, m->dec0_query, B, int(b0), &sp, packed, x, &t1));
            else
                LAMP_CK(mha_core(Y, false, xk, nb, L, T, d, dk, dv, l.enc_attn, &pad_mask, Y, Penc, sci, s, ahead, nullptr,
                                 B, int(b0), &sp, packed, x, &t1));
            if (!t1.done) LAMP_CK(ffn_core(Y, Md, d, dff, l.pos_ffn1, Y, H, s));  // lamp/Layers.py:36
            if (has_slf) {
                LAMP_CK(int_pred());  // dec_output_int, lamp/Decoders.py:149-151
                // label->label messages over the label graph (lamp/Layers.py:40)
                LAMP_CK(mha_core(Y, false, Y, nb, L, L, d, dk, dv, l.slf_attn, &label_mask, Y, Pslf, sc, s, false, nullptr,
                                 B, int(b0), nullptr, false, nullptr, &t2));
            }
            if (!t2.done) {
                if (!last) {
                    LAMP_CK(ffn_core(Y, Md, d, dff, l.pos_ffn2, Y, H, s));  // lamp/Layers.py:45
                } else {
                    // last layer: the read-out (lamp/Models.py:124-126) is fused into this LayerNorm
                    LAMP_CK(ffn_core(Y, Md, d, dff, l.pos_ffn2, Y, H, s, m->w_out, L, logits + b0 * L));
                }
            }
            if (!last) LAMP_CK(int_pred());                                 // all but the last (lamp/Models.py:130)
        }
    }
    return 0;
}

int lamp_forward(const lamp_model* m, const int64_t* src_seq, const int64_t* src_pos, int32_t B, int32_t T,
                 float* logits, float* enc_output, const lamp_aux* aux, void* workspace, size_t workspace_bytes,
                 lamp_stream_t stream) {
    hipStream_t s = hipStream_t(stream);
    if (!m || !src_seq || !logits || !enc_output || !workspace) return LAMP_E_NULL;
    if (B <= 0 || T <= 0) return LAMP_E_DIMS;
    if (!m->src_word_emb || !m->tgt_word_emb || !m->w_out) return LAMP_E_NULL;
    if (m->position_enc && !src_pos) return LAMP_E_NULL;
    if (m->enc0_emb_w1 && m->position_enc && !m->enc0_pos_w1) return LAMP_E_NULL;
    if (m->n_layers_dec <= 0) return LAMP_E_DIMS;
    FwdPlan pl;
    LAMP_CK(make_plan(m, T, aux && aux->enc_self_attn, &pl));
    if ((m->d_model & 3) || (m->d_inner & 3) || (m->d_k & 3) || (m->d_v & 3)) return LAMP_E_UNSUPPORTED;
    // One launch for every decoder layer's enc-attention K/V projection (they all read the finished encoder output)
    // when the whole batch still fits the workspace with the extra K/V buffers and the weights fit one segment list.
    const bool kv_ahead = 2 * m->n_layers_dec <= GEMM_MAX_SEG && m->n_layers_dec > 1 && m->n_layers_dec <= MAX_AHEAD_LAYERS &&
                          workspace_bytes >= (pl.fixed_floats + (pl.per_sample_floats + pl.side_kv_floats) * size_t(B) +
                                              size_t(64) * (8 + 2 * m->n_layers_dec)) * sizeof(float);
    return forward_range(m, pl, src_seq, src_pos, B, T, logits, enc_output, aux, workspace, workspace_bytes, s, kv_ahead);
}

// ------------------------------------------------------------------ profiling ABI
int lamp_prof_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return 0;
}

int lamp_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) {
        (void)hipEventSynchronize(r.e1);
        g_event_pool.push_back(r.e0);
        g_event_pool.push_back(r.e1);
    }
    g_prof.clear();
    return 0;
}

int lamp_prof_read(int32_t kernel_class, int64_t* launches, double* total_ms, double* flops, double* bytes) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int64_t n = 0;
    double ms = 0, fl = 0, by = 0;
    for (auto& r : g_prof) {
        if (r.cls != kernel_class) continue;
        hipError_t e = hipEventSynchronize(r.e1);
        if (e != hipSuccess) return int(e);
        float t = 0.f;
        e = hipEventElapsedTime(&t, r.e0, r.e1);
        if (e != hipSuccess) return int(e);
        ms += t;
        fl += r.flops;
        by += r.bytes;
        ++n;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (flops) *flops = fl;
    if (bytes) *bytes = by;
    return 0;
}

}  // extern "C"
