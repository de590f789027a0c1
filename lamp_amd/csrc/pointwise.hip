// Bandwidth-bound pieces of the path: embedding gather (+ position add), LayerNorm, diagonal
// label read-out.  One 64-lane wave per row, 16-byte accesses per lane, four rows per workgroup.
#include <atomic>

#include "lamp_kernels.h"

namespace lamp {

__device__ __forceinline__ float wave_sum(float v) { return wave64_sum(v); }

// One gathered row by one wave: out[dst] = emb[tok] (+ pos_table[ps]); NaN rows for out-of-range indices.  With `fold`
// (EmbedFold, lamp_kernels.h) the same wave also writes the first encoder layer's HIDDEN row
//     hid[dst] = relu(e1[tok] (+ p1[ps]))      e1 = emb . W1^T, p1 = pos_table . W1^T + b1  (weights-only tables),
// i.e. relu((emb[tok] + pos_table[ps]) . W1^T + b1) of lamp/SubLayers.py:135 re-associated: the gather is a one-hot product,
// so the first FFN GEMM of the encoder folds into the tables and is not launched (api.hip: ffn_core(hidden_ready)).
__device__ __forceinline__ void gather_row(int64_t tok, int64_t ps, const float* __restrict__ emb, int n_vocab,
                                           const float* __restrict__ pos_table, int n_position, int d, float* __restrict__ out,
                                           int64_t dst, const EmbedFold& fold, int lane) {
    const bool ok = tok >= 0 && tok < n_vocab && ps >= 0 && (!pos_table || ps < n_position);
    const float4* e = reinterpret_cast<const float4*>(emb + (ok ? tok : 0) * d);
    const float4* q = pos_table ? reinterpret_cast<const float4*>(pos_table + (ok ? ps : 0) * d) : nullptr;
    float4* o = reinterpret_cast<float4*>(out + dst * d);
    const float nan = __builtin_nanf("");
    if (!fold.hid) {
        for (int c = lane; c < d / 4; c += 64) {
            float4 v = e[c];
            if (q) {
                const float4 w = q[c];
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            if (!ok) v = make_float4(nan, nan, nan, nan);
            o[c] = v;
        }
        return;
    }
    const float4* e1 = reinterpret_cast<const float4*>(fold.e1 + (ok ? tok : 0) * fold.dff);
    const float4* q1 = fold.p1 ? reinterpret_cast<const float4*>(fold.p1 + (ok ? ps : 0) * fold.dff) : nullptr;
    float4* h = reinterpret_cast<float4*>(fold.hid + dst * fold.dff);
    // the embedded row itself is read by ONE consumer, the residual of the layer's second GEMM: with the row maps that GEMM
    // gathers it in its epilogue and the row is not written at all (an out-of-range token's hidden row is NaN, so its output is)
    const int nx = fold.row_tok ? 0 : d / 4, nh = fold.dff / 4;
    if (fold.row_tok && lane == 0) {
        fold.row_tok[dst] = ok ? int(tok) : 0;
        fold.row_pos[dst] = ok ? int(ps) : 0;
    }
    // Two plain passes, the embedded row then the hidden row.  Both "all table reads in flight before the first store"
    // forms (one fused loop; clamped loads into registers up front) measured SLOWER in the forward: 23.6-28 us against 21-22
    // (profiles/r06_rejected_experiments.txt #1); the plain form is kept because it measures fastest.
    for (int c = lane; c < nx; c += 64) {
        float4 v = e[c];
        if (q) {
            const float4 w = q[c];
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        if (!ok) v = make_float4(nan, nan, nan, nan);
        o[c] = v;
    }
    for (int c = lane; c < nh; c += 64) {
        float4 u = e1[c];
        if (q1) {
            const float4 w = q1[c];
            u.x += w.x; u.y += w.y; u.z += w.z; u.w += w.w;
        }
        u = make_float4(fmaxf(u.x, 0.f), fmaxf(u.y, 0.f), fmaxf(u.z, 0.f), fmaxf(u.w, 0.f));
        if (!ok) u = make_float4(nan, nan, nan, nan);
        h[c] = u;
    }
}

// out[t, :] = emb[seq[t], :] (+ pos_table[pos[t], :])          lamp/Encoders.py:66,75
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ seq,
                                                    const int64_t* __restrict__ pos, int64_t n_tok,
                                                    const float* __restrict__ emb, int n_vocab,
                                                    const float* __restrict__ pos_table, int n_position,
                                                    int d, float* __restrict__ out, EmbedFold fold) {
    const int lane = threadIdx.x & 63;
    const int64_t t = int64_t(xcd_remap(blockIdx.x, gridDim.x)) * 4 + (threadIdx.x >> 6);  // XCD order of the next GEMM
    if (t >= n_tok) return;
    gather_row(seq[t], pos_table ? pos[t] : 0, emb, n_vocab, pos_table, n_position, d, out, t, fold, lane);
}

// One sample's extents, by one wave (see seq_plan_kernel): kl = 1 + its last non-PAD token, pl >= kl additionally covers the
// positions that carry a position index; writes the sample's bit-packed key mask.  The sample's 64-position chunks are
// fetched eight at a time -- sixteen loads in flight before the first ballot, from clamped addresses (a guarded load compiles
// to a branch plus a full s_waitcnt per element, which would serialise the round trips again) -- so a sample of up to 512
// positions costs ONE memory round trip.
__device__ __forceinline__ void plan_scan_sample(const int64_t* __restrict__ seq, const int64_t* __restrict__ pos, int b, int T,
                                                 int64_t seq_stride, const SeqPlan& sp, int lane, int& kl, int& pl) {
    constexpr int GROUP = 8;
    const int64_t* pos_or_seq = pos ? pos : seq;   // always a loadable address: no branch around the load
    const int64_t row0 = int64_t(b) * seq_stride;
    kl = 0;
    pl = 0;
    for (int base = 0; base < T; base += 64 * GROUP) {
        int64_t tok[GROUP], ps[GROUP];
#pragma unroll
        for (int u = 0; u < GROUP; ++u) {
            const int j = base + 64 * u + lane;
            const int64_t at = row0 + (j < T ? j : 0);
            tok[u] = seq[at];
            ps[u] = pos_or_seq[at];
        }
#pragma unroll
        for (int u = 0; u < GROUP; ++u) {
            const int c0 = base + 64 * u, j = c0 + lane;
            const bool t = j < T && tok[u] != 0;
            const bool a = t || (pos && j < T && ps[u] != 0);
            const unsigned long long mt = __ballot(t), ma = __ballot(a);
            if (mt) kl = c0 + 64 - __builtin_clzll(mt);
            if (ma) pl = c0 + 64 - __builtin_clzll(ma);
            // bit-packed key mask of this sample (bit = PAD token = blocked key; positions past T count as PAD)
            if (lane < 2 && c0 / 32 + lane < sp.words)
                sp.padbits[int64_t(b) * sp.words + c0 / 32 + lane] = ~unsigned(mt >> (32 * lane));
        }
    }
}

// Per-sample extents of a (possibly ragged) token batch, counted on the device -- no host round trip (SeqPlan,
// lamp_kernels.h).  klen[b] = 1 + the last position whose token is not PAD: keys past it are exactly masked
// (lamp/utils.py:26-34), so the enc-dec attention stops there.  plen[b] >= klen[b] additionally covers every position
// with a non-zero POSITION index: rows past plen[b] are all the same row, emb[PAD] + pos_table[0] pushed through the
// row-wise encoder (lamp/Encoders.py:64-79), which the packed encoder computes once (row n_tok) instead of per
// position.  packed: off[b] = sum of plen before b (the sample's first row of the packed matrix); else off[b] = b * T
// and plen[b] = T (padded layout, only klen is of interest).  rows[0] = off[nb]; rows[1] = off[nb] + 1 when some
// position is skipped (the shared PAD row is then live), else off[nb].  padbits[b][w]: the key mask of sample b, one bit per
// position (set = PAD token) -- the attention kernels then fetch ONE word per 32-key tile instead of testing 32 int64
// tokens (LAMP_MASK_BITS_U32 with a zero query stride).  ONE workgroup: a wave per sample, then a wave-level prefix sum.
__global__ __launch_bounds__(1024) void seq_plan_kernel(const int64_t* __restrict__ seq, const int64_t* __restrict__ pos,
                                                        int nb, int T, int64_t seq_stride, int packed, SeqPlan sp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (int b = wave; b < nb; b += nwave) {
        int kl, pl;
        plan_scan_sample(seq, pos, b, T, seq_stride, sp, lane, kl, pl);
        if (lane == 0) {
            sp.klen[b] = kl;
            sp.plen[b] = packed ? pl : T;
        }
    }
    __syncthreads();
    if (wave != 0) return;
    int carry = 0;
    bool skipped = false;
    for (int base = 0; base < nb; base += 64) {
        const int b = base + lane;
        const int v = b < nb ? sp.plen[b] : 0;
        skipped = skipped || (b < nb && v < T);
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int u = __shfl_up(incl, d, 64);
            if (lane >= d) incl += u;
        }
        if (b < nb) sp.off[b] = packed ? carry + incl - v : b * T;
        carry += __shfl(incl, 63, 64);
    }
    const bool any_skipped = __any(skipped);
    if (lane == 0) {
        const int n_tok = packed ? carry : nb * T;
        sp.off[nb] = n_tok;
        sp.rows[0] = n_tok;
        sp.rows[1] = n_tok + ((packed && any_skipped) ? 1 : 0);
    }
}

// Embedding gather of the packed encoder: flat position (b, j) with j < plen[b] -> packed row off[b] + j; the extra
// work item nb * T writes the shared PAD row (row n_tok = emb[PAD] + pos_table[0]) when some position is skipped.
__global__ __launch_bounds__(256) void embed_packed_kernel(const int64_t* __restrict__ seq, const int64_t* __restrict__ pos,
                                                           int nb, int T, const float* __restrict__ emb, int n_vocab,
                                                           const float* __restrict__ pos_table, int n_position, int d,
                                                           SeqPlan sp, float* __restrict__ out, EmbedFold fold) {
    const int lane = threadIdx.x & 63;
    const int64_t flat = int64_t(xcd_remap(blockIdx.x, gridDim.x)) * 4 + (threadIdx.x >> 6);
    const int64_t n_flat = int64_t(nb) * T;
    if (flat > n_flat) return;
    int64_t tok = 0, ps = 0, dst;
    if (flat == n_flat) {
        if (sp.rows[1] == sp.rows[0]) return;   // no position skipped: no PAD row
        dst = sp.rows[0];
    } else {
        const int b = int(flat / T), j = int(flat - int64_t(b) * T);
        if (j >= sp.plen[b]) return;
        tok = seq[flat];
        ps = pos_table ? pos[flat] : 0;
        dst = int64_t(sp.off[b]) + j;
    }
    gather_row(tok, ps, emb, n_vocab, pos_table, n_position, d, out, dst, fold, lane);
}

// The sequence plan AND the packed embedding gather in ONE launch (round 4: a dependent launch costs 5-8 us on this chain
// whatever it does -- the one-workgroup plan kernel was 7.9 us in front of a 10.7 us gather).
//   * workgroups 0 .. n_plan - 1 are the plan: a wave per sample computes (klen, plen, key-mask bits) exactly as
//     seq_plan_kernel and publishes plen[b] as an 8-byte granule {plen, epoch} with ONE agent-scope atomic store; wave 0 of
//     workgroup 0 then collects the granules, takes the prefix sum and publishes {off[b], epoch} the same way (and writes the
//     plain SeqPlan arrays the LATER kernels read).
//   * every other workgroup is four positions of the gather, as embed_packed_kernel: it requests its token first, then
//     polls the two granules of its sample (relaxed agent-scope 8-byte loads: no fence, no flag -- the tag IS the data's
//     validity; MI355X_MICROARCH.md "handoff-1to1"), and goes on with plen / off.  `epoch` is unique per launch, so a
//     granule left in the workspace by an earlier launch (or by nothing at all) never matches.
// Nothing here DEPENDS on dispatch order or placement: a wave polls a granule a few times (microseconds) and then computes what
// is missing itself (plan_prefix).  In the ordinary case -- one process on the GPU -- the plan workgroups are the launch's
// first and the granules are there at the first or second poll.
struct PlanGranules {
    unsigned long long* plen;   // [nb]
    unsigned long long* off;    // [nb + 2]: off[b]; [nb] = n_tok; [nb + 1] = 1 when some position is skipped
    unsigned epoch;
};
__device__ __forceinline__ void granule_put(unsigned long long* g, unsigned v, unsigned epoch) {
    __hip_atomic_store(g, (static_cast<unsigned long long>(epoch) << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// -> true and v when the granule carries this launch's epoch
__device__ __forceinline__ bool granule_try(const unsigned long long* g, unsigned epoch, unsigned& v) {
    const unsigned long long x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v = unsigned(x);
    return unsigned(x >> 32) == epoch;
}

// The prefix sum over the samples' row counts by ONE wave, from the plan waves' granules -- and, for a granule that has not
// arrived after a few polls, from this wave's own scan of that sample (plan_scan_sample: same values, also rewrites the same
// key-mask words).  No wave of this launch ever waits on another for longer than those few polls: a plan workgroup that has
// not been dispatched yet (a busy XCD, other processes sharing the GPU) costs its consumers one sample scan each, not a stall.
//   want >= 0: -> plen[want] in `mine`, off[want] in `off_mine` (samples 0 .. want are visited)
//   want <  0: all nb samples: -> total rows in `off_mine`, `any_skipped`; with `publish` the SeqPlan arrays and the off granules
__device__ __forceinline__ void plan_prefix(const int64_t* __restrict__ seq, const int64_t* __restrict__ plan_pos, int nb, int T,
                                            const SeqPlan& sp, const PlanGranules& gr, int lane, int want, bool publish,
                                            int polls, int& mine, int& off_mine, bool& any_skipped) {
    const int upto = want >= 0 ? want + 1 : nb;
    int carry = 0;
    bool skipped = false;
    mine = 0;
    off_mine = 0;
    for (int base = 0; base < upto; base += 64) {
        const int l = base + lane;
        const bool valid = l < upto;
        unsigned v = 0;
        bool have = !valid;
        for (int t = 0; t < polls && !__all(have); ++t) {
            if (!have) have = granule_try(gr.plen + l, gr.epoch, v);
            if (!__all(have)) {
                const int naps = t < 5 ? (1 << t) : 32;
                for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(4);
            }
        }
        unsigned long long missing = __ballot(!have);
        while (missing) {   // this wave's own count of a sample whose plan wave has not delivered
            const int i = __builtin_ctzll(missing);
            int kl, pl;
            plan_scan_sample(seq, plan_pos, base + i, T, T, sp, lane, kl, pl);
            if (lane == i) v = unsigned(pl);
            missing &= missing - 1;
        }
        const int pl = valid ? int(v) : 0;
        skipped = skipped || (valid && pl < T);
        int incl = pl;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const int u = __shfl_up(incl, dd, 64);
            if (lane >= dd) incl += u;
        }
        const int off = carry + incl - pl;
        if (publish && valid) {
            sp.off[l] = off;
            granule_put(gr.off + l, unsigned(off), gr.epoch);
        }
        if (want >= 0 && want >= base && want < base + 64) {
            mine = __shfl(pl, want - base, 64);
            off_mine = __shfl(off, want - base, 64);
        }
        carry += __shfl(incl, 63, 64);
    }
    any_skipped = __any(skipped);
    if (want < 0) off_mine = carry;
}

__global__ __launch_bounds__(256) void embed_plan_kernel(const int64_t* __restrict__ seq, const int64_t* __restrict__ pos,
                                                         const int64_t* __restrict__ plan_pos, int nb, int T,
                                                         const float* __restrict__ emb, int n_vocab,
                                                         const float* __restrict__ pos_table, int n_position, int d,
                                                         SeqPlan sp, PlanGranules gr, int n_plan, float* __restrict__ out,
                                                         EmbedFold fold) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (int(blockIdx.x) < n_plan) {
        for (int b = int(blockIdx.x) * 4 + wave; b < nb; b += n_plan * 4) {
            int kl, pl;
            plan_scan_sample(seq, plan_pos, b, T, T, sp, lane, kl, pl);
            if (lane == 0) {
                sp.klen[b] = kl;
                sp.plen[b] = pl;
                granule_put(gr.plen + b, unsigned(pl), gr.epoch);
            }
        }
        if (blockIdx.x != 0 || wave != 0) return;
        // the scanner: prefix sum over all samples, the plain SeqPlan arrays for the LATER kernels, the off granules for this one
        int mine, total;
        bool any_skipped;
        plan_prefix(seq, plan_pos, nb, T, sp, gr, lane, -1, true, 12, mine, total, any_skipped);
        if (lane == 0) {
            sp.off[nb] = total;
            sp.rows[0] = total;
            sp.rows[1] = total + (any_skipped ? 1 : 0);
            granule_put(gr.off + nb, unsigned(total), gr.epoch);
            granule_put(gr.off + nb + 1, any_skipped ? 1u : 0u, gr.epoch);
        }
        return;
    }
    const int nwg = int(gridDim.x) - n_plan;
    const int64_t flat = int64_t(xcd_remap(int(blockIdx.x) - n_plan, nwg)) * 4 + wave;
    const int64_t n_flat = int64_t(nb) * T;
    if (flat > n_flat) return;
    const bool pad_row = flat == n_flat;
    const int b = pad_row ? 0 : int(flat / T), j = pad_row ? 0 : int(flat - int64_t(b) * T);
    // the token first (its round trip overlaps the wait for the plan), then this sample's granules
    int64_t tok = pad_row ? 0 : seq[flat];
    int64_t ps = (pos_table && !pad_row) ? pos[flat] : 0;
    const unsigned long long* ga = pad_row ? gr.off + nb + 1 : gr.plen + b;   // plen[b]  | any position skipped
    const unsigned long long* gb = pad_row ? gr.off + nb : gr.off + b;        // off[b]   | n_tok
    unsigned va = 0, vb = 0;
    bool ok = false;
    {
        // Dense batch?  Every sample whose LAST position holds a token (or a position index) has plen = T; if that is all of
        // them nothing is skipped, off[b] = b T, and the gather needs nothing from the plan -- the fixed-length case pays
        // one more load beside the token's instead of the hand-off's round trips.
        bool dense = true;
        for (int base = 0; base < nb; base += 64) {
            const int bb = base + lane;
            const int64_t at = int64_t(bb < nb ? bb : 0) * T + (T - 1);
            const int64_t lt = seq[at], lp = plan_pos ? plan_pos[at] : 0;
            dense = dense && __all(bb >= nb || lt != 0 || lp != 0);
        }
        if (dense) {
            va = pad_row ? 0u : unsigned(T);
            vb = pad_row ? unsigned(n_flat) : unsigned(b) * unsigned(T);
            ok = true;
        }
    }
    // The scanner's two granules of this sample, polled a few times with a doubling interval (thousands of resident waves
    // polling every hundred cycles are a load of their own on the L2s) ...
    for (int spin = 0; spin < 10 && !ok; ++spin) {
        unsigned ta = 0, tb = 0;
        bool ka = false, kb = false;
        if (lane == 0) {
            ka = granule_try(ga, gr.epoch, ta);
            kb = granule_try(gb, gr.epoch, tb);
        }
        ok = __builtin_amdgcn_readfirstlane(int(ka && kb)) != 0;
        va = unsigned(__builtin_amdgcn_readfirstlane(int(ta)));
        vb = unsigned(__builtin_amdgcn_readfirstlane(int(tb)));
        if (!ok) {
            const int naps = 1 << (spin < 5 ? spin : 5);
            for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(4);   // 256 cycles per nap: ~0.1 us .. 3.4 us
        }
    }
    if (!ok) {
        // ... then this wave's own prefix sum over the plan waves' granules (plan_prefix: what is missing there it counts itself).
        // Seen when several PROCESSES share the GPU: the scanner, or a plan workgroup, queues behind other kernels' waves.
        int mine, off_mine;
        bool any_skipped;
        plan_prefix(seq, plan_pos, nb, T, sp, gr, lane, pad_row ? -1 : b, false, 4, mine, off_mine, any_skipped);
        va = pad_row ? (any_skipped ? 1u : 0u) : unsigned(mine);
        vb = unsigned(off_mine);
    }
    int64_t dst;
    if (pad_row) {
        if (va == 0) return;   // no position skipped: no PAD row
        dst = int64_t(vb);
    } else {
        if (j >= int(va)) return;
        dst = int64_t(vb) + j;
    }
    gather_row(tok, ps, emb, n_vocab, pos_table, n_position, d, out, dst, fold, lane);
}

// y = (x - mean) / sqrt(var_biased + eps) * g + b over the last dim.  Two-pass (mean, then centred
// sum of squares) on a register-resident row; NV float4 per lane.
// RG (ragged encoder, packed rows -- see seq_plan_kernel):  0 = plain rows 0 .. M-1.  1 = the row count comes from device
// memory (*m_dev <= M; M only sizes the launch).  2 = the last encoder LayerNorm: the launch walks the FLAT positions
// (b, j) of the padded [nb, T, d] encoder output; position (b, j) normalises packed row off[b] + j (or the shared PAD row
// n_tok for j >= plen[b]), writes it to y2[b, j] and -- live positions only -- back to its packed row in y, which the
// K/V projections read.  The PAD row is read by many positions and written by none.
template <int NV, bool DROP, int RG>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t M, int d,
                                                        const float* __restrict__ g,
                                                        const float* __restrict__ bta, float eps,
                                                        const float* __restrict__ res, int64_t r_mod,
                                                        float* __restrict__ y,
                                                        const float* __restrict__ w_out, int n_labels,
                                                        float* __restrict__ logits, DropoutSpec drop,
                                                        const int* __restrict__ m_dev, SeqPlan sp, int T,
                                                        float* __restrict__ y2) {
    const int lane = threadIdx.x & 63;
    const int nv = d / 4;
    float4 v[NV];
    auto load_row = [&](int64_t r) {
        const float4* xr = reinterpret_cast<const float4*>(x + r * d);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            v[i] = c < nv ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // rows in the XCD-contiguous order of the GEMM that produced x (and of the one that reads y): most of a row's
    // cache lines are then still in THIS XCD's L2 instead of a round trip to the Infinity Cache away
    int64_t row;
    int64_t flat = 0;
    bool live = true;
    if constexpr (RG == 1) {
        const int64_t Mrt = *m_dev < M ? *m_dev : M;
        const int nblk = int((Mrt + 3) / 4);
        if (int(blockIdx.x) >= nblk) return;
        row = int64_t(xcd_remap(blockIdx.x, nblk)) * 4 + (threadIdx.x >> 6);
        if (row >= Mrt) return;
    } else if constexpr (RG == 2) {
        if (sp.granules && blockIdx.x == 0)   // embed_plan_kernel's hand-off words: spent, back to "no epoch"
            for (int i = threadIdx.x; i < 2 * int(M / T) + 2; i += 256) sp.granules[i] = 0ull;
        flat = int64_t(xcd_remap(blockIdx.x, gridDim.x)) * 4 + (threadIdx.x >> 6);
        if (flat >= M) return;
        const int b = int(flat / T), j = int(flat - int64_t(b) * T);
        live = j < sp.plen[b];
        row = live ? int64_t(sp.off[b]) + j : int64_t(sp.rows[0]);
        // no position skipped anywhere: packed row == flat position, and the K / V projections read y2 (GemmParams::A_dense)
        if (sp.rows[1] == sp.rows[0]) y = nullptr;
    } else {
        row = int64_t(xcd_remap(blockIdx.x, gridDim.x)) * 4 + (threadIdx.x >> 6);
        if (row >= M) return;
    }
    load_row(row);
    const float4* rr = res ? reinterpret_cast<const float4*>(res + (r_mod > 0 ? row % r_mod : row) * d) : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        if constexpr (DROP) v[i] = drop4(v[i], row * d + 4 * c, drop);  // training: dropout on the sub-layer output
        if (rr && c < nv) {
            const float4 w = rr[c];
            v[i].x += w.x; v[i].y += w.y; v[i].z += w.z; v[i].w += w.w;
        }
    }
    float mean, rstd;
    ln_row_stats<NV>(v, lane, nv, d, eps, mean, rstd);   // lamp_kernels.h: shared with chain.hip, contraction off
    float4* yr = (y && live) ? reinterpret_cast<float4*>(y + row * d) : nullptr;
    float4* yf = RG == 2 ? reinterpret_cast<float4*>(y2 + flat * d) : nullptr;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const float4* b4 = reinterpret_cast<const float4*>(bta);
    // optional fused label read-out (lamp/Models.py:124-126): logits[row] = <LN(row), w_out[row % L]>
    const float4* w4 = w_out ? reinterpret_cast<const float4*>(w_out + (row % n_labels) * d) : nullptr;
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const float4 o = ln_row_apply(v[i], mean, rstd, g4[c], b4[c]);
            if (yr) yr[c] = o;
            if constexpr (RG == 2) yf[c] = o;
            if (w4) dot += dot4_nocontract(o, w4[c]);
        }
    }
    if (w4) {
        dot = wave_sum(dot);
        if (lane == 0) logits[row] = dot;
    }
}

// logits[b, i] = <y[b, i, :], w[i, :]>        lamp/Models.py:124-126 (diagonal of y . w^T)
__global__ __launch_bounds__(256) void diag_kernel(const float* __restrict__ y, const float* __restrict__ w,
                                                   int64_t n_rows, int L, int d, float* __restrict__ logits) {
    const int lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int i = int(row % L);
    const float4* yr = reinterpret_cast<const float4*>(y + row * d);
    const float4* wr = reinterpret_cast<const float4*>(w + int64_t(i) * d);
    float s = 0.f;
    for (int c = lane; c < d / 4; c += 64) {
        const float4 a = yr[c], b = wr[c];
        s += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
    }
    s = wave_sum(s);
    if (lane == 0) logits[row] = s;
}

// test_epoch post-processing (test.py:49-51): probs = sigmoid(logits) and, per row, the summed
// binary-cross-entropy-with-logits  max(x,0) - x*z + log1p(exp(-|x|))  against the gold 0/1 targets.
__global__ __launch_bounds__(256) void sigmoid_bce_kernel(const float* __restrict__ logits,
                                                          const float* __restrict__ targets, int64_t n_rows, int L,
                                                          float* __restrict__ probs, float* __restrict__ row_loss) {
    const int lane = threadIdx.x & 63;
    const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    float loss = 0.f;
    for (int i = lane; i < L; i += 64) {
        const float x = logits[row * L + i];
        if (probs) probs[row * L + i] = 1.0f / (1.0f + expf(-x));
        if (targets) {
            const float z = targets[row * L + i];
            loss += fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
        }
    }
    if (row_loss) {
        loss = wave_sum(loss);
        if (lane == 0) row_loss[row] = loss;
    }
}

// Prior label graph (utils/data_loader.py:37-47).  Pass 1: adj = I, blocked = 1 - I.  Pass 2: one workgroup per
// training sample walks its n^2 ordered label pairs and marks both directions; every writer stores the same
// value, so the races are benign and the result does not depend on scheduling.
__global__ __launch_bounds__(256) void prior_graph_init_kernel(int L, float* __restrict__ adj,
                                                               uint8_t* __restrict__ blocked) {
    const int64_t n = int64_t(L) * L;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
        const bool diag = (i / L) == (i % L);
        adj[i] = diag ? 1.f : 0.f;
        if (blocked) blocked[i] = diag ? 0 : 1;
    }
}

__global__ __launch_bounds__(256) void prior_graph_pairs_kernel(const int64_t* __restrict__ ids,
                                                                const int64_t* __restrict__ offsets,
                                                                int64_t n_samples, int L, float* __restrict__ adj,
                                                                uint8_t* __restrict__ blocked) {
    for (int64_t s = blockIdx.x; s < n_samples; s += gridDim.x) {
        const int64_t lo = offsets[s];
        const int64_t n = offsets[s + 1] - lo;
        for (int64_t p = threadIdx.x; p < n * n; p += 256) {
            const int64_t a = ids[lo + p / n], b = ids[lo + p % n];
            if (a == b || uint64_t(a) >= uint64_t(L) || uint64_t(b) >= uint64_t(L)) continue;
            adj[a * L + b] = 1.f;
            if (blocked) blocked[a * L + b] = 0;
        }
    }
}

static inline int grid4(int64_t rows, unsigned* g) {
    const int64_t n = (rows + 3) / 4;
    if (n > 0x7fffffffLL) return LAMP_E_DIMS;
    *g = unsigned(n);
    return 0;
}

// fold (nullable): also write the first encoder layer's hidden rows from the folded tables (gather_row)
static int check_fold(const EmbedFold* fold, const float* pos_table, EmbedFold* out) {
    *out = EmbedFold{nullptr, nullptr, 0, nullptr, nullptr, nullptr};
    if (!fold || !fold->hid) return 0;
    if (!fold->e1 || (pos_table && !fold->p1)) return LAMP_E_NULL;
    if (fold->dff <= 0 || (fold->dff & 3)) return LAMP_E_UNSUPPORTED;
    if (!aligned16(fold->e1) || !aligned16(fold->hid) || (fold->p1 && !aligned16(fold->p1))) return LAMP_E_ALIGN;
    if ((fold->row_tok == nullptr) != (fold->row_pos == nullptr)) return LAMP_E_NULL;
    *out = *fold;
    if (!pos_table) out->p1 = nullptr;
    return 0;
}
static inline double embed_bytes(int64_t n_tok, int d, bool pos, const EmbedFold& f) {
    if (f.hid && f.row_tok) return double(n_tok) * (16.0 + 8.0 + 4.0 * f.dff * (pos ? 3 : 2));   // hidden rows + row maps only
    return double(n_tok) * (16.0 + 4.0 * d * (pos ? 3 : 2) + (f.hid ? 4.0 * f.dff * (pos ? 3 : 2) : 0.0));
}

int launch_embed(const int64_t* seq, const int64_t* pos, int64_t n_tok, const float* emb, int n_vocab,
                 const float* pos_table, int n_position, int d, float* out, hipStream_t s, const EmbedFold* fold) {
    if (n_tok <= 0 || d <= 0 || n_vocab <= 0) return LAMP_E_DIMS;
    if (d & 3) return LAMP_E_UNSUPPORTED;
    if (!seq || !emb || !out || (pos_table && !pos)) return LAMP_E_NULL;
    if (!aligned16(emb) || !aligned16(out) || (pos_table && !aligned16(pos_table))) return LAMP_E_ALIGN;
    EmbedFold f;
    if (int e = check_fold(fold, pos_table, &f)) return e;
    unsigned g;
    if (int e = grid4(n_tok, &g)) return e;
    ProfScope prof(LAMP_K_EMBED, 0.0, embed_bytes(n_tok, d, pos_table != nullptr, f), s);
    hipLaunchKernelGGL(embed_kernel, dim3(g), dim3(256), 0, s, seq, pos, n_tok, emb, n_vocab, pos_table,
                       n_position, d, out, f);
    return int(hipGetLastError());
}

int launch_layernorm(const float* x, int64_t M, int d, const float* g, const float* b, float eps,
                     const float* residual, int64_t r_mod, float* y, hipStream_t s, const float* w_out, int n_labels,
                     float* logits, const DropoutSpec* drop, const int* m_dev, const SeqPlan* scatter, int T,
                     float* y_flat) {
    if (M <= 0 || d <= 0) return LAMP_E_DIMS;
    if ((d & 3) || d > 4096) return LAMP_E_UNSUPPORTED;
    if (!x || !g || !b || (!y && !w_out) || (w_out && (!logits || n_labels <= 0))) return LAMP_E_NULL;
    if (scatter && (!y_flat || !y || T <= 0 || residual || w_out || (drop && drop->threshold > 0))) return LAMP_E_UNSUPPORTED;
    if (m_dev && (scatter || residual || w_out || (drop && drop->threshold > 0))) return LAMP_E_UNSUPPORTED;
    if (!aligned16(x) || (y && !aligned16(y)) || !aligned16(g) || !aligned16(b) || (residual && !aligned16(residual)) ||
        (w_out && !aligned16(w_out)) || (y_flat && !aligned16(y_flat)))
        return LAMP_E_ALIGN;
    unsigned grid;
    if (int e = grid4(M, &grid)) return e;
    ProfScope prof(LAMP_K_LAYERNORM, 0.0, (scatter ? 12.0 : 8.0) * double(M) * d, s);
    const int nv = (d / 4 + 63) / 64;
    const bool dr = drop && drop->threshold > 0;
    const DropoutSpec ds = dr ? *drop : DropoutSpec{0u, 1.f, 0u};
    const SeqPlan sp = scatter ? *scatter : SeqPlan{};
#define LAMP_LN_LAUNCH(NV_)                                                                                          \
    do {                                                                                                              \
        if (dr)                                                                                                       \
            hipLaunchKernelGGL((layernorm_kernel<NV_, true, 0>), dim3(grid), dim3(256), 0, s, x, M, d, g, b, eps,      \
                               residual, r_mod, y, w_out, n_labels, logits, ds, m_dev, sp, T, y_flat);                \
        else if (scatter)                                                                                             \
            hipLaunchKernelGGL((layernorm_kernel<NV_, false, 2>), dim3(grid), dim3(256), 0, s, x, M, d, g, b, eps,     \
                               residual, r_mod, y, w_out, n_labels, logits, ds, m_dev, sp, T, y_flat);                \
        else if (m_dev)                                                                                               \
            hipLaunchKernelGGL((layernorm_kernel<NV_, false, 1>), dim3(grid), dim3(256), 0, s, x, M, d, g, b, eps,     \
                               residual, r_mod, y, w_out, n_labels, logits, ds, m_dev, sp, T, y_flat);                \
        else                                                                                                          \
            hipLaunchKernelGGL((layernorm_kernel<NV_, false, 0>), dim3(grid), dim3(256), 0, s, x, M, d, g, b, eps,     \
                               residual, r_mod, y, w_out, n_labels, logits, ds, m_dev, sp, T, y_flat);                \
    } while (0)
    if (nv <= 1)
        LAMP_LN_LAUNCH(1);
    else if (nv <= 2)
        LAMP_LN_LAUNCH(2);
    else if (nv <= 4)
        LAMP_LN_LAUNCH(4);
    else if (nv <= 8)
        LAMP_LN_LAUNCH(8);
    else
        LAMP_LN_LAUNCH(16);
#undef LAMP_LN_LAUNCH
    return int(hipGetLastError());
}

// seq: [nb, T] tokens (row stride seq_stride), pos: nullable [nb, T] position indices (same stride).
int launch_seq_plan(const int64_t* seq, const int64_t* pos, int nb, int T, int64_t seq_stride, bool packed,
                    const SeqPlan& sp, hipStream_t s) {
    if (nb <= 0 || T <= 0) return LAMP_E_DIMS;
    if (!seq || !sp.klen || !sp.plen || !sp.off || !sp.rows || !sp.padbits) return LAMP_E_NULL;
    if (sp.words != (T + 31) / 32) return LAMP_E_DIMS;
    if (int64_t(nb) * T >= 0x7fffffffLL) return LAMP_E_DIMS;   // packed row indices are 32-bit
    const int threads = nb >= 16 ? 1024 : (nb >= 4 ? 256 : 64);
    hipLaunchKernelGGL(seq_plan_kernel, dim3(1), dim3(threads), 0, s, seq, pos, nb, T, seq_stride, packed ? 1 : 0, sp);
    return int(hipGetLastError());
}

int launch_embed_packed(const int64_t* seq, const int64_t* pos, int nb, int T, const float* emb, int n_vocab,
                        const float* pos_table, int n_position, int d, const SeqPlan& sp, float* out, hipStream_t s,
                        const EmbedFold* fold) {
    if (nb <= 0 || T <= 0 || d <= 0 || n_vocab <= 0) return LAMP_E_DIMS;
    if (d & 3) return LAMP_E_UNSUPPORTED;
    if (!seq || !emb || !out || (pos_table && !pos)) return LAMP_E_NULL;
    if (!aligned16(emb) || !aligned16(out) || (pos_table && !aligned16(pos_table))) return LAMP_E_ALIGN;
    EmbedFold f;
    if (int e = check_fold(fold, pos_table, &f)) return e;
    unsigned g;
    const int64_t n_tok = int64_t(nb) * T;
    if (int e = grid4(n_tok + 1, &g)) return e;
    ProfScope prof(LAMP_K_EMBED, 0.0, embed_bytes(n_tok, d, pos_table != nullptr, f), s);
    hipLaunchKernelGGL(embed_packed_kernel, dim3(g), dim3(256), 0, s, seq, pos, nb, T, emb, n_vocab, pos_table,
                       n_position, d, sp, out, f);
    return int(hipGetLastError());
}

// Plan + packed gather in one launch (embed_plan_kernel).  `granules`: 2 * nb + 2 unsigned 64-bit words of workspace, any content.
int launch_embed_plan(const int64_t* seq, const int64_t* pos, bool plan_uses_pos, int nb, int T, const float* emb, int n_vocab,
                      const float* pos_table, int n_position, int d, const SeqPlan& sp, unsigned long long* granules, float* out,
                      hipStream_t s, const EmbedFold* fold) {
    if (nb <= 0 || T <= 0 || d <= 0 || n_vocab <= 0) return LAMP_E_DIMS;
    if (d & 3) return LAMP_E_UNSUPPORTED;
    if (!seq || !emb || !out || !granules || (pos_table && !pos) || (plan_uses_pos && !pos)) return LAMP_E_NULL;
    if (!aligned16(emb) || !aligned16(out) || (pos_table && !aligned16(pos_table)) || (reinterpret_cast<uintptr_t>(granules) & 7u))
        return LAMP_E_ALIGN;
    if (!sp.klen || !sp.plen || !sp.off || !sp.rows || !sp.padbits) return LAMP_E_NULL;
    EmbedFold f;
    if (int e = check_fold(fold, pos_table, &f)) return e;
    unsigned g;
    const int64_t n_tok = int64_t(nb) * T;
    if (int e = grid4(n_tok + 1, &g)) return e;
    const int n_plan = nb <= 4 ? 1 : (nb + 3) / 4 < 64 ? (nb + 3) / 4 : 64;   // a wave per sample up to 256 samples, then several each
    static std::atomic<unsigned> epoch_counter{0x5eed0001u};
    unsigned epoch = epoch_counter.fetch_add(1, std::memory_order_relaxed);
    if (epoch == 0) epoch = epoch_counter.fetch_add(1, std::memory_order_relaxed);
    PlanGranules gr{granules, granules + nb, epoch};
    ProfScope prof(LAMP_K_EMBED, 0.0, 16.0 * double(n_tok) + embed_bytes(n_tok, d, pos_table != nullptr, f), s);
    hipLaunchKernelGGL(embed_plan_kernel, dim3(g + unsigned(n_plan)), dim3(256), 0, s, seq, pos, plan_uses_pos ? pos : nullptr, nb, T, emb,
                       n_vocab, pos_table, n_position, d, sp, gr, n_plan, out, f);
    return int(hipGetLastError());
}

int launch_sigmoid_bce(const float* logits, const float* targets, int64_t n_rows, int L, float* probs,
                       float* row_loss, hipStream_t s) {
    if (n_rows <= 0 || L <= 0) return LAMP_E_DIMS;
    if (!logits || (!probs && !row_loss) || (row_loss && !targets)) return LAMP_E_NULL;
    unsigned g;
    if (int e = grid4(n_rows, &g)) return e;
    hipLaunchKernelGGL(sigmoid_bce_kernel, dim3(g), dim3(256), 0, s, logits, targets, n_rows, L, probs, row_loss);
    return int(hipGetLastError());
}

int launch_diag(const float* y, const float* w, int B, int L, int d, float* logits, hipStream_t s) {
    if (B <= 0 || L <= 0 || d <= 0) return LAMP_E_DIMS;
    if (d & 3) return LAMP_E_UNSUPPORTED;
    if (!y || !w || !logits) return LAMP_E_NULL;
    if (!aligned16(y) || !aligned16(w)) return LAMP_E_ALIGN;
    unsigned g;
    const int64_t rows = int64_t(B) * L;
    if (int e = grid4(rows, &g)) return e;
    ProfScope prof(LAMP_K_DIAG, 2.0 * rows * d, 4.0 * (double(rows) * d + double(L) * d + rows), s);
    hipLaunchKernelGGL(diag_kernel, dim3(g), dim3(256), 0, s, y, w, rows, L, d, logits);
    return int(hipGetLastError());
}

int launch_prior_graph(const int64_t* ids, const int64_t* offsets, int64_t n_samples, int L, float* adj,
                       uint8_t* blocked, hipStream_t s) {
    if (n_samples < 0 || L <= 0) return LAMP_E_DIMS;
    if (!adj || (n_samples > 0 && (!ids || !offsets))) return LAMP_E_NULL;
    const int64_t cells = int64_t(L) * L;
    const unsigned g0 = unsigned(cells / 256 + 1 < 4096 ? cells / 256 + 1 : 4096);
    hipLaunchKernelGGL(prior_graph_init_kernel, dim3(g0), dim3(256), 0, s, L, adj, blocked);
    if (n_samples > 0) {
        const unsigned g1 = unsigned(n_samples < 65536 ? n_samples : 65536);
        hipLaunchKernelGGL(prior_graph_pairs_kernel, dim3(g1), dim3(256), 0, s, ids, offsets, n_samples, L, adj,
                           blocked);
    }
    return int(hipGetLastError());
}

}  // namespace lamp
