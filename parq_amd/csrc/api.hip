// C ABI of libparq_hip.so (include/parq_hip.h): handle, weight arena, workspace carving and
// the host-side orchestration of the kernel chain.  Nothing here synchronises with the
// device, allocates device memory or touches torch; the caller owns every buffer.
#include "../../include/parq_hip.h"
#include "common.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace parq;

namespace {
// attention mode 4: smallest row sum of probabilities (relative to the row's reference maximum) it keeps.  Measured on the kernels at
// BASELINE cfg 3's size (profiles/r05_split8_guard_sweep.txt): rows down to a sum of 255 keep the mode within 2.4e-5 of mode 1 on smooth
// (FPN-like) features and within 6e-6 on white noise; at sums of 75 .. 87 smooth features are at 3.5e-5 (the round-4 threshold of 64 let
// those through).  The benchmark's synthetic workload: smallest sum 3200 .. 4600.
constexpr float kPeakyL = 256.f;

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(PARQ_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr int64_t kAlign = 64;   // floats (256 B)
int64_t align_up(int64_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

struct WeightRef { const float* p; int64_t n; };

// offsets (in floats) into the packed arena
struct LayerW {
    int64_t self_in_w, self_in_b, self_out_w, self_out_b;
    int64_t cross_in_w, cross_in_b, cross_out_w, cross_out_b;
    int64_t lin1_w, lin1_b, lin2_w, lin2_b;
    int64_t n1_w, n1_b, n2_w, n2_b, n3_w, n3_b;
    int64_t kv_whi, kv_wlo;           // fp16 hi/lo split of in_proj_weight[C:3C] (each 2C*C halfs)
    // the position MLP's last layer folded into its two consumers (inference): (x + h W2^T + b2) W^T = x W^T + h (W W2)^T + W b2
    int64_t self_in_w2, self_in_b2;   // [3C][C]: rows < 2C = W_qk W2, rows >= 2C zero | [3C]: b_qk + W_qk b2, then b_v
    int64_t cross_q_w2, cross_q_b2;   // [C][C] = W_q W2 | [C] = b_q + W_q b2
    // norm1 pushed through the cross-attention query projection (chain.hip seam_tile; inference):
    //   q  = rstd1 (U - mean1 q_s) + V:   U = sa (Wq g1 Wo_s)^T + tgt (Wq g1)^T + qu_b,   V = pe_h cross_q_w2^T + qv_b
    int64_t qg_w, qo_w, qu_b, q_s, qv_b;       // [C][C] Wq diag(g1) | [C][C] (Wq diag(g1)) Wo_self | [C] (Wq diag(g1)) bo_self | [C] row sums | [C] cross_q_b2 + Wq beta1
};
struct Arena {
    std::vector<LayerW> layers;
    int64_t refpoint, pe0_w, pe0_b, pe2_w, pe2_b;
    int64_t heads1_w, heads1_b;       // [NH1][C]: centre.0 | rotation.0 | sem_cls | size | zero pad
    int64_t gn1_g, gn1_b;             // [2][C]  (centre, rotation) GroupNorm after layer 0
    int64_t heads2_w;                 // [2][C][C]  centre.4 | rotation.4
    int64_t gn2_g, gn2_b;             // [2][C]
    int64_t heads3_w, heads3_b;       // [2][6][C], [2][6]: centre.8 padded to 6 rows | rotation.8
    int64_t mean_sizes, dim_t;
    int64_t early_begin = 0, early_end = 0;   // [early_begin, early_end): gradients final after phase 1 of parq_backward (nl = 1)
    int64_t rowmajor_total;           // end of the tensors above = size of the gradient arena (same layout)
    int64_t tile_off;                 // start of the tile-ordered mirror: matrix at offset o has its chain.hip copy at tile_off + o
    int64_t half_off = -1;            // start of the fp16 hi / lo mirror (LinearArgs::Wh) or -1: widths whose chain contracts over >= 768
    int64_t hscale_off = -1;          // its column scales: matrix at offset o has them at hscale_off + hscale_slot(o)
    int64_t total;
};

struct Workspace {
    int64_t T_cl, kv, ref, ref_next, emb, pe_h, pos, tgt, qkv, attn, xa, qc, xb, ffn, xc;
    int64_t h1, h2, gn_sums, ln1, ln2, flash;     // gn_sums: [2 layers][B][2 heads][2] fp64 moments; ln*: [M][2]
    int64_t kvc, flags;               // split-fp16 K/V cache, int flags (overflow)
    int64_t seam_flags, lnp1;         // the norm1 seam inside one launch (chain.hip seam_tile): flags [M / 16][4], fp64 partial row sums of xa ([M][C/64][2])
    int64_t xsplit;                   // fp16 hi/lo copy of the tokens for the large-C K/V projection (kvproj_big.hip), else empty
    int64_t cam, ind;                 // a captured forward: this call's cameras and pointer block (common.hpp CallPtrs), left by its prologue
    int64_t total;
    int self_split, cross_split;
    // per-iteration activations live in [iter_begin, iter_end); a training forward keeps one copy per iteration:
    // iteration k > 0 uses the same offsets shifted by stash + (k - 1) * (iter_end - iter_begin) - iter_begin
    int64_t sa, ln3, lse_s, lse_c, refk;
    int64_t iter_begin, iter_end, stash;
    // backward scratch (training workspace only)
    int64_t g_a, g_b, g_c, g_pos, g_tmp, g_ffh, g_h1, g_h2, g_z, g_act, g_h3, g_qkv, g_emb, g_ref, g_D, g_bs, wT, g_kv, g_dqp, g_drop, kv_train, g_do, g_res, g_dq, g_Dall, g_kvmax, g_pack, g_mat;
    int64_t g_set_stride;             // floats between two copies of the per-iteration backward scratch [g_a, g_drop]
    int g_sets;                       // copies of it: iterations of the batched backward run on that many streams at once
    bool bwd_batched;                 // cross-attention backward of all iterations in one launch (shared layer weights, split cache)
    int64_t train_total;
    int64_t shift(int k) const { return k == 0 ? 0 : stash + (int64_t)(k - 1) * (iter_end - iter_begin) - iter_begin; }
};

struct ProfEvent { hipEvent_t a, b; int which; };

}  // namespace

struct parq_ctx {
    parq_config cfg;
    int C, Q, H, dh, F, I, ncls, NH1, nl;
    ScaleBox sb;
    std::map<std::string, WeightRef> named;
    Arena ar;
    const float* arena = nullptr;     // device arena after pack
    bool packed = false;
    bool derived_valid = false;       // build_derived_weights ran since the last parq_pack_weights
    bool prepared = false;
    bool emb_valid = false;           // workspace emb holds pos2posemb3d of the chained reference points
    int ref_state = 0;                // 0: none, 1: ws.ref valid
    int attn_mode = 1;                // 0: fp32 MFMA, 1: split fp16x3, 2: fp16, 3: bf16 (1..3: head dims 64 and 256), 4: split with fp8 cross terms
    int kv16_state = 1;               // what the arena's 16-bit W_kv copy currently holds (same numbering)
    float drop_p = 0.f;               // training dropout (decoder layer, transformer_parq.py:339-386) and its base seed
    uint32_t drop_seed = 0;
    uint32_t site_seed(int k, int site) const { return rng_stream(drop_seed, (uint32_t)(k * 8 + site)); }
    // split K/V cache in use: head dim 64 (all cache modes) or head dim 256 in split mode (a head = 4 virtual heads of 64)
    bool cache_mode() const { return attn_mode >= 1 && (dh == 64 || dh == 256); }
    int vheads() const { return C / 64; }            // heads of the cache layout
    int terms() const { return attn_mode == 1 || attn_mode == 4 ? 3 : 1; }
    // mode 4 (flash_split8.hip) where its kernels apply — inference, head dim 64, d = 256, whole 64-key stages — and mode 1 otherwise:
    // 8 = the stage cache with fp8 cross-term planes (same size as the split cache for such N)
    // training: only where the backward reads the forward's cache itself (bwd_reads_cache: `stage_bwd`) — there is no fp32 rebuild from stages
    // per-head tiers (parq_set_head_tiers): heads of `safe_mask` run the fp16 x 3 kernel on the split layout inside a mode-4 forward
    // (inference only; a training forward with any safe head runs as mode 1, and so does a forward whose heads are all safe)
    int terms_for(int64_t N, bool train, bool stage_bwd = false) const {
        const bool m4 = attn_mode == 4 && (!train || stage_bwd) && C == 256 && flash_split8_supported(dh, (int)(N > INT32_MAX ? 0 : N));
        if (!m4 || safe_heads() == all_heads() || (train && safe_heads() != 0)) return terms();
        return 8;
    }
    uint32_t seam_epoch = 0;          // one value per seam launch of this handle (chain.hip seam_tile flags); 0 is never used
    uint32_t next_epoch() { if (++seam_epoch == 0) ++seam_epoch; return seam_epoch; }
    uint32_t safe_mask = 0;           // bit h: head h runs the fp16 x 3 kernel where the handle is in attention mode 4
    int peaky_poison = 0;             // mode-4 heads that meet a too-peaked row: write that iteration's outputs (and what follows) as NaN
    uint32_t all_heads() const { return H >= 32 ? 0xffffffffu : ((1u << H) - 1u); }
    uint32_t safe_heads() const { return safe_mask & all_heads(); }
    bool mixed_tiers(int64_t N, bool train, bool stage_bwd = false) const { return terms_for(N, train, stage_bwd) == 8 && safe_heads() != 0; }
    int w16_state() const { return attn_mode == 4 ? 1 : attn_mode; }      // what the 16-bit copy of W_kv has to hold
    int kind() const { return attn_mode == 3 ? kBF16 : kF16; }
    int* range_mirror = nullptr;      // host-visible word raised when outputs are poisoned (parq_set_range_mirror)
    int* progress_word = nullptr;     // host-visible word that receives progress_epoch once a forward can raise no more flags (parq_set_progress)
    int progress_epoch = 0;
    bool seam_fusion = false;         // parq_set_seam_fusion: in-launch hand-offs of the chain (off by default since round 6: every dependent stage its own launch)
    bool bwd_batched_env = true;      // parq_set_backward_batched (the parity test compares the two settings)
    int bwd_streams = 8;              // parq_set_backward_streams: iterations of the chain backward in flight at once (1 = in turn)
    float dim_t_host[128];            // 10000^(2*(i//2)/128): uploaded by parq_pack_weights from this persistent buffer (no stream sync)
    hipStream_t cap_stream = nullptr;       // parq_forward_capture records on a stream of the handle's own
    hipStream_t aux_stream[7] = {nullptr};  // batched backward: iterations 1 .. g_sets-1 (mod g_sets) of a phase run here, 0 on the caller's stream
    hipEvent_t fork_ev = nullptr, join_ev[7] = {nullptr};
    hipEvent_t bucket_done[2] = {nullptr, nullptr};   // parq_backward: gradient bucket 0 / 1 final on the caller's stream (parq_backward_wait_bucket)
    bool bucket_recorded = false;
    hipEvent_t iter_done[16] = {nullptr};   // parq_forward_train records one after every iteration (parq_wait_iteration)
    bool iter_recorded[16] = {false};
    bool profiling = false;
    std::vector<ProfEvent> events;
    double prof_ms[PARQ_PROF_COUNT] = {0};
    int64_t prof_n[PARQ_PROF_COUNT] = {0};
};

namespace {

void build_arena(parq_ctx* c) {
    Arena& a = c->ar;
    int64_t off = 0;
    auto take = [&](int64_t n) { int64_t o = off; off += align_up(n); return o; };
    const int64_t C = c->C, F = c->F, Q = c->Q;
    // Order = when a tensor's GRADIENT is final inside parq_backward, so that data-parallel training can all-reduce the arena
    // in two contiguous buckets (parq_grad_bucket): first everything that is only complete at the END of the backward (phase 2
    // of the iterations and the hoisted K/V projection: reference points, position MLP, both in-projections, self out-proj,
    // norm1), then what is complete after PHASE 1 (cross out-proj, FFN, norm2, norm3, all heads) — with shared layer weights
    // (nl = 1) the split is one offset, `early_begin`; the derived inference tensors (no gradient) come last.
    a.refpoint = take(Q * 3);
    a.pe0_w = take(C * 384); a.pe0_b = take(C); a.pe2_w = take(C * C); a.pe2_b = take(C);
    a.layers.resize(c->nl);
    for (auto& L : a.layers) {
        L.self_in_w = take(3 * C * C); L.self_in_b = take(3 * C);
        L.self_out_w = take(C * C);    L.self_out_b = take(C);
        L.cross_in_w = take(3 * C * C); L.cross_in_b = take(3 * C);
        L.n1_w = take(C); L.n1_b = take(C);
        if (&L == &a.layers.back()) a.early_begin = off;
        L.cross_out_w = take(C * C);   L.cross_out_b = take(C);
        L.lin1_w = take(F * C); L.lin1_b = take(F);
        L.lin2_w = take(C * F); L.lin2_b = take(C);
        L.n2_w = take(C); L.n2_b = take(C); L.n3_w = take(C); L.n3_b = take(C);
    }
    a.heads1_w = take((int64_t)c->NH1 * C); a.heads1_b = take(c->NH1);
    a.gn1_g = take(2 * C); a.gn1_b = take(2 * C);
    a.heads2_w = take(2 * C * C);
    a.gn2_g = take(2 * C); a.gn2_b = take(2 * C);
    a.heads3_w = take(2 * 6 * C); a.heads3_b = take(12);
    a.early_end = off;
    a.mean_sizes = take((int64_t)c->cfg.num_mean_sizes * 3);
    a.dim_t = take(128);
    for (auto& L : a.layers) {
        L.kv_whi = take(C * C); L.kv_wlo = take(C * C);
        L.self_in_w2 = take(3 * C * C); L.self_in_b2 = take(3 * C);
        L.cross_q_w2 = take(C * C); L.cross_q_b2 = take(C);
        L.qg_w = take(C * C); L.qo_w = take(C * C); L.qu_b = take(C); L.q_s = take(C); L.qv_b = take(C);
    }
    a.rowmajor_total = off;
    // tile-ordered copies of the matrices the per-iteration chain multiplies by (LinearArgs::Wp): same offsets, shifted
    a.tile_off = off;
    a.total = 2 * off;
    // fp16 hi / lo mirror for the fp16 x 3 chain tile (chain.hip chain_linear_h3_kernel: contractions over 1024 or 768 only)
    if (C % 256 == 0 && C >= 768) {
        a.half_off = 2 * off;
        a.hscale_off = (3 * off + 63) / 64 * 64;
        a.total = a.hscale_off + ((off + 255) / 256 + 63) / 64 * 64 + 64;
    }
}

// column scales of the fp16 hi / lo mirror: the matrix at arena offset o ([N][K], K >= 768) keeps its N scales at hscale_off + this
// (16-byte aligned; disjoint: the next matrix starts N K / 256 >= 3 N slots further)
inline int64_t hscale_slot(int64_t off) { return ((off >> 8) + 3) & ~(int64_t)3; }

bool chain_h3_on() {
    static const bool on = [] { const char* e = dev_env("PARQ_CHAIN_H3"); return !(e && e[0] == '0'); }();     // 0: fp32 MFMA tiles at every width (A/B)
    return on;
}

bool kvproj_big_on() {
    static const bool on = [] { const char* e = dev_env("PARQ_KVPROJ_BIG"); return !(e && e[0] == '0'); }();   // 0: keep the tiled kernel at C > 256 (A/B)
    return on;
}

// Training: the cross-attention backward of all recurrent iterations can run as ONE launch when the iterations share the layer
// weights (hence K / V) and the split-precision kernel applies (head dim 64, long key axis); parq_set_backward_batched(h, 0) restores the
// per-iteration launches.
int bwd_sets(int I) {          // concurrent iterations of the batched chain backward (development build: PARQ_BWD_SETS)
    static const int want = [] { const char* e = dev_env("PARQ_BWD_SETS"); return e && atoi(e) > 0 ? atoi(e) : 8; }();
    int n = want < I ? want : I;
    return n < 1 ? 1 : (n > 8 ? 8 : n);
}
bool bwd_batched_ok(const parq_ctx* c, int64_t N);
// the batched cross-attention backward at head dim 64 takes K / V straight from the forward's 16-bit cache: no fp32 rebuild
// (kvsplit_to_f32: 0.57 ms and 1.5 GB of traffic per 4-scene step at cfg 3), no second copy of K / V in the training workspace
bool bwd_reads_cache(const parq_ctx* c, const Workspace& ws) { return ws.bwd_batched && c->cache_mode() && c->dh == 64 && c->nl == 1; }
bool bwd_batched_ok(const parq_ctx* c, int64_t N) {
    // head dim 64: the register-resident split kernel (long key axes); head dim 256: the composition from split-precision GEMMs
    return c->bwd_batched_env && c->nl == 1 && ((c->dh == 64 && N >= 2048) || c->dh == 256) && c->I > 1 && c->I <= 16;
}

int carve_workspace(const parq_ctx* c, int B, int V, int h, int w, Workspace* ws) {
    const int64_t C = c->C, Q = c->Q, F = c->F;
    const int64_t M = (int64_t)B * Q;
    const int64_t N = (int64_t)V * h * w;
    int64_t off = 0;
    auto take = [&](int64_t n) { int64_t o = off; off += align_up(n); return o; };
    ws->T_cl = take((int64_t)B * V * 12 * 2);      // float64 poses
    const bool split_mode = c->cache_mode();
    ws->kv = take(split_mode ? 0 : (int64_t)c->nl * B * 2 * N * C);      // fp32 head-major K/V (fp32 mode only)
    ws->ref = take(M * 3); ws->ref_next = take(M * 3);
    ws->emb = take(M * 384); ws->pe_h = take(M * C); ws->pos = take(M * C);
    ws->tgt = take(M * C); ws->qkv = take(M * 3 * C); ws->attn = take(M * C);
    ws->xa = take(M * C); ws->qc = take(M * C);
    ws->xb = take(M * C); ws->ffn = take(M * F);
    ws->xc = take(M * C);
    ws->h1 = take(M * c->NH1); ws->h2 = take(M * 2 * C);
    ws->gn_sums = take((int64_t)2 * B * 4 * kGnSlots * 2);  // doubles: [2 layers][B][2 heads][kGnSlots][2]
    ws->ln1 = take(M * 2); ws->ln2 = take(M * 2); ws->ln3 = take(M * 2);
    ws->sa = take(M * C);
    ws->lse_s = take((int64_t)B * c->H * flash_lq_pad((int)Q)); ws->lse_c = take((int64_t)B * c->H * flash_lq_pad((int)Q));
    ws->refk = take(M * 3);
    ws->iter_begin = ws->emb; ws->iter_end = off;
    const int cus = device_num_cus();
    ws->self_split = flash_pick_splits(B, c->H, c->Q, c->Q, c->dh, cus);
    ws->cross_split = split_mode ? (c->dh == 256 ? flash_split256_pick_splits(B, c->H, c->Q, (int)N, cus)
                                                 : flash_split_pick_splits(B, c->H, c->Q, (int)N, cus))
                                 : flash_pick_splits(B, c->H, c->Q, (int)N, c->dh, cus);
    ws->kvc = take(split_mode ? (int64_t)(c->nl * kvsplit_cache_bytes(B, c->vheads(), (int)N, c->terms()) / sizeof(float)) : 0);
    ws->flags = take(64);
    ws->seam_flags = take((M / 16 + 1) * 4);         // directly behind `flags`: the forward prologue clears both in one go
    ws->lnp1 = take(M * (C / 64) * 4);
    ws->xsplit = take(split_mode && kvproj_big_on() ? (int64_t)kvproj_big_scratch_floats(B, (int)N, C) : 0);
    const size_t fs = flash_scratch_bytes(B, c->H, c->Q, c->dh, ws->self_split);
    size_t fc = flash_scratch_bytes(B, c->H, c->Q, c->dh, ws->cross_split);
    if (c->attn_mode == 4 && c->dh == 64 && c->H <= 16) {
        // per-head tiers: the mode-4 heads and the fp16 x 3 heads run as two launches, each with the key-split count that fills the
        // chip with ITS heads, each with its own partials — room for the worst division of the heads
        for (int nf = 1; nf < c->H; ++nf) {
            const size_t two = flash_scratch_bytes(B, nf, c->Q, c->dh, flash_split_pick_splits(B, nf, c->Q, (int)N, cus)) +
                               flash_scratch_bytes(B, c->H - nf, c->Q, c->dh, flash_split_pick_splits(B, c->H - nf, c->Q, (int)N, cus));
            fc = two > fc ? two : fc;
        }
    }
    ws->flash = take((int64_t)((fs > fc ? fs : fc) / sizeof(float)));
    ws->cam = take((int64_t)B * V * 6);
    ws->ind = take(16);                               // 8 pointers
    ws->total = off;
    // ---- training extras: activation stash of iterations 1..I-1, backward scratch
    ws->stash = take((int64_t)(c->I - 1) * (ws->iter_end - ws->iter_begin));
    const int64_t NH1 = c->NH1;
    const int64_t g_set_begin = off;
    ws->g_a = take(M * C); ws->g_b = take(M * C); ws->g_c = take(M * C); ws->g_pos = take(M * C); ws->g_tmp = take(M * C);
    ws->g_ffh = take(M * F); ws->g_h1 = take(M * NH1); ws->g_h2 = take(M * 2 * C); ws->g_z = take(M * 2 * C);
    ws->g_act = take(M * 2 * C); ws->g_h3 = take(M * 16); ws->g_qkv = take(M * 3 * C); ws->g_emb = take(M * 384);
    ws->g_ref = take(M * 3); ws->g_D = take((int64_t)B * c->H * flash_lq_pad((int)Q)); ws->g_bs = take((int64_t)B * 2 * 2 * 2);
    ws->g_drop = take(M * C);
    // Given the stash the iterations are independent (reference points are detached between them), and the chain backward of one
    // iteration is ~70 small dependent launches: in the batched backward up to 8 iterations run at once (measured 1 / 2 / 4 / 8 at once: 29.3 / 27.5 / 27.7 / 27.2 ms per step), each on its own stream
    // with its own copy of the scratch above (weight gradients meet in the arena through atomics).  Only where every dW product
    // takes the row-split kernel (its plain read-modify-write form and the 64 x 64-tile kernel are not safe for that).
    ws->g_set_stride = off - g_set_begin;
    ws->g_sets = (bwd_batched_ok(c, N) && c->dh == 64 && (int64_t)F * C < (1 << 19) && (int64_t)3 * C * C < (1 << 19) && true)
                     ? (bwd_sets(c->I) < c->bwd_streams ? bwd_sets(c->I) : c->bwd_streams) : 1;
    take((ws->g_sets - 1) * ws->g_set_stride);
    // transposed weight copies of one layer: heads1 [C][NH1], heads2 2x[C][C], lin1^T [C][F], lin2^T [F][C],
    // cross_out^T, cross_q^T, self_out^T [C][C] each, self_in^T [C][3C], pe2^T [C][C], pe0^T [384][C]
    ws->wT = take(C * NH1 + 2 * C * C + 2 * C * F + 3 * C * C + 3 * C * C + C * C + 384 * C);
    ws->g_kv = take((int64_t)c->nl * B * 2 * N * C);
    ws->bwd_batched = bwd_batched_ok(c, N);
    // fp32 K / V rebuilt from the 16-bit cache for the backward — not needed where the backward reads the cache (bwd_reads_cache)
    ws->kv_train = take(split_mode && !bwd_reads_cache(c, *ws) ? (int64_t)c->nl * B * 2 * N * C : 0);
    // batched cross-attention backward (see parq_backward): per-iteration dO, residual gradient, dQ and D rows, and one set of
    // dQ partials per iteration
    ws->bwd_batched = bwd_batched_ok(c, N);
    const int64_t nit = ws->bwd_batched ? c->I : 1;
    // (batched head-dim-64 backward: a workgroup of attn_bwd_split2_kernel keeps one slot for its group of key blocks)
    ws->g_dqp = take(nit * (int64_t)attn_bwd_dq_partial_floats(B, c->H, (int)Q, (int)N, c->dh, ws->bwd_batched && c->dh == 64));
    ws->g_do = take(ws->bwd_batched ? nit * M * C : 0);
    ws->g_res = take(ws->bwd_batched ? nit * M * C : 0);
    ws->g_dq = take(ws->bwd_batched ? nit * M * C : 0);
    ws->g_Dall = take(ws->bwd_batched ? nit * (int64_t)B * c->H * flash_lq_pad((int)Q) : 0);
    ws->g_pack = take(ws->bwd_batched ? (int64_t)attn_bwd_pack_floats(B, c->H, (int)Q, (int)nit) : 0);
    // head dims without a register-resident attention backward (not 32 / 64): score-matrix scratch of the materialised path
    {
        int64_t mat = (c->dh == 64 || c->dh == 32) ? 0 : (int64_t)attn_bwd_mat_scratch_floats((int)Q, (int)(N > Q ? N : Q), c->dh);
        if (ws->bwd_batched && c->dh == 256) {        // S^T / dP^T of all iterations of one (scene, head): 3.1 GB at cfg 3
            const int64_t b256 = (int64_t)attn_bwd_batched256_scratch_floats(c->I, (int)Q, (int)N);
            mat = b256 > mat ? b256 : mat;
        }
        ws->g_mat = take(mat);
    }
    ws->g_kvmax = take(4);                            // [0] bits of max |dK|, |dV| (batched backward), [1] the derived scale
    ws->train_total = off;
    return PARQ_OK;
}

int check_scene(const parq_ctx* c, const parq_scene* s) {
    if (!s) return fail(PARQ_ERR_ARG, "scene is NULL");
    if (s->B < 1 || s->V < 1 || s->h < 2 || s->w < 2) return fail(PARQ_ERR_ARG, "bad scene dims B=%d V=%d h=%d w=%d", s->B, s->V, s->h, s->w);
    if (!s->tokens || !s->camera || !s->T_camera_pseudoCam || !s->T_world_pseudoCam || !s->T_world_local)
        return fail(PARQ_ERR_ARG, "scene has a NULL tensor");
    if ((int64_t)s->B * c->Q > (1 << 22)) return fail(PARQ_ERR_ARG, "B*Q too large");
    return PARQ_OK;
}

struct Prof {
    parq_ctx* c; hipStream_t s; int which; hipEvent_t a = nullptr, b = nullptr;
    Prof(parq_ctx* c_, hipStream_t s_, int w) : c(c_), s(s_), which(w) {
        if (c->profiling) {
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
            (void)hipEventRecord(a, s);
        }
    }
    ~Prof() {
        if (a && b) {
            (void)hipEventRecord(b, s);
            c->events.push_back({a, b, which});
        }
    }
};

LinearArgs lin(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
               int M, int N, int K) {
    LinearArgs a;
    memset(&a, 0, sizeof(a));
    a.X = X; a.ldx = ldx; a.W = W; a.ldw = ldw; a.bias = bias; a.Y = Y;
    a.M = M; a.N = N; a.K = K;
    a.rows_per_batch = M; a.y_batch = 0; a.y_row = ldy; a.col_blk = N; a.y_blk = 0;
    return a;
}

// the arena's 16-bit copy of W_kv follows the attention mode: hi/lo split, or one round-to-nearest fp16 / bf16 copy
int settle_weight_state(parq_ctx* c, hipStream_t s) {
    if (c->cache_mode() && c->kv16_state != c->w16_state()) {
        const float* A = c->arena;
        const int C = c->C;
        for (int li = 0; li < c->nl; ++li) {
            const LayerW& L = c->ar.layers[li];
            float* Aw = const_cast<float*>(A);
            if (c->terms() == 3) HIPCHK(launch_split_f32(A + L.cross_in_w + C * C, Aw + L.kv_whi, Aw + L.kv_wlo, 2 * C * C, s));
            else HIPCHK(launch_cvt16(A + L.cross_in_w + C * C, Aw + L.kv_whi, 2 * C * C, c->kind(), s));
        }
        c->kv16_state = c->w16_state();
    }
    return PARQ_OK;
}

int do_prepare(parq_ctx* c, const parq_scene* sc, float* wsp, const Workspace& ws, hipStream_t s, bool train = false,
               const parq_outputs* call_outs = nullptr, bool forward_call = false) {
    const float* A = c->arena;
    const int B = sc->B, V = sc->V;
    const int64_t N = (int64_t)V * sc->h * sc->w;
    const int C = c->C;
    {
        // T_camera_local (float64), sigmoid(refpoint) tiled over the scenes, its sine embedding for iteration 0 and the cleared range
        // flags: one launch
        Prof p(c, s, PARQ_PROF_OTHER);
        // call_outs (parq_forward_replay): the launch also leaves this call's token / output pointers and cameras in the workspace for
        // the recorded iterations (common.hpp CallPtrs)
        PrologueCall pc;
        memset(&pc, 0, sizeof(pc));
        if (forward_call) {             // this call's epoch (the progress word's value once no flag can follow, LinearArgs::pub_epoch)
            pc.ind = reinterpret_cast<const void**>(wsp + ws.ind);
            pc.ptrs.p[7] = reinterpret_cast<const void*>((uintptr_t)(unsigned)c->progress_epoch);
        }
        if (call_outs) {
            pc.cam_src = sc->camera; pc.cam_dst = wsp + ws.cam; pc.ncam = B * V * 6;
            pc.ind = reinterpret_cast<const void**>(wsp + ws.ind);
            pc.ptrs.p[0] = sc->tokens; pc.ptrs.p[1] = call_outs->pred_logits; pc.ptrs.p[2] = call_outs->center_unnormalized;
            pc.ptrs.p[3] = call_outs->size_unnormalized; pc.ptrs.p[4] = call_outs->ortho6d; pc.ptrs.p[5] = call_outs->sem_cls_prob;
            pc.ptrs.p[6] = call_outs->coord_pos;
        }
        HIPCHK(launch_forward_prologue(sc->T_camera_pseudoCam, sc->T_world_pseudoCam, sc->T_world_local, B, V,
                                       reinterpret_cast<double*>(wsp + ws.T_cl), A + c->ar.refpoint, c->Q, wsp + ws.ref, A + c->ar.dim_t,
                                       wsp + ws.emb, wsp + ws.flags, (int)(ws.lnp1 - ws.flags), s,      // the 64 flag words and the seam flags behind them
                                       pc.ind ? &pc : nullptr));
    }
    // hoisted K/V in-projection of the memory tokens: one GEMM per distinct layer,
    // written head-major [b][{K heads, V heads}][N][dh] so the attention kernel streams contiguous panels
    if ((int64_t)B * N > (int64_t)INT32_MAX) return fail(PARQ_ERR_ARG, "B*N too large");
    { const int rc = settle_weight_state(c, s); if (rc) return rc; }
    for (int li = 0; li < c->nl; ++li) {
        Prof p(c, s, PARQ_PROF_KV_PROJ);
        const LayerW& L = c->ar.layers[li];
        if (c->cache_mode()) {
            char* cache = reinterpret_cast<char*>(wsp + ws.kvc) + (size_t)li * kvsplit_cache_bytes(B, c->vheads(), (int)N, c->terms());
            if (kvproj_big_on() && kvproj_big_scratch_floats(B, (int)N, C) > 0)
                HIPCHK(launch_kvproj_big(sc->tokens, A + L.kv_whi, A + L.kv_wlo, A + L.cross_in_b + C, B, (int)N, C, cache,
                                         reinterpret_cast<int*>(wsp + ws.flags), wsp + ws.xsplit, s, c->terms(), c->kind()));
            else
                HIPCHK(launch_kvproj_split(sc->tokens, A + L.kv_whi, A + L.kv_wlo, A + L.cross_in_b + C, B, (int)N, C, c->vheads(),
                                           cache, reinterpret_cast<int*>(wsp + ws.flags), s,
                                           c->mixed_tiers(N, train, train && bwd_reads_cache(c, ws)) ? 11 : c->terms_for(N, train, train && bwd_reads_cache(c, ws)),
                                           c->kind(), c->safe_heads()));
        } else {
            LinearArgs a = lin(sc->tokens, C, A + L.cross_in_w + (int64_t)C * C, C, A + L.cross_in_b + C,
                               wsp + ws.kv + (int64_t)li * B * 2 * N * C, 0, (int)(B * N), 2 * C, C);
            a.rows_per_batch = (int)N; a.y_batch = 2 * N * C; a.y_row = c->dh; a.col_blk = c->dh; a.y_blk = N * c->dh;
            HIPCHK(launch_linear(a, 1, s));
        }
    }
    c->prepared = true;
    c->emb_valid = true;              // the prologue left pos2posemb3d(ws.ref) in ws.emb
    c->ref_state = 1;
    return PARQ_OK;
}

// Weights DERIVED from the packed tensors for the inference chain (chain.hip): the position MLP's last layer folded into its two
// consumers and the tile-ordered mirror of every matrix the chain multiplies by.  Built lazily by the first inference iteration
// after parq_pack_weights (a training loop re-packs every step and never reads them: its forward keeps the reference's layer order
// and row-major weights), on the caller's stream, into the caller's arena.
int build_derived_weights(parq_ctx* c, hipStream_t s) {
    float* A = const_cast<float*>(c->arena);
    const Arena& ar = c->ar;
    const int64_t C = c->C, F = c->F;
    // float64-accumulated products, rounded once
    for (int li = 0; li < c->nl; ++li) {
        const LayerW& L = c->ar.layers[li];
        HIPCHK(launch_fold_pos_weights(A + L.self_in_w, A + L.self_in_b, A + ar.pe2_w, A + ar.pe2_b, (int)(2 * C), (int)C, A + L.self_in_w2, A + L.self_in_b2, s));
        HIPCHK(hipMemcpyAsync(A + L.self_in_b2 + 2 * C, A + L.self_in_b + 2 * C, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, s));   // b_v unchanged
        HIPCHK(launch_fold_pos_weights(A + L.cross_in_w, A + L.cross_in_b, A + ar.pe2_w, A + ar.pe2_b, (int)C, (int)C, A + L.cross_q_w2, A + L.cross_q_b2, s));
        // norm1 pushed through the cross-attention query projection (LayerW)
        HIPCHK(launch_ln_fold(A + L.cross_in_w, A + L.cross_q_b2, A + L.n1_w, A + L.n1_b, (int)C, (int)C, A + L.qg_w, A + L.q_s, A + L.qv_b, s));
        HIPCHK(launch_matmul_fold(A + L.qg_w, A + L.self_out_w, A + L.self_out_b, (int)C, (int)C, A + L.qo_w, A + L.qu_b, s));
    }
    // tile-ordered mirror of the chain's matrices (chain.hip: one contiguous KB per wave-wide fragment load)
    {
        const int64_t T = ar.tile_off;
        auto tile = [&](int64_t off, int64_t N, int64_t K) { return launch_pack_w_tiles(A + off, K, (int)N, (int)K, A + T + off, s); };
        for (int li = 0; li < c->nl; ++li) {
            const LayerW& L = c->ar.layers[li];
            HIPCHK(tile(L.self_in_w, 3 * C, C)); HIPCHK(tile(L.self_out_w, C, C));
            HIPCHK(tile(L.cross_in_w, C, C));                                      // the query rows (K / V rows: kv_whi / kv_wlo)
            HIPCHK(tile(L.cross_out_w, C, C));
            HIPCHK(tile(L.self_in_w2, 3 * C, C)); HIPCHK(tile(L.cross_q_w2, C, C));
            HIPCHK(tile(L.qg_w, C, C)); HIPCHK(tile(L.qo_w, C, C));
            if (F % 16 == 0) { HIPCHK(tile(L.lin1_w, F, C)); HIPCHK(tile(L.lin2_w, C, F)); }
        }
        HIPCHK(tile(ar.pe0_w, C, 384)); HIPCHK(tile(ar.pe2_w, C, C));
        if (c->NH1 % 16 == 0) HIPCHK(tile(ar.heads1_w, c->NH1, C));
        HIPCHK(tile(ar.heads2_w, C, C)); HIPCHK(tile(ar.heads2_w + C * C, C, C));
    }
    if (ar.half_off >= 0) {
        const int64_t H = ar.half_off, SC = ar.hscale_off;
        // gamma / beta / bias: the LayerNorm in front of the launch that multiplies by this matrix, folded (chain.hip pack_w_half_kernel):
        // gamma into the mirror unless an addend follows the LayerNorm, W beta + bias into the N floats behind the column scales
        auto half = [&](int64_t off, int64_t N, int64_t K, const float* gamma = nullptr, const float* beta = nullptr, const float* bias = nullptr) {
            if (N % 16 != 0 || !(K == 1024 || K == 768)) return hipSuccess;
            float* scp = A + SC + hscale_slot(off);
            return launch_pack_w_half(A + off, K, (int)N, (int)K, A + H + off, scp, s, gamma, beta, bias, beta ? scp + N : nullptr);
        };
        for (int li = 0; li < c->nl; ++li) {
            const LayerW& L = c->ar.layers[li];
            HIPCHK(half(L.self_in_w, 3 * C, C)); HIPCHK(half(L.self_out_w, C, C));
            HIPCHK(half(L.cross_in_w, C, C, nullptr, A + L.n1_b, A + L.cross_in_b));                 // q rows: norm1, then + pos
            HIPCHK(half(L.cross_out_w, C, C));
            HIPCHK(half(L.lin1_w, F, C, A + L.n2_w, A + L.n2_b, A + L.lin1_b)); HIPCHK(half(L.lin2_w, C, F));
        }
        HIPCHK(half(ar.pe2_w, C, C));
        if (c->nl == 1) HIPCHK(half(ar.heads1_w, c->NH1, C, A + ar.layers[0].n3_w, A + ar.layers[0].n3_b, A + ar.heads1_b));      // one norm3 in front of the shared heads
        else HIPCHK(half(ar.heads1_w, c->NH1, C));                     // a norm3 per layer: no fold (that launch keeps the fp32 tile)
        HIPCHK(half(ar.heads2_w, C, C)); HIPCHK(half(ar.heads2_w + C * C, C, C));
    }
    c->derived_valid = true;
    return PARQ_OK;
}

// shift: offset (floats) of this iteration's activation copy (0 in inference: every iteration reuses one set);
// emb_next: where the decode kernel leaves the next iteration's sine embedding; train: also keep what backward needs
// View-sharded scenes (parq_iterate_sharded): an iteration runs in three phases around two exchanges the caller performs.
//   phase bit 1: position MLP + project/sample of the LOCAL views -> `out` = [M*C undivided sums | M valid-view counts]
//   phase bit 2: `in` = the ranks' pairs added up -> tgt; self-attention block, cross-attention over the LOCAL keys ->
//                `out` = [M*C normalised attention outputs | B*H*Lq_pad log2 log-sum-exp rows]
//   phase bit 4: `in` = `nranks` such records -> merged attention output; cross out-proj, FFN, heads, decode
// mask 7 with in = out = nullptr is the ordinary iteration.
struct ShardIO { int mask = 7; const float* in = nullptr; float* out = nullptr; int nranks = 1; };

int do_iterate(parq_ctx* c, const parq_scene* sc, float* wsp, const Workspace& ws, int layer_num, const float* ref,
               bool emb_valid, const parq_outputs* o, float* ref_out, hipStream_t s, int64_t shift = 0,
               float* emb_next = nullptr, bool train = false, const ShardIO& sh = ShardIO(), int64_t captured_row0 = -1,
               bool last_of_forward = false) {
    // captured_row0 >= 0 (parq_forward_capture): the iteration is being RECORDED — tokens, cameras and the six outputs are taken from the
    // workspace's CallPtrs block / camera copy that the prologue of every replay fills in (`sc->tokens`, `sc->camera`, `o` are unused);
    // captured_row0 = first output row of this iteration (k * B * Q)
    const void* const* ind = captured_row0 >= 0 ? reinterpret_cast<const void* const*>(wsp + ws.ind) : nullptr;
    const float* cam_in = ind ? wsp + ws.cam : sc->camera;
    const int64_t row0 = captured_row0 >= 0 ? captured_row0 : 0;
    float* wi = wsp + shift;
    const bool sharded = sh.mask != 7;
    if (!emb_next) emb_next = wi + ws.emb;
    const float dp = train ? c->drop_p : 0.f;          // dropout exists in training only (nn.Dropout / MHA dropout)
    const float* A = c->arena;
    const Arena& ar = c->ar;
    if (!train && !c->derived_valid) { const int rc = build_derived_weights(c, s); if (rc) return rc; }
    // tile-ordered mirror of the chain's matrices (LinearArgs::Wp); the training forward reads the row-major tensors
    const float* TP = train ? nullptr : A + ar.tile_off;
    // fp16 hi / lo mirror (LinearArgs::Wh: the fp16 x 3 tile of chain.hip for contractions over 1024 / 768) — inference, every attention
    // mode but the exact-fp32 one (mode 0 keeps fp32 MFMA products in the chain as well)
    const bool h3 = !train && ar.half_off >= 0 && c->attn_mode != 0 && chain_h3_on();
    // fold: what build_derived_weights folded of the LayerNorm in front of this launch (LinearArgs::wh_fold)
    auto halfw = [&](LinearArgs& la, int64_t off, int fold = 0) {
        if (h3) {
            la.Wh = A + ar.half_off + off; la.wh_scale = A + ar.hscale_off + hscale_slot(off);
            if (fold) { la.wh_bias = la.wh_scale + la.N; la.wh_fold = fold; }
        }
    };
    const int li = c->cfg.share_weights ? 0 : layer_num;
    const LayerW& L = ar.layers[li];
    const int B = sc->B, C = c->C, Q = c->Q, H = c->H, dh = c->dh, F = c->F;
    const int M = B * Q;
    const int64_t N = (int64_t)sc->V * sc->h * sc->w;
    const float eps = 1e-5f;
    double* gn1 = reinterpret_cast<double*>(wi + ws.gn_sums);          // [B][2][2]
    double* gn2 = gn1 + (int64_t)B * 4 * kGnSlots;
    // the two consumers of the position embedding, either with pos = pe2(h) as an addend on the A operand (reference order,
    // transformer_parq.py:372-377) or with the position MLP's last layer folded into them (second operand pair h x (W W2)^T)
    auto self_in_args = [&](bool fold) {
        LinearArgs a = lin(wi + ws.tgt, C, A + L.self_in_w, C, A + (fold ? L.self_in_b2 : L.self_in_b), wi + ws.qkv, 3 * C, M, 3 * C, C);
        a.ldx2 = C; a.x2_ncols = 2 * C; a.Wp = TP ? TP + L.self_in_w : nullptr; halfw(a, L.self_in_w);
        if (fold) { a.X2 = wi + ws.pe_h; a.W2 = A + L.self_in_w2; a.W2p = TP ? TP + L.self_in_w2 : nullptr; }
        else a.X2 = wi + ws.pos;
        return a;
    };
    auto cross_q_args = [&](bool fold) {
        LinearArgs a = lin(wi + ws.xa, C, A + L.cross_in_w, C, A + (fold ? L.cross_q_b2 : L.cross_in_b), wi + ws.qc, C, M, C, C);
        a.ln_gamma = A + L.n1_w; a.ln_beta = A + L.n1_b; a.ln_stats_out = wi + ws.ln1; a.norm_eps = eps;
        a.ldx2 = C; a.x2_ncols = C; a.Wp = TP ? TP + L.cross_in_w : nullptr; halfw(a, L.cross_in_w, 2);
        if (fold) { a.X2 = wi + ws.pe_h; a.W2 = A + L.cross_q_w2; a.W2p = TP ? TP + L.cross_q_w2 : nullptr; }
        else a.X2 = wi + ws.pos;
        return a;
    };
    // inference only (the backward differentiates the unfolded layers from the stashed pos), and only where chain.hip has the kernels
    static const bool fold_off = [] { const char* e = dev_env("PARQ_FOLD_POS"); return e && e[0] == '0'; }();
    const bool fold_pos = !train && !fold_off && chain_linear_supported(self_in_args(true), 1) && chain_linear_supported(cross_q_args(true), 1);
    // the norm1 seam (self out-projection | cross-attention query projection) as ONE launch (chain.hip seam_tile): inference at d = 256
    static const int seams = [] { const char* e = dev_env("PARQ_FUSE_SEAMS"); return e ? atoi(e) : 1; }();      // 0: self out-projection and query projection as two launches (A/B)
    const bool seam_ok = !train && !sharded && fold_pos && C == 256 && TP != nullptr && M % 16 == 0 && c->seam_fusion;

    if (sh.mask & 1) {
    // K3: sine embedding (written by the previous iteration's decode kernel when chained) -> position MLP
    // (transformer_parq.py:317)
    if (!emb_valid) { Prof p(c, s, PARQ_PROF_OTHER); HIPCHK(launch_posemb(ref, A + ar.dim_t, M, wi + ws.emb, s)); }
    LinearArgs pe1 = lin(wi + ws.emb, 384, A + ar.pe0_w, 384, A + ar.pe0_b, wi + ws.pe_h, C, M, C, 384);
    pe1.relu = 1; pe1.Wp = TP ? TP + ar.pe0_w : nullptr;
    float* sample_out = sharded ? sh.out : wi + ws.tgt;
    float* sample_cnt = sharded ? sh.out + (int64_t)M * C : nullptr;
    // the position MLP's first layer and project + sample are independent (both read only what the previous decode left): one
    // launch when the chain kernels apply; the profiled pass keeps them apart so that the per-kernel groups stay per kernel
    static const bool fuse_off = [] { const char* e = dev_env("PARQ_FUSE_PE1_SAMPLE"); return e && e[0] == '0'; }();
    bool fused = false;
    if (!train && !fuse_off && !c->profiling) {
        const hipError_t e = launch_pe1_sample(pe1, sc->tokens, reinterpret_cast<const double*>(wsp + ws.T_cl), cam_in, ref, c->sb, B, sc->V,
                                               sc->h, sc->w, C, Q, sample_out, o->coord_pos, gn1, B * 8 * kGnSlots, sample_cnt, s, ind, row0 * 3);
        if (e == hipSuccess) fused = true;
        else if (e != hipErrorNotSupported) return fail(PARQ_ERR_HIP, "launch_pe1_sample failed: %s", hipGetErrorString(e));
    }
    {
        Prof p(c, s, PARQ_PROF_LINEAR);
        if (!fused) HIPCHK(launch_linear(pe1, 1, s));
        if (!fold_pos) {
            LinearArgs a = lin(wi + ws.pe_h, C, A + ar.pe2_w, C, A + ar.pe2_b, wi + ws.pos, C, M, C, C);
            a.Wp = TP ? TP + ar.pe2_w : nullptr; halfw(a, ar.pe2_w);
            HIPCHK(launch_linear(a, 1, s));
        }
    }
    // K4+K5: project + sample (transformer_parq.py:321); also clears this iteration's GroupNorm moments
    if (!fused) {
        Prof p(c, s, PARQ_PROF_PROJECT_SAMPLE);
        HIPCHK(launch_project_sample_f64(sc->tokens, reinterpret_cast<const double*>(wsp + ws.T_cl), cam_in, ref, c->sb,
                                         B, sc->V, sc->h, sc->w, C, Q, sample_out, o->coord_pos, gn1, B * 8 * kGnSlots, s, sample_cnt, ind, row0 * 3));
    }
    // view-sharded: this rank's fp16-range flag travels as the last float of the record (the caller's all-reduce adds the ranks')
    if (sharded) HIPCHK(launch_shard_range_flag(c->cache_mode() && c->kind() == kF16 ? reinterpret_cast<const int*>(wsp + ws.flags) : nullptr,
                                                sh.out + (int64_t)M * C + M, s));
    }   // phase bit 1
    if (sh.mask & 2) {
    if (sharded) {
        Prof p(c, s, PARQ_PROF_PROJECT_SAMPLE);
        HIPCHK(launch_sample_finalize(sh.in, sh.in + (int64_t)M * C, M, C, wi + ws.tgt, s, sh.in + (int64_t)M * C + M,
                                      c->cache_mode() && c->kind() == kF16 ? reinterpret_cast<int*>(wsp + ws.flags) : nullptr));
    }
    // K6: self-attention, q = k = tgt + pos, v = tgt (transformer_parq.py:372-376)
    {
        Prof p(c, s, PARQ_PROF_LINEAR);
        LinearArgs a = self_in_args(fold_pos);
        HIPCHK(launch_linear(a, 1, s));
    }
    FlashArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.B = B; fa.H = H; fa.Lq = Q; fa.dh = dh;
    fa.out = wi + ws.sa; fa.out_batch = (int64_t)Q * C; fa.out_row = C;
    {
        Prof p(c, s, PARQ_PROF_SELF_ATTN);
        static const bool self_one = [] { const char* e = dev_env("PARQ_SELF_ONE_LAUNCH"); return !(e && e[0] == '0'); }();
        if (dh <= 64 || (self_one && (dh == 128 || dh == 256) && Q <= 1024)) {
            // one launch: a workgroup = 16 queries of one head against all keys (the 256-key self-attention has no long axis to split)
            HIPCHK(launch_self_attn(wi + ws.qkv, 3 * C, B, H, Q, dh, wi + ws.sa, C, s, train ? wi + ws.lse_s : nullptr, dp,
                                    c->site_seed(layer_num, 0)));
        } else {
            fa.q = wi + ws.qkv;         fa.q_batch = (int64_t)Q * 3 * C; fa.q_head = dh; fa.q_row = 3 * C;
            fa.k = wi + ws.qkv + C;     fa.k_batch = fa.q_batch; fa.k_head = dh; fa.k_row = 3 * C;
            fa.v = wi + ws.qkv + 2 * C; fa.v_batch = fa.q_batch; fa.v_head = dh; fa.v_row = 3 * C;
            fa.Lk = Q; fa.nsplit = ws.self_split;
            const int64_t lp = flash_lq_pad(Q);
            fa.o_part = wsp + ws.flash;
            fa.m_part = fa.o_part + (int64_t)B * H * fa.nsplit * dh * lp;
            fa.l_part = fa.m_part + (int64_t)B * H * fa.nsplit * lp;
            fa.lse = train ? wi + ws.lse_s : nullptr;
            fa.drop_p = dp; fa.drop_seed = c->site_seed(layer_num, 0);
            HIPCHK(launch_flash(fa, s));
            HIPCHK(launch_flash_merge(fa, s));
        }
    }
    fa.out = sharded ? sh.out : wi + ws.attn;
    fa.lse = sharded ? sh.out + (int64_t)M * C : (train ? wi + ws.lse_c : nullptr);
    fa.drop_p = dp; fa.drop_seed = c->site_seed(layer_num, 2);
    bool fused_q = false;
    {
        // xa = tgt + self_attn @ Wo  (pre-LayerNorm; norm1 is applied by the consumers)
        Prof p(c, s, PARQ_PROF_LINEAR);
        LinearArgs a = lin(wi + ws.sa, C, A + L.self_out_w, C, A + L.self_out_b, wi + ws.xa, C, M, C, C);
        a.R = wi + ws.tgt; a.ldr = C; a.Wp = TP ? TP + L.self_out_w : nullptr; halfw(a, L.self_out_w);
        a.drop_p = dp; a.drop_seed = c->site_seed(layer_num, 1);
        if (seam_ok && (seams & 1)) {
            // ... and, in the same launch, the query projection behind norm1 (chain.hip seam_tile): q tiles contract sa, tgt and pe_h with
            // pack-time weight products and take norm1's statistics from the partial row sums the xa tiles publish
            a.lnp_out = reinterpret_cast<double*>(wi + ws.lnp1);
            a.lnp_flags = reinterpret_cast<unsigned*>(wsp + ws.seam_flags);
            a.lnp_epoch = c->next_epoch();
            SeamArgs q;
            memset(&q, 0, sizeof(q));
            q.X1 = wi + ws.sa; q.ldx1 = C; q.W1p = TP + L.qo_w;
            q.X2 = wi + ws.tgt; q.ldx2 = C; q.W2p = TP + L.qg_w;
            q.X3 = wi + ws.pe_h; q.ldx3 = C; q.W3p = TP + L.cross_q_w2;
            q.b1 = A + L.qu_b; q.srow = A + L.q_s; q.b2 = A + L.qv_b;
            q.Y = wi + ws.qc; q.ldy = C; q.M = M; q.N = C;
            q.part = a.lnp_out; q.flags = a.lnp_flags; q.epoch = a.lnp_epoch; q.nparts = C / 64; q.width = C; q.eps = eps;
            q.ln_out = wi + ws.ln1;               // norm1's statistics for the residual of the cross out-projection
            q.err = reinterpret_cast<int*>(wsp + ws.flags);
            const hipError_t e = launch_seam_q(a, q, s);
            if (e == hipSuccess) fused_q = true;
            else if (e != hipErrorNotSupported) return fail(PARQ_ERR_HIP, "launch_seam_q failed: %s", hipGetErrorString(e));
            a.lnp_out = nullptr; a.lnp_flags = nullptr;
        }
        if (!fused_q) {
            HIPCHK(launch_linear(a, 1, s));
            // K7: cross-attention query = (norm1(xa) + pos) @ Wq; publishes norm1's row statistics
            a = cross_q_args(fold_pos);
            HIPCHK(launch_linear(a, 1, s));
        }
    }
    bool merged = false;              // the cross-attention launches below already merged their partials (per-head tiers)
    {
        // dense cross-attention against the cached K/V (transformer_parq.py:377-382)
        Prof p(c, s, PARQ_PROF_CROSS_ATTN);
        fa.q = wi + ws.qc; fa.q_batch = (int64_t)Q * C; fa.q_head = dh; fa.q_row = C;
        fa.Lk = (int)N; fa.nsplit = ws.cross_split;
        // odd iterations sweep the K/V cache backwards (Infinity Cache reuse); development: PARQ_FLASH_ALT_PHASE=1 flips the parity,
        // so that iteration 0 starts on the blocks the K/V projection wrote last
        const int alt_phase = [] { const char* e = dev_env("PARQ_FLASH_ALT_PHASE"); return e && e[0] == '1' ? 1 : 0; }();
        fa.flags = ((layer_num + alt_phase) & 1) ? 2 : 0;
        const int64_t lp = flash_lq_pad(Q);
        fa.o_part = wsp + ws.flash;
        fa.m_part = fa.o_part + (int64_t)B * H * fa.nsplit * dh * lp;
        fa.l_part = fa.m_part + (int64_t)B * H * fa.nsplit * lp;
        if (c->cache_mode()) {
            const char* cache = reinterpret_cast<const char*>(wsp + ws.kvc) + (size_t)li * kvsplit_cache_bytes(B, c->vheads(), (int)N, c->terms());
            if (dh == 256) HIPCHK(launch_flash_split256(fa, cache, s, c->terms(), c->kind()));
            else if (c->terms_for(N, train, train && bwd_reads_cache(c, ws)) == 8) {
                // rows whose probability sum (relative to the row's reference maximum) is under kPeakyL are carried by too few keys for
                // this mode's error model: bit h of flags[1] (and of flags[8 + iteration]) -> bits 1 and 8 + h of
                // the range mirror -> the caller moves head h to the fp16 x 3 tier (parq_set_head_tiers)
                int* flg = reinterpret_cast<int*>(wsp + ws.flags);
                fa.peaky = sharded ? nullptr : flg + 1;
                fa.peaky_it = sharded ? nullptr : flg + 8 + (layer_num % 56);
                fa.peaky_min = sharded ? nullptr : flg + 2;
                fa.head_min = (sharded || H > 16) ? nullptr : flg + 32;    // flags[32 + h]: every head's smallest row sum of this forward
                fa.peaky_l = kPeakyL;
                const uint32_t safe = c->safe_heads();
                if (safe == 0) HIPCHK(launch_flash_split8(fa, cache, s));
                else {
                    // per-head tiers: two launches over complementary head sets, each with the split count that fills the chip with its
                    // heads and its own partials; every (scene, head) region of the cache spans the split layout's size (launch_kvproj_split)
                    FlashArgs f8 = fa, f3 = fa;
                    for (int hh = 0; hh < H; ++hh) {
                        FlashArgs& t = ((safe >> hh) & 1u) ? f3 : f8;
                        t.hmap |= (unsigned long long)hh << (4 * t.nh);
                        ++t.nh;
                    }
                    const int cus = device_num_cus();
                    f8.nsplit = flash_split_pick_splits(B, f8.nh, Q, (int)N, cus);
                    f3.nsplit = flash_split_pick_splits(B, f3.nh, Q, (int)N, cus);
                    f8.cache_head_bytes = f3.cache_head_bytes = (int64_t)kvsplit_cache_bytes(1, 1, (int)N, 3);
                    f8.m_part = f8.o_part + (int64_t)B * f8.nh * f8.nsplit * dh * lp;
                    f8.l_part = f8.m_part + (int64_t)B * f8.nh * f8.nsplit * lp;
                    f3.o_part = f8.l_part + (int64_t)B * f8.nh * f8.nsplit * lp;
                    f3.m_part = f3.o_part + (int64_t)B * f3.nh * f3.nsplit * dh * lp;
                    f3.l_part = f3.m_part + (int64_t)B * f3.nh * f3.nsplit * lp;
                    f3.peaky = nullptr; f3.peaky_it = nullptr; f3.peaky_min = nullptr;
                    HIPCHK(launch_flash_split8(f8, cache, s));
                    HIPCHK(launch_flash_split(f3, cache, s, 3, kF16));
                    { Prof pm(c, s, PARQ_PROF_MERGE); HIPCHK(launch_flash_merge(f8, s)); HIPCHK(launch_flash_merge(f3, s)); }
                    merged = true;
                }
            }
            else {
                // a mode-4 handle whose heads are all on the fp16 x 3 tier: the per-head row sums are still recorded (a head may return)
                if (c->attn_mode == 4 && !train && !sharded && H <= 16 && dh == 64) fa.head_min = reinterpret_cast<int*>(wsp + ws.flags) + 32;
                HIPCHK(launch_flash_split(fa, cache, s, c->terms(), c->kind()));
            }
        } else {
            const float* kv = wsp + ws.kv + (int64_t)li * B * 2 * N * C;
            fa.k = kv;                       fa.k_batch = 2 * N * C; fa.k_head = N * dh; fa.k_row = dh;
            fa.v = kv + (int64_t)H * N * dh; fa.v_batch = 2 * N * C; fa.v_head = N * dh; fa.v_row = dh;
            HIPCHK(launch_flash(fa, s));
        }
    }
    if (!merged) { Prof p(c, s, PARQ_PROF_MERGE); HIPCHK(launch_flash_merge(fa, s)); }
    }   // phase bit 2
    if (!(sh.mask & 4)) return PARQ_OK;
    if (sharded) {
        Prof p(c, s, PARQ_PROF_MERGE);
        const int64_t rec = (int64_t)M * C + (int64_t)B * H * flash_lq_pad(Q);
        HIPCHK(launch_attn_combine(sh.in, sh.nranks, rec, B, H, Q, flash_lq_pad(Q), dh, wi + ws.attn, s));
    }
    {
        Prof p(c, s, PARQ_PROF_LINEAR);
        // xb = norm1(xa) + cross_attn @ Wo   (residual recomputed from the published statistics)
        LinearArgs a = lin(wi + ws.attn, C, A + L.cross_out_w, C, A + L.cross_out_b, wi + ws.xb, C, M, C, C);
        a.R = wi + ws.xa; a.ldr = C; a.rln_stats = wi + ws.ln1; a.rln_gamma = A + L.n1_w; a.rln_beta = A + L.n1_b; a.Wp = TP ? TP + L.cross_out_w : nullptr; halfw(a, L.cross_out_w);
        a.drop_p = dp; a.drop_seed = c->site_seed(layer_num, 3);
        if (last_of_forward && c->progress_word != nullptr) {
            // the last iteration's cross-attention has been merged: nothing behind this launch raises a flag (range: the K/V projection;
            // hand-off timeouts: the seam launch in front of the cross-attention; too-peaked rows: the merge).  Tell the host now.
            a.pub_flags = reinterpret_cast<const int*>(wsp + ws.flags);
            a.pub_mirror = c->range_mirror; a.pub_word = c->progress_word;
            a.pub_epoch = reinterpret_cast<const int*>(wsp + ws.ind) + 14;        // CallPtrs::p[7], low word
            a.pub_mask = (c->cache_mode() && c->kind() == kF16) ? ~0 : 4;
            a.pub_peaky = c->terms_for(N, false) == 8 ? 1 : 0;
        }
        HIPCHK(launch_linear(a, 1, s));
        // K8: FFN (transformer_parq.py:383-385): relu(norm2(xb) @ W1), publishes norm2's statistics
        a = lin(wi + ws.xb, C, A + L.lin1_w, C, A + L.lin1_b, wi + ws.ffn, F, M, F, C);
        a.ln_gamma = A + L.n2_w; a.ln_beta = A + L.n2_b; a.ln_stats_out = wi + ws.ln2; a.norm_eps = eps;
        a.relu = 1; a.Wp = TP ? TP + L.lin1_w : nullptr; halfw(a, L.lin1_w, 1);
        a.drop_p = dp; a.drop_seed = c->site_seed(layer_num, 4);
        HIPCHK(launch_linear(a, 1, s));
        // xc = norm2(xb) + ffn @ W2
        a = lin(wi + ws.ffn, F, A + L.lin2_w, F, A + L.lin2_b, wi + ws.xc, C, M, C, F);
        a.R = wi + ws.xb; a.ldr = C; a.rln_stats = wi + ws.ln2; a.rln_gamma = A + L.n2_w; a.rln_beta = A + L.n2_b; a.Wp = TP ? TP + L.lin2_w : nullptr; halfw(a, L.lin2_w);
        a.drop_p = dp; a.drop_seed = c->site_seed(layer_num, 5);
        HIPCHK(launch_linear(a, 1, s));
        // K9: heads (transformer_parq.py:234-252; generic_mlp.py:85-110) on norm3(xc); the first layers of the
        // four heads are one GEMM, which also accumulates the GroupNorm moments of the two hidden blocks
        const int NH1 = c->NH1;
        a = lin(wi + ws.xc, C, A + ar.heads1_w, C, A + ar.heads1_b, wi + ws.h1, NH1, M, NH1, C);
        a.ln_gamma = A + L.n3_w; a.ln_beta = A + L.n3_b; a.norm_eps = eps;
        if (train) a.ln_stats_out = wi + ws.ln3;
        if (NH1 % 16 == 0) { a.Wp = TP ? TP + ar.heads1_w : nullptr; halfw(a, ar.heads1_w, c->nl == 1 ? 1 : 0); }
        a.gn_out_sums = gn1; a.gn_out_ncols = 2 * C; a.gn_out_group_cols = C; a.gn_out_rows_per_scene = Q; a.gn_out_ngroups = 2;
        HIPCHK(launch_linear(a, 1, s));
        a = lin(wi + ws.h1, NH1, A + ar.heads2_w, C, nullptr, wi + ws.h2, 2 * C, M, C, C);
        a.gn_sums = gn1; a.gn_gamma = A + ar.gn1_g; a.gn_beta = A + ar.gn1_b; a.norm_eps = eps;
        a.gn_rows_per_scene = Q; a.gn_ngroups = 2;
        a.gX = C; a.gW = (int64_t)C * C; a.gY = C; a.gGamma = C; a.Wp = TP ? TP + ar.heads2_w : nullptr; halfw(a, ar.heads2_w);
        a.gn_out_sums = gn2; a.gn_out_ncols = C; a.gn_out_group_cols = C; a.gn_out_rows_per_scene = Q; a.gn_out_ngroups = 2;
        HIPCHK(launch_linear(a, 2, s));
    }
    // K10: last head layers + box decode + reference point update + next sine embedding
    // (transformer_parq.py:242-279, 331-332)
    {
        Prof p(c, s, PARQ_PROF_OTHER);
        BoxDecodeArgs d;
        memset(&d, 0, sizeof(d));
        d.h1 = wi + ws.h1 + 2 * C; d.ld1 = c->NH1;
        d.h2 = wi + ws.h2; d.ld2 = 2 * C;
        d.gn_sums = gn2; d.gn_gamma = A + ar.gn2_g; d.gn_beta = A + ar.gn2_b;
        d.w3 = A + ar.heads3_w; d.b3 = A + ar.heads3_b; d.C = C; d.rows_per_scene = Q; d.eps = eps;
        d.ref = ref; d.mean_sizes = A + ar.mean_sizes; d.n_mean = c->cfg.num_mean_sizes; d.dim_t = A + ar.dim_t;
        // fp16-operand modes: a range violation seen while the cache was built must not produce plausible wrong numbers
        d.poison = reinterpret_cast<const int*>(wsp + ws.flags);
        d.poison_mask = (c->cache_mode() && c->kind() == kF16) ? ~0 : 4;     // (a hand-off timeout reaches the mirror in every attention mode)
        // (the last iteration of a forward with an early completion signal has published its bits already — LinearArgs::pub_mirror — and
        // the host may have taken them by now: a second copy from here would be charged to the NEXT forward of the workspace)
        d.poison_mirror = (last_of_forward && c->progress_word != nullptr) ? nullptr : c->range_mirror;
        d.peaky = (!sharded && c->terms_for(N, train, train && bwd_reads_cache(c, ws)) == 8) ? reinterpret_cast<const int*>(wsp + ws.flags) + 1 : nullptr;
        d.peaky_poison = c->peaky_poison;
        d.sb = c->sb; d.M = M; d.ncls = c->ncls;
        d.logits = o->pred_logits; d.center = o->center_unnormalized; d.size = o->size_unnormalized;
        d.rot = o->ortho6d; d.prob = o->sem_cls_prob; d.ref_next = ref_out; d.emb_next = emb_next;
        d.ind = ind; d.out_row0 = row0;
        HIPCHK(launch_box_decode(d, s));
    }
    return PARQ_OK;
}


// =============================================================================== training: backward chain
// One iteration of the decoder, reversed.  Activations come from the training forward's stash
// (Workspace::shift(k)); gradients of the packed weights accumulate into `G`, an arena with the layout of the weight
// arena; dK/dV of the hoisted K/V projection accumulate in ws.g_kv over the iterations and are pushed through the
// projection once at the end.  Reference points are detached between iterations (transformer_parq.py:331-332): only
// iteration 0 sends gradient to refpoint.weight, through the sine embedding, the projection and the centre update.
struct BwdIO {
    const float* g_logits; const float* g_center; const float* g_size; const float* g_rot;     // this iteration, may be null
    const float* center; const float* size;                                                    // forward outputs
};

// transposed copies of layer li's weights for the dX GEMMs (layout: see do_backward_iter)
int do_backward_transposes(parq_ctx* c, float* wsp, const Workspace& ws, int li, hipStream_t s) {
    const float* A = c->arena;
    const Arena& ar = c->ar;
    const LayerW& L = ar.layers[li];
    const int C = c->C, F = c->F, NH1 = c->NH1;
    float* wT = wsp + ws.wT;
    float* h1T = wT;
    float* h2T = h1T + (int64_t)C * NH1;
    float* l1T = h2T + 2 * (int64_t)C * C;
    float* l2T = l1T + (int64_t)C * F;
    float* coT = l2T + (int64_t)C * F;
    float* cqT = coT + (int64_t)C * C;
    float* soT = cqT + (int64_t)C * C;
    float* siT = soT + (int64_t)C * C;
    float* p2T = siT + 3 * (int64_t)C * C;
    float* p0T = p2T + (int64_t)C * C;
    HIPCHK(launch_transpose(A + ar.heads1_w, C, h1T, NH1, NH1, C, s));
    HIPCHK(launch_transpose(A + ar.heads2_w, C, h2T, C, C, C, s));
    HIPCHK(launch_transpose(A + ar.heads2_w + (int64_t)C * C, C, h2T + (int64_t)C * C, C, C, C, s));
    HIPCHK(launch_transpose(A + L.lin1_w, C, l1T, F, F, C, s));
    HIPCHK(launch_transpose(A + L.lin2_w, F, l2T, C, C, F, s));
    HIPCHK(launch_transpose(A + L.cross_out_w, C, coT, C, C, C, s));
    HIPCHK(launch_transpose(A + L.cross_in_w, C, cqT, C, C, C, s));
    HIPCHK(launch_transpose(A + L.self_out_w, C, soT, C, C, C, s));
    HIPCHK(launch_transpose(A + L.self_in_w, C, siT, 3 * C, 3 * C, C, s));
    HIPCHK(launch_transpose(A + ar.pe2_w, C, p2T, C, C, C, s));
    HIPCHK(launch_transpose(A + ar.pe0_w, 384, p0T, C, C, 384, s));
    return PARQ_OK;
}

// phase 0: the whole iteration.  Batched mode (Workspace::bwd_batched) cuts the chain at the cross-attention: phase 1 runs from the
// outputs down to dO (kept per iteration in ws.g_do, with the residual gradient in ws.g_res and the D rows in ws.g_Dall), the
// attention backward of all iterations is one launch in parq_backward, and phase 2 continues from that iteration's dQ (ws.g_dq).
int do_backward_iter(parq_ctx* c, const parq_scene* sc, float* wsp, const Workspace& ws, int k, const BwdIO& io, float* G,
                     float* g_tokens, hipStream_t s, int phase = 0, int set = 0) {
    const float* A = c->arena;
    const Arena& ar = c->ar;
    const int li = c->cfg.share_weights ? 0 : k;
    const LayerW& L = ar.layers[li];
    const int B = sc->B, C = c->C, Q = c->Q, H = c->H, dh = c->dh, F = c->F, NH1 = c->NH1;
    const int M = B * Q;
    const int64_t N = (int64_t)sc->V * sc->h * sc->w;
    const float eps = 1e-5f;
    float* wi = wsp + ws.shift(k);
    const double* gn1 = reinterpret_cast<const double*>(wi + ws.gn_sums);
    const double* gn2 = gn1 + (int64_t)B * 4 * kGnSlots;
    const float* ref = wi + ws.refk;
    float* gsp = wsp + (int64_t)set * ws.g_set_stride;      // this stream's copy of the scratch (set 0 when the iterations run in turn)
    const int acc = ws.g_sets > 1 ? 2 : 1;                   // dW accumulation: 2 = always by atomics (other iterations add concurrently)
    float *gA = gsp + ws.g_a, *gB = gsp + ws.g_b, *gC = gsp + ws.g_c, *gPos = gsp + ws.g_pos, *tmp = gsp + ws.g_tmp;
    float *gFfh = gsp + ws.g_ffh, *gH1 = gsp + ws.g_h1, *gH2 = gsp + ws.g_h2, *gZ = gsp + ws.g_z, *act = gsp + ws.g_act;
    float *gH3 = gsp + ws.g_h3, *gQkv = gsp + ws.g_qkv, *gEmb = gsp + ws.g_emb, *gRef = gsp + ws.g_ref, *Dd = gsp + ws.g_D;
    double* bs = reinterpret_cast<double*>(gsp + ws.g_bs);
    float* gD = gsp + ws.g_drop;                       // gradient entering a dropout site's branch
    const float dp = c->drop_p;
    const float inv_keep = dp > 0.f ? 1.f / (1.f - dp) : 1.f;
    // the gradient that flows into a dropped branch: g * keep / (1 - p); the residual path keeps the plain g
    auto through_dropout = [&](const float* g, int site) -> const float* {
        if (!(dp > 0.f)) return g;
        if (launch_dropout_apply(g, gD, M, C, dp, c->site_seed(k, site), s) != hipSuccess) return nullptr;
        return gD;
    };
    // transposed weight copies of this layer
    float* wT = wsp + ws.wT;
    float* h1T = wT;                      // [C][NH1]
    float* h2T = h1T + (int64_t)C * NH1;  // 2 x [C][C]
    float* l1T = h2T + 2 * (int64_t)C * C;   // lin1^T [C][F]
    float* l2T = l1T + (int64_t)C * F;       // lin2^T [F][C]
    float* coT = l2T + (int64_t)C * F;       // cross out^T
    float* cqT = coT + (int64_t)C * C;       // cross q^T
    float* soT = cqT + (int64_t)C * C;       // self out^T
    float* siT = soT + (int64_t)C * C;       // self in^T [C][3C]
    float* p2T = siT + 3 * (int64_t)C * C;   // pe2^T
    float* p0T = p2T + (int64_t)C * C;       // pe0^T [384][C]
    if (phase == 0) {
        int rc = do_backward_transposes(c, wsp, ws, li, s);
        if (rc) return rc;
    }

    auto mm = [&](const float* X, int64_t ldx, const float* WT, int K, int Nn, float* Y, int64_t ldy) {   // Y = X W  (W^T given [Nn][K])
        return lin(X, ldx, WT, K, nullptr, Y, ldy, M, Nn, K);
    };
    const int64_t MC = (int64_t)M * C;
    if (phase == 2) {                                  // resume below the cross-attention with this iteration's dQ and residual gradient
        gC = wsp + ws.g_dq + k * MC;
        gB = wsp + ws.g_res + k * MC;
    }
    if (phase != 2) {
    HIPCHK(hipMemsetAsync(gRef, 0, (size_t)M * 3 * sizeof(float), s));

    // ---- heads (transformer_parq.py:234-279)
    HIPCHK(launch_decode_bwd(io.g_logits, io.g_center, io.g_size, io.g_rot, io.center, io.size, ref, c->sb, M, c->ncls, NH1, C, gH3, gH1,
                             k == 0 ? gRef : nullptr, s));
    // last layers: w3 rows 0..2 centre (input h2act[:, :C]), rows 6..11 rotation (input h2act[:, C:])
    HIPCHK(launch_gn_apply(wi + ws.h2, 2 * C, gn2, A + ar.gn2_g, A + ar.gn2_b, M, C, 2, Q, eps, act, 2 * C, s));
    // (every dW product below also leaves the bias gradient = column sums of its dY operand: launch_gemm_tn's last arguments)
    HIPCHK(launch_gemm_tn(gH3, 16, act, 2 * C, G + ar.heads3_w, C, M, 3, C, acc, s, G + ar.heads3_b));
    HIPCHK(launch_gemm_tn(gH3 + 3, 16, act + C, 2 * C, G + ar.heads3_w + 6 * (int64_t)C, C, M, 6, C, acc, s, G + ar.heads3_b + 6));
    HIPCHK(launch_head3_bwd(gH3, A + ar.heads3_w, gZ, M, C, s));                       // gZ = d loss / d h2act
    HIPCHK(launch_gn_bwd(wi + ws.h2, 2 * C, gn2, A + ar.gn2_g, A + ar.gn2_b, M, C, 2, Q, eps, gZ, 2 * C, act, bs, gH2, 2 * C,
                         G + ar.gn2_g, G + ar.gn2_b, s));
    // second layers (grouped): h2[:, g] = h1act[:, g] W2g^T
    HIPCHK(launch_gn_apply(wi + ws.h1, NH1, gn1, A + ar.gn1_g, A + ar.gn1_b, M, C, 2, Q, eps, act, 2 * C, s));   // act = h1act
    for (int g = 0; g < 2; ++g) {
        HIPCHK(launch_gemm_tn(gH2 + g * C, 2 * C, act + g * C, 2 * C, G + ar.heads2_w + (int64_t)g * C * C, C, M, C, C, acc, s));
        LinearArgs a = mm(gH2 + g * C, 2 * C, h2T + (int64_t)g * C * C, C, C, gZ + g * C, 2 * C);       // gZ = d / d h1act
        HIPCHK(launch_linear(a, 1, s));
    }
    HIPCHK(launch_gn_bwd(wi + ws.h1, NH1, gn1, A + ar.gn1_g, A + ar.gn1_b, M, C, 2, Q, eps, gZ, 2 * C, act, bs, gH1, NH1,
                         G + ar.gn1_g, G + ar.gn1_b, s));
    // fused first layers on x3 = norm3(xc)
    HIPCHK(launch_layernorm(wi + ws.xc, A + L.n3_w, A + L.n3_b, tmp, M, C, eps, s));           // tmp = x3
    HIPCHK(launch_gemm_tn(gH1, NH1, tmp, C, G + ar.heads1_w, C, M, NH1, C, acc, s, G + ar.heads1_b, 2 * C));   // columns < 2C: GroupNorm follows, no bias
    {
        LinearArgs a = mm(gH1, NH1, h1T, NH1, C, gA, C);                                       // gA = d / d x3
        HIPCHK(launch_linear(a, 1, s));
    }
    // ---- norm3 / FFN (transformer_parq.py:383-385)
    HIPCHK(launch_ln_bwd(gA, wi + ws.xc, wi + ws.ln3, A + L.n3_w, gB, M, C, 0, G + L.n3_w, G + L.n3_b, s));   // gB = d / d xc
    {
        const float* gd = through_dropout(gB, 5);
        if (!gd) return fail(PARQ_ERR_HIP, "dropout_apply failed");
        HIPCHK(launch_gemm_tn(gd, C, wi + ws.ffn, F, G + L.lin2_w, F, M, C, F, acc, s, G + L.lin2_b));
        LinearArgs a = mm(gd, C, l2T, C, F, gFfh, F);                // d / d ffn hidden, through dropout (the stash holds the dropped,
        a.relu_mask = wi + ws.ffn; a.ldmask = F; a.mask_scale = inv_keep;   // rescaled hidden: zero = dropped or ReLU-inactive) and the ReLU
        HIPCHK(launch_linear(a, 1, s));
    }
    HIPCHK(launch_layernorm(wi + ws.xb, A + L.n2_w, A + L.n2_b, tmp, M, C, eps, s));           // tmp = x2
    HIPCHK(launch_gemm_tn(gFfh, F, tmp, C, G + L.lin1_w, C, M, F, C, acc, s, G + L.lin1_b));
    {
        LinearArgs a = mm(gFfh, F, l1T, F, C, gA, C);                                          // gA = d / d x2 = gB + gFfh W1
        a.R = gB; a.ldr = C;
        HIPCHK(launch_linear(a, 1, s));
    }
    // ---- norm2 / cross-attention (transformer_parq.py:377-382)
    if (phase == 1) gB = wsp + ws.g_res + k * MC;                                              // the residual gradient outlives this call
    float* gDo = phase == 1 ? wsp + ws.g_do + k * MC : gA;
    float* Dc = phase == 1 ? wsp + ws.g_Dall + (int64_t)k * B * H * flash_lq_pad(Q) : Dd;
    HIPCHK(launch_ln_bwd(gA, wi + ws.xb, wi + ws.ln2, A + L.n2_w, gB, M, C, 0, G + L.n2_w, G + L.n2_b, s));   // gB = d / d xb
    {
        const float* gd = through_dropout(gB, 3);
        if (!gd) return fail(PARQ_ERR_HIP, "dropout_apply failed");
        HIPCHK(launch_gemm_tn(gd, C, wi + ws.attn, C, G + L.cross_out_w, C, M, C, C, acc, s, G + L.cross_out_b));
        LinearArgs a = mm(gd, C, coT, C, C, gDo, C);                                           // d / d attention output
        HIPCHK(launch_linear(a, 1, s));
    }
    HIPCHK(launch_attn_bwd_rowdot(gDo, wi + ws.attn, (int64_t)Q * C, C, B, H, Q, dh, Dc, s));
    if (phase == 1) return PARQ_OK;
    HIPCHK(hipMemsetAsync(gC, 0, (size_t)M * C * sizeof(float), s));                           // gC = d / d q (atomics)
    {
        const float* kv = wsp + (c->cache_mode() ? ws.kv_train : ws.kv) + (int64_t)li * B * 2 * N * C;
        // dK | dV accumulate token-major: g_kv[layer][b][n][2C], K heads at columns h*dh, V heads at C + h*dh
        float* gkv = wsp + ws.g_kv + (int64_t)li * B * 2 * N * C;
        HIPCHK(launch_attn_bwd(wi + ws.qc, (int64_t)Q * C, dh, C, kv, 2 * N * C, N * dh, dh, kv + (int64_t)H * N * dh, 2 * N * C, N * dh, dh,
                               gA, (int64_t)Q * C, dh, C, wi + ws.lse_c, Dd, gC, (int64_t)Q * C, dh, C, gkv, 2 * N * C, dh, 2 * C,
                               gkv + C, 2 * N * C, dh, 2 * C, B, H, Q, (int)N, dh, 1, s,
                               attn_bwd_dq_partial_floats(B, H, Q, (int)N, dh) ? wsp + ws.g_dqp : nullptr, dp, c->site_seed(k, 2),
                               reinterpret_cast<unsigned int*>(wsp + ws.g_bs),        // g_bs: free between the GroupNorm passes
                               (dh == 64 || dh == 32) ? nullptr : wsp + ws.g_mat));
    }
    }   // phase != 2
    // q = (x1 + pos) Wq^T + bq,  x1 = norm1(xa)
    HIPCHK(launch_layernorm(wi + ws.xa, A + L.n1_w, A + L.n1_b, tmp, M, C, eps, s));
    HIPCHK(launch_add(tmp, wi + ws.pos, tmp, (int64_t)M * C, s));                              // tmp = x1 + pos
    HIPCHK(launch_gemm_tn(gC, C, tmp, C, G + L.cross_in_w, C, M, C, C, acc, s, G + L.cross_in_b));
    {
        LinearArgs a = mm(gC, C, cqT, C, C, gPos, C);                                          // gPos = d / d (x1 + pos) [cross]
        HIPCHK(launch_linear(a, 1, s));
    }
    HIPCHK(launch_add(gB, gPos, gA, (int64_t)M * C, s));                                       // gA = d / d x1 (residual + query path)
    // ---- norm1 / self-attention (transformer_parq.py:372-376)
    HIPCHK(launch_ln_bwd(gA, wi + ws.xa, wi + ws.ln1, A + L.n1_w, gB, M, C, 0, G + L.n1_w, G + L.n1_b, s));   // gB = d / d xa
    {
        const float* gd = through_dropout(gB, 1);
        if (!gd) return fail(PARQ_ERR_HIP, "dropout_apply failed");
        HIPCHK(launch_gemm_tn(gd, C, wi + ws.sa, C, G + L.self_out_w, C, M, C, C, acc, s, G + L.self_out_b));
        LinearArgs a = mm(gd, C, soT, C, C, gA, C);                                            // gA = d / d self-attention output
        HIPCHK(launch_linear(a, 1, s));
    }
    HIPCHK(launch_attn_bwd_rowdot(gA, wi + ws.sa, (int64_t)Q * C, C, B, H, Q, dh, Dd, s));
    HIPCHK(hipMemsetAsync(gQkv, 0, (size_t)M * 3 * C * sizeof(float), s));
    HIPCHK(launch_attn_bwd(wi + ws.qkv, (int64_t)Q * 3 * C, dh, 3 * C, wi + ws.qkv + C, (int64_t)Q * 3 * C, dh, 3 * C,
                           wi + ws.qkv + 2 * C, (int64_t)Q * 3 * C, dh, 3 * C, gA, (int64_t)Q * C, dh, C, wi + ws.lse_s, Dd,
                           gQkv, (int64_t)Q * 3 * C, dh, 3 * C, gQkv + C, (int64_t)Q * 3 * C, dh, 3 * C, gQkv + 2 * C,
                           (int64_t)Q * 3 * C, dh, 3 * C, B, H, Q, Q, dh, 0, s, nullptr, dp, c->site_seed(k, 0), nullptr,
                           (dh == 64 || dh == 32) ? nullptr : wsp + ws.g_mat));
    // in-projection: [q | k] = (tgt + pos) Wqk^T, v = tgt Wv^T
    HIPCHK(launch_add(wi + ws.tgt, wi + ws.pos, tmp, (int64_t)M * C, s));                      // tmp = tgt + pos
    HIPCHK(launch_gemm_tn(gQkv, 3 * C, tmp, C, G + L.self_in_w, C, M, 2 * C, C, acc, s, G + L.self_in_b));
    HIPCHK(launch_gemm_tn(gQkv + 2 * C, 3 * C, wi + ws.tgt, C, G + L.self_in_w + 2 * (int64_t)C * C, C, M, C, C, acc, s, G + L.self_in_b + 2 * C));
    {
        // d / d (tgt + pos) through q and k: rows 0 .. 2C-1 of W_in, i.e. columns 0 .. 2C-1 of W_in^T ([C][3C])
        LinearArgs a = lin(gQkv, 3 * C, siT, 3 * C, nullptr, gA, C, M, C, 2 * C);
        HIPCHK(launch_linear(a, 1, s));
        HIPCHK(launch_add(gPos, gA, gPos, (int64_t)M * C, s));                                  // gPos = total d / d pos
        // d / d tgt = gB (residual) + gA (q, k path) + gQkv_v Wv
        a = lin(gQkv + 2 * C, 3 * C, siT + 2 * C, 3 * C, nullptr, gC, C, M, C, C);
        a.R = gA; a.ldr = C;
        HIPCHK(launch_linear(a, 1, s));
        HIPCHK(launch_add(gC, gB, gC, (int64_t)M * C, s));                                      // gC = d / d tgt
    }
    // ---- project + sample (transformer_parq.py:321)
    HIPCHK(launch_sample_bwd(sc->tokens, reinterpret_cast<const double*>(wsp + ws.T_cl), sc->camera, ref, c->sb, B, sc->V, sc->h, sc->w,
                             C, Q, gC, g_tokens, k == 0 ? gRef : nullptr, s));
    // ---- position MLP (transformer_parq.py:176-180,317): pos = relu(emb W0^T + b0) W2^T + b2
    HIPCHK(launch_gemm_tn(gPos, C, wi + ws.pe_h, C, G + ar.pe2_w, C, M, C, C, acc, s, G + ar.pe2_b));
    {
        LinearArgs a = mm(gPos, C, p2T, C, C, gA, C);                                          // gA = d / d pe hidden
        a.relu_mask = wi + ws.pe_h; a.ldmask = C;
        HIPCHK(launch_linear(a, 1, s));
    }
    HIPCHK(launch_gemm_tn(gA, C, wi + ws.emb, 384, G + ar.pe0_w, 384, M, C, 384, acc, s, G + ar.pe0_b));
    if (k == 0) {
        LinearArgs a = lin(gA, C, p0T, C, nullptr, gEmb, 384, M, 384, C);
        HIPCHK(launch_linear(a, 1, s));
        HIPCHK(launch_posemb_bwd(gEmb, ref, A + ar.dim_t, M, gRef, s));
        HIPCHK(launch_refpoint_bwd(gRef, ref, B, Q, G + ar.refpoint, s));
    }
    return PARQ_OK;
}

// hoisted K/V projection backward: [K | V] = tokens W_kv^T + b_kv per layer; g_kv is token-major [layer][b][n][2C], so the
// whole projection is one TN GEMM (dW_kv = g^T tokens), one column sum (db_kv) and one GEMM (d tokens += g W_kv)
int do_backward_kvproj(parq_ctx* c, const parq_scene* sc, float* wsp, const Workspace& ws, float* G, float* g_tokens, hipStream_t s) {
    const float* A = c->arena;
    const int B = sc->B, C = c->C;
    const int64_t N = (int64_t)sc->V * sc->h * sc->w;
    if ((int64_t)B * N > (int64_t)INT32_MAX) return fail(PARQ_ERR_ARG, "B*N too large");
    float* wT = wsp + ws.wT;            // W_kv^T [C][2C]
    for (int li = 0; li < c->nl; ++li) {
        const LayerW& L = c->ar.layers[li];
        const float* g = wsp + ws.g_kv + (int64_t)li * B * 2 * N * C;          // [B*N][2C]
        const int Mr = (int)(B * N);
        static const bool split_off = [] { const char* e = dev_env("PARQ_KVPROJ_BWD"); return e && e[0] == 'f'; }();   // "fp32": generic kernels
        const bool split_ok = ws.bwd_batched && !split_off;                      // the batched backward leaves max |g| in g_kvmax
        if (split_ok && kvproj_bwd_split_supported(C)) {
            // dW, db on the fp16 matrix pipe (hi/lo split); g is scaled by the power of two derived from max |g| (attention epilogue)
            HIPCHK(launch_kvproj_bwd_split(g, sc->tokens, Mr, C, G + L.cross_in_w + (int64_t)C * C, G + L.cross_in_b + C,
                                           reinterpret_cast<const unsigned int*>(wsp + ws.g_kvmax), wsp + ws.g_kvmax + 1, s));
        } else if (split_ok && C % 256 == 0) {
            // wider models (C = 1024: 805 GFLOP): the same kernel per 512 x 256 block of dW_kv; the bias gradient falls out of the
            // blocks of the first column slice
            float* dW = G + L.cross_in_w + (int64_t)C * C;
            float* db = G + L.cross_in_b + C;
            for (int n0 = 0; n0 < 2 * C; n0 += 512)
                for (int k0 = 0; k0 < C; k0 += 256)
                    HIPCHK(launch_tn_split_512x256(g + n0, 2 * C, sc->tokens + k0, C, Mr, dW + (int64_t)n0 * C + k0, C, k0 == 0 ? db + n0 : nullptr,
                                                   reinterpret_cast<const unsigned int*>(wsp + ws.g_kvmax), wsp + ws.g_kvmax + 1, s));
        } else {
            HIPCHK(launch_gemm_tn(g, 2 * C, sc->tokens, C, G + L.cross_in_w + (int64_t)C * C, C, Mr, 2 * C, C, 1, s));
            HIPCHK(launch_colsum(g, 2 * C, Mr, 2 * C, G + L.cross_in_b + C, 1, s));
        }
        if (g_tokens) {
            HIPCHK(launch_transpose(A + L.cross_in_w + (int64_t)C * C, C, wT, 2 * C, 2 * C, C, s));
            if (split_ok && (kvproj_bwd_split_supported(C) || C % 256 == 0)) {        // (the dW launch above left the scale in g_kvmax[1])
                // d tokens += g W_kv on the fp16 matrix pipe (201 GFLOP per scene at C = 256, 805 at C = 1024): W_kv^T split hi/lo
                // once, g scaled by the power of two of the dW kernel (g_kvmax[1], written by the launch above) while it is split
                _Float16* whi = reinterpret_cast<_Float16*>(wT + (int64_t)C * 2 * C);
                _Float16* wlo = whi + (int64_t)C * 2 * C;
                HIPCHK(launch_split_f32(wT, whi, wlo, (int64_t)C * 2 * C, s));
                HIPCHK(launch_gemm_split(g, 2 * C, whi, wlo, nullptr, g_tokens, C, Mr, C, 2 * C, 0, nullptr, 1, s, wsp + ws.g_kvmax + 1, 1.f,
                                         wsp + ws.g_kvmax + 1, 1));
            } else {
                LinearArgs a = lin(g, 2 * C, wT, 2 * C, nullptr, g_tokens, C, Mr, C, 2 * C);
                a.R = g_tokens; a.ldr = C;
                HIPCHK(launch_linear(a, 1, s));
            }
        }
    }
    return PARQ_OK;
}

int check_outs(const parq_outputs* o) {
    if (!o || !o->pred_logits || !o->center_unnormalized || !o->size_unnormalized || !o->ortho6d || !o->sem_cls_prob ||
        !o->coord_pos)
        return fail(PARQ_ERR_ARG, "outputs has a NULL tensor");
    return PARQ_OK;
}

}  // namespace

// =============================================================================== C ABI

extern "C" {

const char* parq_last_error(void) { return g_err; }
const char* parq_version(void) {
#ifdef PARQ_DEV_PROBES
    return "parq_hip 0.3-dev (gfx950; DEVELOPMENT build: environment A/B switches, probe kernels and time stamps compiled in)";
#else
    return "parq_hip 0.4 (gfx950; cross-attention: fp16 hi/lo split MFMA products, cross terms as MX-fp8 products in mode 4, exact fp32 MFMA on request)";
#endif
}

int parq_create(const parq_config* cfg, parq_handle* out) {
    if (!cfg || !out) return fail(PARQ_ERR_ARG, "NULL argument");
    if (cfg->dim < 32 || cfg->dim % 32 != 0 || cfg->dim > 1024) return fail(PARQ_ERR_ARG, "dim=%d must be a multiple of 32 in [32,1024]", cfg->dim);
    if (cfg->num_heads < 1 || cfg->dim % cfg->num_heads != 0) return fail(PARQ_ERR_ARG, "dim %% num_heads != 0");
    const int dh = cfg->dim / cfg->num_heads;
    if (!(dh == 32 || dh == 64 || dh == 128 || dh == 256)) return fail(PARQ_ERR_ARG, "head dim %d unsupported (32/64/128/256)", dh);
    if (cfg->ffn_dim < 32 || cfg->ffn_dim % 32 != 0) return fail(PARQ_ERR_ARG, "ffn_dim must be a multiple of 32");
    if (cfg->num_queries < 1 || cfg->num_layers < 1) return fail(PARQ_ERR_ARG, "num_queries/num_layers must be >= 1");
    if (cfg->num_classes < 2 || cfg->num_classes > 32) return fail(PARQ_ERR_ARG, "num_classes must be in [2,32]");
    if (cfg->num_mean_sizes < 1) return fail(PARQ_ERR_ARG, "num_mean_sizes must be >= 1");
    for (int i = 0; i < 3; ++i)
        if (!(cfg->scale[2 * i + 1] > cfg->scale[2 * i])) return fail(PARQ_ERR_ARG, "scale must be increasing per axis");
    parq_ctx* c = new parq_ctx();
    c->cfg = *cfg;
    c->C = cfg->dim; c->Q = cfg->num_queries; c->H = cfg->num_heads; c->dh = dh; c->F = cfg->ffn_dim;
    c->I = cfg->num_layers; c->ncls = cfg->num_classes;
    c->nl = cfg->share_weights ? 1 : cfg->num_layers;
    c->NH1 = 2 * c->C + ((c->ncls + 3 + 3) / 4) * 4;
    for (int i = 0; i < 3; ++i) { c->sb.lo[i] = cfg->scale[2 * i]; c->sb.hi[i] = cfg->scale[2 * i + 1]; }
    // dim_t[i] = 10000^(2*(i//2)/128) in float32 (transformer_parq.py:49-50)
    for (int i = 0; i < 128; ++i) c->dim_t_host[i] = powf(10000.0f, 2.0f * (float)(i / 2) / 128.0f);
    build_arena(c);
    *out = c;
    return PARQ_OK;
}

#ifdef PARQ_DEV_PROBES
// development build only (not declared in include/parq_hip.h): device buffer of in-kernel time stamps, see common.hpp.
// buf = [cursor, 3 unused, records of 4 x u64 ...] with room for `cap` records; NULL switches the stamps off.  Synchronises.
extern "C" int parq_dev_timeline(unsigned long long* buf, unsigned int cap) {
    if (buf) HIPCHK(hipMemset(buf, 0, 32));
    HIPCHK(tl_set_linear(buf, cap)); HIPCHK(tl_set_elementwise(buf, cap)); HIPCHK(tl_set_flash(buf, cap));
    HIPCHK(tl_set_flash_split(buf, cap)); HIPCHK(tl_set_flash_split8(buf, cap)); HIPCHK(tl_set_kvproj_split(buf, cap)); HIPCHK(tl_set_chain(buf, cap));
    return PARQ_OK;
}
#endif

int parq_set_backward_batched(parq_handle h, int32_t on) {
    if (!h) return fail(PARQ_ERR_ARG, "NULL handle");
    h->bwd_batched_env = on != 0;
    return PARQ_OK;
}

/* Iterations of the chain backward in flight at once (default 8 = up to eight streams; their weight gradients meet in the arena
 * through float atomics, so the summation order — and the last bits of the gradients — vary from run to run).  1 runs the
 * iterations in turn on the caller's stream with plain accumulation: reproducible gradients for the chain (the hoisted K/V
 * projection's row-split and the set loss still reduce with atomics).  Changes the training workspace size: call before
 * parq_train_workspace_bytes. */
int parq_set_backward_streams(parq_handle h, int32_t n) {
    if (!h || n < 1 || n > 8) return fail(PARQ_ERR_ARG, "parq_set_backward_streams: 1 <= n <= 8");
    h->bwd_streams = n;
    return PARQ_OK;
}

int parq_destroy(parq_handle h) {
    if (!h) return PARQ_OK;
    for (hipEvent_t e : h->bucket_done) if (e) (void)hipEventDestroy(e);
    for (auto& e : h->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (hipEvent_t e : h->iter_done) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->join_ev) if (e) (void)hipEventDestroy(e);
    if (h->fork_ev) (void)hipEventDestroy(h->fork_ev);
    for (hipStream_t st : h->aux_stream) if (st) (void)hipStreamDestroy(st);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    delete h;
    return PARQ_OK;
}

int parq_set_weight(parq_handle h, const char* name, const float* dev, int64_t numel) {
    if (!h || !name || !dev || numel <= 0) return fail(PARQ_ERR_ARG, "bad argument to parq_set_weight");
    std::string n(name);
    const std::string dup = "parq_module.decoder.mlp_heads.";      // same storage as mlp_heads.* (parq_decoder.py:66)
    if (n.compare(0, dup.size(), dup) == 0) n = "mlp_heads." + n.substr(dup.size());
    h->named[n] = WeightRef{dev, numel};
    h->packed = false;
    return PARQ_OK;
}

size_t parq_packed_weights_bytes(parq_handle h) { return h ? (size_t)h->ar.total * sizeof(float) : 0; }

int parq_pack_weights(parq_handle h, void* arena_v, size_t arena_bytes, parq_stream stream) {
    if (!h || !arena_v) return fail(PARQ_ERR_ARG, "NULL argument");
    if (arena_bytes < (size_t)h->ar.total * sizeof(float)) return fail(PARQ_ERR_WORKSPACE, "weight arena too small");
    hipStream_t s = (hipStream_t)stream;
    float* A = (float*)arena_v;
    parq_ctx* c = h;
    const int64_t C = c->C, F = c->F, Q = c->Q;
    int rc = PARQ_OK;
    auto get = [&](const std::string& name, int64_t numel) -> const float* {
        auto it = c->named.find(name);
        if (it == c->named.end()) { rc = fail(PARQ_ERR_STATE, "weight '%s' was never set", name.c_str()); return nullptr; }
        if (it->second.n != numel) { rc = fail(PARQ_ERR_ARG, "weight '%s' has %lld elements, expected %lld", name.c_str(), (long long)it->second.n, (long long)numel); return nullptr; }
        return it->second.p;
    };
    GatherArgs gq;
    gq.count = 0;
    auto copy = [&](const std::string& name, int64_t dst, int64_t numel) -> bool {
        const float* p = get(name, numel);
        if (!p) return false;
        // queued: the tensors go in launches of up to kGatherMax copies (a training step re-packs ~50 tensors after every
        // optimizer step; one hipMemcpyAsync each was ~50 dependent 3 us operations)
        if (gq.count == kGatherMax) {
            hipError_t e = launch_gather_copy(gq, s);
            if (e != hipSuccess) { rc = fail(PARQ_ERR_HIP, "gather copy (%s): %s", name.c_str(), hipGetErrorString(e)); return false; }
            gq.count = 0;
        }
        gq.src[gq.count] = p; gq.dst[gq.count] = A + dst; gq.n[gq.count] = numel; ++gq.count;
        return true;
    };
    auto flush = [&]() -> int {
        HIPCHK(launch_gather_copy(gq, s));
        gq.count = 0;
        return PARQ_OK;
    };
    HIPCHK(hipMemsetAsync(A, 0, (size_t)c->ar.total * sizeof(float), s));
    for (int li = 0; li < c->nl; ++li) {
        const LayerW& L = c->ar.layers[li];
        const std::string p = "parq_module.decoder.layers." + std::to_string(li) + ".";
        if (!copy(p + "self_attn.in_proj_weight", L.self_in_w, 3 * C * C) || !copy(p + "self_attn.in_proj_bias", L.self_in_b, 3 * C) ||
            !copy(p + "self_attn.out_proj.weight", L.self_out_w, C * C) || !copy(p + "self_attn.out_proj.bias", L.self_out_b, C) ||
            !copy(p + "multihead_attn.in_proj_weight", L.cross_in_w, 3 * C * C) || !copy(p + "multihead_attn.in_proj_bias", L.cross_in_b, 3 * C) ||
            !copy(p + "multihead_attn.out_proj.weight", L.cross_out_w, C * C) || !copy(p + "multihead_attn.out_proj.bias", L.cross_out_b, C) ||
            !copy(p + "linear1.weight", L.lin1_w, F * C) || !copy(p + "linear1.bias", L.lin1_b, F) ||
            !copy(p + "linear2.weight", L.lin2_w, C * F) || !copy(p + "linear2.bias", L.lin2_b, C) ||
            !copy(p + "norm1.weight", L.n1_w, C) || !copy(p + "norm1.bias", L.n1_b, C) ||
            !copy(p + "norm2.weight", L.n2_w, C) || !copy(p + "norm2.bias", L.n2_b, C) ||
            !copy(p + "norm3.weight", L.n3_w, C) || !copy(p + "norm3.bias", L.n3_b, C))
            return rc;
    }
    c->kv16_state = 1;
    const Arena& ar = c->ar;
    const std::string d = "parq_module.decoder.";
    const std::string hc = "mlp_heads.center_head.layers.", hr = "mlp_heads.rotation_head.layers.";
    const int ncls = c->ncls;
    if (!copy("refpoint.weight", ar.refpoint, Q * 3) ||
        !copy(d + "position_encoder.0.weight", ar.pe0_w, C * 384) || !copy(d + "position_encoder.0.bias", ar.pe0_b, C) ||
        !copy(d + "position_encoder.2.weight", ar.pe2_w, C * C) || !copy(d + "position_encoder.2.bias", ar.pe2_b, C) ||
        // fused first head layer: [centre.0 ; rotation.0 ; sem_cls ; size]
        !copy(hc + "0.weight", ar.heads1_w, C * C) || !copy(hr + "0.weight", ar.heads1_w + C * C, C * C) ||
        !copy("mlp_heads.sem_cls_head.layers.0.weight", ar.heads1_w + 2 * C * C, ncls * C) ||
        !copy("mlp_heads.size_head.layers.0.weight", ar.heads1_w + (2 * C + ncls) * C, 3 * C) ||
        !copy("mlp_heads.sem_cls_head.layers.0.bias", ar.heads1_b + 2 * C, ncls) ||
        !copy("mlp_heads.size_head.layers.0.bias", ar.heads1_b + 2 * C + ncls, 3) ||
        !copy(hc + "1.weight", ar.gn1_g, C) || !copy(hr + "1.weight", ar.gn1_g + C, C) ||
        !copy(hc + "1.bias", ar.gn1_b, C) || !copy(hr + "1.bias", ar.gn1_b + C, C) ||
        !copy(hc + "4.weight", ar.heads2_w, C * C) || !copy(hr + "4.weight", ar.heads2_w + C * C, C * C) ||
        !copy(hc + "5.weight", ar.gn2_g, C) || !copy(hr + "5.weight", ar.gn2_g + C, C) ||
        !copy(hc + "5.bias", ar.gn2_b, C) || !copy(hr + "5.bias", ar.gn2_b + C, C) ||
        !copy(hc + "8.weight", ar.heads3_w, 3 * C) || !copy(hr + "8.weight", ar.heads3_w + 6 * C, 6 * C) ||
        !copy(hc + "8.bias", ar.heads3_b, 3) || !copy(hr + "8.bias", ar.heads3_b + 6, 6) ||
        !copy("mean_sizes", ar.mean_sizes, (int64_t)c->cfg.num_mean_sizes * 3))
        return rc;
    if ((rc = flush()) != PARQ_OK) return rc;
    for (int li = 0; li < c->nl; ++li) {                // hi/lo split of the K/V projection weights, from the packed copy
        const LayerW& L = c->ar.layers[li];
        HIPCHK(launch_split_f32(A + L.cross_in_w + C * C, A + L.kv_whi, A + L.kv_wlo, 2 * C * C, s));
    }
    // small host table kept in the handle, so the copy needs no synchronisation with the stream
    HIPCHK(hipMemcpyAsync(A + ar.dim_t, c->dim_t_host, sizeof(c->dim_t_host), hipMemcpyHostToDevice, s));
    c->derived_valid = false;          // folded position-MLP weights + tile-ordered mirror: built by the first inference iteration
    c->arena = A;
    c->packed = true;
    c->prepared = false;
    return PARQ_OK;
}

size_t parq_workspace_bytes(parq_handle h, int32_t B, int32_t V, int32_t hh, int32_t ww) {
    if (!h || B < 1 || V < 1 || hh < 1 || ww < 1) return 0;
    Workspace ws;
    carve_workspace(h, B, V, hh, ww, &ws);
    return (size_t)ws.total * sizeof(float);
}

int parq_prepare(parq_handle h, const parq_scene* scene, void* workspace, size_t workspace_bytes, parq_stream stream) {
    if (!h || !workspace) return fail(PARQ_ERR_ARG, "NULL argument");
    if (!h->packed) return fail(PARQ_ERR_STATE, "parq_pack_weights must be called first");
    int rc = check_scene(h, scene);
    if (rc) return rc;
    Workspace ws;
    carve_workspace(h, scene->B, scene->V, scene->h, scene->w, &ws);
    if (workspace_bytes < (size_t)ws.total * sizeof(float)) return fail(PARQ_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, (size_t)ws.total * sizeof(float));
    return do_prepare(h, scene, (float*)workspace, ws, (hipStream_t)stream);
}

int parq_iterate(parq_handle h, const parq_scene* scene, void* workspace, size_t workspace_bytes, int32_t layer_num,
                 const float* ref_in, const parq_outputs* outs, float* ref_out, parq_stream stream) {
    if (!h || !workspace) return fail(PARQ_ERR_ARG, "NULL argument");
    if (!h->packed || !h->prepared) return fail(PARQ_ERR_STATE, "parq_prepare must be called first");
    int rc = check_scene(h, scene);
    if (rc) return rc;
    rc = check_outs(outs);
    if (rc) return rc;
    if (layer_num < 0 || layer_num >= h->I) return fail(PARQ_ERR_ARG, "layer_num out of range");
    Workspace ws;
    carve_workspace(h, scene->B, scene->V, scene->h, scene->w, &ws);
    if (workspace_bytes < (size_t)ws.total * sizeof(float)) return fail(PARQ_ERR_WORKSPACE, "workspace too small");
    float* wsp = (float*)workspace;
    hipStream_t s = (hipStream_t)stream;
    const float* ref = ref_in ? ref_in : wsp + ws.ref;
    rc = do_iterate(h, scene, wsp, ws, layer_num, ref, ref_in == nullptr && h->emb_valid, outs, wsp + ws.ref_next, s);
    h->emb_valid = (rc == PARQ_OK);        // the decode kernel left pos2posemb3d(ref_next) in the workspace
    if (rc) return rc;
    const size_t rb = (size_t)scene->B * h->Q * 3 * sizeof(float);
    HIPCHK(hipMemcpyAsync(wsp + ws.ref, wsp + ws.ref_next, rb, hipMemcpyDeviceToDevice, s));
    if (ref_out) HIPCHK(hipMemcpyAsync(ref_out, wsp + ws.ref_next, rb, hipMemcpyDeviceToDevice, s));
    return PARQ_OK;
}

size_t parq_shard_exchange_floats(parq_handle h, int32_t B, int32_t which) {
    if (!h || B < 1 || which < 0 || which > 1) return 0;
    const int64_t M = (int64_t)B * h->Q;
    // which = 0: [M*C sample sums | M valid-view counts | 1 fp16-range flag] (SUM all-reduce); 1: [M*C outputs | lse rows] (all-gather)
    return (size_t)(which == 0 ? M * h->C + M + 1 : M * h->C + (int64_t)B * h->H * flash_lq_pad(h->Q));
}

int parq_iterate_sharded(parq_handle h, const parq_scene* scene, void* workspace, size_t workspace_bytes, int32_t layer_num,
                         int32_t phase, const float* ref_in, const parq_outputs* outs, float* ref_out, const float* xchg_in,
                         float* xchg_out, int32_t nranks, parq_stream stream) {
    if (!h || !workspace) return fail(PARQ_ERR_ARG, "NULL argument");
    if (!h->packed || !h->prepared) return fail(PARQ_ERR_STATE, "parq_prepare must be called first");
    int rc = check_scene(h, scene);
    if (rc) return rc;
    rc = check_outs(outs);
    if (rc) return rc;
    if (layer_num < 0 || layer_num >= h->I) return fail(PARQ_ERR_ARG, "layer_num out of range");
    if (phase < 0 || phase > 2) return fail(PARQ_ERR_ARG, "phase must be 0, 1 or 2");
    if ((phase <= 1 && !xchg_out) || (phase >= 1 && !xchg_in)) return fail(PARQ_ERR_ARG, "exchange buffer is NULL for phase %d", phase);
    if (phase == 2 && nranks < 1) return fail(PARQ_ERR_ARG, "nranks must be >= 1");
    Workspace ws;
    carve_workspace(h, scene->B, scene->V, scene->h, scene->w, &ws);
    if (workspace_bytes < (size_t)ws.total * sizeof(float)) return fail(PARQ_ERR_WORKSPACE, "workspace too small");
    float* wsp = (float*)workspace;
    hipStream_t s = (hipStream_t)stream;
    const float* ref = ref_in ? ref_in : wsp + ws.ref;
    ShardIO sh;
    sh.mask = 1 << phase; sh.in = xchg_in; sh.out = xchg_out; sh.nranks = nranks;
    rc = do_iterate(h, scene, wsp, ws, layer_num, ref, ref_in == nullptr && h->emb_valid, outs, wsp + ws.ref_next, s, 0, nullptr, false, sh);
    if (rc) return rc;
    if (phase == 2) {
        h->emb_valid = true;                   // the decode kernel left pos2posemb3d(ref_next) in the workspace
        const size_t rb = (size_t)scene->B * h->Q * 3 * sizeof(float);
        HIPCHK(hipMemcpyAsync(wsp + ws.ref, wsp + ws.ref_next, rb, hipMemcpyDeviceToDevice, s));
        if (ref_out) HIPCHK(hipMemcpyAsync(ref_out, wsp + ws.ref_next, rb, hipMemcpyDeviceToDevice, s));
    }
    return PARQ_OK;
}

// the I iterations behind the prologue; captured: being recorded by parq_forward_capture (do_iterate)
static int forward_iterations(parq_ctx* h, const parq_scene* scene, float* wsp, const Workspace& ws, const parq_outputs* outs, hipStream_t s,
                              bool captured) {
    const int64_t M = (int64_t)scene->B * h->Q;
    float* ra = wsp + ws.ref;
    float* rb = wsp + ws.ref_next;
    for (int k = 0; k < h->I; ++k) {
        parq_outputs o;
        memset(&o, 0, sizeof(o));
        if (!captured) {
            o.pred_logits = outs->pred_logits + k * M * h->ncls;
            o.center_unnormalized = outs->center_unnormalized + k * M * 3;
            o.size_unnormalized = outs->size_unnormalized + k * M * 3;
            o.ortho6d = outs->ortho6d + k * M * 6;
            o.sem_cls_prob = outs->sem_cls_prob + k * M * h->ncls;
            o.coord_pos = outs->coord_pos + k * M * 3;
        }
        const int rc = do_iterate(h, scene, wsp, ws, k, ra, true, &o, rb, s, 0, nullptr, false, ShardIO(),      // ping-pong the reference points; the sine
                                  captured ? k * M : -1, k + 1 == h->I);                                        // embedding of iteration 0 comes from the
        if (rc) return rc;                                                                                      // prologue, later ones from the previous decode
        float* t = ra; ra = rb; rb = t;
    }
    h->ref_state = 0;       // ws.ref / ws.ref_next roles depend on parity; stepping must re-prepare
    h->prepared = false;
    return PARQ_OK;
}

static int forward_checks(parq_handle h, const parq_scene* scene, void* workspace, size_t workspace_bytes, const parq_outputs* outs, Workspace* ws) {
    if (!h || !workspace) return fail(PARQ_ERR_ARG, "NULL argument");
    if (!h->packed) return fail(PARQ_ERR_STATE, "parq_pack_weights must be called first");
    int rc = check_scene(h, scene);
    if (rc) return rc;
    rc = check_outs(outs);
    if (rc) return rc;
    carve_workspace(h, scene->B, scene->V, scene->h, scene->w, ws);
    if (workspace_bytes < (size_t)ws->total * sizeof(float)) return fail(PARQ_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, (size_t)ws->total * sizeof(float));
    return PARQ_OK;
}

int parq_forward(parq_handle h, const parq_scene* scene, void* workspace, size_t workspace_bytes,
                 const parq_outputs* outs, parq_stream stream) {
    Workspace ws;
    int rc = forward_checks(h, scene, workspace, workspace_bytes, outs, &ws);
    if (rc) return rc;
    rc = do_prepare(h, scene, (float*)workspace, ws, (hipStream_t)stream, false, nullptr, true);
    if (rc) return rc;
    return forward_iterations(h, scene, (float*)workspace, ws, outs, (hipStream_t)stream, false);
}

/* ---- a captured forward (include/parq_hip.h) ---------------------------------------------------------------------------------- */
// what a graph was recorded with: parq_forward_replay refuses a graph whose recording no longer matches the handle or the call
struct GraphKey {
    int B, V, h, w, mode, seam, poison;
    uint32_t safe;
    const void* ws; const void* arena; const void* mirror; const void* progress;
    bool operator==(const GraphKey& o) const {
        return B == o.B && V == o.V && h == o.h && w == o.w && mode == o.mode && seam == o.seam && poison == o.poison && safe == o.safe &&
               ws == o.ws && arena == o.arena && mirror == o.mirror && progress == o.progress;
    }
};
static GraphKey graph_key(const parq_ctx* c, int B, int V, int hh, int ww, const void* wsp) {
    return GraphKey{B, V, hh, ww, c->attn_mode, c->seam_fusion ? 1 : 0, c->peaky_poison, c->safe_heads(), wsp, c->arena, c->range_mirror, c->progress_word};
}
struct parq_graph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; size_t nodes = 0; GraphKey key; };

int parq_forward_capture(parq_handle h, int32_t B, int32_t V, int32_t hh, int32_t ww, void* workspace, size_t workspace_bytes,
                         parq_stream stream, parq_graph_t* out) {
    if (!h || !workspace || !out) return fail(PARQ_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (!h->packed) return fail(PARQ_ERR_STATE, "parq_pack_weights must be called first");
    if (h->profiling) return fail(PARQ_ERR_STATE, "parq_forward_capture: switch parq_profile_enable off (its events are host-side records)");
    parq_scene scene;
    memset(&scene, 0, sizeof(scene));
    scene.B = B; scene.V = V; scene.h = hh; scene.w = ww;
    // (the recorded iterations take tokens / cameras / outputs from the workspace; the scene's pointers are never read)
    scene.tokens = scene.camera = scene.T_camera_pseudoCam = scene.T_world_pseudoCam = scene.T_world_local = reinterpret_cast<const float*>(workspace);
    int rc = check_scene(h, &scene);
    if (rc) return rc;
    Workspace ws;
    carve_workspace(h, B, V, hh, ww, &ws);
    if (workspace_bytes < (size_t)ws.total * sizeof(float)) return fail(PARQ_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, (size_t)ws.total * sizeof(float));
    // Lazily built state must not end up inside the graph (it would be rebuilt by every replay): the derived inference weights are
    // settled on the caller's stream, in front of the first replay.  (The mode's 16-bit copy of W_kv belongs to the K/V projection,
    // which every replay launches directly.)
    hipStream_t s = (hipStream_t)stream;
    if (!h->derived_valid) { rc = build_derived_weights(h, s); if (rc) return rc; }
    // recorded on a stream of the handle's own (the caller's may be the legacy default stream, which cannot capture); relaxed mode:
    // other threads of the process (a data loader allocating, another module's launches) are not affected by the capture
    if (!h->cap_stream) HIPCHK(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
    HIPCHK(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeRelaxed));
    rc = forward_iterations(h, &scene, (float*)workspace, ws, nullptr, h->cap_stream, true);
    parq_graph* g = new parq_graph();
    const hipError_t e = hipStreamEndCapture(h->cap_stream, &g->graph);
    if (rc != PARQ_OK || e != hipSuccess || !g->graph) {
        if (g->graph) (void)hipGraphDestroy(g->graph);
        delete g;
        (void)hipGetLastError();
        return rc != PARQ_OK ? rc : fail(PARQ_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    }
    (void)hipGraphGetNodes(g->graph, nullptr, &g->nodes);
    const hipError_t e2 = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e2 != hipSuccess) {
        (void)hipGraphDestroy(g->graph);
        delete g;
        return fail(PARQ_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e2));
    }
    g->key = graph_key(h, B, V, hh, ww, workspace);
    *out = g;
    return PARQ_OK;
}

int parq_forward_replay(parq_handle h, parq_graph_t g, const parq_scene* scene, void* workspace, size_t workspace_bytes,
                        const parq_outputs* outs, parq_stream stream) {
    if (!g || !g->exec) return fail(PARQ_ERR_ARG, "NULL graph");
    Workspace ws;
    int rc = forward_checks(h, scene, workspace, workspace_bytes, outs, &ws);
    if (rc) return rc;
    if (!(g->key == graph_key(h, scene->B, scene->V, scene->h, scene->w, workspace)) || !h->derived_valid)
        return fail(PARQ_ERR_STATE, "parq_forward_replay: the graph was recorded for another shape / workspace / weight arena / range mirror or under "
                                    "other attention settings (mode, head tiers, seam fusion): capture again");
    hipStream_t s = (hipStream_t)stream;
    // launched directly with THIS call's pointers: prologue (which also leaves them in the workspace for the recorded part) + K/V projection
    rc = do_prepare(h, scene, (float*)workspace, ws, s, false, outs, true);
    if (rc) return rc;
    HIPCHK(hipGraphLaunch(g->exec, s));
    h->ref_state = 0;
    h->prepared = false;
    return PARQ_OK;
}

int64_t parq_graph_nodes(parq_graph_t g) { return g ? (int64_t)g->nodes : 0; }

int parq_graph_destroy(parq_graph_t g) {
    if (!g) return PARQ_OK;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return PARQ_OK;
}

int parq_set_progress(parq_handle h, int32_t* host_visible_word, int32_t epoch) {
    if (!h) return fail(PARQ_ERR_ARG, "NULL handle");
    h->progress_word = host_visible_word;
    h->progress_epoch = epoch;
    return PARQ_OK;
}

int32_t parq_mirror_take(int32_t* host_visible_flag) {
    return host_visible_flag ? __atomic_exchange_n(host_visible_flag, 0, __ATOMIC_ACQ_REL) : 0;
}

int parq_workspace_lookup(parq_handle h, int32_t B, int32_t V, int32_t hh, int32_t ww, const char* name,
                          size_t* offset_floats, size_t* numel) {
    if (!h || !name || !offset_floats || !numel) return fail(PARQ_ERR_ARG, "NULL argument");
    Workspace ws;
    carve_workspace(h, B, V, hh, ww, &ws);
    const int64_t C = h->C, Q = h->Q, F = h->F, M = (int64_t)B * Q, N = (int64_t)V * hh * ww;
    struct E { const char* n; int64_t off, cnt; };
    const E table[] = {
        {"T_camera_local_f64", ws.T_cl, (int64_t)B * V * 24},
        {"kv_cache", ws.kv, h->cache_mode() ? 0 : (int64_t)h->nl * B * 2 * N * C},
        {"ref", ws.ref, M * 3}, {"ref_next", ws.ref_next, M * 3}, {"posemb", ws.emb, M * 384}, {"pos_hidden", ws.pe_h, M * C}, {"pos_feat", ws.pos, M * C},
        {"tgt", ws.tgt, M * C}, {"self_qkv", ws.qkv, M * 3 * C}, {"attn", ws.attn, M * C}, {"xa_prenorm1", ws.xa, M * C},
        {"cross_q", ws.qc, M * C}, {"xb_prenorm2", ws.xb, M * C}, {"ffn_hidden", ws.ffn, M * F},
        {"xc_prenorm3", ws.xc, M * C}, {"heads1", ws.h1, M * h->NH1}, {"heads2", ws.h2, M * 2 * C},
        {"gn_sums_f64", ws.gn_sums, (int64_t)2 * B * 4 * kGnSlots * 2}, {"ln1_stats", ws.ln1, M * 2}, {"ln2_stats", ws.ln2, M * 2},
        {"flags", ws.flags, 64}};
    for (const E& e : table)
        if (strcmp(e.n, name) == 0) { *offset_floats = (size_t)e.off; *numel = (size_t)e.cnt; return PARQ_OK; }
    return fail(PARQ_ERR_ARG, "unknown workspace buffer '%s'", name);
}

int parq_set_attention_mode(parq_handle h, int32_t mode) {
    if (!h || mode < 0 || mode > 4) return fail(PARQ_ERR_ARG, "attention mode must be 0 (fp32 MFMA), 1 (split fp16x3), 2 (fp16), 3 (bf16) or 4 (split, fp8 cross terms)");
    if (mode == 4 && !(h->dh == 64 || h->dh == 256)) return fail(PARQ_ERR_ARG, "attention mode 4 needs a head dim of 64 (256: runs as mode 1)");
    if (mode >= 2 && mode <= 3 && !((h->dh == 64 && h->C <= 256 && (2 * h->C) % 256 == 0) || (h->dh == 256 && h->C % 128 == 0 && (h->C == 256 || kvproj_big_on())))) 
        return fail(PARQ_ERR_ARG, "the fp16 / bf16 attention modes need head dim 64 with dim in {128, 256}, or head dim 256 with dim a multiple of 128");
    h->attn_mode = mode;
    h->prepared = false;
    return PARQ_OK;
}

int parq_set_head_tiers(parq_handle h, uint32_t safe_mask, int32_t poison_on_peaked) {
    if (!h) return fail(PARQ_ERR_ARG, "NULL handle");
    if (safe_mask != 0 && h->H > 16) return fail(PARQ_ERR_ARG, "per-head tiers need at most 16 heads");
    if ((safe_mask & h->all_heads()) != h->safe_heads()) h->prepared = false;      // the K/V cache is laid out per tier
    h->safe_mask = safe_mask;
    h->peaky_poison = poison_on_peaked ? 1 : 0;
    return PARQ_OK;
}

int parq_set_seam_fusion(parq_handle h, int32_t on) {
    if (!h) return fail(PARQ_ERR_ARG, "NULL handle");
    h->seam_fusion = on != 0;
    return PARQ_OK;
}

int parq_set_range_mirror(parq_handle h, int32_t* host_visible_flag) {
    if (!h) return fail(PARQ_ERR_ARG, "NULL handle");
    h->range_mirror = host_visible_flag;
    return PARQ_OK;
}

/* ---- training ---------------------------------------------------- */
size_t parq_train_workspace_bytes(parq_handle h, int32_t B, int32_t V, int32_t hh, int32_t ww) {
    if (!h || B < 1 || V < 1 || hh < 1 || ww < 1) return 0;
    Workspace ws;
    carve_workspace(h, B, V, hh, ww, &ws);
    return (size_t)ws.train_total * sizeof(float);
}

int parq_set_dropout(parq_handle h, float p, uint32_t seed) {
    if (!h || !(p >= 0.f) || !(p < 1.f)) return fail(PARQ_ERR_ARG, "dropout probability must be in [0, 1)");
    h->drop_p = p;
    h->drop_seed = seed;
    return PARQ_OK;
}

int parq_k_dropout_mask(parq_handle h, int32_t iteration, int32_t site, int64_t rows, int64_t cols, float* out, parq_stream stream) {
    if (!h || !out || rows < 1 || cols < 1 || rows * cols > INT32_MAX) return fail(PARQ_ERR_ARG, "bad argument");
    if (!(h->drop_p > 0.f)) return fail(PARQ_ERR_STATE, "dropout is off");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(launch_fill(out, 1.f, rows * cols, s));
    HIPCHK(launch_dropout_apply(out, out, (int)rows, (int)cols, h->drop_p, h->site_seed(iteration, site), s));
    return PARQ_OK;
}

size_t parq_grad_arena_bytes(parq_handle h) { return h ? (size_t)h->ar.rowmajor_total * sizeof(float) : 0; }

int parq_forward_train(parq_handle h, const parq_scene* scene, void* workspace, size_t workspace_bytes, const parq_outputs* outs,
                       parq_stream stream) {
    if (!h || !workspace) return fail(PARQ_ERR_ARG, "NULL argument");
    if (!h->packed) return fail(PARQ_ERR_STATE, "parq_pack_weights must be called first");
    if (h->dh % 16 != 0) return fail(PARQ_ERR_ARG, "training needs a head dim that is a multiple of 16");
    int rc = check_scene(h, scene);
    if (rc) return rc;
    rc = check_outs(outs);
    if (rc) return rc;
    Workspace ws;
    carve_workspace(h, scene->B, scene->V, scene->h, scene->w, &ws);
    if (workspace_bytes < (size_t)ws.train_total * sizeof(float)) return fail(PARQ_ERR_WORKSPACE, "training workspace too small: %zu < %zu", workspace_bytes, (size_t)ws.train_total * sizeof(float));
    float* wsp = (float*)workspace;
    hipStream_t s = (hipStream_t)stream;
    rc = do_prepare(h, scene, wsp, ws, s, true);
    if (rc) return rc;
    const int64_t M = (int64_t)scene->B * h->Q;
    // the reference points of iteration k live in that iteration's stash (initial_ref wrote ws.ref)
    HIPCHK(hipMemcpyAsync(wsp + ws.shift(0) + ws.refk, wsp + ws.ref, (size_t)M * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    for (int k = 0; k < h->I; ++k) {
        parq_outputs o;
        o.pred_logits = outs->pred_logits + k * M * h->ncls;
        o.center_unnormalized = outs->center_unnormalized + k * M * 3;
        o.size_unnormalized = outs->size_unnormalized + k * M * 3;
        o.ortho6d = outs->ortho6d + k * M * 6;
        o.sem_cls_prob = outs->sem_cls_prob + k * M * h->ncls;
        o.coord_pos = outs->coord_pos + k * M * 3;
        const bool last = k + 1 == h->I;
        float* ref_next = last ? wsp + ws.ref_next : wsp + ws.shift(k + 1) + ws.refk;
        float* emb_next = last ? wsp + ws.g_emb : wsp + ws.shift(k + 1) + ws.emb;
        rc = do_iterate(h, scene, wsp, ws, k, wsp + ws.shift(k) + ws.refk, true, &o, ref_next, s, ws.shift(k), emb_next, true);   // ws.emb of iteration 0: prologue
        if (rc) return rc;
        // outputs of iteration k are final from here on: a host that matches predictions to boxes per iteration (the set loss)
        // can start on them while the later iterations run (parq_wait_iteration)
        if (k < 16) {
            if (!h->iter_done[k]) HIPCHK(hipEventCreateWithFlags(&h->iter_done[k], hipEventDisableTiming));
            HIPCHK(hipEventRecord(h->iter_done[k], s));
            h->iter_recorded[k] = true;
        }
    }
    // (after the iterations, which do not read it: the stream gets to this while the host evaluates the loss)
    if (h->cache_mode() && !bwd_reads_cache(h, ws)) {
        // cache modes: the forward streams the 16-bit cache; the backward gets fp32 K / V rebuilt from it (hi + lo; in the fp16 /
        // bf16 modes the rounded values themselves: the gradient is taken straight through the rounding).  Not at head dim 64 with
        // the batched backward: attn_bwd_split2_kernel reads the cache itself (bwd_reads_cache)
        const int64_t N = (int64_t)scene->V * scene->h * scene->w;
        const int C = h->C, dh = h->dh, H = h->H, B = scene->B;
        for (int li = 0; li < h->nl; ++li) {
            const char* cache = reinterpret_cast<const char*>(wsp + ws.kvc) + (size_t)li * kvsplit_cache_bytes(B, h->vheads(), (int)N, h->terms());
            float* kv = wsp + ws.kv_train + (int64_t)li * B * 2 * N * C;
            HIPCHK(launch_kvsplit_to_f32(cache, B, h->vheads(), (int)N, kv, kv + (int64_t)H * N * dh, 2 * N * C, N * dh, 2 * N * C, N * dh, s,
                                         dh / 64, h->terms(), h->kind()));
        }
    }
    h->ref_state = 0;
    h->prepared = false;
    return PARQ_OK;
}

int parq_wait_iteration(parq_handle h, int32_t k) {
    if (!h || k < 0 || k >= 16 || k >= h->I) return fail(PARQ_ERR_ARG, "bad iteration index");
    if (!h->iter_recorded[k]) return fail(PARQ_ERR_STATE, "parq_wait_iteration: no parq_forward_train has been enqueued");
    HIPCHK(hipEventSynchronize(h->iter_done[k]));
    return PARQ_OK;
}

int parq_backward(parq_handle h, const parq_scene* scene, void* workspace, size_t workspace_bytes, const parq_outputs* outs,
                  const parq_output_grads* g, float* grad_arena, float* d_tokens, parq_stream stream) {
    if (!h || !workspace || !outs || !g || !grad_arena) return fail(PARQ_ERR_ARG, "NULL argument");
    if (!h->packed) return fail(PARQ_ERR_STATE, "parq_pack_weights must be called first");
    int rc = check_scene(h, scene);
    if (rc) return rc;
    Workspace ws;
    carve_workspace(h, scene->B, scene->V, scene->h, scene->w, &ws);
    if (workspace_bytes < (size_t)ws.train_total * sizeof(float)) return fail(PARQ_ERR_WORKSPACE, "training workspace too small");
    float* wsp = (float*)workspace;
    hipStream_t s = (hipStream_t)stream;
    const int64_t M = (int64_t)scene->B * h->Q;
    const int64_t N = (int64_t)scene->V * scene->h * scene->w;
    HIPCHK(hipMemsetAsync(grad_arena, 0, (size_t)h->ar.rowmajor_total * sizeof(float), s));
    if (!ws.bwd_batched) HIPCHK(hipMemsetAsync(wsp + ws.g_kv, 0, (size_t)h->nl * scene->B * 2 * N * h->C * sizeof(float), s));
    if (d_tokens) HIPCHK(hipMemsetAsync(d_tokens, 0, (size_t)scene->B * N * h->C * sizeof(float), s));
    auto iter_io = [&](int k) {
        BwdIO io;
        io.g_logits = g->pred_logits ? g->pred_logits + k * M * h->ncls : nullptr;
        io.g_center = g->center_unnormalized ? g->center_unnormalized + k * M * 3 : nullptr;
        io.g_size = g->size_unnormalized ? g->size_unnormalized + k * M * 3 : nullptr;
        io.g_rot = g->ortho6d ? g->ortho6d + k * M * 6 : nullptr;
        io.center = outs->center_unnormalized + k * M * 3;
        io.size = outs->size_unnormalized + k * M * 3;
        return io;
    };
    for (int i = 0; i < 2; ++i)
        if (!h->bucket_done[i]) HIPCHK(hipEventCreateWithFlags(&h->bucket_done[i], hipEventDisableTiming));
    auto finish = [&](bool early_too) -> int {          // the end of the backward: every gradient is final
        const int r = do_backward_kvproj(h, scene, wsp, ws, grad_arena, d_tokens, s);
        if (r) return r;
        if (early_too) HIPCHK(hipEventRecord(h->bucket_done[0], s));
        HIPCHK(hipEventRecord(h->bucket_done[1], s));
        h->bucket_recorded = true;
        return PARQ_OK;
    };
    if (!ws.bwd_batched) {
        for (int k = h->I - 1; k >= 0; --k) {
            rc = do_backward_iter(h, scene, wsp, ws, k, iter_io(k), grad_arena, d_tokens, s);
            if (rc) return rc;
        }
        return finish(true);
    }
    // ---- batched: (1) every iteration from its outputs down to dO, (2) ONE cross-attention backward over all iterations
    // (dK / dV written once, accumulated in registers), (3) every iteration from its dQ down to the sampled features
    const int I = h->I, B = scene->B, C = h->C, H = h->H, dh = h->dh, Q = h->Q;
    rc = do_backward_transposes(h, wsp, ws, 0, s);
    if (rc) return rc;
    // the iterations of a phase on ws.g_sets streams (the caller's + the handle's own), joined back before the phase ends
    const int ns = h->profiling ? 1 : ws.g_sets;
    if (ns > 1 && !h->fork_ev) {
        HIPCHK(hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
        for (int i = 0; i < 7; ++i) {
            HIPCHK(hipStreamCreateWithFlags(&h->aux_stream[i], hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&h->join_ev[i], hipEventDisableTiming));
        }
    }
    auto run_phase = [&](int phase) -> int {
        if (ns > 1) {
            HIPCHK(hipEventRecord(h->fork_ev, s));
            for (int i = 0; i + 1 < ns; ++i) HIPCHK(hipStreamWaitEvent(h->aux_stream[i], h->fork_ev, 0));
        }
        int r = PARQ_OK;
        for (int k = I - 1, n = 0; k >= 0 && r == PARQ_OK; --k, ++n) {
            const int set = n % ns;
            r = do_backward_iter(h, scene, wsp, ws, k, iter_io(k), grad_arena, d_tokens, set == 0 ? s : h->aux_stream[set - 1], phase, set);
        }
        // joined on the error path too: whatever the auxiliary streams were given must not outlive the call on a stream the
        // caller does not know about
        for (int i = 0; i + 1 < ns; ++i) {
            const hipError_t e1 = hipEventRecord(h->join_ev[i], h->aux_stream[i]);
            const hipError_t e2 = hipStreamWaitEvent(s, h->join_ev[i], 0);
            if (r == PARQ_OK && (e1 != hipSuccess || e2 != hipSuccess)) r = fail(PARQ_ERR_HIP, "joining the backward's auxiliary streams failed");
        }
        return r;
    };
    rc = run_phase(1);
    if (rc) return rc;
    // gradients of everything above the cross-attention (cross out-proj, FFN, norm2 / norm3, the heads) are final: a data-parallel
    // caller may start all-reducing that bucket on another stream while the attention backward runs (parq_backward_wait_bucket)
    HIPCHK(hipEventRecord(h->bucket_done[0], s));
    {
        int64_t q_off[16], lse_off[16];
        uint32_t seeds[16];
        for (int k = 0; k < I; ++k) {
            q_off[k] = ws.shift(k);
            lse_off[k] = ws.shift(k);
            seeds[k] = h->site_seed(k, 2);
        }
        const float* kv = wsp + (h->cache_mode() ? ws.kv_train : ws.kv);
        float* gkv = wsp + ws.g_kv;
        const int64_t MC = M * C;
        HIPCHK(hipMemsetAsync(wsp + ws.g_dq, 0, (size_t)I * MC * sizeof(float), s));      // the partial-sum reduction accumulates
        HIPCHK(launch_attn_bwd_batched(wsp + ws.qc, q_off, (int64_t)Q * C, dh, C, kv, 2 * N * C, N * dh, dh, kv + (int64_t)H * N * dh,
                                       2 * N * C, N * dh, dh, wsp + ws.g_do, MC, (int64_t)Q * C, dh, C, wsp + ws.lse_c, lse_off,
                                       wsp + ws.g_Dall, (int64_t)B * H * flash_lq_pad(Q), wsp + ws.g_dq, MC, (int64_t)Q * C, dh, C, gkv,
                                       2 * N * C, dh, 2 * C, gkv + C, 2 * N * C, dh, 2 * C, B, H, Q, (int)N, dh, I, s, wsp + ws.g_dqp,
                                       h->drop_p, seeds, reinterpret_cast<unsigned int*>(wsp + ws.g_bs),
                                       reinterpret_cast<unsigned int*>(wsp + ws.g_kvmax), wsp + ws.g_pack, wsp + ws.g_mat,
                                       bwd_reads_cache(h, ws) ? reinterpret_cast<const void*>(wsp + ws.kvc) : nullptr,
                                       h->terms_for(N, true, bwd_reads_cache(h, ws)), h->kind()));
        if (dh == 256) {                          // max |dK|, |dV| for the split-precision dW_kv GEMM (the dh = 64 kernel records it itself)
            HIPCHK(hipMemsetAsync(wsp + ws.g_kvmax, 0, sizeof(unsigned int), s));
            HIPCHK(launch_absmax(gkv, (int64_t)B * 2 * N * C, reinterpret_cast<unsigned int*>(wsp + ws.g_kvmax), s));
        }
    }
    rc = run_phase(2);
    if (rc) return rc;
    return finish(false);
}

/* Data-parallel gradient buckets (train.py:103 DDP buckets its all-reduces and overlaps them with the backward): the gradient
 * arena in the order its parts become final inside parq_backward.  Bucket 0 = [offset, offset + count) floats final after phase
 * 1 of the batched backward (cross out-proj, FFN, norm2, norm3, every head: everything above the cross-attention), bucket 1 =
 * the rest of the arena's front, final at the end.  With unshared layer weights (several layers interleave) bucket 0 is empty
 * and bucket 1 is the whole arena. */
int parq_grad_bucket(parq_handle h, int32_t bucket, int64_t* offset, int64_t* count) {
    if (!h || !offset || !count || bucket < 0 || bucket > 1) return fail(PARQ_ERR_ARG, "bad argument");
    const Arena& ar = h->ar;
    const bool two = h->nl == 1 && ar.early_end > ar.early_begin;
    if (bucket == 0) { *offset = two ? ar.early_begin : 0; *count = two ? ar.early_end - ar.early_begin : 0; }
    else { *offset = 0; *count = two ? ar.early_begin : ar.rowmajor_total; }
    return PARQ_OK;
}

int parq_backward_wait_bucket(parq_handle h, int32_t bucket, parq_stream stream) {
    if (!h || bucket < 0 || bucket > 1) return fail(PARQ_ERR_ARG, "bad argument");
    if (!h->bucket_recorded) return fail(PARQ_ERR_STATE, "parq_backward_wait_bucket: no parq_backward has been enqueued");
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, h->bucket_done[bucket], 0));
    return PARQ_OK;
}

/* offset (in floats) and element count of a named reference tensor inside the packed weight / gradient arena;
 * `rows` x `cols` with row stride `ld` (fused head matrices are row slices of a wider block) */
int parq_arena_lookup(parq_handle h, const char* name, int64_t* offset, int64_t* rows, int64_t* cols, int64_t* ld) {
    if (!h || !name || !offset || !rows || !cols || !ld) return fail(PARQ_ERR_ARG, "NULL argument");
    const int64_t C = h->C, F = h->F, Q = h->Q, ncls = h->ncls;
    const Arena& ar = h->ar;
    struct E { std::string n; int64_t off, r, c; };
    std::vector<E> t;
    for (int li = 0; li < h->nl; ++li) {
        const LayerW& L = ar.layers[li];
        const std::string p = "parq_module.decoder.layers." + std::to_string(li) + ".";
        t.push_back({p + "self_attn.in_proj_weight", L.self_in_w, 3 * C, C}); t.push_back({p + "self_attn.in_proj_bias", L.self_in_b, 1, 3 * C});
        t.push_back({p + "self_attn.out_proj.weight", L.self_out_w, C, C}); t.push_back({p + "self_attn.out_proj.bias", L.self_out_b, 1, C});
        t.push_back({p + "multihead_attn.in_proj_weight", L.cross_in_w, 3 * C, C}); t.push_back({p + "multihead_attn.in_proj_bias", L.cross_in_b, 1, 3 * C});
        t.push_back({p + "multihead_attn.out_proj.weight", L.cross_out_w, C, C}); t.push_back({p + "multihead_attn.out_proj.bias", L.cross_out_b, 1, C});
        t.push_back({p + "linear1.weight", L.lin1_w, F, C}); t.push_back({p + "linear1.bias", L.lin1_b, 1, F});
        t.push_back({p + "linear2.weight", L.lin2_w, C, F}); t.push_back({p + "linear2.bias", L.lin2_b, 1, C});
        t.push_back({p + "norm1.weight", L.n1_w, 1, C}); t.push_back({p + "norm1.bias", L.n1_b, 1, C});
        t.push_back({p + "norm2.weight", L.n2_w, 1, C}); t.push_back({p + "norm2.bias", L.n2_b, 1, C});
        t.push_back({p + "norm3.weight", L.n3_w, 1, C}); t.push_back({p + "norm3.bias", L.n3_b, 1, C});
    }
    const std::string d = "parq_module.decoder.";
    const std::string hc = "mlp_heads.center_head.layers.", hr = "mlp_heads.rotation_head.layers.";
    t.push_back({"refpoint.weight", ar.refpoint, Q, 3});
    t.push_back({d + "position_encoder.0.weight", ar.pe0_w, C, 384}); t.push_back({d + "position_encoder.0.bias", ar.pe0_b, 1, C});
    t.push_back({d + "position_encoder.2.weight", ar.pe2_w, C, C}); t.push_back({d + "position_encoder.2.bias", ar.pe2_b, 1, C});
    t.push_back({hc + "0.weight", ar.heads1_w, C, C}); t.push_back({hr + "0.weight", ar.heads1_w + C * C, C, C});
    t.push_back({"mlp_heads.sem_cls_head.layers.0.weight", ar.heads1_w + 2 * C * C, ncls, C});
    t.push_back({"mlp_heads.size_head.layers.0.weight", ar.heads1_w + (2 * C + ncls) * C, 3, C});
    t.push_back({"mlp_heads.sem_cls_head.layers.0.bias", ar.heads1_b + 2 * C, 1, ncls});
    t.push_back({"mlp_heads.size_head.layers.0.bias", ar.heads1_b + 2 * C + ncls, 1, 3});
    t.push_back({hc + "1.weight", ar.gn1_g, 1, C}); t.push_back({hr + "1.weight", ar.gn1_g + C, 1, C});
    t.push_back({hc + "1.bias", ar.gn1_b, 1, C}); t.push_back({hr + "1.bias", ar.gn1_b + C, 1, C});
    t.push_back({hc + "4.weight", ar.heads2_w, C, C}); t.push_back({hr + "4.weight", ar.heads2_w + C * C, C, C});
    t.push_back({hc + "5.weight", ar.gn2_g, 1, C}); t.push_back({hr + "5.weight", ar.gn2_g + C, 1, C});
    t.push_back({hc + "5.bias", ar.gn2_b, 1, C}); t.push_back({hr + "5.bias", ar.gn2_b + C, 1, C});
    t.push_back({hc + "8.weight", ar.heads3_w, 3, C}); t.push_back({hr + "8.weight", ar.heads3_w + 6 * C, 6, C});
    t.push_back({hc + "8.bias", ar.heads3_b, 1, 3}); t.push_back({hr + "8.bias", ar.heads3_b + 6, 1, 6});
    for (const E& e : t)
        if (e.n == name) { *offset = e.off; *rows = e.r; *cols = e.c; *ld = e.c; return PARQ_OK; }
    return fail(PARQ_ERR_ARG, "unknown tensor '%s'", name);
}

int parq_profile_enable(parq_handle h, int32_t on) {
    if (!h) return fail(PARQ_ERR_ARG, "NULL handle");
    h->profiling = on != 0;
    return PARQ_OK;
}

int parq_profile_read(parq_handle h, int32_t which, double* total_ms, int64_t* launches) {
    if (!h || which < 0 || which >= PARQ_PROF_COUNT) return fail(PARQ_ERR_ARG, "bad argument");
    for (auto& e : h->events) {
        HIPCHK(hipEventSynchronize(e.b));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
        h->prof_ms[e.which] += ms;
        h->prof_n[e.which] += 1;
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    h->events.clear();
    if (total_ms) *total_ms = h->prof_ms[which];
    if (launches) *launches = h->prof_n[which];
    h->prof_ms[which] = 0;
    h->prof_n[which] = 0;
    return PARQ_OK;
}

// ------------------------------------------------------------------ single kernels

int parq_k_camera_local(const float* T_cp, const float* T_wp, const float* T_wl, int32_t B, int32_t V, float* T_cl,
                        parq_stream stream) {
    if (!T_cp || !T_wp || !T_wl || !T_cl || B < 1 || V < 1) return fail(PARQ_ERR_ARG, "bad argument");
    HIPCHK(launch_camera_local(T_cp, T_wp, T_wl, B, V, T_cl, (hipStream_t)stream));
    return PARQ_OK;
}

int parq_k_project_sample(const float* tokens, const float* T_cl, const float* camera, const float* ref,
                          const float* scale6_host, int32_t B, int32_t V, int32_t hh, int32_t ww, int32_t C, int32_t Q,
                          float* tgt, float* coord_pos, parq_stream stream) {
    if (!tokens || !T_cl || !camera || !ref || !scale6_host || !tgt) return fail(PARQ_ERR_ARG, "NULL argument");
    if (B < 1 || V < 1 || hh < 2 || ww < 2 || Q < 1 || C % 4 != 0 || C > 1024) return fail(PARQ_ERR_ARG, "bad dims");
    ScaleBox sb;
    for (int i = 0; i < 3; ++i) { sb.lo[i] = scale6_host[2 * i]; sb.hi[i] = scale6_host[2 * i + 1]; }
    HIPCHK(launch_project_sample(tokens, T_cl, camera, ref, sb, B, V, hh, ww, C, Q, tgt, coord_pos, (hipStream_t)stream));
    return PARQ_OK;
}

int parq_k_linear(const float* X, const float* X2, const float* W, const float* bias, const float* R, float* Y,
                  int32_t M, int32_t N, int32_t K, int32_t relu, parq_stream stream) {
    if (!X || !W || !Y || M < 1 || N < 1 || K < 32 || K % 32 != 0) return fail(PARQ_ERR_ARG, "bad argument (K must be a multiple of 32)");
    LinearArgs a = lin(X, K, W, K, bias, Y, N, M, N, K);
    a.X2 = X2; a.ldx2 = K; a.x2_ncols = N;
    a.R = R; a.ldr = N; a.relu = relu;
    HIPCHK(launch_linear(a, 1, (hipStream_t)stream));
    return PARQ_OK;
}

int parq_k_linear_half(const float* X, const float* X2, const float* W, const float* bias, const float* R, float* Y,
                       int32_t M, int32_t N, int32_t K, int32_t relu, void* scratch, size_t scratch_bytes, parq_stream stream) {
    if (!X || !W || !Y || !scratch || M < 16 || N < 16 || M % 16 != 0 || N % 16 != 0 || !(K == 1024 || K == 768))
        return fail(PARQ_ERR_ARG, "bad argument (K in {1024, 768}, M and N multiples of 16)");
    if (scratch_bytes < ((size_t)N * K + N) * sizeof(float) || (reinterpret_cast<uintptr_t>(scratch) & 15u))
        return fail(PARQ_ERR_ARG, "scratch: N * K * 4 + N * 4 bytes, 16-byte aligned");
    float* wh = static_cast<float*>(scratch);
    float* sc = wh + (size_t)N * K;
    HIPCHK(launch_pack_w_half(W, K, N, K, wh, sc, (hipStream_t)stream));
    LinearArgs a = lin(X, K, W, K, bias, Y, N, M, N, K);
    a.X2 = X2; a.ldx2 = K; a.x2_ncols = N;
    a.R = R; a.ldr = N; a.relu = relu;
    a.Wh = wh; a.wh_scale = sc;
    const hipError_t e = launch_chain_linear(a, 1, (hipStream_t)stream);
    if (e == hipErrorNotSupported) return fail(PARQ_ERR_ARG, "no fp16 x 3 tile for this combination of addend / bias / ReLU / residual");
    HIPCHK(e);
    return PARQ_OK;
}

size_t parq_k_attention_scratch_bytes(int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t dh) {
    const int ns = flash_pick_splits(B, H, Lq, Lk, dh, device_num_cus());
    return flash_scratch_bytes(B, H, Lq, dh, ns);
}

int parq_k_attention(const float* q, const float* k, const float* v, float* out, int32_t B, int32_t H, int32_t Lq,
                     int32_t Lk, int32_t dh, void* scratch, size_t scratch_bytes, parq_stream stream) {
    if (!q || !k || !v || !out || !scratch) return fail(PARQ_ERR_ARG, "NULL argument");
    if (!(dh == 32 || dh == 64 || dh == 128 || dh == 256) || B < 1 || H < 1 || Lq < 1 || Lk < 1) return fail(PARQ_ERR_ARG, "bad dims");
    FlashArgs fa;
    memset(&fa, 0, sizeof(fa));
    const int64_t C = (int64_t)H * dh;
    fa.B = B; fa.H = H; fa.Lq = Lq; fa.Lk = Lk; fa.dh = dh;
    fa.q = q; fa.q_batch = Lq * C; fa.q_head = dh; fa.q_row = C;
    fa.k = k; fa.k_batch = Lk * C; fa.k_head = dh; fa.k_row = C;
    fa.v = v; fa.v_batch = Lk * C; fa.v_head = dh; fa.v_row = C;
    fa.out = out; fa.out_batch = Lq * C; fa.out_row = C;
    fa.nsplit = flash_pick_splits(B, H, Lq, Lk, dh, device_num_cus());
    if (scratch_bytes < flash_scratch_bytes(B, H, Lq, dh, fa.nsplit)) return fail(PARQ_ERR_WORKSPACE, "attention scratch too small");
    const int64_t lp = flash_lq_pad(Lq);
    fa.o_part = (float*)scratch;
    fa.m_part = fa.o_part + (int64_t)B * H * fa.nsplit * dh * lp;
    fa.l_part = fa.m_part + (int64_t)B * H * fa.nsplit * lp;
    HIPCHK(launch_flash(fa, (hipStream_t)stream));
    HIPCHK(launch_flash_merge(fa, (hipStream_t)stream));
    return PARQ_OK;
}

size_t parq_k_attention_split_scratch_bytes(int32_t B, int32_t H, int32_t Lq, int32_t Lk) {
    const int ns = flash_split_pick_splits(B, H, Lq, Lk, device_num_cus());
    return flash_scratch_bytes(B, H, Lq, 64, ns) + kvsplit_cache_bytes(B, H, Lk) + 256;
}

int parq_k_attention_split(const float* q, const float* k, const float* v, float* out, int32_t B, int32_t H, int32_t Lq,
                           int32_t Lk, void* scratch, size_t scratch_bytes, parq_stream stream) {
    if (!q || !k || !v || !out || !scratch) return fail(PARQ_ERR_ARG, "NULL argument");
    if (B < 1 || H < 1 || Lq < 1 || Lk < 1) return fail(PARQ_ERR_ARG, "bad dims");
    if (scratch_bytes < parq_k_attention_split_scratch_bytes(B, H, Lq, Lk)) return fail(PARQ_ERR_WORKSPACE, "attention scratch too small");
    hipStream_t s = (hipStream_t)stream;
    const int dh = 64;
    const int64_t C = (int64_t)H * dh;
    FlashArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.B = B; fa.H = H; fa.Lq = Lq; fa.Lk = Lk; fa.dh = dh;
    fa.q = q; fa.q_batch = Lq * C; fa.q_head = dh; fa.q_row = C;
    fa.out = out; fa.out_batch = Lq * C; fa.out_row = C;
    fa.nsplit = flash_split_pick_splits(B, H, Lq, Lk, device_num_cus());
    const int64_t lp = flash_lq_pad(Lq);
    char* base = (char*)scratch;
    int* flag = (int*)base;
    char* cache = base + 256;
    fa.o_part = (float*)(cache + kvsplit_cache_bytes(B, H, Lk));
    fa.m_part = fa.o_part + (int64_t)B * H * fa.nsplit * dh * lp;
    fa.l_part = fa.m_part + (int64_t)B * H * fa.nsplit * lp;
    HIPCHK(hipMemsetAsync(flag, 0, 256, s));
    HIPCHK(launch_kvsplit_convert(k, v, Lk * C, dh, C, Lk * C, dh, C, B, H, Lk, cache, flag, s));
    HIPCHK(launch_flash_split(fa, cache, s));
    HIPCHK(launch_flash_merge(fa, s));
    return PARQ_OK;
}

/* mode 4: hi.hi on the fp16 matrix pipe, the cross terms as MX-scaled fp8 products (flash_split8.hip); Lk % 64 == 0 */
int parq_k_attention_split8(const float* q, const float* k, const float* v, float* out, int32_t B, int32_t H, int32_t Lq,
                            int32_t Lk, void* scratch, size_t scratch_bytes, parq_stream stream) {
    if (!q || !k || !v || !out || !scratch) return fail(PARQ_ERR_ARG, "NULL argument");
    if (B < 1 || H < 1 || Lq < 1 || Lk < 1) return fail(PARQ_ERR_ARG, "bad dims");
    if (!flash_split8_supported(64, Lk)) return fail(PARQ_ERR_ARG, "attention mode 4 needs a key count that is a multiple of 64");
    if (scratch_bytes < parq_k_attention_split_scratch_bytes(B, H, Lq, Lk)) return fail(PARQ_ERR_WORKSPACE, "attention scratch too small");
    hipStream_t s = (hipStream_t)stream;
    const int dh = 64;
    const int64_t C = (int64_t)H * dh;
    FlashArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.B = B; fa.H = H; fa.Lq = Lq; fa.Lk = Lk; fa.dh = dh;
    fa.q = q; fa.q_batch = Lq * C; fa.q_head = dh; fa.q_row = C;
    fa.out = out; fa.out_batch = Lq * C; fa.out_row = C;
    fa.nsplit = flash_split_pick_splits(B, H, Lq, Lk, device_num_cus());
    const int64_t lp = flash_lq_pad(Lq);
    char* cache = (char*)scratch + 256;
    fa.o_part = (float*)(cache + kvsplit_cache_bytes(B, H, Lk));
    fa.m_part = fa.o_part + (int64_t)B * H * fa.nsplit * dh * lp;
    fa.l_part = fa.m_part + (int64_t)B * H * fa.nsplit * lp;
    HIPCHK(launch_kvsplit8_convert(k, v, Lk * C, dh, C, Lk * C, dh, C, B, H, Lk, cache, s));
    HIPCHK(launch_flash_split8(fa, cache, s));
    HIPCHK(launch_flash_merge(fa, s));
    return PARQ_OK;
}

size_t parq_k_attention_split256_scratch_bytes(int32_t B, int32_t H, int32_t Lq, int32_t Lk) {
    const int ns = flash_split256_pick_splits(B, H, Lq, Lk, device_num_cus());
    return flash_scratch_bytes(B, H, Lq, 256, ns) + kvsplit_cache_bytes(B, 4 * H, Lk) + 256;
}

/* head dim 256: a head is stored as 4 virtual heads of 64 in the split cache (flash_split256.hip) */
int parq_k_attention_split256(const float* q, const float* k, const float* v, float* out, int32_t B, int32_t H, int32_t Lq,
                              int32_t Lk, void* scratch, size_t scratch_bytes, parq_stream stream) {
    if (!q || !k || !v || !out || !scratch) return fail(PARQ_ERR_ARG, "NULL argument");
    if (B < 1 || H < 1 || Lq < 1 || Lk < 1) return fail(PARQ_ERR_ARG, "bad dims");
    if (scratch_bytes < parq_k_attention_split256_scratch_bytes(B, H, Lq, Lk)) return fail(PARQ_ERR_WORKSPACE, "attention scratch too small");
    hipStream_t s = (hipStream_t)stream;
    const int dh = 256;
    const int64_t C = (int64_t)H * dh;
    FlashArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.B = B; fa.H = H; fa.Lq = Lq; fa.Lk = Lk; fa.dh = dh;
    fa.q = q; fa.q_batch = Lq * C; fa.q_head = dh; fa.q_row = C;
    fa.out = out; fa.out_batch = Lq * C; fa.out_row = C;
    fa.nsplit = flash_split256_pick_splits(B, H, Lq, Lk, device_num_cus());
    const int64_t lp = flash_lq_pad(Lq);
    char* base = (char*)scratch;
    int* flag = (int*)base;
    char* cache = base + 256;
    fa.o_part = (float*)(cache + kvsplit_cache_bytes(B, 4 * H, Lk));
    fa.m_part = fa.o_part + (int64_t)B * H * fa.nsplit * dh * lp;
    fa.l_part = fa.m_part + (int64_t)B * H * fa.nsplit * lp;
    HIPCHK(hipMemsetAsync(flag, 0, 256, s));
    HIPCHK(launch_kvsplit_convert(k, v, Lk * C, 64, C, Lk * C, 64, C, B, 4 * H, Lk, cache, flag, s));     // 4 H virtual heads of 64
    HIPCHK(launch_flash_split256(fa, cache, s));
    HIPCHK(launch_flash_merge(fa, s));
    return PARQ_OK;
}

size_t parq_k_attention_half_scratch_bytes(int32_t B, int32_t H, int32_t Lq, int32_t Lk) {
    return parq_k_attention_split_scratch_bytes(B, H, Lq, Lk);      // the single-term cache is half the split one
}

int parq_k_attention_half(const float* q, const float* k, const float* v, float* out, int32_t B, int32_t H, int32_t Lq,
                          int32_t Lk, int32_t bf16, void* scratch, size_t scratch_bytes, parq_stream stream) {
    if (!q || !k || !v || !out || !scratch) return fail(PARQ_ERR_ARG, "NULL argument");
    if (B < 1 || H < 1 || Lq < 1 || Lk < 1) return fail(PARQ_ERR_ARG, "bad dims");
    if (scratch_bytes < parq_k_attention_half_scratch_bytes(B, H, Lq, Lk)) return fail(PARQ_ERR_WORKSPACE, "attention scratch too small");
    hipStream_t s = (hipStream_t)stream;
    const int dh = 64, kind = bf16 ? kBF16 : kF16;
    const int64_t C = (int64_t)H * dh;
    FlashArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.B = B; fa.H = H; fa.Lq = Lq; fa.Lk = Lk; fa.dh = dh;
    fa.q = q; fa.q_batch = Lq * C; fa.q_head = dh; fa.q_row = C;
    fa.out = out; fa.out_batch = Lq * C; fa.out_row = C;
    fa.nsplit = flash_split_pick_splits(B, H, Lq, Lk, device_num_cus());
    const int64_t lp = flash_lq_pad(Lq);
    char* base = (char*)scratch;
    int* flag = (int*)base;
    char* cache = base + 256;
    fa.o_part = (float*)(cache + kvsplit_cache_bytes(B, H, Lk, 3));
    fa.m_part = fa.o_part + (int64_t)B * H * fa.nsplit * dh * lp;
    fa.l_part = fa.m_part + (int64_t)B * H * fa.nsplit * lp;
    HIPCHK(hipMemsetAsync(flag, 0, 256, s));
    HIPCHK(launch_kvsplit_convert(k, v, Lk * C, dh, C, Lk * C, dh, C, B, H, Lk, cache, flag, s, 1, kind));
    HIPCHK(launch_flash_split(fa, cache, s, 1, kind));
    HIPCHK(launch_flash_merge(fa, s));
    return PARQ_OK;
}

// ------------------------------------------------------------------ AddRayPE + tokenisation
static int64_t raype_align(int64_t x) { return (x + 63) / 64 * 64; }

// Workspace layout: [hidden [M][C] — only when the hidden layer is kept] | W1 hi/lo (C*K1 halfs each) | W2 hi/lo (C*C halfs each)
// | W2 in fragment order (one-pass path) | pose + depth tables (float64) | points [M][K1] (generic path only).
// The hidden layer is kept (first region: parq_ray_pe_backward reads it there) unless the caller passes PARQ_RAYPE_NO_HIDDEN on
// the one-pass path (C = 256, 64 samples), where it never leaves the CU; the generic path needs it as an intermediate.
static bool raype_keeps_hidden(int32_t C, int32_t num_samples, int32_t flags) {
    return !((flags & PARQ_RAYPE_NO_HIDDEN) && C == 256 && num_samples == 64);
}
size_t parq_ray_pe_workspace_bytes_flags(int32_t B, int32_t V, int32_t hh, int32_t ww, int32_t C, int32_t num_samples, int32_t flags) {
    if (B < 1 || V < 1 || hh < 1 || ww < 1 || C < 1 || num_samples < 1) return 0;
    const int64_t M = (int64_t)B * V * hh * ww, K1 = 3 * (int64_t)num_samples;
    const bool fused = (C == 256 && num_samples == 64);
    const int64_t floats = (raype_keeps_hidden(C, num_samples, flags) ? raype_align(M * C) : 0) + raype_align(C * K1) +
                           2 * raype_align((int64_t)C * C) + raype_align(2 * ((int64_t)B * V * 12 + num_samples)) +
                           (fused ? 0 : raype_align(M * K1));
    return (size_t)floats * sizeof(float);
}
size_t parq_ray_pe_workspace_bytes(int32_t B, int32_t V, int32_t hh, int32_t ww, int32_t C, int32_t num_samples) {
    return parq_ray_pe_workspace_bytes_flags(B, V, hh, ww, C, num_samples, 0);
}

int parq_ray_pe(const float* camera, const float* T_cp, const float* T_wp, const float* T_wl, const float* w1,
                const float* b1, const float* w2, const float* b2, const float* scale6_host, float min_depth,
                float max_depth, int32_t num_samples, int32_t B, int32_t V, int32_t hh, int32_t ww, int32_t C,
                const float* features_nchw, float* tokens_out, int32_t flags, void* workspace, size_t workspace_bytes,
                parq_stream stream) {
    const int32_t nchw_out = flags & PARQ_RAYPE_NCHW_OUT;
    if (!camera || !T_cp || !T_wp || !T_wl || !w1 || !b1 || !w2 || !b2 || !scale6_host || !tokens_out || !workspace)
        return fail(PARQ_ERR_ARG, "NULL argument");
    if (B < 1 || V < 1 || hh < 1 || ww < 1 || num_samples < 1) return fail(PARQ_ERR_ARG, "bad dims");
    if ((3 * num_samples) % 64 != 0 || C % 64 != 0) return fail(PARQ_ERR_ARG, "3*num_samples and C must be multiples of 64");
    if (!(max_depth > min_depth && min_depth > 0.f)) return fail(PARQ_ERR_ARG, "need 0 < min_depth < max_depth");
    if (workspace_bytes < parq_ray_pe_workspace_bytes_flags(B, V, hh, ww, C, num_samples, flags)) return fail(PARQ_ERR_WORKSPACE, "ray-PE workspace too small");
    const int64_t M64 = (int64_t)B * V * hh * ww;
    if (M64 > INT32_MAX) return fail(PARQ_ERR_ARG, "too many tokens");
    const int M = (int)M64, K1 = 3 * num_samples;
    hipStream_t s = (hipStream_t)stream;
    float* wsp = (float*)workspace;
    const bool keep = raype_keeps_hidden(C, num_samples, flags);
    float* Hd = keep ? wsp : nullptr;
    float* W1s = wsp + (keep ? raype_align((int64_t)M * C) : 0);
    float* W2s = W1s + raype_align((int64_t)C * K1);
    float* W2f = W2s + raype_align((int64_t)C * C);
    double* tabs = reinterpret_cast<double*>(W2f + raype_align((int64_t)C * C));
    float* P = reinterpret_cast<float*>(tabs) + raype_align(2 * ((int64_t)B * V * 12 + num_samples));
    char* w1hi = (char*)W1s; char* w1lo = w1hi + (size_t)C * K1 * 2;
    char* w2hi = (char*)W2s; char* w2lo = w2hi + (size_t)C * C * 2;
    const bool cached = (flags & PARQ_RAYPE_WEIGHTS_CACHED) != 0;       // the workspace still holds this w1 / w2, split and packed
    if (!cached) {
        HIPCHK(launch_split_f32(w1, w1hi, w1lo, (int64_t)C * K1, s));
        HIPCHK(launch_split_f32(w2, w2hi, w2lo, (int64_t)C * C, s));
    }
    if (C == 256 && num_samples == 64) {
        // development: PARQ_RAYPE_TWO_KERNELS=1 runs the round-1 form (hidden tensor written and re-read) for A/B
        static const int two = [] { const char* e = dev_env("PARQ_RAYPE_TWO_KERNELS"); return e && e[0] == '1' ? 1 : 0; }();
        if (two && !Hd) return fail(PARQ_ERR_ARG, "the two-kernel form needs the hidden region (do not pass the no-hidden flag)");
        HIPCHK(launch_raype_fused(camera, T_cp, T_wp, T_wl, scale6_host, min_depth, max_depth, B, V, hh, ww, w1hi, w1lo, b1,
                                  w2hi, w2lo, b2, features_nchw, Hd, tabs, tabs + (int64_t)B * V * 12, tokens_out,
                                  nchw_out ? 1 : 0, s, W2f, two | (cached ? 2 : 0)));
        return PARQ_OK;
    }
    if (nchw_out) return fail(PARQ_ERR_ARG, "NCHW output needs the fused path (C = 256, 64 samples)");
    HIPCHK(launch_raype_points(camera, T_cp, T_wp, T_wl, scale6_host, min_depth, max_depth, B, V, hh, ww, num_samples, P, s));
    HIPCHK(launch_gemm_split(P, K1, w1hi, w1lo, b1, Hd, C, M, C, K1, 1, nullptr, 1, s));
    HIPCHK(launch_gemm_split(Hd, C, w2hi, w2lo, b2, tokens_out, C, M, C, C, 0, features_nchw, hh * ww, s));
    return PARQ_OK;
}

size_t parq_ray_pe_backward_workspace_bytes(int32_t B, int32_t V, int32_t hh, int32_t ww, int32_t C, int32_t num_samples) {
    if (B < 1 || V < 1 || hh < 1 || ww < 1 || C < 1 || num_samples < 1) return 0;
    const int64_t M = (int64_t)B * V * hh * ww, K1 = 3 * (int64_t)num_samples;
    // points [M][K1] | d hidden [M][C] | W2^T [C][C]
    return (size_t)(raype_align(M * K1) + raype_align(M * C) + raype_align((int64_t)C * C)) * sizeof(float);
}

int parq_ray_pe_backward(const float* camera, const float* T_cp, const float* T_wp, const float* T_wl, const float* w2,
                         const float* scale6_host, float min_depth, float max_depth, int32_t num_samples, int32_t B, int32_t V,
                         int32_t hh, int32_t ww, int32_t C, const float* d_tokens, const void* fwd_workspace, void* bwd_workspace,
                         size_t bwd_workspace_bytes, float* dw1, float* db1, float* dw2, float* db2, float* d_features_nchw,
                         parq_stream stream) {
    if (!camera || !T_cp || !T_wp || !T_wl || !w2 || !scale6_host || !d_tokens || !fwd_workspace || !bwd_workspace || !dw1 || !db1 ||
        !dw2 || !db2)
        return fail(PARQ_ERR_ARG, "NULL argument");
    if (B < 1 || V < 1 || hh < 1 || ww < 1 || num_samples < 1 || (3 * num_samples) % 64 != 0 || C % 64 != 0) return fail(PARQ_ERR_ARG, "bad dims");
    if (bwd_workspace_bytes < parq_ray_pe_backward_workspace_bytes(B, V, hh, ww, C, num_samples)) return fail(PARQ_ERR_WORKSPACE, "ray-PE backward workspace too small");
    const int64_t M64 = (int64_t)B * V * hh * ww;
    if (M64 > INT32_MAX) return fail(PARQ_ERR_ARG, "too many tokens");
    const int M = (int)M64, K1 = 3 * num_samples;
    hipStream_t s = (hipStream_t)stream;
    const float* Hd = (const float*)fwd_workspace;                  // hidden activations of the forward (first region)
    float* P = (float*)bwd_workspace;
    float* gHd = P + raype_align((int64_t)M * K1);
    float* W2T = gHd + raype_align((int64_t)M * C);
    // tokens = features + relu(p W1^T + b1) W2^T + b2   (ray_positional_encoding.py:128-136, parq_lightning.py:75)
    HIPCHK(hipMemsetAsync(db2, 0, (size_t)C * sizeof(float), s));
    HIPCHK(hipMemsetAsync(db1, 0, (size_t)C * sizeof(float), s));
    HIPCHK(launch_colsum(d_tokens, C, M, C, db2, 1, s));
    HIPCHK(hipMemsetAsync(dw2, 0, (size_t)C * C * sizeof(float), s));
    HIPCHK(launch_gemm_tn(d_tokens, C, Hd, C, dw2, C, M, C, C, 1, s));
    HIPCHK(launch_transpose(w2, C, W2T, C, C, C, s));
    {
        LinearArgs a = lin(d_tokens, C, W2T, C, nullptr, gHd, C, M, C, C);
        a.relu_mask = Hd; a.ldmask = C;
        HIPCHK(launch_linear(a, 1, s));
    }
    HIPCHK(launch_colsum(gHd, C, M, C, db1, 1, s));
    HIPCHK(launch_raype_points(camera, T_cp, T_wp, T_wl, scale6_host, min_depth, max_depth, B, V, hh, ww, num_samples, P, s));
    HIPCHK(hipMemsetAsync(dw1, 0, (size_t)C * K1 * sizeof(float), s));
    HIPCHK(launch_gemm_tn(gHd, C, P, K1, dw1, K1, M, C, K1, 1, s));
    if (d_features_nchw) {
        const int hw = hh * ww;
        for (int i = 0; i < B * V; ++i)                                 // (hw, C) -> (C, hw) per image
            HIPCHK(launch_transpose(d_tokens + (int64_t)i * hw * C, C, d_features_nchw + (int64_t)i * C * hw, hw, hw, C, s));
    }
    return PARQ_OK;
}

int parq_parse_pred(const float* center, const float* size, const float* ortho6d, const float* sem_cls_prob, int32_t B, int32_t Q,
                    int32_t num_classes, const float* track_scale6_host, int32_t for_vis, int32_t enable_nms, float* obbs_out,
                    unsigned char* mask_out, parq_stream stream) {
    if (!center || !size || !ortho6d || !sem_cls_prob || !track_scale6_host || !obbs_out || !mask_out) return fail(PARQ_ERR_ARG, "NULL argument");
    if (B < 1 || Q < 1 || Q > 1024 || num_classes < 2) return fail(PARQ_ERR_ARG, "bad dims (1 <= Q <= 1024)");
    HIPCHK(launch_parse_pred(center, size, ortho6d, sem_cls_prob, B, Q, num_classes, num_classes - 1, track_scale6_host, for_vis,
                             enable_nms, obbs_out, mask_out, (hipStream_t)stream));
    return PARQ_OK;
}

int parq_set_loss(const float* pred_logits, const float* center_unnormalized, const float* size_unnormalized, const float* ortho6d,
                  int32_t I, int32_t B, int32_t Q, int32_t num_classes, const float* t_center, const float* t_size, const float* t_rot,
                  const int32_t* t_label, const int32_t* t_sym, int32_t nmax, const int32_t* pairs, const float* pair_coef, int32_t P,
                  const float* row_weight, const float* class_weight, const float* loss_weight4_host, float* terms, float* g_logits,
                  float* g_center, float* g_size, float* g_ortho6d, int32_t* class_scratch, parq_stream stream) {
    if (!pred_logits || !center_unnormalized || !size_unnormalized || !ortho6d || !t_center || !t_size || !t_rot || !t_label || !row_weight ||
        !class_weight || !loss_weight4_host || !terms || !g_logits || !g_center || !g_size || !g_ortho6d || !class_scratch)
        return fail(PARQ_ERR_ARG, "NULL argument");
    if (I < 1 || B < 1 || Q < 1 || num_classes < 2 || nmax < 1 || P < 0 || (P > 0 && (!pairs || !pair_coef))) return fail(PARQ_ERR_ARG, "bad dims");
    HIPCHK(launch_set_loss(pred_logits, center_unnormalized, size_unnormalized, ortho6d, I, B, Q, num_classes, t_center, t_size, t_rot,
                           t_label, t_sym, nmax, pairs, pair_coef, P, row_weight, class_weight, loss_weight4_host, num_classes - 1, terms,
                           g_logits, g_center, g_size, g_ortho6d, class_scratch, (hipStream_t)stream));
    return PARQ_OK;
}

int parq_k_layernorm(const float* X, const float* gamma, const float* beta, float* Y, int32_t M, int32_t C, float eps,
                     parq_stream stream) {
    if (!X || !gamma || !beta || !Y || M < 1 || C < 1 || C > 1024) return fail(PARQ_ERR_ARG, "bad argument");
    HIPCHK(launch_layernorm(X, gamma, beta, Y, M, C, eps, (hipStream_t)stream));
    return PARQ_OK;
}

}  // extern "C"
