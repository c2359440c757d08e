/*
 * parq_hip.h — C ABI of the MI355X-native PARQ decoder path (libparq_hip.so).
 *
 * The reference (ymingxie/PARQ) is pure Python/PyTorch: it has no plugin, operator or
 * FFI interface for this path.  Its boundary is the Python call
 *     PARQDecoder.forward(intput_tokens, camera, T_camera_pseudoCam,
 *                         T_world_pseudoCam, T_world_local)      model/parq_decoder.py:134-163
 * and, once per forward before it,
 *     AddRayPE.forward(images_feat, camera, T_cp, T_wp, T_wl)   model/ray_positional_encoding.py:61-139
 * The entry points below are what a binding for those two calls needs: plain pointers,
 * sizes and a hipStream_t.  No torch types, no exceptions, no ownership of caller memory.
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to float32 unless stated otherwise;
 *   - every call only ENQUEUES work on `stream` (a hipStream_t) and returns; nothing
 *     synchronises, allocates or frees device memory;
 *   - return value: PARQ_OK or an error code; parq_last_error() gives the message
 *     (thread-local);
 *   - tokens are channels-last (B, V*h*w, C), token index (v*h + y)*w + x
 *     (model/parq_lightning.py:78-85); Pose = 12 floats [R row-major | t]
 *     (utils/wrappers.py:194-213); Camera = 6 floats [w,h,fx,fy,cx,cy] (utils/wrappers.py:441-476).
 */
#ifndef PARQ_HIP_H
#define PARQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct parq_ctx *parq_handle;
typedef void *parq_stream; /* hipStream_t */

enum {
    PARQ_OK = 0,
    PARQ_ERR_ARG = 1,         /* bad argument / unsupported shape */
    PARQ_ERR_HIP = 2,         /* a HIP runtime call failed */
    PARQ_ERR_STATE = 3,       /* call order violated (e.g. forward before pack_weights) */
    PARQ_ERR_WORKSPACE = 4    /* workspace too small */
};

/* Fields PARQDecoder.__init__ reads from cfg (model/parq_decoder.py:40-82). */
typedef struct parq_config {
    int32_t dim;            /* DIM_IN = TRANSFORMER.DEC_DIM = QUERIES_DIM (C)            */
    int32_t num_queries;    /* NUM_QUERIES (Q)                                           */
    int32_t num_classes;    /* NUM_SEMCLS + 1 (background last)                          */
    int32_t num_heads;      /* TRANSFORMER.DEC_HEADS; head dim C/H must be 32, 64, 128 or 256 */
    int32_t ffn_dim;        /* TRANSFORMER.DEC_FFN_DIM                                   */
    int32_t num_layers;     /* TRANSFORMER.DEC_LAYERS = recurrent iterations (I)         */
    int32_t share_weights;  /* TRANSFORMER.SHARE_WEIGHTS                                 */
    int32_t num_mean_sizes; /* rows of the mean-size table (BoxProcessor: 10)            */
    float scale[6];         /* TRANSFORMER.SCALE = [x0,x1,y0,y1,z0,z1]                   */
} parq_config;

/* Inputs of one PARQDecoder.forward call (model/parq_decoder.py:134). */
typedef struct parq_scene {
    int32_t B, V, h, w;              /* scenes, views, feature-map height/width           */
    const float *tokens;             /* (B, V*h*w, C)                                     */
    const float *camera;             /* (B, V, 6)  feature-scale cameras                  */
    const float *T_camera_pseudoCam; /* (B, V, 12)                                        */
    const float *T_world_pseudoCam;  /* (B, V, 12)                                        */
    const float *T_world_local;      /* (B, 1, 12)                                        */
} parq_scene;

/* The six tensors of one iteration's output dict (model/transformer_parq.py:271-279).
 * For parq_forward each pointer addresses I consecutive (B,Q,·) blocks. */
typedef struct parq_outputs {
    float *pred_logits;          /* (B,Q,num_classes) */
    float *center_unnormalized;  /* (B,Q,3) */
    float *size_unnormalized;    /* (B,Q,3) */
    float *ortho6d;              /* (B,Q,6) */
    float *sem_cls_prob;         /* (B,Q,num_classes) */
    float *coord_pos;            /* (B,Q,3) */
} parq_outputs;

const char *parq_last_error(void);
const char *parq_version(void);

/* ---- lifetime -------------------------------------------------------------------- */
int parq_create(const parq_config *cfg, parq_handle *out);
int parq_destroy(parq_handle h);

/* ---- weights ---------------------------------------------------------------------
 * Names are the reference decoder's state_dict keys, e.g.
 *   "refpoint.weight", "mlp_heads.center_head.layers.0.weight",
 *   "parq_module.decoder.layers.0.multihead_attn.in_proj_weight",
 *   "parq_module.decoder.position_encoder.2.bias",
 * plus "mean_sizes" = the (num_mean_sizes,3) float32 table BoxProcessor gathers from
 * (utils/parq_utils.py:88,96-98).  parq_set_weight only records the pointer;
 * parq_pack_weights copies every recorded tensor into the caller-provided arena
 * (fusing the four first-layer head matrices into one), after which the original
 * tensors are no longer referenced.  Call it again whenever the parameters change. */
int parq_set_weight(parq_handle h, const char *name, const float *dev, int64_t numel);
size_t parq_packed_weights_bytes(parq_handle h);
int parq_pack_weights(parq_handle h, void *arena, size_t arena_bytes, parq_stream stream);

/* Arithmetic of the dense cross-attention (call before sizing the workspace):
 *   0  v_mfma_f32_32x32x2_f32 on fp32 operands (exact fp32 products) — and the only mode in which the per-iteration GEMMs of an
 *      inference forward keep fp32 MFMA products at widths whose chain contracts over 1024 / 768: in modes 1..4 those run as fp16
 *      hi / lo 3-term products on pre-split weights with exact power-of-two scales per row and per output column (no range
 *      condition; see parq_k_linear_half);
 *   1  (default when head dim is 64 or 256) every fp32 operand split as hi+lo fp16 and each product
 *      evaluated as hi*hi + hi*lo + lo*hi on the fp16 matrix pipe with fp32 accumulation:
 *      ~2^-22 relative product error, i.e. fp32-rounding class (measured against float64 the
 *      two modes are indistinguishable).
 *      GUARANTEED OPERAND RANGE of modes 1 and 2: every input token element and every projected K / V
 *      element must satisfy |x| < 60000 (fp16 range with margin).  An element is carried with absolute
 *      error <= max(2^-24, 2^-22 |x|) (both halves round toward zero; below 2^-3 the lo half is an fp16
 *      subnormal with quantum 2^-24): fp32-class as long as the large elements of K / V are >= 2^-3;
 *      tensors that are tiny as a whole lose relative accuracy gradually — measured on the attention
 *      kernel alone, V scaled by 1e-3 gives 2.4e-5 of the output scale instead of 2e-7, while the whole
 *      decoder stays within 1e-4 of the float64 oracle for feature scales 1e-3 ... 1e2 and in-projection
 *      scales 0.1 ... 10 (tests/test_gpu_range.py).  A violation of the upper bound is NOT silent: the kernels that build the
 *      cache raise the int at workspace buffer "flags"[0], and while it is set the last kernel of every
 *      iteration writes NaN into all five computed outputs of parq_iterate / parq_forward instead of
 *      plausible wrong numbers.  Re-run such inputs in mode 0 (the Python class does this: see
 *      parq_amd.PARQDecoder.range_check);
 *   2  single fp16 products, 3  single bf16 products (head dim 64 with dim 128 / 256, or head dim 256 with dim a multiple
 *      of 128 — the reference's shipped DEC_DIM 1024 / 4 heads included): the reduced-precision
 *      configurations of the benchmark (BASELINE.json configs 2 and 5; the reference defines no mixed
 *      precision).  Q, K, V and the probabilities are rounded to nearest 16-bit
 *      once, accumulation stays fp32; the K/V cache shrinks to half.  Outputs agree with the fp32 path to
 *      ~1e-3 (fp16) / ~1e-2 (bf16) on unit-scale features (tests state the tolerances); on key counts that are a multiple of 64
 *      the kernel is mode 4's without cross terms (probabilities normalised by the sum of their rounded values);
 *   4  the scores as mode 1's hi.hi fp16 product plus the two CROSS terms (hi.lo, lo.hi) as MX-scaled fp8 (e4m3) products — one
 *      v_mfma_scale_f32_32x32x64_f8f6f4 per cross term and 64-long contraction instead of four fp16 instructions (the cross terms
 *      are 2^-11 of a product, e4m3 rounds them at 2^-4: ~2^-15 per score) — and P V as ONE fp16 product of probabilities and values
 *      rounded to nearest, normalised by the sum of those same rounded probabilities (flash_split8.hip).  4e-6 .. 1.3e-5 at the
 *      decoder outputs on the reference's fixtures (tests/emulate_attention_arithmetic.py, tests/test_gpu_split8.py).  Head dim 64,
 *      dim 256, key counts that are a multiple of 64 — inference, and training steps whose batched backward reads the forward's cache
 *      (it then reads the stage cache: K = hi16 + e4m3 lo, V = the fp16 value); every other case of a handle in this mode runs as
 *      mode 1.  (The Python class trains in mode 1 unless asked: PARQDecoder.train_split8.)  Range: as mode 1; |K|, |q| past 448 saturate in their fp8 forms only (those elements keep the
 *      accuracy of mode 2, nothing is poisoned).  The mode's error model assumes rows that spread over many keys (within 2.4e-5
 *      of mode 1 at the outputs while every row's probability sum, relative to its maximum, is above 256; 1e-4 and more for rows that
 *      two or three keys carry): the merge kernel raises bit h of workspace "flags"[1] (and of "flags"[8 + iteration]; "flags"[2] keeps the smallest
 *      row sum seen as 0x7fffffff - its float bits) — and bits 1 and 8 + h of the range mirror — when a row of head h has a sum
 *      under 256; what happens then is the caller's policy, see parq_set_head_tiers. */
int parq_set_attention_mode(parq_handle h, int32_t mode);
/* Per-head tiers of attention mode 4 (inference; head dim 64, dim 256, at most 16 heads).  Bit h of `safe_mask` moves head h to the
 * fp16 x 3 arithmetic of mode 1 INSIDE a mode-4 forward: the K/V projection writes that head's cache region in the split layout, the
 * cross-attention of an iteration runs as two launches over complementary head sets (flash_split8_kernel over the others,
 * flash_split_pipe_kernel over these; each with the key-split count that fills the chip with its heads), merged per set.  All heads
 * safe = mode 1 exactly; a training forward with any safe head runs as mode 1.  `poison_on_peaked` != 0: an iteration in which a
 * mode-4 head meets a too-peaked row (see mode 4 above) writes NaN outputs from that iteration on, like a range violation — the
 * forward then never returns plausible numbers outside the mode's error model; the mirror's bits 8 + h say which heads to move.
 * 0: outputs stay numbers (reduced accuracy on those rows), flags and mirror are raised all the same.
 * The reference has one arithmetic (fp32, model/transformer_parq.py:377-380); this call only chooses how its result is approximated. */
int parq_set_head_tiers(parq_handle h, uint32_t safe_mask, int32_t poison_on_peaked);
/* In-launch hand-offs of the small-op chain (default OFF since round 6: under graph replay the fused launch no longer pays for the
 * spin-wait it contains, profiles/r06_ab_seam_q_under_graph_replay.txt).  With `on` != 0, at d = 256 the self-attention out-projection and the cross-attention
 * query projection behind norm1 (model/transformer_parq.py:375-377) run as ONE launch whose query tiles wait, in their epilogue, for the
 * LayerNorm row sums the other tiles publish (chain.hip seam_tile).  The wait is bounded: if a producer workgroup has not published
 * within ~0.1 s — which needs a dispatch order no HIP implementation has shown, the waiting tiles are placed behind the tiles they wait
 * for — the launch gives up, the forward's outputs are NaN, workspace "flags"[0] gets bit 2 and the range mirror bit 2.  `on` = 0 runs
 * every dependent stage as its own launch (placement-independent; about 1 % slower at BASELINE cfg 3); the Python class does this by
 * itself after such a timeout.  Results of the two forms differ by fp32 rounding (the LayerNorm is pushed through the projection). */
int parq_set_seam_fusion(parq_handle h, int32_t on);
/* Optional: a host-visible, device-writable int32 (pinned host memory, e.g. hipHostMalloc) in which the device sets bit 0 whenever
 * it poisons outputs because of a range violation (above), bit 2 when an in-launch hand-off timed out (parq_set_seam_fusion), and bit 1 plus bit 8 + h when head h of attention mode 4 met a
 * too-peaked row (above; outputs poisoned only if parq_set_head_tiers asked for it).  Lets a host poll for events of earlier, already finished calls with a plain load — no stream synchronisation,
 * nothing extra on the forward path.  NULL switches it off. */
int parq_set_range_mirror(parq_handle h, int32_t *host_visible_flag);
/* Host side of that word: returns its value and clears it in ONE atomic exchange, so that a bit the device raises between a plain
 * load and a plain store of the host is never lost (several forwards in flight may share a word).  The pointer is read at ENQUEUE
 * time (and recorded by parq_forward_capture): a caller with several forwards in flight may give every workspace its own word by
 * calling parq_set_range_mirror before each forward. */
int32_t parq_mirror_take(int32_t *host_visible_flag);
/* Early completion signal of parq_forward / parq_forward_replay.  A caller that waits for a forward only to learn whether it has to be
 * re-run (parq_amd.PARQDecoder's default policy) does not need the forward's end: no kernel behind the LAST iteration's cross-attention
 * merge can raise a flag (range violations come from the K/V projection, hand-off timeouts from the launch in front of the
 * cross-attention, too-peaked rows from the merge).  The first launch behind that merge therefore ORs what has been raised into the
 * range mirror word and then stores `epoch` (the value given here before the forward was enqueued; the call's prologue launch carries it
 * to the device) into `host_visible_word` with release semantics: a host that spins on the word sees the mirror bits once it sees the
 * epoch, one chain tail (~36 us at BASELINE cfg 3) before the outputs are complete.  The outputs themselves are stream-ordered as always.
 * The word's address is recorded by parq_forward_capture (part of what parq_forward_replay checks); the epoch is not.  NULL: off. */
int parq_set_progress(parq_handle h, int32_t *host_visible_word, int32_t epoch);

/* ---- PARQDecoder.forward ---------------------------------------------------------- */
size_t parq_workspace_bytes(parq_handle h, int32_t B, int32_t V, int32_t hh, int32_t ww);

/* Whole forward: prologue + I iterations, reference points chained on device
 * (model/transformer_parq.py:283-337).  No host synchronisation inside. */
int parq_forward(parq_handle h, const parq_scene *scene, void *workspace, size_t workspace_bytes,
                 const parq_outputs *outs, parq_stream stream);

/* A captured forward.  The reference's loop stalls the host every iteration
 * (model/transformer_parq.py:135,301; utils/parq_utils.py:96-98); parq_forward only enqueues, but its ~90 launches still cost the
 * host ~0.35 ms per call.  parq_forward_capture records the I ITERATIONS of a forward of shape (B, V, hh, ww) in `workspace` — the
 * same kernels with the same arguments as parq_forward enqueues, hence bit-identical results — into a HIP graph, WITHOUT running
 * anything.  parq_forward_replay is then parq_forward for a call of that shape: it launches the prologue and the K/V projection
 * directly, with THIS call's pointers, and replays the graph behind them (3 host calls instead of ~90; the graph launch overlaps the
 * K/V projection on the device).  The recorded iterations hold no pointer of any particular call: the prologue launch leaves the
 * call's token / output pointers and a copy of its cameras in the workspace, and the kernels that need them load them from there — so
 * one graph serves every call of its shape, whatever tensors the caller passes.  What the graph DOES hold: the workspace, the packed
 * weight arena, the range mirror word, and the handle's attention settings (mode, head tiers, seam fusion) at capture time;
 * parq_forward_replay checks all of them and returns PARQ_ERR_STATE on a mismatch (capture again; parq_amd.PARQDecoder keys its graphs
 * the same way and captures on the second forward of a kind).
 * `stream` of parq_forward_capture: the stream of the first replay; the derived inference weights are built on it first if they are
 * not yet (they must not become part of the graph).  The capture itself runs on a stream of the handle's own in relaxed mode (the
 * caller's may be the legacy default stream) and launches nothing.  Not while parq_profile_enable is on.
 * parq_graph_destroy: only after every replay of the graph has completed. */
typedef struct parq_graph *parq_graph_t;
int parq_forward_capture(parq_handle h, int32_t B, int32_t V, int32_t hh, int32_t ww, void *workspace, size_t workspace_bytes,
                         parq_stream stream, parq_graph_t *out);
int parq_forward_replay(parq_handle h, parq_graph_t g, const parq_scene *scene, void *workspace, size_t workspace_bytes,
                        const parq_outputs *outs, parq_stream stream);
int64_t parq_graph_nodes(parq_graph_t g);
int parq_graph_destroy(parq_graph_t g);

/* Stepping interface (teacher-forced parity tests, custom drivers):
 *   parq_prepare  : T_camera_local and the hoisted K/V cache (transformer_parq.py:298-305)
 *   parq_iterate  : one loop body (:310-335).  ref_in (B,Q,3) normalised reference points,
 *                   or NULL to continue from the previous iteration (or from
 *                   sigmoid(refpoint.weight) after parq_prepare).  ref_out (B,Q,3) may be NULL. */
int parq_prepare(parq_handle h, const parq_scene *scene, void *workspace, size_t workspace_bytes,
                 parq_stream stream);
int parq_iterate(parq_handle h, const parq_scene *scene, void *workspace, size_t workspace_bytes,
                 int32_t layer_num, const float *ref_in, const parq_outputs *outs, float *ref_out,
                 parq_stream stream);

/* Intra-scene view sharding ("split-N"; no reference analogue — the reference holds all views of a scene in one
 * process, model/transformer_parq.py:129-161, 377-380).  A scene whose K/V stream is too large or too slow for one GPU (BASELINE
 * cfg 5: 20 views of 240x320 features = 1.5 M keys) is split by VIEWS: every rank calls parq_prepare with ITS views (tokens,
 * camera, poses of those views only: the K/V projection and cache are sharded) and runs each iteration in three phases around
 * two small exchanges the caller performs with its own collective library (RCCL all-reduce / all-gather over xGMI):
 *   phase 0  position MLP, project + sample of the local views      -> xchg_out: [B*Q*C undivided sums | B*Q valid-view counts |
 *            1 fp16-range flag of this rank's K/V shard]; caller: SUM all-reduce of that buffer over the ranks.  A rank whose
 *            shard left the fp16 operand range thereby poisons the iteration on EVERY rank (phase 1 raises the local device flag
 *            from the reduced value, phase 2's decode writes NaN outputs and raises the host mirror): all ranks return NaN and
 *            take the same fallback decision, none returns unflagged numbers merged from another rank's inf / NaN record
 *   phase 1  xchg_in = the reduced buffer -> view mean; self-attention block; cross-attention over the local keys
 *                                                                   -> xchg_out: [B*Q*C outputs | B*H*pad32(Q) log2 log-sum-exp]
 *            caller: all-gather of that buffer (rank-major)
 *   phase 2  xchg_in = the nranks gathered records -> merged attention; out-proj, FFN, heads, decode: outs, ref_out
 * Everything outside the two sharded stages is computed redundantly and identically on every rank.
 * parq_shard_exchange_floats(h, B, which): floats of the phase-0 (which = 0) / phase-1 (which = 1) record.
 * ref_in / outs / ref_out as in parq_iterate (pass the same ref_in to all three phases). */
size_t parq_shard_exchange_floats(parq_handle h, int32_t B, int32_t which);
int parq_iterate_sharded(parq_handle h, const parq_scene *scene, void *workspace, size_t workspace_bytes, int32_t layer_num,
                         int32_t phase, const float *ref_in, const parq_outputs *outs, float *ref_out, const float *xchg_in,
                         float *xchg_out, int32_t nranks, parq_stream stream);

/* Introspection for parity tests: where a named intermediate of the last parq_iterate lives
 * inside the workspace (offset and element count in floats).  Names: "T_camera_local_f64" and
 * "gn_sums_f64" (float64 payloads), "kv_cache", "ref", "ref_next", "posemb" (after an iteration: pos2posemb3d of the NEXT
 * reference points, written by the decode kernel), "pos_hidden" (relu of the position MLP's first layer for THIS iteration's
 * reference points), "pos_feat" (the position MLP's output; only written where its last layer is not folded into the two
 * in-projections that consume it, i.e. by the training forward), "tgt",
 * "self_qkv", "attn", "xa_prenorm1", "cross_q", "xb_prenorm2", "ffn_hidden", "xc_prenorm3",
 * "heads1", "heads2", "ln1_stats", "ln2_stats", "flags" (int32 words: [0] fp16 operand range exceeded, [1] attention mode 4 met a
 * row carried by too few keys). */
int parq_workspace_lookup(parq_handle h, int32_t B, int32_t V, int32_t hh, int32_t ww, const char *name,
                          size_t *offset_floats, size_t *numel);

/* Optional instrumentation: when enabled, parq_forward brackets each kernel group with
 * hipEvents on `stream`; parq_profile_read synchronises those events and returns the
 * accumulated milliseconds and launch count of group `which` (see PARQ_PROF_*). */
enum {
    PARQ_PROF_KV_PROJ = 0, PARQ_PROF_PROJECT_SAMPLE = 1, PARQ_PROF_CROSS_ATTN = 2,
    PARQ_PROF_SELF_ATTN = 3, PARQ_PROF_LINEAR = 4, PARQ_PROF_OTHER = 5, PARQ_PROF_MERGE = 6, PARQ_PROF_COUNT = 7
};
int parq_profile_enable(parq_handle h, int32_t on);
int parq_profile_read(parq_handle h, int32_t which, double *total_ms, int64_t *launches);

/* ---- training: forward with saved activations + backward (model/parq_lightning.py:97-100) ----------
 * The backward of the whole decoder chain as HIP kernels.  Every attention mode (in the cache modes 1 - 3 the forward streams the
 * 16-bit cache and the backward gets fp32 K / V rebuilt from it: hi + lo in mode 1, the rounded values themselves in the fp16 /
 * bf16 modes 2 / 3, i.e. the gradient is taken straight through the operand rounding of the reduced-precision forward).  Head dims 32 / 64 have register-resident attention backward kernels; head dim 256
 * (the reference's shipped size) with shared layer weights composes the backward of all iterations from split-precision GEMMs (the training
 * workspace then holds two (N x 512-padded I*Q) fp32 score matrices: 3.1 GB at 10 views of 120x160); any other multiple of 16 runs a
 * materialised fp32 path (functional, ~10x slower).  parq_forward_train = parq_forward that keeps every iteration's activations in the (larger)
 * training workspace; parq_backward consumes them: `grads` holds d loss / d output per iteration (same (I,B,Q,k) layout as
 * the outputs, NULL = zero), `grad_arena` receives d loss / d weight in the layout of the packed weight arena
 * (parq_arena_lookup maps reference tensor names to offsets), `d_tokens` (B,N,C) or NULL receives d loss / d input tokens.
 * Reference points are detached between iterations as in the reference (transformer_parq.py:331-332), which makes the iterations
 * independent given the stash: with shared layer weights (one K / V per scene) parq_backward runs the cross-attention backward of
 * ALL iterations as one launch (dK / dV written once) and the K/V-projection weight gradient on the fp16 matrix pipe (hi/lo split);
 * the training workspace then also holds per-iteration dO / dQ and packed query tiles (parq_train_workspace_bytes accounts for it).
 * parq_set_backward_batched(h, 0) (before sizing the training workspace) selects the per-iteration launches instead: same
 * gradients up to summation order, kept as the cross-check of the one-launch form.
 * The library reads NO environment variables: development A/B switches and the kernels-with-ingredients-removed probes exist
 * only in the separate development build (-DPARQ_DEV_PROBES, libparq_hip_dev.so, used by tools/). */
typedef struct parq_output_grads {
    const float *pred_logits, *center_unnormalized, *size_unnormalized, *ortho6d;
} parq_output_grads;
/* Dropout of the decoder layer in training (transformer_parq.py:339-386: attention-probability dropout of both attentions,
 * dropout1/2/3 on the sub-layer outputs, dropout on the FFN hidden layer; all with probability p): counter-based masks from
 * `seed` (draw a new one per step), regenerated identically by parq_backward.  p = 0 (default) switches it off.
 * parq_k_dropout_mask writes the keep-mask scaled by 1/(1-p) of (iteration, site) over a rows x cols index space — sites:
 * 0 self-attention probabilities (rows = B*H*Q, cols = Q), 1 self out-proj (B*Q x C), 2 cross-attention probabilities
 * (B*H*Q x N), 3 cross out-proj, 4 FFN hidden (B*Q x F), 5 FFN output — for tests that feed the same masks to an oracle. */
int parq_set_dropout(parq_handle h, float p, uint32_t seed);
int parq_set_backward_batched(parq_handle h, int32_t on);
int parq_k_dropout_mask(parq_handle h, int32_t iteration, int32_t site, int64_t rows, int64_t cols, float *out,
                        parq_stream stream);
size_t parq_train_workspace_bytes(parq_handle h, int32_t B, int32_t V, int32_t hh, int32_t ww);
size_t parq_grad_arena_bytes(parq_handle h);
int parq_forward_train(parq_handle h, const parq_scene *scene, void *workspace, size_t workspace_bytes,
                       const parq_outputs *outs, parq_stream stream);
/* Blocks the calling host thread until iteration k of the last enqueued parq_forward_train has written its outputs (an event
 * recorded on the stream after every iteration).  The reference evaluates its set loss per iteration with a host-side Hungarian
 * matcher (model/matcher.py, scipy): a caller can match iteration k while the device runs iterations k+1 .. I-1 (reading the
 * outputs of iteration k on a second stream), instead of idling the device for the whole matcher after the last iteration. */
int parq_wait_iteration(parq_handle h, int32_t k);
int parq_backward(parq_handle h, const parq_scene *scene, void *workspace, size_t workspace_bytes, const parq_outputs *outs,
                  const parq_output_grads *grads, float *grad_arena, float *d_tokens, parq_stream stream);
int parq_arena_lookup(parq_handle h, const char *name, int64_t *offset, int64_t *rows, int64_t *cols, int64_t *ld);
/* Data-parallel gradient buckets (the reference trains under DDP, train.py:103-108: bucketed all-reduces overlapped with the
 * backward).  The gradient arena is laid out in the order its parts become final inside parq_backward:
 *   bucket 0 = floats [offset, offset + count) final after phase 1 of the batched backward — cross out-proj, FFN, norm2, norm3
 *              and every head, i.e. everything above the cross-attention — while the cross-attention backward of all iterations
 *              (half of the step) has not started yet;
 *   bucket 1 = the front of the arena (reference points, position MLP, both in-projections, self out-proj, norm1), final at the end.
 * With unshared layer weights bucket 0 is empty (count 0) and bucket 1 is the whole arena.
 * parq_backward_wait_bucket makes `stream` wait (hipStreamWaitEvent, no host synchronisation) until the last enqueued
 * parq_backward has finished writing that bucket: the caller then starts its collective (RCCL all-reduce) on `stream` beside the
 * rest of the backward. */
int parq_grad_bucket(parq_handle h, int32_t bucket, int64_t *offset, int64_t *count);
int parq_backward_wait_bucket(parq_handle h, int32_t bucket, parq_stream stream);
/* Iterations of the chain backward in flight at once (default 8: up to eight streams, weight gradients met in the arena through
 * float atomics — the last bits of the gradients then vary from run to run).  1 = in turn on the caller's stream with plain
 * accumulation (reproducible chain gradients).  Changes the training workspace size: call before parq_train_workspace_bytes. */
int parq_set_backward_streams(parq_handle h, int32_t n);

/* ---- AddRayPE.forward + tokenisation (model/ray_positional_encoding.py:61-139,
 *      model/parq_lightning.py:72-85), once per forward -------------------------------------
 * tokens_out (B, V*h*w, C) channels-last = ray-point positional encoding of every pixel
 *   encoder.2(relu(encoder.0(p))), p = inverse_sigmoid(box-normalised 3-D samples along the pixel ray)
 * plus, when features_nchw (B, V, C, h, w) is given, the feature maps (the `+` and the einops
 * rearrange of parq_lightning.py:75-85).  w1 (C, 3*num_samples), b1 (C), w2 (C, C), b2 (C) are
 * AddRayPE.encoder.{0,2}.{weight,bias}; scale6_host = RAY_POINTS_SCALE (host pointer).
 * flags: PARQ_RAYPE_NCHW_OUT writes the result as (B, V, C, h, w) instead (what AddRayPE.forward returns); available on the
 * one-pass path (C = 256, 64 samples: ONE persistent kernel generates the operand tile, keeps the 64-token hidden tile in LDS
 * between the two layers and adds the feature maps — the hidden tensor is never written).  PARQ_RAYPE_NO_HIDDEN (inference):
 * the fp32 hidden layer that parq_ray_pe_backward reads from the forward's workspace is not kept either, and the workspace is
 * smaller by B*V*h*w*C floats (parq_ray_pe_workspace_bytes_flags; ignored off the one-pass path, which needs the tensor as an
 * intermediate).  Requires (3*num_samples) % 64 == 0 and C % 64 == 0 (shipped: 64 samples, C = 1024 or 256). */
/* PARQ_RAYPE_WEIGHTS_CACHED: the workspace is the one the previous call used (same flags otherwise, same C and num_samples) and
 * w1 / w2 have not changed since: their hi/lo split and fragment-ordered copies inside it are reused (three small launches less). */
enum { PARQ_RAYPE_NCHW_OUT = 1, PARQ_RAYPE_NO_HIDDEN = 2, PARQ_RAYPE_WEIGHTS_CACHED = 4 };
size_t parq_ray_pe_workspace_bytes(int32_t B, int32_t V, int32_t hh, int32_t ww, int32_t C, int32_t num_samples);   /* flags = 0 */
size_t parq_ray_pe_workspace_bytes_flags(int32_t B, int32_t V, int32_t hh, int32_t ww, int32_t C, int32_t num_samples,
                                         int32_t flags);
int parq_ray_pe(const float *camera, const float *T_camera_pseudoCam, const float *T_world_pseudoCam,
                const float *T_world_local, const float *w1, const float *b1, const float *w2, const float *b2,
                const float *scale6_host, float min_depth, float max_depth, int32_t num_samples, int32_t B,
                int32_t V, int32_t hh, int32_t ww, int32_t C, const float *features_nchw, float *tokens_out,
                int32_t flags, void *workspace, size_t workspace_bytes, parq_stream stream);

/* Backward of the encoding + tokenisation (training): given d loss / d tokens (B, V*h*w, C) it returns the gradients of
 * encoder.{0,2}.{weight,bias} and, optionally, d loss / d features (B, V, C, h, w).  `fwd_workspace` is the workspace the
 * forward call left behind (it holds the hidden layer); the geometry has no learnable part. */
size_t parq_ray_pe_backward_workspace_bytes(int32_t B, int32_t V, int32_t hh, int32_t ww, int32_t C, int32_t num_samples);
int parq_ray_pe_backward(const float *camera, const float *T_camera_pseudoCam, const float *T_world_pseudoCam,
                         const float *T_world_local, const float *w2, const float *scale6_host, float min_depth,
                         float max_depth, int32_t num_samples, int32_t B, int32_t V, int32_t hh, int32_t ww, int32_t C,
                         const float *d_tokens, const void *fwd_workspace, void *bwd_workspace, size_t bwd_workspace_bytes,
                         float *dw1, float *db1, float *dw2, float *db2, float *d_features_nchw, parq_stream stream);

/* ---- eval post-processing (model/parq_decoder.py:372-424 parse_pred + utils/nms.py), one workgroup per scene ----------
 * From the LAST iteration's outputs: obbs_out (B,Q,19) = [-s/2, s/2 per axis | R(ortho6d) centre | arg-max class],
 * mask_out (B,Q) bytes = NMS keep mask AND validity window (centre x in (ts[0], ts[1]), z in (ts[4], ts[5]); all valid when
 * for_vis).  NMS on the axis-aligned bounds of the boxes, background class (num_classes - 1) excluded: IoU > 0.1
 * class-agnostic, or IoU > 0.2 within a class when for_vis. */
int parq_parse_pred(const float *center, const float *size, const float *ortho6d, const float *sem_cls_prob, int32_t B,
                    int32_t Q, int32_t num_classes, const float *track_scale6_host, int32_t for_vis, int32_t enable_nms,
                    float *obbs_out, unsigned char *mask_out, parq_stream stream);

/* ---- set loss for a given matching (model/parq_decoder.py:264-370), three launches --------------------------------------------
 * The reference matches predictions to boxes per (iteration, scene) on the host (utils/matcher.py: scipy LSAP + a capped random
 * neighbourhood) and then evaluates ~100 tiny tensor operations per step, forward and in autograd.  Given the matching, this entry
 * writes the four loss terms (centre L1, size L1, rotation MSE minimised over the box's y-symmetry candidates, weighted class
 * cross-entropy; terms[0..3], already divided by valid_bs and scaled by loss_weight) AND d term / d output of the one output
 * tensor each term depends on (same (I,B,Q,k) layout as the outputs; rows without a match get zero).
 *   pairs [4][P] int32: iteration, scene, query, box of every matched pair (a prediction appears at most once per iteration);
 *   pair_coef [P] = 1 / (number of pairs of that (iteration, scene) * valid_bs);
 *   row_weight [I*B*Q] = valid(iteration, scene) * punish_mask / sum_q punish_mask / valid_bs  (parq_decoder.py:341-362);
 *   t_center / t_size (B,nmax,3), t_rot (B,nmax,3,3), t_label / t_sym (B,nmax) int32 (t_sym NULL: no symmetry classes);
 *   the background class is num_classes - 1; class_scratch: I*B*Q int32.  loss_weight4_host is a host pointer. */
int parq_set_loss(const float *pred_logits, const float *center_unnormalized, const float *size_unnormalized, const float *ortho6d,
                  int32_t I, int32_t B, int32_t Q, int32_t num_classes, const float *t_center, const float *t_size, const float *t_rot,
                  const int32_t *t_label, const int32_t *t_sym, int32_t nmax, const int32_t *pairs, const float *pair_coef, int32_t P,
                  const float *row_weight, const float *class_weight, const float *loss_weight4_host, float *terms, float *g_logits,
                  float *g_center, float *g_size, float *g_ortho6d, int32_t *class_scratch, parq_stream stream);

/* ---- single kernels (parity tests, roofline measurements) --------------------------- */

/* K4+K5: project (B,Q,3) normalised reference points into every view and bilinearly
 * sample + view-average the channels-last feature stack (transformer_parq.py:129-161).
 * T_camera_local (B,V,12).  tgt (B,Q,C); coord_pos (B,Q,3) may be NULL. */
int parq_k_project_sample(const float *tokens, const float *T_camera_local, const float *camera,
                          const float *ref, const float *scale6_host, int32_t B, int32_t V, int32_t hh,
                          int32_t ww, int32_t C, int32_t Q, float *tgt, float *coord_pos,
                          parq_stream stream);

/* T_camera_local = T_cp ∘ (inv(T_wp) ∘ T_wl)  (transformer_parq.py:298-300). */
int parq_k_camera_local(const float *T_cp, const float *T_wp, const float *T_wl, int32_t B, int32_t V,
                        float *T_cl, parq_stream stream);

/* Y[M,N] = act(X[M,K] (+ X2[M,K]) @ W[N,K]^T + bias) (+ R[M,N]); fp32 MFMA.  K % 32 == 0. */
int parq_k_linear(const float *X, const float *X2, const float *W, const float *bias, const float *R,
                  float *Y, int32_t M, int32_t N, int32_t K, int32_t relu, parq_stream stream);

/* The same product on the fp16 matrix pipe with fp32-class accuracy (chain.hip chain_linear_h3_kernel: the tile the inference chain uses
 * for contractions over 1024 / 768 — the reference's shipped decoder width, config/train.yaml:37-56 — in every attention mode but 0):
 * operands carried as fp16 hi + lo, a_hi w_hi + a_hi w_lo + a_lo w_hi with fp32 accumulation, rows of X and rows of W scaled by exact
 * powers of two (no range condition).  K in {1024, 768}, M % 16 == 0, N % 16 == 0; `scratch` (>= N * K * 4 + N * 4 bytes, 16-byte aligned)
 * receives the packed weights (what parq_pack_weights' arena holds for the chain).  PARQ_ERR_ARG otherwise. */
int parq_k_linear_half(const float *X, const float *X2, const float *W, const float *bias, const float *R,
                       float *Y, int32_t M, int32_t N, int32_t K, int32_t relu, void *scratch, size_t scratch_bytes,
                       parq_stream stream);

/* softmax(Q K^T / sqrt(dh)) V for (B,H) heads; q (B,Lq,H*dh), k/v (B,Lk,H*dh) row-major;
 * out (B,Lq,H*dh).  scratch must hold parq_k_attention_scratch_bytes(). */
size_t parq_k_attention_scratch_bytes(int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t dh);
int parq_k_attention(const float *q, const float *k, const float *v, float *out, int32_t B, int32_t H,
                     int32_t Lq, int32_t Lk, int32_t dh, void *scratch, size_t scratch_bytes,
                     parq_stream stream);

/* the same attention through the split-fp16 path (head dim 64): converts k/v to the split cache
 * layout inside `scratch`, then runs the split kernel + merge. */
size_t parq_k_attention_split_scratch_bytes(int32_t B, int32_t H, int32_t Lq, int32_t Lk);
int parq_k_attention_split(const float *q, const float *k, const float *v, float *out, int32_t B, int32_t H,
                           int32_t Lq, int32_t Lk, void *scratch, size_t scratch_bytes, parq_stream stream);

/* the same through the mode-4 path (scores with fp8 cross terms, P V in fp16 with a self-consistent normaliser); Lk % 64 == 0; scratch
 * as for parq_k_attention_split */
int parq_k_attention_split8(const float *q, const float *k, const float *v, float *out, int32_t B, int32_t H,
                            int32_t Lq, int32_t Lk, void *scratch, size_t scratch_bytes, parq_stream stream);

/* the split-fp16 path at head dim 256 (the reference's shipped DEC_DIM 1024 / 4 heads, config/train.yaml:49-50): a head is
 * stored as 4 virtual heads of 64 in the split cache, a pair of waves shares each 32-query tile. */
size_t parq_k_attention_split256_scratch_bytes(int32_t B, int32_t H, int32_t Lq, int32_t Lk);
int parq_k_attention_split256(const float *q, const float *k, const float *v, float *out, int32_t B, int32_t H,
                              int32_t Lq, int32_t Lk, void *scratch, size_t scratch_bytes, parq_stream stream);

/* the single-product reduced-precision variants (attention modes 2 / 3): bf16 != 0 selects bf16, else fp16 */
size_t parq_k_attention_half_scratch_bytes(int32_t B, int32_t H, int32_t Lq, int32_t Lk);
int parq_k_attention_half(const float *q, const float *k, const float *v, float *out, int32_t B, int32_t H,
                          int32_t Lq, int32_t Lk, int32_t bf16, void *scratch, size_t scratch_bytes,
                          parq_stream stream);

int parq_k_layernorm(const float *X, const float *gamma, const float *beta, float *Y, int32_t M,
                     int32_t C, float eps, parq_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* PARQ_HIP_H */
