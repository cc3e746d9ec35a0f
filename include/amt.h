/* amt.h — C ABI of the MI355X-native audio-to-MIDI hot path (libamt_hip.so).
 *
 * What this boundary replaces in the reference (kasper0406/audio-to-midi @ 2025-03-28):
 *   - OutputSequenceGenerator.__call__ / .predict            model.py:740-773
 *     (always called under jax.vmap over the batch/window axis: train.py:56-58, infer.py:40)
 *   - compute_loss (value_and_grad of the batched BCE loss)   train.py:39-62
 *   - the fwd/bwd part of compute_training_step               train.py:259-332
 *     (scale, accumulate in fp32, unscale, finite check)
 * The reference has no FFI under the model (it is JAX/XLA); the conventions below copy the only
 * native boundary it has, rust-plugins/src/cbinds.rs:9-91: POD structs, plain pointers + sizes,
 * dtype implied by configuration, paired create/free, no global state, re-entrant per context.
 * Improvement over cbinds.rs: errors are returned as codes + amt_last_error() instead of a panic.
 *
 * Conventions
 *   - All tensor pointers are DEVICE pointers, contiguous, row-major; the CALLER owns them.
 *   - The context owns weights, the RoPE table and the activation workspace.
 *   - Every call is asynchronous on the given HIP stream; nothing synchronises internally
 *     (the workspace is (re)allocated on the first call with a new (B, T): do that call outside
 *     hipGraph capture).
 *   - One context per device per process; a context is not thread-safe.
 *   - Activation dtype (AMT_F32 or AMT_BF16) is fixed at creation.  `audio` is float32 in AMT_F32 contexts and
 *     IEEE float16 in AMT_BF16 contexts: the reference feeds its model float16 samples (grain_loader.py:88;
 *     python.rs:251-261 rounds them through f16 at load time), so the bf16 path adds no input quantisation of its
 *     own (bf16 samples alone moved the frame logits by up to 2e-2: profiles/r04_notes.md).  The same two-byte IEEE
 *     half is the storage type of the forward residual stream inside a bf16 context (csrc/common.h: Res<T>);
 *     GEMM operands and gradients are bf16.  Logits, probs, losses and parameter gradients are always float32; labels are
 *     float32 unless amt_set_label_dtype selected IEEE float16 - what the reference's loader hands over (grain_loader.py:88).
 *   - Return value 0 = success; otherwise amt_last_error(ctx) (or amt_last_error(NULL) for a
 *     failed amt_create) describes the failure.
 */
#ifndef AMT_H_
#define AMT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMT_F32 0
#define AMT_BF16 1
#define AMT_F16 2   /* label dtype only (amt_set_label_dtype) */
#define AMT_MAX_STAGES 8

/* mirrors model_config (model.py:20-34) + MIDI_EVENT_VOCCAB_SIZE (audio_to_midi_dataset.py:26) */
typedef struct amt_config {
  int32_t n_stages;                 /* 7 */
  int32_t dims[AMT_MAX_STAGES];     /* 4,8,16,32,64,128,256 (dims[i] = 4 << i is required) */
  int32_t depths[AMT_MAX_STAGES];   /* 3,3,3,3,3,21,3 */
  int32_t n_layer_pairs;            /* num_transformer_layers = 8 (local + global each) */
  int32_t n_heads;                  /* 4 */
  int32_t head_dim;                 /* attention_size = 64 */
  int32_t kv_compress;              /* compressed_attention_kv_size = 64 */
  int32_t ffn_inter;                /* hidden * transformer_hidden_expansion = 512 */
  int32_t n_events;                 /* 90 */
  int32_t local_window;             /* 16 (model.py:635) */
  float dropout_rate;               /* transformer_dropout_rate = 0.1 */
  int32_t act_dtype;                /* AMT_F32 | AMT_BF16 */
} amt_config;

typedef struct amt_ctx amt_ctx;

/* one leaf (or fused group of leaves) of the flat parameter vector, in device layout */
typedef struct amt_segment {
  char name[96];      /* e.g. "layers.5.3.point_conv_1.weight", "tl.4.wqc" */
  uint64_t offset;    /* element offset into the flat fp32 vector */
  int32_t rows, cols; /* device-layout shape (cols == 1 for vectors) */
  int32_t lr_depth;   /* conv depth label for layer-wise LR decay (train.py:646-672), -1 = default */
} amt_segment;

void amt_default_config(amt_config* cfg);

/* Flat parameter layout (no GPU needed). */
uint64_t amt_param_count(const amt_config* cfg);
int32_t amt_layout_count(const amt_config* cfg);
int32_t amt_layout_segment(const amt_config* cfg, int32_t index, amt_segment* out);

int32_t amt_create(const amt_config* cfg, int32_t device, amt_ctx** out);
void amt_destroy(amt_ctx* ctx);
const char* amt_last_error(const amt_ctx* ctx);

/* Copy `n` fp32 parameters (device layout) into the context and derive the compute copies.
 * `flat` is a host pointer if on_device == 0, else a device pointer. */
int32_t amt_load_weights(amt_ctx* ctx, const float* flat, uint64_t n, int32_t on_device, void* stream);
/* Device pointer to the context's fp32 master parameters (for in-place optimizer updates)... */
float* amt_weights_ptr(amt_ctx* ctx);
/* ...after which the bf16 / transposed compute copies must be re-derived: */
int32_t amt_refresh_weights(amt_ctx* ctx, void* stream);

/* RoPE table (rope.py:12-22): host pointers, [n_pos][head_dim/2] each.  If never called the
 * context builds the table itself for max(512, frames) positions with the reference's float32
 * arithmetic. */
int32_t amt_set_rope(amt_ctx* ctx, const float* cos_table, const float* sin_table, int32_t n_pos);

/* BASELINE configs[4] (the reference has no fp8 path; its attention is model.py:333-386): run S = QK^T and
 * O = PV of the GLOBAL-attention forward on fp8 (e4m3) MFMA with per-tensor delayed scaling (the scales of a
 * call come from the |max| of q, k, v seen in the context's previous forward; the first call assumes 16).
 * bf16 contexts only; softmax, backward and everything else are unchanged.  Off by default: results differ
 * from the bf16 path by fp8 rounding (tests/test_model_gpu.py states the bound). */
int32_t amt_set_attn_fp8(amt_ctx* ctx, int32_t enable);

/* frames produced for T samples (T must be a multiple of 320) */
int32_t amt_num_frames(const amt_config* cfg, int32_t T);

/* Inference: model.predict under vmap (model.py:771-773, infer.py:40).
 * audio [B][2][T] (float32 | float16, see Conventions) -> logits, probs [B][S][n_events] float32 (either may be NULL). */
int32_t amt_forward(amt_ctx* ctx, const void* audio, int32_t B, int32_t T, float* logits, float* probs,
                    void* stream);

/* Train-mode forward only: OutputSequenceGenerator.__call__(..., enable_dropout=True) under vmap (model.py:740-769, the
 * call compute_loss makes at train.py:56-58): the forward pass of the training step (attention-weight and FFN-output
 * dropout drawn from `seed` + the context's seed word, the masks amt_forward_backward would draw for the same seed),
 * logits / probs [B][S][n_events] float32 (either may be NULL).  Uses the training workspace plan. */
int32_t amt_forward_train(amt_ctx* ctx, const void* audio, int32_t B, int32_t T, uint64_t seed, float* logits,
                          float* probs, void* stream);

/* Training forward + backward: compute_loss + the gradient half of compute_training_step.
 *   labels     [B][S][n_events] float32, or IEEE float16 after amt_set_label_dtype(ctx, AMT_F16)
 *   scale      loss scale (reference grad_scale; 1.0 for bf16)
 *   seed       dropout seed; train != 0 enables attention-weight and FFN-output dropout
 *   loss_out   device float[2]: {loss (mean over batch of the summed BCE), scaled_loss}
 *   grads_flat device float[amt_param_count]: d loss / d params, already divided by `scale`
 *   grads_valid device int32: 1 if every gradient is finite, else 0
 *   logits     optional [B][S][n_events] float32 */
int32_t amt_forward_backward(amt_ctx* ctx, const void* audio, const void* labels, int32_t B, int32_t T,
                             float scale, uint64_t seed, int32_t train, float* loss_out, float* grads_flat,
                             int32_t* grads_valid, float* logits, void* stream);

/* The element type of `labels` in every amt_forward_backward* call that follows: AMT_F32 (default) or AMT_F16 - the
 * reference's batches carry float16 labels (`midi_events.astype(np.float16)`, grain_loader.py:88), so a reference batch goes
 * in as it is and the per-step upload carries half the label bytes.  The BCE converts on load; everything else is unchanged. */
int32_t amt_set_label_dtype(amt_ctx* ctx, int32_t dtype);

/* The same step in two calls, for a data-parallel host that overlaps the gradient all-reduce with the
 * backward pass (the reference leaves that to XLA's GSPMD scheduler, train.py:238-244, 280):
 *   phase 1 = forward, loss, backward of decoder + transformer + final CNN LayerNorm: on return (in
 *             stream order) grads_flat[amt_grad_bucket_split(ctx) ..) is complete, unscaled and checked;
 *   phase 2 = backward of the CNN stages: completes grads_flat[0 .. split).
 * Same arguments in both calls; loss_out / logits are written by phase 1, grads_valid is AND-ed by both.
 * Phase 1 followed by phase 2 computes what amt_forward_backward computes (same kernels; the weight gradients of the
 * first transformer layer run as their own launch at the end of phase 1 instead of riding on a phase-2 launch).
 * Neither phase touches the part of grads_flat that belongs to the other. */
int32_t amt_forward_backward_phase(amt_ctx* ctx, const void* audio, const void* labels, int32_t B, int32_t T,
                                   float scale, uint64_t seed, int32_t train, float* loss_out, float* grads_flat,
                                   int32_t* grads_valid, float* logits, int32_t phase, void* stream);
uint64_t amt_grad_bucket_split(const amt_ctx* ctx);

/* Gradient accumulation over minibatches on the device (train.py:275-310: lax.scan over batch / minibatch chunks, fp32
 * accumulation, grads / (grad_scale * steps), train.py:319): the general form of the two entry points above.
 *   phase        0 = whole step, 1 / 2 = its halves (as amt_forward_backward_phase), 3 = whole step + first-half signal
 *                (fused protocol only, see amt_first_half_wait)
 *   accum_index  0 .. accum_count - 1, in order, same grads_flat / grads_valid / loss_out pointers in every call
 * Call 0 zeroes the context's fp32 accumulator; every call adds its gradients to it (no second buffer, no extra pass);
 * only the last call writes grads_flat = accumulator / (scale * accum_count), loss_out = mean over the minibatches of
 * their mean loss, and AND-s grads_valid.  The reference hands every minibatch of a step the SAME key (split once per
 * step, train.py:274, 283-290): pass the same seed in every call to reproduce that (audio_to_midi_amd.train does); a
 * different seed per call is allowed and draws independent masks.  accum_count == 1 is exactly amt_forward_backward / _phase. */
int32_t amt_forward_backward_accum(amt_ctx* ctx, const void* audio, const void* labels, int32_t B, int32_t T,
                                   float scale, uint64_t seed, int32_t train, float* loss_out, float* grads_flat,
                                   int32_t* grads_valid, float* logits, int32_t phase, int32_t accum_index,
                                   int32_t accum_count, void* stream);

/* Bytes of activation workspace currently held (diagnostics). */
uint64_t amt_workspace_bytes(const amt_ctx* ctx);

/* Workspace and hipGraph capture.  The first call for a new (B, T, inference | training) shape plans and allocates
 * (synchronously: make it outside stream capture); every later call for a shape planned before allocates nothing,
 * whatever other shapes were run in between: the workspace only grows and each plan keeps its descriptors.  A graph
 * captured for one shape therefore stays valid while other, already planned shapes run eagerly.  What DOES invalidate
 * every graph captured from this context is a first-time plan that needs a larger workspace (the old one is freed):
 * amt_plan_generation() changes exactly then.  Record it at capture time and re-capture when it differs. */
uint64_t amt_plan_generation(const amt_ctx* ctx);

/* Dropout seed under hipGraph replay.  `seed` is a by-value kernel argument, so a captured step would replay the masks
 * it was captured with.  Every mask is therefore drawn from seed + *amt_seed_word(ctx): a device uint64 owned by the
 * context (0 after amt_create).  Change it between replays with a copy to that address or with amt_advance_seed, a
 * one-thread kernel (*word += delta) that may itself be captured as the first node of the step: each replay then draws
 * fresh masks with no host involvement.  Backward regenerates the forward masks from the same word, so the word must
 * not change between the forward and the backward pass of one step. */
uint64_t* amt_seed_word(amt_ctx* ctx);
int32_t amt_advance_seed(amt_ctx* ctx, uint64_t delta, void* stream);

/* (loss, all-finite flag) of the step just issued into the control words of a data-parallel gradient bucket
 * [valid_all, pad x 3 | grads | loss, valid_first] (audio_to_midi_amd/train.py: GradBucket; the reference reduces loss and
 * flag with the gradients, train.py:292-322): tail_loss[0] = loss[0], tail_valid[0] = head_valid[0] = 1.0 or 0.0.  Null
 * destinations are skipped.  One launch on `stream`, graph-capturable; all pointers are device pointers. */
int32_t amt_pack_tail(amt_ctx* ctx, const float* loss, const int32_t* grads_valid, float* head_valid, float* tail_loss,
                      float* tail_valid, void* stream);

/* In-launch XCD-team hand-offs (csrc/team.h) spin with a bound.  0 = none has ever given up in this context; non-zero =
 * a launch timed out (its outputs are undefined): treat that step as lost.  The timeout word is sticky on the device
 * (every later team launch gives up at its first poll and its kernels fall through) and, once this call has seen it, on
 * the host: the context launches one kernel per Block from then on.  Host-synchronous read; poll it where the training
 * loop synchronises anyway (audio_to_midi_amd.train.DataParallelStep.result does). */
int32_t amt_team_status(const amt_ctx* ctx);

/* The forward of the transformer stack (model.py:559-670) as ONE XCD-team launch (csrc/tlayer.hip): 1 if this context takes
 * it for training-size calls (the default on an 8-XCD x 32-CU device; AMT_TLAYER=0 or a timed-out team launch: 0 = one
 * launch per operator chain).  amt_tlayer_stamps: diagnostic (AMT_TLAYER_STAMPS=1 when the context was created): copies
 * the per-phase wall-clock stamps of the last such launch, [workgroup][layer][32] 64-bit words (100 MHz), returns the
 * number of words written (0: stamps are off). */
int32_t amt_tlayer_active(const amt_ctx* ctx);
int32_t amt_tlayer_stamps(const amt_ctx* ctx, uint64_t* out, int32_t max_words);

/* ---- next-row: optimizer step of compute_training_step (train.py:324-325, 646-729) ----------
 * optax.adamw(b1, b2, eps, weight_decay) per depth label, then clip_by_global_norm(1.0) on the
 * UPDATES, then apply.  lr_base is the scheduled base learning rate for this step; conv leaves
 * use lr_base * layer_lr_decay^(max_depth - depth).  State (m, v) lives in the context.
 * amt_optimizer_init allocates that state (synchronous; call it before any hipGraph capture —
 * amt_adamw_step calls it on first use otherwise). */
int32_t amt_optimizer_init(amt_ctx* ctx, float layer_lr_decay);
int32_t amt_adamw_step(amt_ctx* ctx, const float* grads_flat, float lr_base, float layer_lr_decay, float b1,
                       float b2, float eps, float weight_decay, float clip_norm, int32_t step, void* stream);
/* The same step gated on DEVICE flags (no host sync, hipGraph-capturable): parameters, m and v are left untouched
 * unless *valid_i32 != 0 (the grads_valid word of amt_forward_backward) and *valid_f32 >= 0.999 (that flag as a float
 * after a mean all-reduce over ranks: 1.0 iff every rank's gradients are finite).  Either pointer may be NULL.
 * This is the skip of train.py:369-377 (the reference rolls back to a copy instead).
 * `step` counts optimizer steps from 1 (optax); 0 and values below -1 are errors in both entry points (the bias
 * corrections 1 - b^step would be zero).  step = -1: the count lives ON THE DEVICE - the context's counter of APPLIED
 * updates + 1; an update behind an open gate advances it (an explicit step sets it), a skipped one does not, as optax's
 * count under the reference's roll-back (train.py:369-377).  amt_optimizer_count reads it (host-synchronous). */
int32_t amt_adamw_step_gated(amt_ctx* ctx, const float* grads_flat, float lr_base, float layer_lr_decay, float b1,
                             float b2, float eps, float weight_decay, float clip_norm, int32_t step,
                             const int32_t* valid_i32, const float* valid_f32, void* stream);

int32_t amt_optimizer_count(amt_ctx* ctx);
/* Resume (the reference restores its optimizer state through its checkpoint manager, train.py:799-831): the first / second
 * moments as device float[amt_param_count] (which = 0 / 1; NULL before amt_optimizer_init) - read them to save, write them
 * to restore - and the count of applied updates, which the bias corrections and the host's schedule are evaluated at
 * (host-synchronous). */
float* amt_optimizer_moment_ptr(amt_ctx* ctx, int32_t which);
int32_t amt_optimizer_set_count(amt_ctx* ctx, int32_t count);

/* ---- the fused single-rank protocol: no gradient vector leaves the context ----------------------------------------
 * amt_forward_backward / _phase / _accum with grads_flat == NULL leave their raw fp32 gradient sums in the context's
 * accumulator and skip the pass that would write grads / (scale * steps) to the caller (train.py:319); grads_valid is
 * then only reset to 1.  amt_adamw_step_gated with grads_flat == NULL consumes them: a norm pass (the global norm of the
 * Adam updates, nothing written) and a commit pass that applies the clipped update behind the gate
 *     (valid_f32 == NULL || *valid_f32 >= 0.999)  &&  every gradient finite  &&  loss finite
 * and ZEROES the accumulator in every case.  valid_i32 is an OUTPUT in this form: 1 iff the update was applied (the
 * grads_valid of train.py:320-322 together with the loss test of train.py:369).  52 instead of 60 bytes of HBM traffic
 * per parameter, two launches instead of three, and no zero fill at the start of the next step: the accumulator is
 * clean between steps by construction (zeroed at creation, by the commit pass, by amt_grad_accumulator_reset, and at the
 * end of every backward pass that does write grads_flat).  A backward pass whose gradients are NOT to be applied (a
 * warm-up run, an aborted step) must be followed by amt_grad_accumulator_reset, otherwise the next one adds to it -
 * which is also how train.py:275-310's accumulation over minibatches looks in this protocol.
 * By default the two passes are ONE launch with a grid barrier between them (one workgroup per compute unit; the new moments
 * stay in registers across it): it needs the device to itself for those ~100 us - issue it in stream order behind the
 * step's collectives.  The barrier's verdict is one word per launch that every workgroup reads (all-or-nothing).  If it does
 * not open within ~15 ms (another process or stream holds compute units) the step is STILL APPLIED, with the same bits as
 * the two-launch form: the workgroups leave without touching anything and the one whose arrival completes the count commits
 * the whole update alone (a few ms, once) - a rank of a data-parallel job therefore never parts from the others because of a
 * time-out.  A sticky device word and a mapped host word are set, and the next optimizer call of the context (or
 * amt_optimizer_status / amt_optimizer_count) switches it to the two launches for good and records the reason in
 * amt_last_error.  Under stream capture the two launches are taken (a replayed launch would have no host between replays
 * to switch forms).  AMT_OPT_PERSIST=0 selects the two launches from the start. */
int32_t amt_grad_accumulator_reset(amt_ctx* ctx, void* stream);
/* Number of barrier time-outs of the one-launch form this context has seen (0 in normal operation; no step is lost to
 * one, see above; reads a mapped host word: no synchronisation, but only steps whose launch has completed are seen). */
int32_t amt_optimizer_status(amt_ctx* ctx);

/* ---- the same protocol across ranks: the collectives run ON the context's accumulator ---------------------------------
 * The accumulator is bucket-shaped: [head control floats | n gradient sums | tail control floats], contiguous, the
 * gradient range 16-byte aligned.  amt_grad_bucket returns its first float and the three lengths (head = 4, tail = 2:
 * audio_to_midi_amd/train.py: GradBucket uses [0] of the head and [0], [1] of the tail; amt_pack_tail writes them).  A
 * data-parallel step that replaces train.py:238-244, 280, 292-325 is then, per rank and in stream order:
 *     amt_forward_backward*(grads_flat = NULL)          raw fp32 sums stay in the accumulator (no unscale pass, no copy)
 *     amt_pack_tail(.., tail_loss = tail)               the rank's loss next to the gradients
 *     SUM all-reduce over [gradients | tail]            (RCCL; in one piece, or the two ranges of amt_forward_backward_phase)
 *     amt_adamw_step_reduced(grad_mul = 1 / world, loss_sum = tail)
 * The optimizer reads accumulator * grad_mul / (scale * accum_count), tests isfinite(|update|^2) && isfinite(*loss_sum) -
 * a non-finite gradient on ANY rank makes the reduced sum, hence |update|^2, non-finite on EVERY rank: the all-finite flag
 * of train.py:320-322 needs no collective of its own - and leaves the accumulator zeroed.  This is the single-rank fused
 * step plus the collectives: the N > 1 rank runs the program the N = 1 benchmark times.  valid_out: device int32, 1 iff
 * the update was applied (the same on every rank). */
float* amt_grad_bucket(amt_ctx* ctx, uint64_t* head, uint64_t* n, uint64_t* tail);
/* Overlapping the first collective with the CNN backward WITHOUT cutting the step in two host calls.
 * amt_forward_backward_accum(phase = 3, grads_flat = NULL) issues the launches of the one-call step (phase 0) plus - right
 * behind the first CNN Block chain, which carries the last weight-gradient riders of the range [amt_grad_bucket_split, n) -
 * the record sums of that range and a one-thread kernel that writes the rank's loss into tail[0] of the bucket and
 * publishes "first half complete" as a counter in MAPPED HOST memory.  amt_first_half_wait blocks the calling HOST thread
 * (busy poll, `timeout_ms`, 0 = 10 s; error on time-out) until the next unconsumed signal has been published: the host
 * then issues the collective of [split, n) + tail[0] on its communication stream, which needs no device-side dependency
 * on the compute stream.  The step stays ONE hipGraph whose node sequence is the single-rank step's (+8 us).
 * Why the host: measured on one MI355X with no collective running (profiles/r04_notes.md; the one-call step: 3.01 ms): a
 * stream waiting for an event recorded on the busy compute stream costs 0.12 ms per step (the two-call form, phases 1 / 2:
 * 3.15 ms), a polling kernel resident on a second queue 0.14 ms, hipStreamWaitValue32 on signal memory 1.4 ms; external
 * event-record nodes are refused in a capture by torch 2.10 on ROCm and a captured RCCL call aborts its watchdog.
 * Every phase-3 step must be paired with exactly one amt_first_half_wait, issued after it in host order;
 * amt_first_half_sync (host-synchronous on the step's stream) declares every signal published so far as consumed: call it
 * after warm-up / rehearsal steps that were not paired with a wait.  amt_first_half_status: waits that timed out. */
int32_t amt_first_half_wait(amt_ctx* ctx, int32_t timeout_ms);
int32_t amt_first_half_sync(amt_ctx* ctx, void* stream);
int32_t amt_first_half_status(amt_ctx* ctx);
int32_t amt_adamw_step_reduced(amt_ctx* ctx, float lr_base, float layer_lr_decay, float b1, float b2, float eps,
                               float weight_decay, float clip_norm, int32_t step, float grad_mul, const float* loss_sum,
                               int32_t* valid_out, void* stream);

/* ---- measurement aid (bench.py roofline leg): per-kernel-class timing with HIP events recorded
 * on the launch stream around every kernel launch.  Process-global; off by default. */
typedef struct amt_prof_entry {
  char name[64];      /* kernel class, e.g. "gemm_nt<bf16,store,64>" */
  int32_t launches;
  double total_ms;    /* sum of launch durations */
  double flops;       /* algorithmic flops summed over launches (2*M*N*K per GEMM) */
  double bytes;       /* algorithmic HBM bytes summed over launches */
} amt_prof_entry;
void amt_profile_begin(void);
/* synchronises the device, aggregates per class, returns the number of entries written */
int32_t amt_profile_end(amt_prof_entry* out, int32_t max_entries);

#ifdef __cplusplus
}
#endif
#endif /* AMT_H_ */
