/* amt_ops.h — op-level C entry points of libamt_hip.so.
 *
 * These expose the individual HIP kernels behind amt.h so that tests/ can check each of them
 * against the oracle in isolation (same conventions as amt.h: device pointers, row-major,
 * channels-last activations, asynchronous on `stream`, return 0 on success).  `dtype` is the
 * activation dtype (AMT_F32 | AMT_BF16) of every `void*` tensor unless stated otherwise.
 * "Res" marks a tensor of the forward RESIDUAL STREAM (Block / layer inputs and outputs, the residual operand of a
 * GEMM epilogue): float32 in AMT_F32 contexts, IEEE half (float16) in AMT_BF16 contexts - the reference's own activation
 * type (train.py:35-37) in the same two bytes (csrc/common.h: Res<T>).
 * Reference op for each entry point is cited in the kernel sources (audio-to-midi_amd/csrc/).
 */
#ifndef AMT_OPS_H_
#define AMT_OPS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct amt_ctx amt_ctx;   /* include/amt.h */

/* Y = epilogue(X[N,K] . W[M,K]^T)  — eqx.nn.Linear / 1x1 Conv1d (model.py:141-153, 213-222, 300-334)
 * epi: 0 store, 1 store f32, 2 Y(Res) = R(Res) + dropout(.), 3 GELU (Y = u, Y2 = gelu(u)), 4 Y2 = z, Y(Res) = R(Res) + gamma z,
 * 5 GLU, 6 GLU', 7 GELU', 8 store Res */
int32_t amt_op_gemm_nt(int32_t dtype, int32_t epi, const void* X, int32_t ldx, const void* W, int32_t ldw,
                       int32_t N, int32_t M, int32_t K, const float* bias, void* Y, int32_t ldy, void* Y2,
                       int32_t ldy2, const void* R, int32_t ldr, const void* U, int32_t ldu, const float* gamma,
                       uint64_t seed, uint32_t stream_id, float drop_p, void* stream);
/* dW[M,Kp] += G[N,M]^T . A[N,Kp] ; dbias[M] += colsum(G) */
int32_t amt_op_gemm_tn_wgrad(int32_t dtype, const void* G, int32_t ldg, const void* A, int32_t lda, int32_t N,
                             int32_t M, int32_t Kp, float* dW, int32_t ldw, float* dbias, int32_t M_store,
                             void* stream);
/* X: Res (every LayerNorm of the model reads the residual stream); Y, dY, dRes, dX: activation dtype */
int32_t amt_op_layernorm_fwd(int32_t dtype, const void* X, const float* w, const float* b, void* Y, float* mean,
                             float* rstd, int32_t N, int32_t C, void* stream);
int32_t amt_op_layernorm_bwd(int32_t dtype, const void* dY, const void* X, const float* mean, const float* rstd,
                             const float* w, const void* dRes, void* dX, float* dw, float* db, int32_t N, int32_t C,
                             void* stream);
/* audio, Y: Res */
int32_t amt_op_stem_fwd(int32_t dtype, const void* audio, const float* w, const float* b, const float* lnw,
                        const float* lnb, void* Y, int32_t B, int32_t T, void* stream);
int32_t amt_op_stem_bwd(int32_t dtype, const void* audio, const void* dY, const float* w, const float* b,
                        const float* lnw, float* dw, float* db, float* dlnw, float* dlnb, int32_t B, int32_t T,
                        void* stream);
/* the four depthwise-conv entry points: X (the Block's input) is Res; everything else the activation dtype */
int32_t amt_op_dwln_fwd(int32_t dtype, const void* X, const float* wdw, const float* bdw, const float* lnw,
                        const float* lnb, void* Y, float* mean, float* rstd, int32_t B, int32_t L, int32_t C,
                        void* stream);
int32_t amt_op_dwln_bwd(int32_t dtype, const void* dY, const void* X, const float* wdw, const float* bdw,
                        const float* mean, const float* rstd, const float* lnw, void* dD, float* dlnw, float* dlnb,
                        int32_t B, int32_t L, int32_t C, void* stream);
int32_t amt_op_dwconv_bwd(int32_t dtype, const void* dD, const void* X, const float* wdw, const void* dRes, void* dX,
                          float* dwdw, float* dbdw, int32_t B, int32_t L, int32_t C, void* stream);
/* fused tail of the Block backward: dX = dY + conv^T(LN'(dLN)); dlnw, dlnb, dwdw[7][C], dbdw, dgamma += ... */
int32_t amt_op_dwblock_bwd(int32_t dtype, const void* dLN, const void* X, const void* dY, const void* Z,
                           const float* wdw, const float* bdw, const float* mean, const float* rstd, const float* lnw,
                           void* dX, float* dlnw, float* dlnb, float* dwdw, float* dbdw, float* dgamma, int32_t B,
                           int32_t L, int32_t C, void* stream);
int32_t amt_op_scale_bwd(int32_t dtype, const void* dY, const void* Z, const float* gamma, void* dZ, float* dgamma,
                         int32_t N, int32_t C, void* stream);
/* attention core on raw projections; local != 0 selects the windowed kernel */
int32_t amt_op_attn_fwd(int32_t dtype, int32_t local, const void* QC, int32_t ldqc, const void* KV, int32_t ldkv,
                        void* O, int32_t ldo, float* LSE, const float* cos_t, const float* sin_t, int32_t B, int32_t S,
                        int32_t H, uint64_t seed, uint32_t stream_id, float drop_p, void* stream);
/* bf16 global attention forward with fp8 (e4m3) QK^T / PV (amt.h: amt_set_attn_fp8).  amax: device float[8],
 * {previous |q|,|k|,|v| maxima, -, this call's, -}; the call first rolls cur -> prev (zeros = no history). */
int32_t amt_op_attn_global_fwd_fp8(const void* QC, int32_t ldqc, const void* KV, int32_t ldkv, void* O, int32_t ldo,
                                   float* LSE, const float* cos_t, const float* sin_t, int32_t B, int32_t S, int32_t H,
                                   uint64_t seed, uint32_t stream_id, float drop_p, float* amax, void* stream);
int32_t amt_op_attn_bwd(int32_t dtype, int32_t local, const void* QC, int32_t ldqc, const void* KV, int32_t ldkv,
                        const void* O, int32_t ldo, const float* LSE, const float* cos_t, const float* sin_t,
                        int32_t B, int32_t S, int32_t H, uint64_t seed, uint32_t stream_id, float drop_p,
                        const void* dO, int32_t lddo, void* dQC, int32_t lddqc, void* dKV, int32_t lddkv, float* delta,
                        void* stream);
int32_t amt_op_bce_loss(int32_t dtype, const float* logits_pad, int32_t ldl, const float* labels, float* loss_acc,
                        void* dlogits, int32_t ldd, int32_t N, int32_t V, float scale, int32_t B, void* stream);
int32_t amt_op_dropout(int32_t dtype, const void* X, void* Y, uint64_t n, uint64_t seed, uint32_t stream_id,
                       float drop_p, void* stream);
/* Residual-stream checkpoints of the LAST training-mode forward of a context (amt_forward_backward* / amt_forward_train),
 * for per-layer error probes (probes/bf16_layer_probe.py): index 0 .. amt_op_stream_count(ctx) - 1 in forward order - the
 * input x of every CNN Block (channels-last [B * L][C]), the input of the final CNN LayerNorm, the input x of every
 * transformer layer ([B * S][256]; layer 0's is the final norm's output) and the decoder's input.  `name` receives a label
 * ("cnn.<stage>.<block>.x", "cnn.norm.x", "tl.<layer>.x", "head.x").  The tensors are Res (float32 | float16) and live in the
 * context's workspace until its next call.  Returns 0, or 1 for a bad index / no saved forward. */
int32_t amt_op_stream_count(amt_ctx* ctx);
int32_t amt_op_stream_checkpoint(amt_ctx* ctx, int32_t index, char name[64], void** ptr, int64_t* rows, int32_t* cols);
/* host mirrors of the counter-based dropout RNG (so the oracle can rebuild the device's masks) */
uint32_t amt_rng_u32(uint64_t seed, uint32_t stream_id, uint32_t idx);
uint32_t amt_drop_threshold24(float drop_p);

#ifdef __cplusplus
}
#endif
#endif /* AMT_OPS_H_ */
