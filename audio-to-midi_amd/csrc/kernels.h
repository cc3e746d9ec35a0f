// Host-side launcher declarations for every HIP kernel of the hot path.
// All tensors are channels-last / token-major: [rows][features], row-major.
// T (activation dtype) is float (verification mode) or bf16 (production mode).
// "Res" below marks a tensor of the forward RESIDUAL STREAM (the CNN stream: Stem / Downsample / Block outputs; the
// transformer stream: layer inputs x, h = x + attention, layer outputs y, and the final CNN LayerNorm output that starts
// it): stored as Res<T> (common.h) - float in float contexts, IEEE half in bf16 contexts.  Everything else is T.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace amt {

enum DType { DT_F32 = 0, DT_BF16 = 1 };

// ----------------------------------------------------------------------------- GEMM  Y = X . W^T
enum EpiKind {
  EPI_STORE = 0,        // Y(T)   = acc + bias
  EPI_STORE_F32 = 1,    // Y(f32) = acc + bias
  EPI_RESID = 2,        // Y(Res) = R(Res) + dropout(acc + bias)
  EPI_GELU = 3,         // Y(T)   = u = acc + bias ; Y2(T) = gelu(u)
  EPI_SCALE_RESID = 4,  // Y2(T)  = z = acc + bias ; Y(Res) = R(Res) + gamma[m] * z
  EPI_GLU = 5,          // rows of W in (16 x u1 | 16 x u2) blocks: Y2 = u (permuted), Y = gelu(u1) * u2
  EPI_GLU_BWD = 6,      // acc = dg ; Y(T)[.. permuted ..] = (dg*u2*gelu'(u1), dg*gelu(u1)), u read from U
  EPI_GELU_BWD = 7,     // Y(T) = acc * gelu'(U)
  EPI_STORE_RES = 8,    // Y(Res) = acc + bias   (Downsample conv: its output starts a stage's residual stream)
  EPI_COUNT
};

struct GemmArgs {
  const void* X; int ldx;   // [N x K] (T)
  const void* W; int ldw;   // [M x K] (T)
  int N, M, K;
  const float* bias;        // [M] or nullptr
  void* Y; int ldy;
  void* Y2; int ldy2;
  const void* R; int ldr;   // residual (Res)
  const void* U; int ldu;   // saved pre-activation (T)
  const float* gamma;       // [M]
  uint64_t seed; uint32_t stream; uint32_t drop_thresh24; float inv_keep;
  const uint64_t* seed_dev;   // optional device word ADDED to `seed` by the kernel (fresh masks under hipGraph replay)
};
// requirements: M % 4 == 0; (K % 8 == 0 and ldx, ldw % 8 == 0) or (K == 4 and ldx, ldw % 4 == 0)
void gemm_nt(hipStream_t s, DType dt, EpiKind epi, const GemmArgs& a);

// ----------------------------------------------------------------------------- weight-grad GEMM
// dW[m][k] += sum_n G[n][m] * A[n][k]  (fp32 atomics) ; dbias[m] += sum_n G[n][m]
struct WgradArgs {
  const void* G; int ldg;   // [N x M]  (T)
  const void* A; int lda;   // [N x Kp] (T)
  int N, M, Kp;
  float* dW; int ldw;       // [M x Kp] fp32
  float* dbias;             // [M] fp32 or nullptr
  int M_store;              // rows of dW / dbias actually written (0 = M); G may be zero-padded past it
  float* part; int part_stride;   // set by the launcher: per-split records instead of atomics
  const float* row_scale;         // optional [M]: dW[m][:] and dbias[m] are scaled by row_scale[m] (layer scale)
};
struct PartialSink;
void gemm_tn_wgrad(hipStream_t s, DType dt, const WgradArgs& a, PartialSink* sink = nullptr);
// up to 16 independent problems in one launch (8 as riders of a chain kernel)
// xcd = 1: workgroup b of the group takes work item (b % 8) * (total / 8) + b / 8 (wgrad.h), so that the tiles of one token
// split - which read the same two operand panels - run on ONE XCD and the panels cross the fabric once, not up to 8 times;
// xcd = 2: the same within each problem.  AMT_WGRAD_XCD; default 0 for riders (fabric reads fall by 40-55 % and the cfg2 step
// gets slower) and 2 for the stand-alone grouped launches of the large-N schedule (gemm.hip: wgrad_group_prepare).
template <int CAP> struct WgradGroupN { int n; int prefix[CAP + 1]; int tiles[CAP]; int tiles_k[CAP]; int rps[CAP]; int xcd; WgradArgs p[CAP]; };
using WgradGroup = WgradGroupN<8>;        // what a chain kernel carries as riders (a kernel argument of every backward chain)
using WgradGroupWide = WgradGroupN<16>;   // stand-alone grouped launches (gemm_tn_wgrad_group): up to 16 problems
// target_wgs > 0: aim for about that many workgroups (default: AMT_WGRAD_TARGET_WGS / 336)
void gemm_tn_wgrad_group(hipStream_t s, DType dt, const WgradArgs* probs, int n, PartialSink* sink = nullptr, int target_wgs = 0);
// fills the group (one-tile / long-N problems get per-split records from `sink`); returns the number of
// workgroups, 0 while planning.  Used by gemm_tn_wgrad_group and by chain kernels that carry the
// previous scope's weight gradients in extra workgroups of their own launch.
// target_override > 0: aim for about that many workgroups instead of the default (a carrier whose own workgroups
// leave compute units free asks for as many riders as there are free units).
int wgrad_group_prepare(DType dt, const WgradArgs* probs, int n, PartialSink* sink, WgradGroup* g, double* flops = nullptr,
                        double* bytes = nullptr, int target_override = 0);
int wgrad_group_prepare(DType dt, const WgradArgs* probs, int n, PartialSink* sink, WgradGroupWide* g, double* flops = nullptr,
                        double* bytes = nullptr, int target_override = 0);

// ----------------------------------------------------------------------------- per-block partial sums
// Row kernels that produce parameter gradients (LayerNorm w/b, depthwise conv w/b, gamma) can write
// one record of partial column sums per block instead of atomically adding into the gradient;
// reduce_partials sums all records of the step in one launch.
struct PartDesc { uint64_t ws_off; uint64_t dst_off; uint32_t nblocks, stride, count, pad; };
struct PartialSink {
  float* ws = nullptr;            // records workspace (device); may be null while planning
  size_t off = 0;                 // bump pointer (floats)
  const float* grads_base = nullptr;
  std::vector<PartDesc>* descs = nullptr;   // filled while planning
  bool dry = false;               // planning pass: account, do not launch
  size_t take(size_t n) { size_t o = off; off += n; return o; }
  void add(size_t ws_off, uint32_t nblocks, uint32_t stride, uint32_t count, const float* dst) {
    if (descs) descs->push_back(PartDesc{ws_off, (uint64_t)(dst - grads_base), nblocks, stride, count, 0});
  }
};
void reduce_partials(hipStream_t s, const PartDesc* descs_dev, int n_desc, const float* ws, float* grads);

// ----------------------------------------------------------------------------- LayerNorm over the last dim
// C in {4, 8, 16, 32, 64, 128, 256}.  X: Res (every LayerNorm of the model reads the residual stream); Y: T, or Res when
// y_res (the final CNN norm: its output is the transformer's stream).
void layernorm_fwd(hipStream_t s, DType dt, const void* X, const float* w, const float* b,
                   void* Y, float* mean, float* rstd, int N, int C, bool y_res = false);
// dX = (dRes ? dRes : 0) + LN'(dY) ; dw += sum dY*xhat ; db += sum dY.  X: Res (saved forward stream); gradients: T
void layernorm_bwd(hipStream_t s, DType dt, const void* dY, const void* X, const float* mean, const float* rstd,
                   const float* w, const void* dRes, void* dX, float* dw, float* db, int N, int C,
                   PartialSink* sink = nullptr);

// ----------------------------------------------------------------------------- CNN pieces
// audio [B][2][T] (Res: the model's input has the residual stream's storage type) -> Y [B][T/5][4] (Res) = LN(conv k5 s5)
void stem_fwd(hipStream_t s, DType dt, const void* audio, const float* w /*[4][2][5]*/, const float* b,
              const float* lnw, const float* lnb, void* Y, int B, int T);
void stem_bwd(hipStream_t s, DType dt, const void* audio, const void* dY, const float* w, const float* b,
              const float* lnw, float* dw, float* db, float* dlnw, float* dlnb, int B, int T, PartialSink* sink = nullptr);
// whole Block backward for C <= 8 in one launch (cnn.hip): data gradient dX and, as per-workgroup records in
// `sink`, every parameter gradient of the Block (pw1 / pw2 weights + biases with the layer scale folded into pw2's,
// LayerNorm, depthwise taps + bias, gamma).  Pointers d* are the destinations inside the gradient buffer.
struct SmallBwdArgs {
  const void *dY, *X /* Res */, *U, *LN;      // (h = gelu(u) is re-formed from u; no z: see dgamma below)
  const float *mean, *rstd;
  const void *w2tg, *w1t;
  const float *wdw, *bdw, *lnw, *gamma;
  void* dX;
  // dW2 / db2 leave RAW (without the layer scale): cnn_gamma_finish forms dgamma from them and scales them afterwards
  float *dW1, *db1, *dW2, *db2, *dlnw, *dlnb, *dwdw, *dbdw;
  int B, L, C;
};
void block_small_bwd_full(hipStream_t s, DType dt, const SmallBwdArgs& a, PartialSink* sink);
// Y = LN(dwconv7(X) + bdw) ; X (Res), Y (T) [B][L][C]; saves mean/rstd [B*L]
void dwln_fwd(hipStream_t s, DType dt, const void* X, const float* wdw /*[7][C]*/, const float* bdw,
              const float* lnw, const float* lnb, void* Y, float* mean, float* rstd, int B, int L, int C);
// dD = LN'(dY) w.r.t. the conv output (recomputed from X: Res) ; dlnw,dlnb accumulated
void dwln_bwd(hipStream_t s, DType dt, const void* dY, const void* X, const float* wdw, const float* bdw,
              const float* mean, const float* rstd, const float* lnw, void* dD, float* dlnw, float* dlnb,
              int B, int L, int C, PartialSink* sink = nullptr);
// dX = dRes + conv^T(dD) ; dwdw[k][c] += sum dD[l][c] X[l+k-3][c] ; dbdw[c] += sum dD.  X: Res; dRes is a GRADIENT (T)
void dwconv_bwd(hipStream_t s, DType dt, const void* dD, const void* X, const float* wdw, const void* dRes,
                void* dX, float* dwdw, float* dbdw, int B, int L, int C, PartialSink* sink = nullptr);
// LayerNorm of a Block's OUTPUT rows, evaluated by the launch that produces them (the LayerNorm a Downsample or the final
// norm opens with, model.py:107-118, 759-762: row-local, so the tile that holds y can finish it): Y = LN(y), mean, rstd
// exactly as layernorm_fwd writes them.  w == nullptr: off.  y_res: Y is stored as Res (the final norm), else T.
struct TailLn { const float* w; const float* b; void* Y; float* mean; float* rstd; int y_res; };
// Whole Block forward per position for C in {4, 8} (dwconv + LN + pw1 + GELU + pw2 + gamma + residual);
// w1 [2C][C], w2 [C][2C] are the T compute copies.  Writes LN, mean, rstd, U, H = gelu(U), Z (T) and Y (Res); X: Res.
void block_small_fwd(hipStream_t s, DType dt, const void* X, const float* wdw, const float* bdw, const float* lnw,
                     const float* lnb, const void* w1, const float* b1, const void* w2, const float* b2,
                     const float* gamma, void* LN, float* mean, float* rstd, void* U, void* Hh, void* Z, void* Y, int B,
                     int L, int C, const TailLn* tail = nullptr);
// dU = (w2tg . dY) * gelu'(U), dLN = w1t . dU per position (C in {4, 8}); w2tg [2C][C], w1t [C][2C] (T)
void block_small_bwd(hipStream_t s, DType dt, const void* dY, const void* U, const void* w2tg, const void* w1t, void* dU,
                     void* dLN, long N, int C);
// Fused tail of the Block backward (replaces dwln_bwd + dwconv_bwd [+ scale_bwd's dgamma]):
//   dD = LN'(dLN) w.r.t. the depthwise-conv output (recomputed from X), kept in LDS with a 3-row halo;
//   dX = dY + conv^T(dD); dlnw, dlnb, dwdw[7][C], dbdw, and dgamma[c] += sum dY * Z (if Z != null).  X: Res.
void dwblock_bwd(hipStream_t s, DType dt, const void* dLN, const void* X, const void* dY, const void* Z,
                 const float* wdw, const float* bdw, const float* mean, const float* rstd, const float* lnw,
                 void* dX, float* dlnw, float* dlnb, float* dwdw, float* dbdw, float* dgamma, int B, int L, int C,
                 PartialSink* sink = nullptr);
// dZ = gamma * dY ; dgamma[c] += sum dY * Z
void scale_bwd(hipStream_t s, DType dt, const void* dY, const void* Z, const float* gamma, void* dZ,
               float* dgamma, int N, int C, PartialSink* sink = nullptr);

// ----------------------------------------------------------------------------- chain kernels (chain.hip)
// Whole Block forward (C >= 16) in one launch; writes LN, mean, rstd, U, H = gelu(U), Z, Y exactly
// as dwln_fwd + gemm_nt<gelu> + gemm_nt<scale_resid> do.  W1 [2C][C], W2 [C][2C] are fragment-PACKED T copies.
struct BlockFwdArgs {
  const void* X /* Res */; const float* wdw; const float* bdw; const float* lnw; const float* lnb;
  const void* W1; const float* b1; const void* W2; const float* b2; const float* gamma;
  void* LN; float* mean; float* rstd; void* U; void* Hh; void* Z; void* Y /* Res */;
  int B, L, C;
  int dbg_slot;   // micro-benchmark only (AMT_CHAIN_TIMING)
  TailLn tail;
};
void block_fwd_fused(hipStream_t s, DType dt, const BlockFwdArgs& a);

// XCD-team form (team.h, chain.hip): the Blocks of one CNN stage (same C, same L) in ONE launch of one 256-thread
// workgroup per compute unit; clip b is owned by the workgroups of one XCD, which meet at a team barrier between Blocks.
// Same tile body, same saved tensors, same bits as n launches of block_fwd_fused.  Training form only (C >= 16).
struct TeamState;
constexpr int BLOCK_RUN_MAX = 24;
struct BlockRunFwdItem {
  const void *W1, *W2; const float *wdw, *bdw, *lnw, *lnb, *b1, *b2, *gamma;
  void *LN, *U, *Hh, *Z, *Y; float *mean, *rstd;
};
struct BlockRunFwdArgs {
  const void* X0;            // input of the first Block
  TeamState* st;
  int B, L, C, n;
  int dbg_base;              // micro-benchmark only (AMT_CHAIN_TIMING)
  BlockRunFwdItem it[BLOCK_RUN_MAX];
};
// false if this (dtype, C, n, device) is not served by the team form: the caller launches the Blocks one by one
bool block_run_fwd(hipStream_t s, DType dt, const BlockRunFwdArgs& r);
// device words of the team protocol (zeroed); free with hipFree
TeamState* team_state_create();
// 0 = no team launch has timed out since the state was created (host-synchronous read)
int team_state_timeout(const TeamState* st);

// Data-gradient chain of a Block (C >= 16) in one launch: du, dln (LDS only), dD, dx and the per-workgroup
// records of dlnw, dlnb, dwdw[7][C], dbdw, dgamma (summed by reduce_partials; `sink` is mandatory).
// W2tg [2C][C] (pw2^T, columns scaled by gamma) and W1t [C][2C] are fragment-PACKED T copies.
struct BlockBwdArgs {
  const void* dY; const void* X /* Res */; const void* U; const void* Z;   // Z may be null (no dgamma)
  const void* W2tg; const void* W1t;
  const float* wdw; const float* bdw; const float* lnw; const float* mean; const float* rstd;
  void* dU; void* dX;
  float* rec; int rec_stride;      // set by the launcher
  int B, L, C, tiles_per_clip;     // tiles_per_clip set by the launcher
  int n_chain;                     // workgroups of the chain; the rest of the grid runs `riders` (launcher)
  int dbg_slot;
};
// `riders` (may be null / n == 0): weight-gradient problems of the PREVIOUS scope, executed by extra
// workgroups of this launch (they only feed the gradient buffer, so they leave the dependent chain).
void block_bwd_fused(hipStream_t s, DType dt, const BlockBwdArgs& a, float* dlnw, float* dlnb, float* dwdw,
                     float* dbdw, float* dgamma, PartialSink* sink, const WgradArgs* riders = nullptr, int n_riders = 0);

// Second half of a transformer layer forward in one launch (model.py:369, 545-556; H = 256, I = 512):
//   h = x + ao . Wo^T ; hn = LN2(h) ; u = hn . W1^T + b1 ; g = gelu(u1) * u2 ; y = h + dropout(g . W2^T + b2)
// Writes h, hn, mean, rstd, u (permuted GLU layout), g, y exactly as gemm_nt<resid> + layernorm_fwd +
// gemm_nt<glu> + gemm_nt<resid> do.  Wo [256][256], W1 [1024][256], W2 [256][512]: fragment-PACKED.
// W1 == nullptr: stop after the LayerNorm (h, hn, mean, rstd) — the production form; the full chain is
// an opt-in (AMT_FUSE_FFN=1) that measured slower at 8 clips per GPU (chain.hip).
struct FfnFwdArgs {
  const void* AO; const void* X;   // X, Hout, Y: Res
  const void* Wo; const float* ln_w; const float* ln_b;
  const void* W1; const float* b1; const void* W2; const float* b2;
  void* Hout; void* HN; float* mean; float* rstd; void* U; void* G; void* Y;
  int N;
  uint64_t seed; uint32_t stream; uint32_t drop_thresh24; float inv_keep;
  const uint64_t* seed_dev;   // optional device word ADDED to `seed` by the kernel (fresh masks under hipGraph replay)
  int dbg_slot;
};
void ffn_fwd_fused(hipStream_t s, DType dt, const FfnFwdArgs& a);
// Large-N form (chain_big.hip): 128-token tiles, FFN hidden dimension in slices, same results bit for bit.  save = false
// (inference) writes nothing but Y.  false: not served (N < 8192 tokens, attention-output-only form, AMT_BIG=0).
bool ffn_fwd_big(hipStream_t s, DType dt, const FfnFwdArgs& a, bool save);

// First half of a transformer layer forward up to the attention core, in one launch (model.py:539,
// 300-334, 353-357; H = 256, 4 heads x 64, kv_compress 64):
//   xn = LN1(x) ; qc = xn . Wqc^T  ([q (256) | c (64)]) ; kv = c . Wkv^T  ([k (256) | v (256)])
// Writes xn, mean, rstd, qc, kv exactly as layernorm_fwd + gemm_nt + gemm_nt do.  Wqc [320][256],
// Wkv [512][64]: fragment-PACKED.
struct QkvFwdArgs {
  const void* X /* Res */; const float* ln_w; const float* ln_b; const void* Wqc; const void* Wkv;
  void* XN; float* mean; float* rstd; void* QC; void* KV;
  int N;
  int dbg_slot;
  // Optional (global-attention layers): q = QC[:, 0:256] and k = KV[:, 0:256] leave the chain ROTATED (RoPE of the
  // token's frame index row % rope_S, rounded to T exactly as the attention kernels round what they rotate while staging),
  // so that the attention kernels (AttnArgs::prerot) copy instead of rotating every K tile once per query tile and every Q
  // tile once per key tile.  nullptr: raw projections (local layers rotate window-relative positions).
  const float* rope_cos; const float* rope_sin; int rope_S;
};
void qkv_fwd_fused(hipStream_t s, DType dt, const QkvFwdArgs& a);
// ffn_fwd_fused (full) of layer l and qkv_fwd_fused of layer l + 1 in ONE launch (same token tiles, y handed over
// through LDS): same outputs as the two launches.  false: not served for this size, launch the two separately.
bool ffn_qkv_fwd_fused(hipStream_t s, DType dt, const FfnFwdArgs& fa, const QkvFwdArgs& qa);

// Backward of the first half of a transformer layer after the attention core, in one launch
// (replaces gemm_nt (k/v-up^T) + gemm_nt (q/kv-down^T) + layernorm_bwd [+ the next layer's dropout_apply]):
//   dc = dkv . Wkv (64)  -> dQC[:, 256:320] ;  dxn = dQC . Wqc ;  dx = dh + LN1'(dxn) ;  dln1 w/b records
// DXD (optional): dx with the FFN-output dropout mask of the layer that consumes it.
// Wkvt [64][512], Wqct [256][320]: fragment-PACKED.
struct QkvBwdArgs {
  const void* dKV; void* dQC;                 // dQC[:, 0:256] in, [:, 256:320] out
  const void* Wkvt; const void* Wqct;
  const void* X /* Res */; const float* mean; const float* rstd; const float* ln_w; const void* dH;
  void* dX; void* dXD;
  float* rec; int rec_stride;                 // set by the launcher
  int N;
  int n_chain;                                // workgroups of the chain; the rest run the riders (launcher)
  uint64_t seed; uint32_t stream; uint32_t drop_thresh24; float inv_keep;
  const uint64_t* seed_dev;   // optional device word ADDED to `seed` by the kernel (fresh masks under hipGraph replay)
  int dbg_slot;
};
void qkv_bwd_fused(hipStream_t s, DType dt, const QkvBwdArgs& a, float* dlnw, float* dlnb, PartialSink* sink,
                   const WgradArgs* riders = nullptr, int n_riders = 0);   // riders: see block_bwd_fused

// Backward of the second half of a transformer layer below the FFN, in one launch (replaces
// layernorm_bwd + gemm_nt (output-projection^T)):  dh = dy + LN2'(dhn) ; dao = dh . Wo ; dln2 w/b records.
// Wot [256][256]: fragment-PACKED.
struct LnOutBwdArgs {
  const void* dHN; const void* Hres /* Res */; const float* mean; const float* rstd; const float* ln_w; const void* dY;
  const void* Wot;
  void* dH; void* dAO;
  float* rec; int rec_stride;
  int N;
  int n_chain;
  int dbg_slot;
};
void lnout_bwd_fused(hipStream_t s, DType dt, const LnOutBwdArgs& a, float* dlnw, float* dlnb, PartialSink* sink,
                     const WgradArgs* riders = nullptr, int n_riders = 0);

// Backward of the WHOLE second half of a transformer layer in one launch (the backward counterpart of the full
// ffn_fwd_fused; replaces gemm_nt<glu_bwd> + gemm_nt<store> + lnout_bwd_fused):
//   dg = dyd . W2 ; du = GLU'(dg, u) (permuted layout, written for the W1 weight gradient) ; dhn = du . W1 ;
//   dh = dy + LN2'(dhn) ; dao = dh . Wo ; dln2 w/b records.
// dYD = dy with the FFN-output dropout mask applied (== dY without dropout).  W2t [512][256], W1t [256][1024],
// Wot [256][256]: fragment-PACKED.  The boundary a launch ends with costs the write-back of what it stored
// (profiles/r02_notes.md): this chain keeps dhn on the chip and ends two launches earlier.
struct FfnBwdArgs {
  const void* dYD; const void* dY; const void* U;
  const void* W2t; const void* W1t; const void* Wot;
  const void* Hres /* Res */; const float* mean; const float* rstd; const float* ln_w;
  void* dU; void* dH; void* dAO;
  float* rec; int rec_stride;      // set by the launcher
  int N;
  int n_chain;                     // workgroups of the chain; the rest run the riders (launcher)
  int dbg_slot;
};
void ffn_bwd_fused(hipStream_t s, DType dt, const FfnBwdArgs& a, float* dlnw, float* dlnb, PartialSink* sink,
                   const WgradArgs* riders = nullptr, int n_riders = 0);

// ----------------------------------------------------------------------------- XCD-team layer forward (tlayer.hip)
// The forward of the WHOLE transformer stack (model.py:529-556, 559-670) in ONE launch of one 512-thread workgroup per
// compute unit.  Clip b is owned by the 32 workgroups of one XCD (team.h).  Inside a team the work of every per-token
// stage is cut in TWO dimensions - token groups of TG rows x 8 feature groups - so that a workgroup streams 1/8 of a
// stage's weights for TG (128) tokens instead of all of them for 16, and the stage outputs are exchanged through that
// XCD's L2 (plain stores, sc1 loads, one flag word per workgroup).  Phases of a layer, all waits on 8 (phase B: 2 n_tg)
// flag words:  A: LN1 + q / c / k / v projections (+ RoPE)   B: attention core (global: (head, 64-query tile) per
// workgroup; local: (stripe, head) per wave)   C: output projection + residual   D: LN2 + W1 / GLU   E: W2 + dropout +
// residual.  Saved tensors, roundings and K orders are those of qkv_fwd_fused / attn_*_fwd / ffn_fwd_fused: same bits.
constexpr int TLAYER_MAX = 16;
struct TLayerItem {
  const void *Wqc, *Wkv, *Wo, *W1, *W2;                    // fragment-PACKED T copies
  const float *ln1w, *ln1b, *ln2w, *ln2b, *b1, *b2;
  void *xn, *qc, *kv, *ao, *h /* Res */, *hn, *u, *g, *y /* Res: the next layer's x */;
  float *m1, *r1, *m2, *r2, *lse;
  unsigned long long* keepbits;
};
struct TLayerArgs {
  const void* X0;            // input of the first layer (Res)
  TeamState* st;
  unsigned long long* stamps;   // optional [grid][n_layers][32] wall-clock stamps (diagnostic builds of the bench)
  const float *rope_cos, *rope_sin;
  const uint64_t* seed_dev; uint64_t seed;
  int B, S, n_layers;
  int prerot;                // global layers: q / k leave phase A rotated (QkvFwdArgs::rope_cos)
  int padL, nW;              // local-attention geometry (attn_fwd.h: local_geom)
  uint32_t drop_thresh24; float inv_keep;
  TLayerItem it[TLAYER_MAX];
};
// false: not served (dtype / size / device), the caller takes the launch-per-operator path
bool tlayer_fwd(hipStream_t s, DType dt, const TLayerArgs& a);
// the device part of that answer: 8 XCDs x 32 compute units (one workgroup per compute unit, all resident)
bool tlayer_fwd_device_ok();

// ----------------------------------------------------------------------------- attention
struct AttnArgs {
  const void* QC; int ldqc;   // q = QC[:, 0:256] (un-roped)
  const void* KV; int ldkv;   // k = KV[:, 0:256] (un-roped), v = KV[:, 256:512]
  void* O; int ldo;           // [B*S][256]
  float* LSE;                 // [B][H][S] (global only)
  const float* cos; const float* sin;  // [n_pos][32]
  int B, S, H;                // H heads of dim 64
  uint64_t seed; uint32_t stream; uint32_t drop_thresh24; float inv_keep;
  const uint64_t* seed_dev;   // optional device word ADDED to `seed` by the kernel (fresh masks under hipGraph replay)
  // Global attention, dropout on (optional): the forward leaves the keep decisions of its weight dropout here, one bit per
  // score - word [(b H + h) * ntile + kt][q] (q < 64 ntile), bit 16 st + 4 g + e = key 64 kt + 16 st + 4 g + e - and both
  // roles of the backward read them back instead of hashing 16 indices per lane and 64 x 64 tile again (attn.hip).
  unsigned long long* keepbits;
  int prerot;                 // global attention: q and k arrive rotated and rounded (QkvFwdArgs::rope_cos): no RoPE while staging
};
struct AttnBwdArgs {
  AttnArgs f;                 // forward tensors (O = forward output, LSE)
  const void* dO; int lddo;   // [B*S][256]
  void* dQC; int lddqc;       // dq -> dQC[:, 0:256]
  void* dKV; int lddkv;       // dk -> [:, 0:256], dv -> [:, 256:512]
  float* delta;               // [B][H][S] scratch
};
void attn_global_fwd(hipStream_t s, DType dt, const AttnArgs& a);
void attn_global_bwd(hipStream_t s, DType dt, const AttnBwdArgs& a);
// bf16 activations, fp8 (e4m3) QK^T / PV with per-tensor delayed scaling: amax_prev[3] (|q|, |k|, |v| maxima of the
// previous call of this layer; 0 = none yet) gives the scales, amax_cur[3] (zeroed by the caller) receives this call's
void attn_global_fwd_fp8(hipStream_t s, const AttnArgs& a, const float* amax_prev, float* amax_cur);
// start of a forward pass: per layer, prev <- cur, cur <- 0 (amax laid out [layers][prev[4] | cur[4]])
void attn_fp8_roll(hipStream_t s, float* amax, int n_layers);
void attn_local_fwd(hipStream_t s, DType dt, const AttnArgs& a);
void attn_local_bwd(hipStream_t s, DType dt, const AttnBwdArgs& a);

// ----------------------------------------------------------------------------- head / loss / utilities
// logits_pad [N][ldl] fp32 -> logits/probs [N][V] (optional, fp32)
void head_probs(hipStream_t s, const float* logits_pad, int ldl, float* logits, float* probs, int N, int V);
// BCE-with-logits: loss_acc[0] += sum over all (n,v) of bce * scale / B ; dlogits(T)[N][ldd] = (sigmoid(z)-y)*scale/B
// (columns >= V of dlogits are zeroed)
// labels: float32, or IEEE half when labels_f16 (the reference's loader hands over float16 labels, grain_loader.py:88)
void bce_loss(hipStream_t s, DType dt, const float* logits_pad, int ldl, const void* labels, float* loss_acc,
              void* dlogits, int ldd, int N, int V, float scale, int B, bool labels_f16 = false);
// elementwise dropout: Y = X * keep / (1-p)
void dropout_apply(hipStream_t s, DType dt, const void* X, void* Y, size_t n, uint64_t seed, uint32_t stream,
                   uint32_t drop_thresh24, float inv_keep, const uint64_t* seed_dev = nullptr);
// g[i] *= mul ; flag[0] = 0 if any non-finite
// dst[i] = src[i] * mul, AND-ing `valid_flag` with "all finite" (src == dst allowed)
void grads_finalize(hipStream_t s, const float* src, float* dst, size_t n, float mul, int* valid_flag);
void convert_f32(hipStream_t s, DType dt, const float* src, void* dst, size_t n);
// dst[c][r] = src[r][c]   (src [R x C] fp32 -> dst [C x R] T)
void convert_transpose(hipStream_t s, DType dt, const float* src, void* dst, int R, int C);
void fill_f32(hipStream_t s, float* p, float v, size_t n);
void copy_f32(hipStream_t s, const float* src, float* dst, size_t n);
void scale_copy_f32(hipStream_t s, const float* src, float* dst, size_t n, float mul);   // dst = src * mul (small vectors)
// dst[0..n) = src, dst[n..n_pad) = 0 in one launch
void copy_pad_f32(hipStream_t s, const float* src, float* dst, size_t n, size_t n_pad);

// Layer scale of the CNN Blocks (model.py:157-158, 166: y = x + gamma * z, z = W2 h + b2) without saving z:
//   dgamma[c] = sum_n dy[n][c] z[n][c] = sum_j W2[c][j] (sum_n dy[n][c] h[n][j]) + b2[c] sum_n dy[n][c]
//             = < W2[c][:], dW2raw[c][:] > + b2[c] db2raw[c]
// i.e. a dot product with the pw2 weight gradient BEFORE its own scaling by gamma.  The weight-gradient launches leave
// dW2raw / db2raw in the gradient buffer; this pass (one launch for all Blocks, at the end of the backward pass) writes
// dgamma and scales the rows in place (dW2 = gamma dW2raw, db2 = gamma db2raw).  W2 is taken as the activation type's
// rounding of the fp32 master - the value the forward multiplied with.  Round 4: the forward stores one tensor less per
// Block (z: 2 of 14.3 MB per C = 128 launch) and the backward reads one less.
struct GammaDesc { uint64_t w2, b2, gamma; int C; int row0; };      // flat offsets (params and gradients share the layout); first row index
void cnn_gamma_finish(hipStream_t s, DType dt, const float* params, float* grads, const GammaDesc* descs_dev, int n_desc, int total_rows);

// AdamW (optax.adamw) on the flat parameter vector with per-segment learning rates
// pass 1: update m, v, write the (unclipped) update into upd, accumulate sum(upd^2) into norm2[0]
void adamw_compute(hipStream_t s, const float* params, const float* grads, float* m, float* v, float* upd,
                   const float* lr_scale, size_t n, float lr, float b1, float b2, float eps, float wd, int step,
                   const int* count, float* norm2, const int* gate_i, const float* gate_f);
// pass 2: params += upd * min(1, clip / sqrt(norm2)); norm2 = float[2] (sum, reader count), left zeroed for the next step
bool step_begin(hipStream_t s, float* grads, size_t n, float* loss_acc, int* valid);   // false: not applicable (alignment)
void pack_tail(hipStream_t s, const float* loss, const int* valid, float* head_valid, float* tail_loss, float* tail_valid);
// AdamW straight from the context's gradient accumulator (head.hip): norm pass + commit pass, g is read as g * gmul and left
// ZEROED; the all-finite flag is isfinite(|update|^2) && isfinite(loss) and goes to valid_out (device int, optional).
// norm2: float[8 + 4096] (control words + one partial-sum slot per workgroup of the norm pass); grid: workgroups of the norm
// pass = of the one-launch form (the number of compute units); persist: take the one-launch form
void adamw_fused(hipStream_t s, float* params, float* gacc, float* m, float* v, const float* lr_scale, size_t n, float gmul,
                 float lr, float b1, float b2, float eps, float wd, int step, int* count, float* norm2, float clip,
                 const int* gate_i, const float* gate_f, const float* loss, int* valid_out, int grid = 0, bool persist = false,
                 const unsigned char* lr_idx = nullptr, const float* lr_tab = nullptr,
                 int* persist_fail_host = nullptr, unsigned spin_limit = 0, int delay_wg = -1, unsigned delay_spins = 0);
// gate_i / gate_f (device, either may be null): both passes are no-ops unless *gate_i != 0 and *gate_f >= 0.999;
// step > 0: explicit optimizer step, step < 0: *count + 1 (device count of applied updates, advanced behind an open gate)
void adamw_apply(hipStream_t s, float* params, const float* upd, size_t n, float* norm2, float clip,
                 const int* gate_i, const float* gate_f, int step, int* count);

// batched fp32 -> T weight conversion (plain copy or transpose), one launch for the whole model
struct ConvDesc { uint64_t src_off; uint64_t dst_off; int rows, cols; int src_rows; int transpose;
                  long long scale_off; /* >= 0: dst[r][c] *= params[scale_off + c] */
                  int pack;            /* != 0: dst in MFMA-fragment order (chain.h: chain_packed_index) */ };
// Fragment-packed weight copy for the chain kernels: [rows/16][ceil(cols/32)][64 lanes][8], lane = 16 g + r
// holds W[16 rb + r][32 ks + 8 g .. + 8) — one wave-level A-fragment load is 1 KiB contiguous
// (row-major gives 16 rows x 64 B per load: half of every cache line per instruction).
static inline constexpr __host__ __device__ size_t chain_packed_elems(int rows, int cols) { return (size_t)(rows / 16) * ((cols + 31) / 32) * 512; }
void convert_batch(hipStream_t s, DType dt, const float* src, void* dst, const ConvDesc* descs_dev,
                   const int* tile_prefix_dev, int n_desc, int total_tiles);

}  // namespace amt
