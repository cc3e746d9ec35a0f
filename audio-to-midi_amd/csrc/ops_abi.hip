// extern "C" wrappers of the individual kernels (include/amt_ops.h) for op-level parity tests.
#include <cmath>
#include <cstring>

#include "../../include/amt.h"
#include "../../include/amt_ops.h"
#include "kernels.h"

using namespace amt;

namespace {
inline DType dtof(int32_t d) { return d == AMT_F32 ? DT_F32 : DT_BF16; }
inline hipStream_t st(void* s) { return static_cast<hipStream_t>(s); }
inline int32_t done() { return hipGetLastError() == hipSuccess ? 0 : 1; }
inline uint32_t thr24(float p) { return p > 0.f ? (uint32_t)lrintf(p * 16777216.f) : 0u; }
inline float inv_keep(float p) { uint32_t t = thr24(p); return t ? 1.f / (1.f - (float)t / 16777216.f) : 1.f; }

uint32_t h_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
}  // namespace

extern "C" {

int32_t amt_op_gemm_nt(int32_t dtype, int32_t epi, const void* X, int32_t ldx, const void* W, int32_t ldw, int32_t N,
                       int32_t M, int32_t K, const float* bias, void* Y, int32_t ldy, void* Y2, int32_t ldy2,
                       const void* R, int32_t ldr, const void* U, int32_t ldu, const float* gamma, uint64_t seed,
                       uint32_t stream_id, float drop_p, void* stream) {
  if (epi < 0 || epi >= EPI_COUNT || M % 4 != 0) return 2;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.X = X; g.ldx = ldx; g.W = W; g.ldw = ldw; g.N = N; g.M = M; g.K = K; g.bias = bias;
  g.Y = Y; g.ldy = ldy; g.Y2 = Y2; g.ldy2 = ldy2; g.R = R; g.ldr = ldr; g.U = U; g.ldu = ldu; g.gamma = gamma;
  g.seed = seed; g.stream = stream_id; g.drop_thresh24 = thr24(drop_p); g.inv_keep = inv_keep(drop_p);
  gemm_nt(st(stream), dtof(dtype), (EpiKind)epi, g);
  return done();
}

int32_t amt_op_gemm_tn_wgrad(int32_t dtype, const void* G, int32_t ldg, const void* A, int32_t lda, int32_t N, int32_t M,
                             int32_t Kp, float* dW, int32_t ldw, float* dbias, int32_t M_store, void* stream) {
  WgradArgs w;
  memset(&w, 0, sizeof(w));
  w.G = G; w.ldg = ldg; w.A = A; w.lda = lda; w.N = N; w.M = M; w.Kp = Kp; w.dW = dW; w.ldw = ldw; w.dbias = dbias;
  w.M_store = M_store;
  gemm_tn_wgrad(st(stream), dtof(dtype), w);
  return done();
}

int32_t amt_op_layernorm_fwd(int32_t dtype, const void* X, const float* w, const float* b, void* Y, float* mean,
                             float* rstd, int32_t N, int32_t C, void* stream) {
  layernorm_fwd(st(stream), dtof(dtype), X, w, b, Y, mean, rstd, N, C);
  return done();
}
int32_t amt_op_layernorm_bwd(int32_t dtype, const void* dY, const void* X, const float* mean, const float* rstd,
                             const float* w, const void* dRes, void* dX, float* dw, float* db, int32_t N, int32_t C,
                             void* stream) {
  layernorm_bwd(st(stream), dtof(dtype), dY, X, mean, rstd, w, dRes, dX, dw, db, N, C);
  return done();
}
int32_t amt_op_stem_fwd(int32_t dtype, const void* audio, const float* w, const float* b, const float* lnw,
                        const float* lnb, void* Y, int32_t B, int32_t T, void* stream) {
  stem_fwd(st(stream), dtof(dtype), audio, w, b, lnw, lnb, Y, B, T);
  return done();
}
int32_t amt_op_stem_bwd(int32_t dtype, const void* audio, const void* dY, const float* w, const float* b,
                        const float* lnw, float* dw, float* db, float* dlnw, float* dlnb, int32_t B, int32_t T,
                        void* stream) {
  stem_bwd(st(stream), dtof(dtype), audio, dY, w, b, lnw, dw, db, dlnw, dlnb, B, T);
  return done();
}
int32_t amt_op_dwln_fwd(int32_t dtype, const void* X, const float* wdw, const float* bdw, const float* lnw,
                        const float* lnb, void* Y, float* mean, float* rstd, int32_t B, int32_t L, int32_t C,
                        void* stream) {
  dwln_fwd(st(stream), dtof(dtype), X, wdw, bdw, lnw, lnb, Y, mean, rstd, B, L, C);
  return done();
}
int32_t amt_op_dwln_bwd(int32_t dtype, const void* dY, const void* X, const float* wdw, const float* bdw,
                        const float* mean, const float* rstd, const float* lnw, void* dD, float* dlnw, float* dlnb,
                        int32_t B, int32_t L, int32_t C, void* stream) {
  dwln_bwd(st(stream), dtof(dtype), dY, X, wdw, bdw, mean, rstd, lnw, dD, dlnw, dlnb, B, L, C);
  return done();
}
int32_t amt_op_dwconv_bwd(int32_t dtype, const void* dD, const void* X, const float* wdw, const void* dRes, void* dX,
                          float* dwdw, float* dbdw, int32_t B, int32_t L, int32_t C, void* stream) {
  dwconv_bwd(st(stream), dtof(dtype), dD, X, wdw, dRes, dX, dwdw, dbdw, B, L, C);
  return done();
}
int32_t amt_op_dwblock_bwd(int32_t dtype, const void* dLN, const void* X, const void* dY, const void* Z,
                           const float* wdw, const float* bdw, const float* mean, const float* rstd, const float* lnw,
                           void* dX, float* dlnw, float* dlnb, float* dwdw, float* dbdw, float* dgamma, int32_t B,
                           int32_t L, int32_t C, void* stream) {
  dwblock_bwd(st(stream), dtof(dtype), dLN, X, dY, Z, wdw, bdw, mean, rstd, lnw, dX, dlnw, dlnb, dwdw, dbdw, dgamma, B, L,
              C);
  return done();
}
int32_t amt_op_scale_bwd(int32_t dtype, const void* dY, const void* Z, const float* gamma, void* dZ, float* dgamma,
                         int32_t N, int32_t C, void* stream) {
  scale_bwd(st(stream), dtof(dtype), dY, Z, gamma, dZ, dgamma, N, C);
  return done();
}

int32_t amt_op_attn_fwd(int32_t dtype, int32_t local, const void* QC, int32_t ldqc, const void* KV, int32_t ldkv,
                        void* O, int32_t ldo, float* LSE, const float* cos_t, const float* sin_t, int32_t B, int32_t S,
                        int32_t H, uint64_t seed, uint32_t stream_id, float drop_p, void* stream) {
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.QC = QC; a.ldqc = ldqc; a.KV = KV; a.ldkv = ldkv; a.O = O; a.ldo = ldo; a.LSE = LSE; a.cos = cos_t; a.sin = sin_t;
  a.B = B; a.S = S; a.H = H; a.seed = seed; a.stream = stream_id; a.drop_thresh24 = thr24(drop_p);
  a.inv_keep = inv_keep(drop_p);
  if (local) attn_local_fwd(st(stream), dtof(dtype), a); else attn_global_fwd(st(stream), dtof(dtype), a);
  return done();
}

int32_t amt_op_attn_global_fwd_fp8(const void* QC, int32_t ldqc, const void* KV, int32_t ldkv, void* O, int32_t ldo,
                                   float* LSE, const float* cos_t, const float* sin_t, int32_t B, int32_t S, int32_t H,
                                   uint64_t seed, uint32_t stream_id, float drop_p, float* amax, void* stream) {
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.QC = QC; a.ldqc = ldqc; a.KV = KV; a.ldkv = ldkv; a.O = O; a.ldo = ldo; a.LSE = LSE; a.cos = cos_t; a.sin = sin_t;
  a.B = B; a.S = S; a.H = H; a.seed = seed; a.stream = stream_id; a.drop_thresh24 = thr24(drop_p);
  a.inv_keep = inv_keep(drop_p);
  attn_fp8_roll(st(stream), amax, 1);
  attn_global_fwd_fp8(st(stream), a, amax, amax + 4);
  return done();
}

int32_t amt_op_attn_bwd(int32_t dtype, int32_t local, const void* QC, int32_t ldqc, const void* KV, int32_t ldkv,
                        const void* O, int32_t ldo, const float* LSE, const float* cos_t, const float* sin_t, int32_t B,
                        int32_t S, int32_t H, uint64_t seed, uint32_t stream_id, float drop_p, const void* dO,
                        int32_t lddo, void* dQC, int32_t lddqc, void* dKV, int32_t lddkv, float* delta, void* stream) {
  AttnBwdArgs b;
  memset(&b, 0, sizeof(b));
  AttnArgs& a = b.f;
  a.QC = QC; a.ldqc = ldqc; a.KV = KV; a.ldkv = ldkv; a.O = const_cast<void*>(O); a.ldo = ldo;
  a.LSE = const_cast<float*>(LSE); a.cos = cos_t; a.sin = sin_t; a.B = B; a.S = S; a.H = H; a.seed = seed;
  a.stream = stream_id; a.drop_thresh24 = thr24(drop_p); a.inv_keep = inv_keep(drop_p);
  b.dO = dO; b.lddo = lddo; b.dQC = dQC; b.lddqc = lddqc; b.dKV = dKV; b.lddkv = lddkv; b.delta = delta;
  if (local) attn_local_bwd(st(stream), dtof(dtype), b); else attn_global_bwd(st(stream), dtof(dtype), b);
  return done();
}

int32_t amt_op_bce_loss(int32_t dtype, const float* logits_pad, int32_t ldl, const float* labels, float* loss_acc,
                        void* dlogits, int32_t ldd, int32_t N, int32_t V, float scale, int32_t B, void* stream) {
  bce_loss(st(stream), dtof(dtype), logits_pad, ldl, labels, loss_acc, dlogits, ldd, N, V, scale, B);
  return done();
}

int32_t amt_op_dropout(int32_t dtype, const void* X, void* Y, uint64_t n, uint64_t seed, uint32_t stream_id,
                       float drop_p, void* stream) {
  dropout_apply(st(stream), dtof(dtype), X, Y, (size_t)n, seed, stream_id, thr24(drop_p), inv_keep(drop_p));
  return done();
}

uint32_t amt_rng_u32(uint64_t seed, uint32_t stream_id, uint32_t idx) {
  uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
  const uint32_t key = h_mix32(h_mix32(s0 ^ (stream_id * 0x85ebca6bU)) + s1 + 0x165667b1U);
  uint32_t h = h_mix32((idx * 0x9e3779b1U) ^ key);
  return h;
}
uint32_t amt_drop_threshold24(float drop_p) { return thr24(drop_p); }

}  // extern "C"
