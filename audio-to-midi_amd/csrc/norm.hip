// Row LayerNorm forward / backward (fp32 statistics via wavefront shuffles).
// Reference: eqx.nn.LayerNorm under jax.vmap over positions / tokens, computed in float32
// (model.py:100,117,162,190,539,546,759).
// Every LayerNorm of the model reads the forward residual stream: X is stored as Res<T> (common.h); the output is T (an
// operand of the GEMM that follows) except for the final CNN norm, whose output IS the transformer's stream (y_res).
#include "rowops.h"
#include "kernels.h"
#include "prof.h"

namespace amt {

template <typename T, typename TY, int C>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const Res<T>* __restrict__ X, const float* __restrict__ w,
                                                     const float* __restrict__ b, TY* __restrict__ Y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int N) {
  using R = RowCfg<C>;
  constexpr int VEC = R::VEC, LPR = R::LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rw = lane / LPR;
  float wv[VEC], bv[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { wv[i] = w[sub * VEC + i]; bv[i] = b[sub * VEC + i]; }
  for (long row = ((long)blockIdx.x * 4 + wave) * R::RPW + rw; row < N; row += (long)gridDim.x * R::RPB) {
    float x[VEC];
    loadvf<VEC>(X + row * C + sub * VEC, x);
    // explicit rounding sequence (rowops.h: ln_stats / ln_apply), shared with the chain kernels
    float mu, rs;
    ln_stats<VEC, LPR>(x, mu, rs);
    float y[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) y[i] = ln_apply(x[i], rs, wv[i], bv[i]);
    storevf<VEC>(Y + row * C + sub * VEC, y);
    if (sub == 0 && mean != nullptr) { mean[row] = mu; rstd[row] = rs; }
  }
}

template <typename T, int C>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dY, const Res<T>* __restrict__ X,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ w, const T* dRes,
                                                     T* dX, float* __restrict__ dw,
                                                     float* __restrict__ db, int N, int pstride) {
  using R = RowCfg<C>;
  constexpr int VEC = R::VEC, LPR = R::LPR;
  __shared__ float smem[4 * C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rw = lane / LPR;
  float wv[VEC], aw[VEC], ab[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { wv[i] = w[sub * VEC + i]; aw[i] = 0.f; ab[i] = 0.f; }
  for (long row = ((long)blockIdx.x * 4 + wave) * R::RPW + rw; row < N; row += (long)gridDim.x * R::RPB) {
    float x[VEC], dy[VEC];
    loadvf<VEC>(X + row * C + sub * VEC, x);
    loadvf<VEC>(dY + row * C + sub * VEC, dy);
    float mu = mean[row], rs = rstd[row];
    // explicit roundings (no compiler-chosen fma contraction): the chain kernels (chain.h:
    // chain_ln_bwd_rows) use the same sequence
    float s1 = 0.f, s2 = 0.f, gy[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      x[i] = __fmul_rn(__fsub_rn(x[i], mu), rs);          // xhat
      gy[i] = __fmul_rn(dy[i], wv[i]);
      s1 = __fadd_rn(s1, gy[i]);
      s2 = __fmaf_rn(gy[i], x[i], s2);
      aw[i] += dy[i] * x[i];
      ab[i] += dy[i];
    }
    s1 = group_sum<LPR>(s1) * (1.f / C);
    s2 = group_sum<LPR>(s2) * (1.f / C);
    float o[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = __fmul_rn(rs, __fmaf_rn(-x[i], s2, __fsub_rn(gy[i], s1)));
    if (dRes != nullptr) {
      float rr[VEC];
      loadvf<VEC>(dRes + row * C + sub * VEC, rr);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = __fadd_rn(o[i], rr[i]);
    }
    storevf<VEC>(dX + row * C + sub * VEC, o);
  }
  reduce_cols_to_global<VEC, LPR>(aw, dw, smem, pstride);
  reduce_cols_to_global<VEC, LPR>(ab, db, smem, pstride);
}

template <typename T, typename TY>
static void ln_fwd_t(hipStream_t s, const void* X, const float* w, const float* b, void* Y, float* mean, float* rstd,
                     int N, int C) {
  AMT_DISPATCH_C(C, hipLaunchKernelGGL((ln_fwd_kernel<T, TY, CC>), dim3(row_grid(N, RowCfg<CC>::RPB)), dim3(256), 0, s,
                                       (const Res<T>*)X, w, b, (TY*)Y, mean, rstd, N));
}
template <typename T>
static void ln_bwd_t(hipStream_t s, const void* dY, const void* X, const float* mean, const float* rstd,
                     const float* w, const void* dRes, void* dX, float* dw, float* db, int N, int C, int nb,
                     int pstride) {
  AMT_DISPATCH_C(C, hipLaunchKernelGGL((ln_bwd_kernel<T, CC>), dim3(nb), dim3(256), 0, s, (const T*)dY, (const Res<T>*)X,
                                       mean, rstd, w, (const T*)dRes, (T*)dX, dw, db, N, pstride));
}

void layernorm_fwd(hipStream_t s, DType dt, const void* X, const float* w, const float* b, void* Y, float* mean,
                   float* rstd, int N, int C, bool y_res) {
  ProfScope ps(s, "layernorm_fwd", 0, 2.0 * N * C * (dt == DT_F32 ? 4 : 2));
  if (dt == DT_F32) ln_fwd_t<float, float>(s, X, w, b, Y, mean, rstd, N, C);
  else if (y_res) ln_fwd_t<bf16, Res<bf16>>(s, X, w, b, Y, mean, rstd, N, C);
  else ln_fwd_t<bf16, bf16>(s, X, w, b, Y, mean, rstd, N, C);
}
void layernorm_bwd(hipStream_t s, DType dt, const void* dY, const void* X, const float* mean, const float* rstd,
                   const float* w, const void* dRes, void* dX, float* dw, float* db, int N, int C,
                   PartialSink* sink) {
  const int nb = bwd_blocks(N, C, sink);
  int pstride = 0;
  if (sink) {   // records [nb][w(C) | b(C)]
    size_t o = sink->take((size_t)nb * 2 * C);
    sink->add(o, nb, 2 * C, C, dw);
    sink->add(o + C, nb, 2 * C, C, db);
    if (sink->dry) return;
    pstride = 2 * C;
    dw = sink->ws + o;
    db = dw + C;
  }
  ProfScope ps(s, "layernorm_bwd", 0, (dRes ? 4.0 : 3.0) * N * C * (dt == DT_F32 ? 4 : 2));
  if (dt == DT_F32) ln_bwd_t<float>(s, dY, X, mean, rstd, w, dRes, dX, dw, db, N, C, nb, pstride);
  else ln_bwd_t<bf16>(s, dY, X, mean, rstd, w, dRes, dX, dw, db, N, C, nb, pstride);
}

}  // namespace amt
