// CNN frontend kernels (channels-last [B][L][C]):
//   stem_fwd / stem_bwd      Stem: Conv1d(2->4, k5, s5) + per-position LayerNorm   model.py:84-100
//   dwln_fwd / dwln_bwd      Block: depthwise Conv1d(k7, SAME) + LayerNorm          model.py:161-162
//   dwconv_bwd               data + weight gradient of the depthwise conv
//   scale_bwd                layer-scale gamma                                      model.py:166
// The pointwise convs of Block and the k2/s2 conv of Downsample are GEMMs (gemm.hip): in
// channels-last a stride-2 kernel-2 conv is a Linear over two concatenated positions.
// The Blocks' input x / output y (and the Stem's output) are the forward residual stream: stored as Res<T> (common.h).
#include "rowops.h"
#include "kernels.h"
#include "prof.h"
#include <algorithm>

namespace amt {

// ------------------------------------------------------------------ stem
// One thread per output position; a block stages 256 positions x 5 taps x 2 channels of audio in
// LDS with 16-byte coalesced loads (the audio is read from HBM exactly once).
constexpr int STEM_POS = 256;
constexpr int STEM_SAMP = STEM_POS * 5;   // 1280 samples per channel per block

// The audio is the model's INPUT: it has the residual stream's storage type (common.h: Res<T>) - IEEE half in bf16 contexts,
// which is what the reference feeds its model (grain_loader.py:88 casts the samples to float16; python.rs:251-261 rounds them
// through f16 at load time): no input quantisation on top of the reference's own.
template <typename T>
AMT_DEV void stem_stage_audio(const T* audio, float (*xs)[STEM_SAMP], int b, long t0, int T_) {
  // 2 channels x 1280 samples = 320 chunks of 8
  for (int id = threadIdx.x; id < 2 * (STEM_SAMP / 8); id += blockDim.x) {
    int ch = id / (STEM_SAMP / 8), c = id % (STEM_SAMP / 8);
    long t = t0 + 8 * c;
    float v[8];
    if (t + 8 <= T_) {
      load8f(audio + ((long)b * 2 + ch) * T_ + t, v);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) xs[ch][8 * c + i] = v[i];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const Res<T>* __restrict__ audio, const float* __restrict__ w,
                                                       const float* __restrict__ bias, const float* __restrict__ lnw,
                                                       const float* __restrict__ lnb, Res<T>* __restrict__ Y, int B, int T_,
                                                       int L, int tiles_per_clip) {
  __shared__ float xs[2][STEM_SAMP];
  const int b = blockIdx.x / tiles_per_clip, tile = blockIdx.x % tiles_per_clip;
  const int l0 = tile * STEM_POS;
  stem_stage_audio(audio, xs, b, (long)l0 * 5, T_);
  __syncthreads();
  const int l = l0 + threadIdx.x;
  if (l >= L) return;
  float c[4];
#pragma unroll
  for (int co = 0; co < 4; ++co) {
    float a = bias[co];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int k = 0; k < 5; ++k) a += w[(co * 2 + ci) * 5 + k] * xs[ci][threadIdx.x * 5 + k];
    c[co] = a;
  }
  float mu = 0.25f * (c[0] + c[1] + c[2] + c[3]);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { c[i] -= mu; q += c[i] * c[i]; }
  float rs = __builtin_amdgcn_rsqf(0.25f * q + kLnEps);
  float y[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) y[i] = c[i] * rs * lnw[i] + lnb[i];
  store4f(Y + ((long)b * L + l) * 4, y);
}

template <typename T>
__global__ __launch_bounds__(256) void stem_bwd_kernel(const Res<T>* __restrict__ audio, const T* __restrict__ dY,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       const float* __restrict__ lnw, float* __restrict__ dw,
                                                       float* __restrict__ db, float* __restrict__ dlnw,
                                                       float* __restrict__ dlnb, int B, int T_, int L,
                                                       int tiles_per_clip, int total_tiles, float* __restrict__ rec) {
  __shared__ float xs[2][STEM_SAMP];
  __shared__ float red[52];
  float acc[52];   // 40 dw, 4 db, 4 dlnw, 4 dlnb
#pragma unroll
  for (int i = 0; i < 52; ++i) acc[i] = 0.f;
  for (int t = blockIdx.x; t < total_tiles; t += gridDim.x) {
    const int b = t / tiles_per_clip, tile = t % tiles_per_clip;
    const int l0 = tile * STEM_POS;
    __syncthreads();
    stem_stage_audio(audio, xs, b, (long)l0 * 5, T_);
    __syncthreads();
    const int l = l0 + threadIdx.x;
    if (l < L) {
      float c[4], dy[4];
#pragma unroll
      for (int co = 0; co < 4; ++co) {
        float a = bias[co];
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
          for (int k = 0; k < 5; ++k) a += w[(co * 2 + ci) * 5 + k] * xs[ci][threadIdx.x * 5 + k];
        c[co] = a;
      }
      load4f(dY + ((long)b * L + l) * 4, dy);
      float mu = 0.25f * (c[0] + c[1] + c[2] + c[3]);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { c[i] -= mu; q += c[i] * c[i]; }
      float rs = __builtin_amdgcn_rsqf(0.25f * q + kLnEps);
      float s1 = 0.f, s2 = 0.f, gy[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        c[i] *= rs;   // xhat
        gy[i] = dy[i] * lnw[i];
        s1 += gy[i];
        s2 += gy[i] * c[i];
        acc[44 + i] += dy[i] * c[i];
        acc[48 + i] += dy[i];
      }
      s1 *= 0.25f; s2 *= 0.25f;
#pragma unroll
      for (int co = 0; co < 4; ++co) {
        float dc = rs * (gy[co] - s1 - c[co] * s2);
        acc[40 + co] += dc;
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
          for (int k = 0; k < 5; ++k) acc[(co * 2 + ci) * 5 + k] += dc * xs[ci][threadIdx.x * 5 + k];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 52) red[threadIdx.x] = 0.f;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 52; ++i) {
    float v = group_sum<64>(acc[i]);
    if ((threadIdx.x & 63) == 0) atomicAdd(&red[i], v);
  }
  __syncthreads();
  if (rec != nullptr) {   // one record [dw 40 | db 4 | dlnw 4 | dlnb 4] per block, summed by reduce_partials
    if (threadIdx.x < 52) rec[(size_t)blockIdx.x * 52 + threadIdx.x] = red[threadIdx.x];
    return;
  }
  if (threadIdx.x < 40) atomicAdd(&dw[threadIdx.x], red[threadIdx.x]);
  else if (threadIdx.x < 44) atomicAdd(&db[threadIdx.x - 40], red[threadIdx.x]);
  else if (threadIdx.x < 48) atomicAdd(&dlnw[threadIdx.x - 44], red[threadIdx.x]);
  else if (threadIdx.x < 52) atomicAdd(&dlnb[threadIdx.x - 48], red[threadIdx.x]);
}

void stem_fwd(hipStream_t s, DType dt, const void* audio, const float* w, const float* b, const float* lnw,
              const float* lnb, void* Y, int B, int T_) {
  int L = T_ / 5, tiles = (L + STEM_POS - 1) / STEM_POS;
  ProfScope ps(s, "stem_fwd", 2.0 * B * L * 40, ((double)B * 2 * T_ + (double)B * L * 4) * (dt == DT_F32 ? 4 : 2));
  if (dt == DT_F32)
    hipLaunchKernelGGL((stem_fwd_kernel<float>), dim3(B * tiles), dim3(256), 0, s, (const float*)audio, w, b, lnw, lnb,
                       (float*)Y, B, T_, L, tiles);
  else
    hipLaunchKernelGGL((stem_fwd_kernel<bf16>), dim3(B * tiles), dim3(256), 0, s, (const Res<bf16>*)audio, w, b, lnw, lnb,
                       (Res<bf16>*)Y, B, T_, L, tiles);
}

void stem_bwd(hipStream_t s, DType dt, const void* audio, const void* dY, const float* w, const float* b,
              const float* lnw, float* dw, float* db, float* dlnw, float* dlnb, int B, int T_, PartialSink* sink) {
  int L = T_ / 5, tiles = (L + STEM_POS - 1) / STEM_POS, total = B * tiles;
  // with per-block records: one 256-position tile per block (the blocks' memory round trips overlap);
  // with atomics: 52 same-address adds per block, so few blocks with several tiles each
  int grid = sink ? (total < 2048 ? total : 2048) : (total < 128 ? total : 128);
  float* rec = nullptr;
  if (sink) {
    size_t o = sink->take((size_t)grid * 52);
    sink->add(o, grid, 52, 40, dw);
    sink->add(o + 40, grid, 52, 4, db);
    sink->add(o + 44, grid, 52, 4, dlnw);
    sink->add(o + 48, grid, 52, 4, dlnb);
    if (sink->dry) return;
    rec = sink->ws + o;
  }
  ProfScope ps(s, "stem_bwd", 4.0 * B * L * 40, ((double)B * 2 * T_ + (double)B * L * 4) * (dt == DT_F32 ? 4 : 2));
  if (dt == DT_F32)
    hipLaunchKernelGGL((stem_bwd_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)audio, (const float*)dY, w,
                       b, lnw, dw, db, dlnw, dlnb, B, T_, L, tiles, total, rec);
  else
    hipLaunchKernelGGL((stem_bwd_kernel<bf16>), dim3(grid), dim3(256), 0, s, (const Res<bf16>*)audio, (const bf16*)dY, w, b,
                       lnw, dw, db, dlnw, dlnb, B, T_, L, tiles, total, rec);
}

// ------------------------------------------------------------------ depthwise conv k7 (+ LayerNorm)
// conv output of row (b, l), feature chunk `sub`: sum_k w[k][c] X[b][l + k - 3][c] + bias[c]
template <typename TX, int VEC>
AMT_DEV void dwconv_row(const TX* __restrict__ X, long rowbase /* b*L */, int l, int L, int C, int sub,
                        const float (&wk)[7][VEC], const float (&bv)[VEC], float (&out)[VEC]) {
  // all seven taps are loaded unconditionally from a clamped row and masked afterwards: a branch
  // per tap would serialise seven L2 round trips (measured ~4 us per row visit)
  float x[7][VEC];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    int ll = l + k - 3;
    int lc = ll < 0 ? 0 : (ll >= L ? L - 1 : ll);
    loadvf<VEC>(X + (rowbase + lc) * C + sub * VEC, x[k]);
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) out[i] = bv[i];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    int ll = l + k - 3;
    float m = (ll >= 0 && ll < L) ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] += wk[k][i] * (x[k][i] * m);
  }
}

template <typename T, int C>
__global__ __launch_bounds__(256) void dwln_fwd_kernel(const Res<T>* __restrict__ X, const float* __restrict__ wdw,
                                                       const float* __restrict__ bdw, const float* __restrict__ lnw,
                                                       const float* __restrict__ lnb, T* __restrict__ Y,
                                                       float* __restrict__ mean, float* __restrict__ rstd, int B,
                                                       int L) {
  using R = RowCfg<C>;
  constexpr int VEC = R::VEC, LPR = R::LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rw = lane / LPR;
  float wk[7][VEC], bv[VEC], wv[VEC], lb[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    bv[i] = bdw[sub * VEC + i]; wv[i] = lnw[sub * VEC + i]; lb[i] = lnb[sub * VEC + i];
#pragma unroll
    for (int k = 0; k < 7; ++k) wk[k][i] = wdw[k * C + sub * VEC + i];
  }
  const long N = (long)B * L;
  for (long row = ((long)blockIdx.x * 4 + wave) * R::RPW + rw; row < N; row += (long)gridDim.x * R::RPB) {
    int b = (int)(row / L), l = (int)(row % L);
    float d[VEC];
    dwconv_row<Res<T>, VEC>(X, (long)b * L, l, L, C, sub, wk, bv, d);
    float mu, rs;
    ln_stats<VEC, LPR>(d, mu, rs);
    float y[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) y[i] = ln_apply(d[i], rs, wv[i], lb[i]);
    storevf<VEC>(Y + row * C + sub * VEC, y);
    if (sub == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

template <typename T, int C>
__global__ __launch_bounds__(256) void dwln_bwd_kernel(const T* __restrict__ dY, const Res<T>* __restrict__ X,
                                                       const float* __restrict__ wdw, const float* __restrict__ bdw,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ lnw, T* __restrict__ dD,
                                                       float* __restrict__ dlnw, float* __restrict__ dlnb, int B,
                                                       int L, int pstride) {
  using R = RowCfg<C>;
  constexpr int VEC = R::VEC, LPR = R::LPR;
  __shared__ float smem[4 * C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rw = lane / LPR;
  float wk[7][VEC], bv[VEC], wv[VEC], aw[VEC], ab[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    bv[i] = bdw[sub * VEC + i]; wv[i] = lnw[sub * VEC + i]; aw[i] = 0.f; ab[i] = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) wk[k][i] = wdw[k * C + sub * VEC + i];
  }
  const long N = (long)B * L;
  for (long row = ((long)blockIdx.x * 4 + wave) * R::RPW + rw; row < N; row += (long)gridDim.x * R::RPB) {
    int b = (int)(row / L), l = (int)(row % L);
    float d[VEC], dy[VEC];
    dwconv_row<Res<T>, VEC>(X, (long)b * L, l, L, C, sub, wk, bv, d);
    loadvf<VEC>(dY + row * C + sub * VEC, dy);
    float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f, gy[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      d[i] = (d[i] - mu) * rs;
      gy[i] = dy[i] * wv[i];
      s1 += gy[i];
      s2 += gy[i] * d[i];
      aw[i] += dy[i] * d[i];
      ab[i] += dy[i];
    }
    s1 = group_sum<LPR>(s1) * (1.f / C);
    s2 = group_sum<LPR>(s2) * (1.f / C);
    float o[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = rs * (gy[i] - s1 - d[i] * s2);
    storevf<VEC>(dD + row * C + sub * VEC, o);
  }
  reduce_cols_to_global<VEC, LPR>(aw, dlnw, smem, pstride);
  reduce_cols_to_global<VEC, LPR>(ab, dlnb, smem, pstride);
}

template <typename T, int C>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const T* __restrict__ dD, const Res<T>* __restrict__ X,
                                                         const float* __restrict__ wdw, const T* dRes, T* dX,
                                                         float* __restrict__ dwdw, float* __restrict__ dbdw, int B,
                                                         int L, int pstride) {
  using R = RowCfg<C>;
  constexpr int VEC = R::VEC, LPR = R::LPR;
  __shared__ float smem[4 * C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rw = lane / LPR;
  float wk[7][VEC], aw[7][VEC], ab[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    ab[i] = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) { wk[k][i] = wdw[k * C + sub * VEC + i]; aw[k][i] = 0.f; }
  }
  const long N = (long)B * L;
  for (long row = ((long)blockIdx.x * 4 + wave) * R::RPW + rw; row < N; row += (long)gridDim.x * R::RPB) {
    int b = (int)(row / L), l = (int)(row % L);
    const long rb = (long)b * L;
    float o[VEC], dd[VEC];
    if (dRes != nullptr) {
      loadvf<VEC>(dRes + row * C + sub * VEC, o);
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = 0.f;
    }
    loadvf<VEC>(dD + row * C + sub * VEC, dd);
#pragma unroll
    for (int i = 0; i < VEC; ++i) ab[i] += dd[i];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      // data grad: dX[l] += w[k] * dD[l - k + 3]
      int ls = l - k + 3;
      if (ls >= 0 && ls < L) {
        float t[VEC];
        loadvf<VEC>(dD + (rb + ls) * C + sub * VEC, t);
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] += wk[k][i] * t[i];
      }
      // weight grad: dw[k] += dD[l] * X[l + k - 3]
      int lx = l + k - 3;
      if (lx >= 0 && lx < L) {
        float x[VEC];
        loadvf<VEC>(X + (rb + lx) * C + sub * VEC, x);
#pragma unroll
        for (int i = 0; i < VEC; ++i) aw[k][i] += dd[i] * x[i];
      }
    }
    storevf<VEC>(dX + row * C + sub * VEC, o);
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) reduce_cols_to_global<VEC, LPR>(aw[k], dwdw + k * C, smem, pstride);
  reduce_cols_to_global<VEC, LPR>(ab, dbdw, smem, pstride);
}

template <typename T, int C>
__global__ __launch_bounds__(256) void scale_bwd_kernel(const T* __restrict__ dY, const T* __restrict__ Z,
                                                        const float* __restrict__ gamma, T* __restrict__ dZ,
                                                        float* __restrict__ dgamma, long N, int pstride) {
  using R = RowCfg<C>;
  constexpr int VEC = R::VEC, LPR = R::LPR;
  __shared__ float smem[4 * C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rw = lane / LPR;
  float gv[VEC], ag[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { gv[i] = gamma[sub * VEC + i]; ag[i] = 0.f; }
  for (long row = ((long)blockIdx.x * 4 + wave) * R::RPW + rw; row < N; row += (long)gridDim.x * R::RPB) {
    float dy[VEC], z[VEC], o[VEC];
    loadvf<VEC>(dY + row * C + sub * VEC, dy);
    loadvf<VEC>(Z + row * C + sub * VEC, z);
#pragma unroll
    for (int i = 0; i < VEC; ++i) { ag[i] += dy[i] * z[i]; o[i] = gv[i] * dy[i]; }
    storevf<VEC>(dZ + row * C + sub * VEC, o);
  }
  reduce_cols_to_global<VEC, LPR>(ag, dgamma, smem, pstride);
}


// ------------------------------------------------------------------ fused tail of the Block backward
// rows kept in LDS per tile (TL interior rows + 3 halo rows each side)
template <int C> struct DwTile {
  static constexpr int ROWS = C <= 8 ? 256 : (C <= 32 ? 128 : (C <= 64 ? 64 : 32));
  static constexpr int TL = ROWS - 6;
};

template <typename T, int C>
__global__ __launch_bounds__(256) void dwblock_bwd_kernel(
    const T* __restrict__ dLN, const Res<T>* __restrict__ X, const T* __restrict__ dY, const T* __restrict__ Z,
    const float* __restrict__ wdw, const float* __restrict__ bdw, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ lnw, T* __restrict__ dX, float* dlnw, float* dlnb,
    float* dwdw, float* dbdw, float* dgamma, int B, int L, int tiles_per_clip, int total_tiles, int pstride) {
  using R = RowCfg<C>;
  constexpr int VEC = R::VEC, LPR = R::LPR, RPB = R::RPB, RPW = R::RPW;
  constexpr int ROWS = DwTile<C>::ROWS, TL = DwTile<C>::TL;
  __shared__ __attribute__((aligned(16))) T dDs[ROWS][C];
  __shared__ float smem[11 * C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rw = lane / LPR;
  float wk[7][VEC], bv[VEC], wv[VEC];
  float aw[7][VEC], ab[VEC], alw[VEC], alb[VEC], ag[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    bv[i] = bdw[sub * VEC + i]; wv[i] = lnw[sub * VEC + i];
    ab[i] = 0.f; alw[i] = 0.f; alb[i] = 0.f; ag[i] = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) { wk[k][i] = wdw[k * C + sub * VEC + i]; aw[k][i] = 0.f; }
  }
  for (int t = blockIdx.x; t < total_tiles; t += gridDim.x) {
    const int b = t / tiles_per_clip, l0 = (t % tiles_per_clip) * TL;
    const long rb = (long)b * L;
    __syncthreads();   // previous tile's phase 2 done with dDs
    // ---- phase 1: dD for rows l0-3 .. l0+TL+2 into LDS.  Loads are unconditional (clamped rows,
    // masked results) and the pass loop is fully unrolled so that the loads of all passes overlap.
#pragma unroll
    for (int pp = 0; pp < ROWS / RPB; ++pp) {
      const int p = pp * RPB + wave * RPW + rw;
      const int l = l0 - 3 + p;
      const bool valid = (l >= 0) && (l < L);
      const int lcl = l < 0 ? 0 : (l >= L ? L - 1 : l);
      float d[VEC], dl[VEC], o[VEC];
      dwconv_row<Res<T>, VEC>(X, rb, lcl, L, C, sub, wk, bv, d);
      loadvf<VEC>(dLN + (rb + lcl) * C + sub * VEC, dl);
      const float mu = mean[rb + lcl], rs = rstd[rb + lcl];
      const float fin = (valid && p >= 3 && p < TL + 3) ? 1.f : 0.f;
      float s1 = 0.f, s2 = 0.f, gy[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        d[i] = (d[i] - mu) * rs;
        gy[i] = dl[i] * wv[i];
        s1 += gy[i];
        s2 += gy[i] * d[i];
        alw[i] += fin * dl[i] * d[i];
        alb[i] += fin * dl[i];
      }
      s1 = group_sum<LPR>(s1) * (1.f / C);
      s2 = group_sum<LPR>(s2) * (1.f / C);
      const float fv = valid ? 1.f : 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = fv * rs * (gy[i] - s1 - d[i] * s2);
      storevf<VEC>(&dDs[p][sub * VEC], o);
    }
    __syncthreads();
    // ---- phase 2: dX = dY + conv^T(dD), depthwise weight / bias / gamma gradients
#pragma unroll
    for (int pp = 0; pp < (TL + RPB - 1) / RPB; ++pp) {
      const int i0 = pp * RPB + wave * RPW + rw;
      const int l = l0 + i0;
      const bool valid = (i0 < TL) && (l < L);
      const int i0c = i0 < TL ? i0 : TL - 1;
      const int lcl = l < L ? l : L - 1;
      const float fv = valid ? 1.f : 0.f;
      float o[VEC], dd[VEC], dy[VEC];
      loadvf<VEC>(dY + (rb + lcl) * C + sub * VEC, dy);
      loadvf<VEC>(&dDs[i0c + 3][sub * VEC], dd);   // value as stored (rounded to T), like the unfused path
#pragma unroll
      for (int i = 0; i < VEC; ++i) { o[i] = dy[i]; dd[i] *= fv; ab[i] += dd[i]; }
      if (Z != nullptr) {
        float z[VEC];
        loadvf<VEC>(Z + (rb + lcl) * C + sub * VEC, z);
#pragma unroll
        for (int i = 0; i < VEC; ++i) ag[i] += fv * dy[i] * z[i];
      }
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        float tq[VEC];
        loadvf<VEC>(&dDs[i0c + 6 - k][sub * VEC], tq);   // dD[l - k + 3] (zero outside the clip)
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] += wk[k][i] * tq[i];
        const int lx = lcl + k - 3;
        const int lxc = lx < 0 ? 0 : (lx >= L ? L - 1 : lx);
        const float mx = (lx >= 0 && lx < L) ? 1.f : 0.f;
        float x[VEC];
        loadvf<VEC>(X + (rb + lxc) * C + sub * VEC, x);
#pragma unroll
        for (int i = 0; i < VEC; ++i) aw[k][i] += dd[i] * (x[i] * mx);
      }
      if (valid) storevf<VEC>(dX + (rb + l) * C + sub * VEC, o);
    }
  }
  // all eleven column sums through LDS with three barriers in total
  __syncthreads();
  for (int t = threadIdx.x; t < 11 * C; t += 256) smem[t] = 0.f;
  __syncthreads();
  auto fold = [&](const float (&acc)[VEC], int slot) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float v = strided_sum<LPR>(acc[i]);
      if (rw == 0) atomicAdd(&smem[slot * C + sub * VEC + i], v);
    }
  };
  fold(alw, 0);
  fold(alb, 1);
#pragma unroll
  for (int k = 0; k < 7; ++k) fold(aw[k], 2 + k);
  fold(ab, 9);
  fold(ag, 10);
  __syncthreads();
  if (pstride > 0) {   // record layout == smem layout: [lnw | lnb | taps 7C | bias | gamma]
    float* rec = dlnw + (size_t)blockIdx.x * pstride;
    const int n = dgamma != nullptr ? 11 * C : 10 * C;
    for (int t = threadIdx.x; t < n; t += 256) rec[t] = smem[t];
  } else {
    for (int t = threadIdx.x; t < 11 * C; t += 256) {
      const int slot = t / C, ch = t % C;
      float* dst = slot == 0 ? dlnw : slot == 1 ? dlnb : slot < 9 ? dwdw + (slot - 2) * C : slot == 9 ? dbdw : dgamma;
      if (dst != nullptr) atomicAdd(&dst[ch], smem[t]);
    }
  }
}

template <typename T>
static void dwblock_bwd_t(hipStream_t s, const void* dLN, const void* X, const void* dY, const void* Z,
                          const float* wdw, const float* bdw, const float* mean, const float* rstd, const float* lnw,
                          void* dX, float* dlnw, float* dlnb, float* dwdw, float* dbdw, float* dgamma, int B, int L, int C,
                          int nb, int tiles_per_clip, int total, int pstride) {
  AMT_DISPATCH_C(C, hipLaunchKernelGGL((dwblock_bwd_kernel<T, CC>), dim3(nb), dim3(256), 0, s, (const T*)dLN,
                                       (const Res<T>*)X, (const T*)dY, (const T*)Z, wdw, bdw, mean, rstd, lnw, (T*)dX, dlnw,
                                       dlnb, dwdw, dbdw, dgamma, B, L, tiles_per_clip, total, pstride));
}

static int dw_tile_len(int C) { return (C <= 8 ? 256 : (C <= 32 ? 128 : (C <= 64 ? 64 : 32))) - 6; }

void dwblock_bwd(hipStream_t s, DType dt, const void* dLN, const void* X, const void* dY, const void* Z,
                 const float* wdw, const float* bdw, const float* mean, const float* rstd, const float* lnw, void* dX,
                 float* dlnw, float* dlnb, float* dwdw, float* dbdw, float* dgamma, int B, int L, int C,
                 PartialSink* sink) {
  const int TL = dw_tile_len(C);
  const int tiles_per_clip = (L + TL - 1) / TL, total = B * tiles_per_clip;
  const int nb = total < 256 ? total : 256;
  int pstride = 0;
  if (sink) {   // records [nb][lnw C | lnb C | taps 7C | bias C | gamma C]
    const int rec = 11 * C;
    size_t o = sink->take((size_t)nb * rec);
    sink->add(o, nb, rec, C, dlnw);
    sink->add(o + C, nb, rec, C, dlnb);
    sink->add(o + 2 * C, nb, rec, 7 * C, dwdw);
    sink->add(o + 9 * C, nb, rec, C, dbdw);
    if (dgamma) sink->add(o + 10 * C, nb, rec, C, dgamma);
    if (sink->dry) return;
    pstride = rec;
    dlnw = sink->ws + o; dlnb = dlnw + C; dwdw = dlnw + 2 * C; dbdw = dlnw + 9 * C;
    if (dgamma) dgamma = dlnw + 10 * C;
  }
  ProfScope ps(s, "dwblock_bwd", 42.0 * B * L * C, 5.0 * B * L * C * (dt == DT_F32 ? 4 : 2));
  if (dt == DT_F32)
    dwblock_bwd_t<float>(s, dLN, X, dY, Z, wdw, bdw, mean, rstd, lnw, dX, dlnw, dlnb, dwdw, dbdw, dgamma, B, L, C, nb,
                         tiles_per_clip, total, pstride);
  else
    dwblock_bwd_t<bf16>(s, dLN, X, dY, Z, wdw, bdw, mean, rstd, lnw, dX, dlnw, dlnb, dwdw, dbdw, dgamma, B, L, C, nb,
                        tiles_per_clip, total, pstride);
}


// ------------------------------------------------------------------ whole Block per position, C <= 8
// Stages 0 and 1 (C = 4, 8; 256 000 / 128 000 positions) have 4..16-wide 1x1 convs: an MFMA GEMM tile
// is almost all padding there (measured 20-25 us per launch).  One thread owns one position:
// depthwise conv + LayerNorm + pw1 + GELU + pw2 + layer scale + residual in registers, weights
// broadcast from LDS.  Every tensor the backward pass needs is written exactly as the unfused
// path writes it (LN output, u, h = gelu(u), z, y, mean, rstd), with the same roundings.
// SAVE = false (inference): only y (and the tail LayerNorm's rows) leave the kernel; the six saved tensors were 400 of the
// 517 MB a C = 4 launch moved at streaming-inference sizes (143 us per launch, 3.6 TB/s: HBM-bound by stores nobody reads).
// (`save` is a run-time flag, not a template parameter: two instantiations contracted their fp32 arithmetic differently
// and predict() stopped matching the training forward bit for bit.)
template <typename T, int C>
__global__ __launch_bounds__(256, 2) void block_small_fwd_kernel(
    const Res<T>* __restrict__ X, const float* __restrict__ wdw, const float* __restrict__ bdw,
    const float* __restrict__ lnw, const float* __restrict__ lnb, const T* __restrict__ w1, const float* __restrict__ b1,
    const T* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ gamma, T* __restrict__ LN,
    float* __restrict__ mean, float* __restrict__ rstd, T* __restrict__ U, T* __restrict__ Hh, T* __restrict__ Z,
    Res<T>* __restrict__ Y, int B, int L, TailLn tail, int save) {
  constexpr int H2 = 2 * C;
  __shared__ __attribute__((aligned(16))) float sw1[H2 * C], sw2[C * H2], sb1[H2], sb2[C], sg[C], sdw[7 * C], sbd[C], slw[C], slb[C];
  __shared__ __attribute__((aligned(16))) float stw[C], stb[C];
  if (tail.w != nullptr && threadIdx.x < C) { stw[threadIdx.x] = tail.w[threadIdx.x]; stb[threadIdx.x] = tail.b[threadIdx.x]; }
  {   // every array has at most 256 entries: one UNCONDITIONAL load per thread and array from a clamped index, all issued
      // back to back, then the LDS stores (guarded loads in loops compiled into one round trip after the other)
    static_assert(H2 * C <= 256 && 7 * C <= 256, "one staging element per thread");
    const int t = threadIdx.x;
    const float r_w1 = to_f(w1[min(t, H2 * C - 1)]), r_w2 = to_f(w2[min(t, H2 * C - 1)]);
    const float r_b1 = b1[min(t, H2 - 1)], r_dw = wdw[min(t, 7 * C - 1)];
    const int tc = min(t, C - 1);
    const float r_b2 = b2[tc], r_g = gamma[tc], r_bd = bdw[tc], r_lw = lnw[tc], r_lb = lnb[tc];
    if (t < H2 * C) { sw1[t] = r_w1; sw2[t] = r_w2; }
    if (t < H2) sb1[t] = r_b1;
    if (t < 7 * C) sdw[t] = r_dw;
    if (t < C) { sb2[t] = r_b2; sg[t] = r_g; sbd[t] = r_bd; slw[t] = r_lw; slb[t] = r_lb; }
  }
  __syncthreads();
  const long N = (long)B * L;
  for (long row = (long)blockIdx.x * 256 + threadIdx.x; row < N; row += (long)gridDim.x * 256) {
    const int b = (int)(row / L), l = (int)(row % L);
    const long rb = (long)b * L;
    // The weights are read from LDS where they are used (same address in every lane: a broadcast).  `wz` is an opaque
    // zero: without it the compiler keeps all 312 weights of C = 8 in registers across the loop (256 VGPRs + 188
    // AGPRs, one wave per SIMD, and the 500 workgroups of stage 1 then run in two rounds).
    int wz = 0;
    asm volatile("" : "+v"(wz));
    wz <<= 2;
    float x[7][C];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      int ll = l + k - 3;
      int lc = ll < 0 ? 0 : (ll >= L ? L - 1 : ll);
      loadvf<C>(X + (rb + lc) * C, x[k]);
    }
    float d[C];
#pragma unroll
    for (int c = 0; c < C; ++c) d[c] = sbd[wz + c];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      int ll = l + k - 3;
      float m = (ll >= 0 && ll < L) ? 1.f : 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) d[c] += sdw[wz + k * C + c] * (x[k][c] * m);
    }
    float mu = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) mu += d[c];
    mu *= (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { d[c] -= mu; q += d[c] * d[c]; }
    const float rs = __builtin_amdgcn_rsqf(q * (1.f / C) + kLnEps);
    float ln[C];
#pragma unroll
    // Everything between the dwconv and y stays in fp32 registers (round 4): this kernel multiplies on the vector unit, so
    // rounding ln / u / h to T "as the MFMA path would" only added error; what is STORED for the backward is rounded by its store
    for (int c = 0; c < C; ++c) ln[c] = d[c] * rs * slw[wz + c] + slb[wz + c];
    float u[H2], h[H2];
#pragma unroll
    for (int j = 0; j < H2; ++j) {
      float a = sb1[wz + j];
#pragma unroll
      for (int c = 0; c < C; ++c) a += sw1[wz + j * C + c] * ln[c];
      u[j] = a;
      h[j] = gelu_tanh(u[j]);
    }
    float z[C], y[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float a = sb2[wz + c];
#pragma unroll
      for (int j = 0; j < H2; ++j) a += sw2[wz + c * H2 + j] * h[j];
      z[c] = a;                                    // unrounded (chain.hip: block_fwd_tile)
      y[c] = x[3][c] + sg[wz + c] * z[c];
    }
    storevf<C>(Y + row * C, y);
    if (save) {
      storevf<C>(LN + row * C, ln);
      if (Z != nullptr) storevf<C>(Z + row * C, z);     // (the training path passes none: dgamma comes from the pw2 weight gradient, head.hip: cnn_gamma_finish)
      mean[row] = mu;
      rstd[row] = rs;
    }
    if (tail.w != nullptr) {   // LayerNorm of the stored y row (kernels.h: TailLn), the roundings of ln_fwd_kernel
      float t[C], tm, tr;
#pragma unroll
      for (int c = 0; c < C; ++c) t[c] = to_f(from_f<Res<T>>(y[c]));
      ln_stats<C, 1>(t, tm, tr);
#pragma unroll
      for (int c = 0; c < C; ++c) t[c] = ln_apply(t[c], tr, stw[wz + c], stb[wz + c]);
      storevf<C>(reinterpret_cast<T*>(tail.Y) + row * C, t);
      tail.mean[row] = tm;
      tail.rstd[row] = tr;
    }
    // (Hh: the training path passes none - block_small_bwd_full re-forms h = gelu(u) from u)
    if (!save) {
    } else if constexpr (H2 == 8) {
      storevf<8>(U + row * H2, u);
      if (Hh != nullptr) storevf<8>(Hh + row * H2, h);
    } else {
      float t0[8], t1[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { t0[i] = u[i]; t1[i] = u[8 + i]; }
      storevf<8>(U + row * H2, t0); storevf<8>(U + row * H2 + 8, t1);
      if (Hh != nullptr) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { t0[i] = h[i]; t1[i] = h[8 + i]; }
        storevf<8>(Hh + row * H2, t0); storevf<8>(Hh + row * H2 + 8, t1);
      }
    }
  }
}

// data-gradient chain of the two 1x1 convs per position (replaces the GELU' GEMM + the store GEMM):
//   du = (w2tg . dy) * gelu'(u)   (w2tg = pw2^T with gamma folded in),   dln = w1t . du
template <typename T, int C>
__global__ __launch_bounds__(256) void block_small_bwd_kernel(const T* __restrict__ dY, const T* __restrict__ U,
                                                              const T* __restrict__ w2tg, const T* __restrict__ w1t,
                                                              T* __restrict__ dU, T* __restrict__ dLN, long N) {
  constexpr int H2 = 2 * C;
  __shared__ float s2[H2 * C], s1[C * H2];
  for (int t = threadIdx.x; t < H2 * C; t += 256) { s2[t] = to_f(w2tg[t]); s1[t] = to_f(w1t[t]); }
  __syncthreads();
  for (long row = (long)blockIdx.x * 256 + threadIdx.x; row < N; row += (long)gridDim.x * 256) {
    float dy[C], u[H2], du[H2];
    loadvf<C>(dY + row * C, dy);
    if constexpr (H2 == 8) {
      loadvf<8>(U + row * H2, u);
    } else {
      float t0[8], t1[8];
      loadvf<8>(U + row * H2, t0); loadvf<8>(U + row * H2 + 8, t1);
#pragma unroll
      for (int i = 0; i < 8; ++i) { u[i] = t0[i]; u[8 + i] = t1[i]; }
    }
#pragma unroll
    for (int j = 0; j < H2; ++j) {
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) a += s2[j * C + c] * dy[c];
      du[j] = to_f(from_f<T>(a * gelu_tanh_grad(u[j])));
    }
    float dl[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < H2; ++j) a += s1[c * H2 + j] * du[j];
      dl[c] = a;
    }
    storevf<C>(dLN + row * C, dl);
    if constexpr (H2 == 8) {
      storevf<8>(dU + row * H2, du);
    } else {
      float t0[8], t1[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { t0[i] = du[i]; t1[i] = du[8 + i]; }
      storevf<8>(dU + row * H2, t0); storevf<8>(dU + row * H2 + 8, t1);
    }
  }
}

// ------------------------------------------------------------------ whole Block backward per position, C <= 8
// One launch per Block instead of three (block_small_bwd + dwblock_bwd + a grouped weight-gradient GEMM with
// 1000-2000 workgroups for 4..16-wide outputs: 54-56 us per Block at 8 clips).  One thread owns one position of
// a 256-row tile (250 interior rows + 3 halo rows each side):
//   phase 1  du = (pw2^T gamma . dy) gelu'(u), dln = pw1^T . du, LayerNorm', dD -> LDS; du, h, LN output and dy of
//            the interior rows are staged in LDS as [position][16] tiles
//   phase 2  dX = dY + dwconv^T(dD), depthwise tap / bias sums; the two 1x1-conv weight gradients are
//            [16 x 16] x 256-position MFMA products of the staged tiles read column-wise (a ones column in the
//            LN tile yields the pw1 bias gradient).  h = gelu(u) is re-formed from u (the forward stores neither h nor z:
//            round 4); the pw2 weight / bias gradients leave RAW (not yet scaled by the layer scale) and dgamma is formed
//            from them afterwards (head.hip: cnn_gamma_finish)
// Every parameter gradient leaves the kernel as one record per workgroup (reduce_partials sums them).
template <int C> struct SmallBwd {
  static constexpr int H2 = 2 * C, ROWS = 256, TL = ROWS - 6, FW = 16;
  // (no layer-scale record: dgamma is formed from the RAW pw2 weight / bias gradients by cnn_gamma_finish, head.hip)
  static constexpr int O_W1 = 0, O_B1 = H2 * C, O_W2 = O_B1 + H2, O_B2 = O_W2 + C * H2, O_LNW = O_B2 + C,
                       O_LNB = O_LNW + C, O_TAPS = O_LNB + C, O_DWB = O_TAPS + 7 * C, REC = O_DWB + C;
};

// ONE: the grid has one workgroup per tile (no tile loop): the per-thread gradient accumulators are then plain
// temporaries of the phase that produces them instead of values carried around a loop, which is what lets C = 8 fit
// two waves per SIMD (512 workgroups resident at once instead of two rounds of 256).
template <typename T, int C, bool ONE>
__global__ __launch_bounds__(256, ONE ? 2 : 1) void block_small_bwd_full_kernel(
    const T* __restrict__ dY, const Res<T>* __restrict__ X, const T* __restrict__ U,
    const T* __restrict__ LNs, const float* __restrict__ mean, const float* __restrict__ rstd,
    const T* __restrict__ w2tg, const T* __restrict__ w1t, const float* __restrict__ wdw, const float* __restrict__ bdw,
    const float* __restrict__ lnw, const float* __restrict__ gamma, T* __restrict__ dX, float* __restrict__ rec, int B,
    int L, int tiles_per_clip, int total_tiles) {
  using P = SmallBwd<C>;
  constexpr int H2 = P::H2, TL = P::TL, FW = P::FW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T (*du_t)[FW] = reinterpret_cast<T (*)[FW]>(smem_raw);
  T (*h_t)[FW] = du_t + 256;
  T (*ln_t)[FW] = h_t + 256;
  T (*dy_t)[FW] = ln_t + 256;
  T (*dDs)[C] = reinterpret_cast<T (*)[C]>(dy_t + 256);
  __shared__ __attribute__((aligned(16))) float s2[H2 * C], s1[C * H2], sdw[7 * C], sbd[C], slw[C], sg[C];
  {   // one unconditional load per thread and array, issued back to back (block_small_fwd_kernel)
    static_assert(H2 * C <= 256 && 7 * C <= 256, "one staging element per thread");
    const int t = threadIdx.x, tc = min(t, C - 1);
    const float r_2 = to_f(w2tg[min(t, H2 * C - 1)]), r_1 = to_f(w1t[min(t, H2 * C - 1)]), r_dw = wdw[min(t, 7 * C - 1)];
    const float r_bd = bdw[tc], r_lw = lnw[tc], r_g = gamma[tc];
    if (t < H2 * C) { s2[t] = r_2; s1[t] = r_1; }
    if (t < 7 * C) sdw[t] = r_dw;
    if (t < C) { sbd[t] = r_bd; slw[t] = r_lw; sg[t] = r_g; }
  }
  const int p = threadIdx.x, lane = p & 63, wave = p >> 6, r = lane & 15, g = lane >> 4;
  float alw[C], alb[C], ab[C], aw[7][C], adb2[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    alw[c] = 0.f; alb[c] = 0.f; ab[c] = 0.f; adb2[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) aw[k][c] = 0.f;
  }
  f32x4 aw1 = f32x4{0.f, 0.f, 0.f, 0.f}, aw2 = aw1;
  for (int t = blockIdx.x; t < total_tiles; t += ONE ? total_tiles : (int)gridDim.x) {
    const int b = t / tiles_per_clip, l0 = (t % tiles_per_clip) * TL;
    const long rb = (long)b * L;
    const int l = l0 - 3 + p;
    const bool valid = (l >= 0) && (l < L);
    const bool inner = valid && p >= 3 && p < TL + 3;
    const int lcl = l < 0 ? 0 : (l >= L ? L - 1 : l);
    const long row = rb + lcl;
    __syncthreads();   // weights staged / previous tile's tiles consumed
    int wz = 0;        // opaque zero: the weights are read from LDS at their uses, not kept in registers (block_small_fwd)
    asm volatile("" : "+v"(wz));
    wz <<= 2;
    float dy[C];
    {
      // Written in consumption order (what is loaded only to be staged or summed is finished first): the LDS stores keep
      // the compiler from moving the loads much, and the order the source gives decides the register peak.
      const float fin = inner ? 1.f : 0.f, fv = valid ? 1.f : 0.f;
      float t0[8], t1[8];
      loadvf<C>(dY + row * C, dy);
      float u[H2];       // u: GELU' below, and h = gelu(u) - an operand tile of the pw2 weight gradient - re-formed here
      if constexpr (H2 == 8) {
        loadvf<8>(U + row * H2, u);
      } else {
        loadvf<8>(U + row * H2, t0); loadvf<8>(U + row * H2 + 8, t1);
#pragma unroll
        for (int i = 0; i < 8; ++i) { u[i] = t0[i]; u[8 + i] = t1[i]; }
      }
      {   // (rounded to T exactly as block_small_fwd_kernel forms the h it multiplies with)
        float hh[H2];
#pragma unroll
        for (int j = 0; j < H2; ++j) hh[j] = to_f(from_f<T>(gelu_tanh(u[j])));
#pragma unroll
        for (int i = 0; i < 8; ++i) { t0[i] = i < H2 ? hh[i] : 0.f; t1[i] = (8 + i) < H2 ? hh[(8 + i) < H2 ? 8 + i : 0] : 0.f; }
        storevf<8>(&h_t[p][0], t0); storevf<8>(&h_t[p][8], t1);
      }
      {   // LN output: operand tile of the pw1 weight gradient (a ones column yields the pw1 bias gradient)
        float ln[C];
        loadvf<C>(LNs + row * C, ln);
#pragma unroll
        for (int i = 0; i < 8; ++i) { t0[i] = i < C ? ln[i < C ? i : 0] : (i == C ? 1.f : 0.f); t1[i] = (8 + i) == C ? 1.f : 0.f; }
        storevf<8>(&ln_t[p][0], t0); storevf<8>(&ln_t[p][8], t1);
      }
      {   // dy: pw2 bias gradient and operand tile
#pragma unroll
        for (int c = 0; c < C; ++c) adb2[c] += fin * dy[c];
#pragma unroll
        for (int i = 0; i < 8; ++i) { t0[i] = i < C ? fin * dy[i < C ? i : 0] : 0.f; t1[i] = 0.f; }
        storevf<8>(&dy_t[p][0], t0); storevf<8>(&dy_t[p][8], t1);
      }
      float dl[C];
      {
        float du[H2];
#pragma unroll
        for (int j = 0; j < H2; ++j) {
          float a = 0.f;
#pragma unroll
          for (int c = 0; c < C; ++c) a += s2[wz + j * C + c] * dy[c];
          du[j] = to_f(from_f<T>(a * gelu_tanh_grad(u[j])));
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float a = 0.f;
#pragma unroll
          for (int j = 0; j < H2; ++j) a += s1[wz + c * H2 + j] * du[j];
          dl[c] = to_f(from_f<T>(a));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { t0[i] = i < H2 ? fin * du[i] : 0.f; t1[i] = (8 + i) < H2 ? fin * du[(8 + i) < H2 ? 8 + i : 0] : 0.f; }
        storevf<8>(&du_t[p][0], t0); storevf<8>(&du_t[p][8], t1);
      }
      const float mu = mean[row], rs = rstd[row];
      float d[C];
#pragma unroll
      for (int c = 0; c < C; ++c) d[c] = sbd[wz + c];
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const int ll = lcl + k - 3;
        const int lc = ll < 0 ? 0 : (ll >= L ? L - 1 : ll);
        const float m = (ll >= 0 && ll < L) ? 1.f : 0.f;
        float x[C];
        loadvf<C>(X + (rb + lc) * C, x);
#pragma unroll
        for (int c = 0; c < C; ++c) d[c] += sdw[wz + k * C + c] * (x[c] * m);
      }
      float q1 = 0.f, q2 = 0.f, gy[C], o[C];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        d[c] = (d[c] - mu) * rs;
        gy[c] = dl[c] * slw[wz + c];
        q1 += gy[c];
        q2 += gy[c] * d[c];
        alw[c] += fin * dl[c] * d[c];
        alb[c] += fin * dl[c];
      }
      q1 *= (1.f / C);
      q2 *= (1.f / C);
#pragma unroll
      for (int c = 0; c < C; ++c) o[c] = fv * rs * (gy[c] - q1 - d[c] * q2);
      storevf<C>(&dDs[p][0], o);
    }
    __syncthreads();
    {
      const float fi = inner ? 1.f : 0.f;
      const int pc = p < 3 ? 3 : (p >= TL + 3 ? TL + 2 : p);     // clamp so that halo threads read inside the tile
      float o[C], dd[C];
      loadvf<C>(&dDs[pc][0], dd);
#pragma unroll
      for (int c = 0; c < C; ++c) { o[c] = dy[c]; dd[c] *= fi; ab[c] += dd[c]; }
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        float tq[C], x[C];
        loadvf<C>(&dDs[pc + 3 - k][0], tq);      // dD[l - k + 3] (zero outside the clip)
        const int lx = lcl + k - 3;
        const int lxc = lx < 0 ? 0 : (lx >= L ? L - 1 : lx);
        const float mx = (lx >= 0 && lx < L) ? 1.f : 0.f;
        loadvf<C>(X + (rb + lxc) * C, x);
#pragma unroll
        for (int c = 0; c < C; ++c) { o[c] += sdw[wz + k * C + c] * tq[c]; aw[k][c] += dd[c] * (x[c] * mx); }
      }
      if (inner) storevf<C>(dX + row * C, o);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int ks = 2 * wave + kk;
        mma(aw1, frag_col16(&du_t[32 * ks + 8 * g][0], FW, r), frag_col16(&ln_t[32 * ks + 8 * g][0], FW, r));
        mma(aw2, frag_col16(&h_t[32 * ks + 8 * g][0], FW, r), frag_col16(&dy_t[32 * ks + 8 * g][0], FW, r));
      }
    }
  }
  // ---- one record per workgroup
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem_raw);     // [4 waves][2][16][16] MFMA sums, then [4][11 C] row sums
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[wave * 512 + (4 * g + e) * 16 + r] = aw1[e];         // [j][n]: n < C pw1 weight, n == C pw1 bias
    red[wave * 512 + 256 + (4 * g + e) * 16 + r] = aw2[e];   // [j][c] = sum h[j] dy[c]
  }
  auto fold = [&](const float (&acc)[C], int slot) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float v = group_sum<64>(acc[c]);
      if (lane == 0) red[2048 + wave * 11 * C + slot * C + c] = v;
    }
  };
  fold(alw, 0);
  fold(alb, 1);
#pragma unroll
  for (int k = 0; k < 7; ++k) fold(aw[k], 2 + k);
  fold(ab, 9);
  fold(adb2, 10);
  __syncthreads();
  float* R = rec + (size_t)blockIdx.x * P::REC;
  {
    const int j = p >> 4, n = p & 15;
    const float v1 = red[p] + red[512 + p] + red[1024 + p] + red[1536 + p];
    const float v2 = red[256 + p] + red[768 + p] + red[1280 + p] + red[1792 + p];
    if (j < H2) {
      if (n < C) { R[P::O_W1 + j * C + n] = v1; R[P::O_W2 + n * H2 + j] = v2; }      // pw2: RAW (cnn_gamma_finish scales it)
      else if (n == C) R[P::O_B1 + j] = v1;
    }
  }
  if (p < 11 * C) {
    const float v = red[2048 + p] + red[2048 + 11 * C + p] + red[2048 + 22 * C + p] + red[2048 + 33 * C + p];
    if (p < 10 * C) R[P::O_LNW + p] = v;                       // lnw | lnb | taps 7C | dw bias
    else R[P::O_B2 + (p - 10 * C)] = v;                         // pw2 bias, RAW
  }
}

template <typename T, int C, bool ONE>
static void block_small_bwd_full_launch(hipStream_t s, const SmallBwdArgs& a, float* rec, int nb, int tiles_per_clip, int total) {
  const size_t lds = (size_t)(4 * 256 * 16 + 256 * C) * sizeof(T);
  if (lds > 48 * 1024) {
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(block_small_bwd_full_kernel<T, C, ONE>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024), true);
    (void)once;
  }
  hipLaunchKernelGGL((block_small_bwd_full_kernel<T, C, ONE>), dim3(nb), dim3(256), lds, s, (const T*)a.dY, (const Res<T>*)a.X,
                     (const T*)a.U, (const T*)a.LN, a.mean, a.rstd, (const T*)a.w2tg,
                     (const T*)a.w1t, a.wdw, a.bdw, a.lnw, a.gamma, (T*)a.dX, rec, a.B, a.L, tiles_per_clip, total);
}
template <typename T, int C>
static void block_small_bwd_full_t(hipStream_t s, const SmallBwdArgs& a, float* rec, int nb, int tiles_per_clip, int total) {
  if (nb == total) block_small_bwd_full_launch<T, C, true>(s, a, rec, nb, tiles_per_clip, total);
  else block_small_bwd_full_launch<T, C, false>(s, a, rec, nb, tiles_per_clip, total);
}

void block_small_bwd_full(hipStream_t s, DType dt, const SmallBwdArgs& a, PartialSink* sink) {
  const int C = a.C, H2 = 2 * C, TL = 250;
  const int tiles_per_clip = (a.L + TL - 1) / TL, total = a.B * tiles_per_clip;
  const int nb = total < 512 ? total : 512;
  const int REC = 4 * C * C + 13 * C;
  const size_t o = sink->take((size_t)nb * REC);
  size_t q = o;
  sink->add(q, nb, REC, H2 * C, a.dW1); q += H2 * C;
  sink->add(q, nb, REC, H2, a.db1); q += H2;
  sink->add(q, nb, REC, C * H2, a.dW2); q += C * H2;
  sink->add(q, nb, REC, C, a.db2); q += C;
  sink->add(q, nb, REC, C, a.dlnw); q += C;
  sink->add(q, nb, REC, C, a.dlnb); q += C;
  sink->add(q, nb, REC, 7 * C, a.dwdw); q += 7 * C;
  sink->add(q, nb, REC, C, a.dbdw);
  if (sink->dry) return;
  float* rec = sink->ws + o;
  const long N = (long)a.B * a.L;
  ProfScope ps(s, "block_small_bwd_full", 2.0 * N * (7 * C * 2 + 12 * C * C), 6.0 * N * C * (dt == DT_F32 ? 4 : 2));
  if (dt == DT_F32) {
    if (C == 4) block_small_bwd_full_t<float, 4>(s, a, rec, nb, tiles_per_clip, total);
    else block_small_bwd_full_t<float, 8>(s, a, rec, nb, tiles_per_clip, total);
  } else {
    if (C == 4) block_small_bwd_full_t<bf16, 4>(s, a, rec, nb, tiles_per_clip, total);
    else block_small_bwd_full_t<bf16, 8>(s, a, rec, nb, tiles_per_clip, total);
  }
}

void block_small_fwd(hipStream_t s, DType dt, const void* X, const float* wdw, const float* bdw, const float* lnw,
                     const float* lnb, const void* w1, const float* b1, const void* w2, const float* b2,
                     const float* gamma, void* LN, float* mean, float* rstd, void* U, void* Hh, void* Z, void* Y, int B,
                     int L, int C, const TailLn* tail) {
  const long N = (long)B * L;
  const int grid = (int)std::min<long>((N + 255) / 256, 2048);
  TailLn tl{};
  if (tail != nullptr) tl = *tail;
  const bool save = U != nullptr;      // inference passes no save tensors
  ProfScope ps(s, "block_small_fwd", 2.0 * N * (7 * C + 4 * C * C), (save ? 5.0 + (Hh ? 2.0 : 0.0) + (Z ? 1.0 : 0.0) : 2.0) * N * C * (dt == DT_F32 ? 4 : 2));
#define AMT_LAUNCH_SF(TT, CC_)                                                                                         \
  hipLaunchKernelGGL((block_small_fwd_kernel<TT, CC_>), dim3(grid), dim3(256), 0, s, (const Res<TT>*)X, wdw, bdw, lnw, lnb,  \
                     (const TT*)w1, b1, (const TT*)w2, b2, gamma, (TT*)LN, mean, rstd, (TT*)U, (TT*)Hh, (TT*)Z, (Res<TT>*)Y, \
                     B, L, tl, save ? 1 : 0)
  if (dt == DT_F32) { if (C == 4) AMT_LAUNCH_SF(float, 4); else AMT_LAUNCH_SF(float, 8); }
  else { if (C == 4) AMT_LAUNCH_SF(bf16, 4); else AMT_LAUNCH_SF(bf16, 8); }
#undef AMT_LAUNCH_SF
}

void block_small_bwd(hipStream_t s, DType dt, const void* dY, const void* U, const void* w2tg, const void* w1t, void* dU,
                     void* dLN, long N, int C) {
  const int grid = (int)std::min<long>((N + 255) / 256, 2048);
  ProfScope ps(s, "block_small_bwd", 8.0 * N * C * C, 6.0 * N * C * (dt == DT_F32 ? 4 : 2));
#define AMT_LAUNCH_SB(TT, CC_)                                                                                       \
  hipLaunchKernelGGL((block_small_bwd_kernel<TT, CC_>), dim3(grid), dim3(256), 0, s, (const TT*)dY, (const TT*)U,     \
                     (const TT*)w2tg, (const TT*)w1t, (TT*)dU, (TT*)dLN, N)
  if (dt == DT_F32) { if (C == 4) AMT_LAUNCH_SB(float, 4); else AMT_LAUNCH_SB(float, 8); }
  else { if (C == 4) AMT_LAUNCH_SB(bf16, 4); else AMT_LAUNCH_SB(bf16, 8); }
#undef AMT_LAUNCH_SB
}

template <typename T>
static void dwln_fwd_t(hipStream_t s, const void* X, const float* wdw, const float* bdw, const float* lnw,
                       const float* lnb, void* Y, float* mean, float* rstd, int B, int L, int C) {
  long N = (long)B * L;
  AMT_DISPATCH_C(C, hipLaunchKernelGGL((dwln_fwd_kernel<T, CC>), dim3(row_grid(N, RowCfg<CC>::RPB)), dim3(256), 0, s,
                                       (const Res<T>*)X, wdw, bdw, lnw, lnb, (T*)Y, mean, rstd, B, L));
}
template <typename T>
static void dwln_bwd_t(hipStream_t s, const void* dY, const void* X, const float* wdw, const float* bdw,
                       const float* mean, const float* rstd, const float* lnw, void* dD, float* dlnw, float* dlnb,
                       int B, int L, int C, int nb, int pstride) {
  AMT_DISPATCH_C(C, hipLaunchKernelGGL((dwln_bwd_kernel<T, CC>), dim3(nb), dim3(256), 0, s, (const T*)dY, (const Res<T>*)X,
                                       wdw, bdw, mean, rstd, lnw, (T*)dD, dlnw, dlnb, B, L, pstride));
}
template <typename T>
static void dwconv_bwd_t(hipStream_t s, const void* dD, const void* X, const float* wdw, const void* dRes, void* dX,
                         float* dwdw, float* dbdw, int B, int L, int C, int nb, int pstride) {
  AMT_DISPATCH_C(C, hipLaunchKernelGGL((dwconv_bwd_kernel<T, CC>), dim3(nb), dim3(256), 0, s, (const T*)dD,
                                       (const Res<T>*)X, wdw, (const T*)dRes, (T*)dX, dwdw, dbdw, B, L, pstride));
}
template <typename T>
static void scale_bwd_t(hipStream_t s, const void* dY, const void* Z, const float* gamma, void* dZ, float* dgamma,
                        int N, int C, int nb, int pstride) {
  AMT_DISPATCH_C(C, hipLaunchKernelGGL((scale_bwd_kernel<T, CC>), dim3(nb), dim3(256), 0, s, (const T*)dY, (const T*)Z,
                                       gamma, (T*)dZ, dgamma, (long)N, pstride));
}

void dwln_fwd(hipStream_t s, DType dt, const void* X, const float* wdw, const float* bdw, const float* lnw,
              const float* lnb, void* Y, float* mean, float* rstd, int B, int L, int C) {
  ProfScope ps(s, "dwln_fwd", 14.0 * B * L * C, 2.0 * B * L * C * (dt == DT_F32 ? 4 : 2));
  if (dt == DT_F32) dwln_fwd_t<float>(s, X, wdw, bdw, lnw, lnb, Y, mean, rstd, B, L, C);
  else dwln_fwd_t<bf16>(s, X, wdw, bdw, lnw, lnb, Y, mean, rstd, B, L, C);
}
void dwln_bwd(hipStream_t s, DType dt, const void* dY, const void* X, const float* wdw, const float* bdw,
              const float* mean, const float* rstd, const float* lnw, void* dD, float* dlnw, float* dlnb, int B, int L,
              int C, PartialSink* sink) {
  const int nb = bwd_blocks((long)B * L, C, sink);
  int pstride = 0;
  if (sink) {   // records [nb][lnw(C) | lnb(C)]
    size_t o = sink->take((size_t)nb * 2 * C);
    sink->add(o, nb, 2 * C, C, dlnw);
    sink->add(o + C, nb, 2 * C, C, dlnb);
    if (sink->dry) return;
    pstride = 2 * C;
    dlnw = sink->ws + o;
    dlnb = dlnw + C;
  }
  ProfScope ps(s, "dwln_bwd", 14.0 * B * L * C, 3.0 * B * L * C * (dt == DT_F32 ? 4 : 2));
  if (dt == DT_F32) dwln_bwd_t<float>(s, dY, X, wdw, bdw, mean, rstd, lnw, dD, dlnw, dlnb, B, L, C, nb, pstride);
  else dwln_bwd_t<bf16>(s, dY, X, wdw, bdw, mean, rstd, lnw, dD, dlnw, dlnb, B, L, C, nb, pstride);
}
void dwconv_bwd(hipStream_t s, DType dt, const void* dD, const void* X, const float* wdw, const void* dRes, void* dX,
                float* dwdw, float* dbdw, int B, int L, int C, PartialSink* sink) {
  const int nb = bwd_blocks((long)B * L, C, sink);
  int pstride = 0;
  if (sink) {   // records [nb][taps(7C) | bias(C)]
    size_t o = sink->take((size_t)nb * 8 * C);
    sink->add(o, nb, 8 * C, 7 * C, dwdw);
    sink->add(o + 7 * C, nb, 8 * C, C, dbdw);
    if (sink->dry) return;
    pstride = 8 * C;
    dwdw = sink->ws + o;
    dbdw = dwdw + 7 * C;
  }
  ProfScope ps(s, "dwconv_bwd", 28.0 * B * L * C, 4.0 * B * L * C * (dt == DT_F32 ? 4 : 2));
  if (dt == DT_F32) dwconv_bwd_t<float>(s, dD, X, wdw, dRes, dX, dwdw, dbdw, B, L, C, nb, pstride);
  else dwconv_bwd_t<bf16>(s, dD, X, wdw, dRes, dX, dwdw, dbdw, B, L, C, nb, pstride);
}
void scale_bwd(hipStream_t s, DType dt, const void* dY, const void* Z, const float* gamma, void* dZ, float* dgamma,
               int N, int C, PartialSink* sink) {
  const int nb = bwd_blocks(N, C, sink);
  int pstride = 0;
  if (sink) {
    size_t o = sink->take((size_t)nb * C);
    sink->add(o, nb, C, C, dgamma);
    if (sink->dry) return;
    pstride = C;
    dgamma = sink->ws + o;
  }
  ProfScope ps(s, "scale_bwd", 0, 3.0 * N * C * (dt == DT_F32 ? 4 : 2));
  if (dt == DT_F32) scale_bwd_t<float>(s, dY, Z, gamma, dZ, dgamma, N, C, nb, pstride);
  else scale_bwd_t<bf16>(s, dY, Z, gamma, dZ, dgamma, N, C, nb, pstride);
}

}  // namespace amt
