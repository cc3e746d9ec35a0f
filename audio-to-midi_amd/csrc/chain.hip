// Token-tile chain kernels: one launch per operator chain (see chain.h for the rationale).
//
//   block_fwd_fused   ConvNeXt Block forward, C >= 16                       model.py:160-167
//                     dwconv k7 -> LN -> 1x1 (C -> 2C) -> GELU -> 1x1 (2C -> C) -> gamma -> + x
//
// Every tensor the backward pass reads is written exactly as the unfused launch sequence
// (dwln_fwd, gemm_nt<gelu>, gemm_nt<scale_resid>) writes it, with the same roundings and the same
// accumulation order, so both paths agree bit for bit (tests/test_model_gpu.py).
#include "chain.h"
#include "kernels.h"
#include "prof.h"
#include "rowops.h"

namespace amt {

namespace {

// depthwise conv k7 (SAME) of row (b, l), feature chunk `sub` (8 features), taps loaded from clamped
// rows and masked afterwards (see cnn.hip: dwconv_row)
template <typename T>
AMT_DEV void dw7_row(const T* __restrict__ X, long rowbase, int l, int L, int C, int sub, const float (&wk)[7][8],
                     const float (&bv)[8], float (&out)[8]) {
  float x[7][8];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    int ll = l + k - 3;
    int lc = ll < 0 ? 0 : (ll >= L ? L - 1 : ll);
    load8f(X + (rowbase + lc) * C + sub * 8, x[k]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = bv[i];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    int ll = l + k - 3;
    float m = (ll >= 0 && ll < L) ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] += wk[k][i] * (x[k][i] * m);
  }
}

}  // namespace

// ------------------------------------------------------------------ Block forward
template <int C> struct BlockTile {   // tokens per workgroup (bf16); chosen for ~250 workgroups at cfg2
  static constexpr int TT = C <= 16 ? 128 : (C <= 32 ? 64 : (C <= 64 ? 32 : 16));
};

template <typename T, int C, int TT>
__global__ __launch_bounds__(256) void block_fwd_fused_kernel(BlockFwdArgs a) {
  constexpr int H2 = 2 * C, LDX = C + 8, LDH = H2 + 8;
  constexpr int LPR = C / 8, RPP = 256 / LPR;   // lanes per row, rows per pass of the row phase
  static_assert(TT % RPP == 0, "row phase covers the tile in whole passes");
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  T* Xs = reinterpret_cast<T*>(chain_smem);            // [TT][LDX]  LN output
  T* Hs = Xs + TT * LDX;                               // [TT][LDH]  gelu(u)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const long N = (long)a.B * a.L;
  const long n0 = (long)blockIdx.x * TT;
  const T* X = reinterpret_cast<const T*>(a.X);

  // ---- row phase: depthwise conv + LayerNorm (fp32 statistics), LN output -> LDS + HBM
  {
    const int sub = tid % LPR, prow = tid / LPR;
    float wk[7][8], bv[8], wv[8], lb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      bv[i] = a.bdw[sub * 8 + i]; wv[i] = a.lnw[sub * 8 + i]; lb[i] = a.lnb[sub * 8 + i];
#pragma unroll
      for (int k = 0; k < 7; ++k) wk[k][i] = a.wdw[k * C + sub * 8 + i];
    }
#pragma unroll
    for (int pp = 0; pp < TT / RPP; ++pp) {
      const int p = pp * RPP + prow;
      const long row = n0 + p;
      const bool valid = row < N;
      const long rc = valid ? row : N - 1;
      const int b = (int)(rc / a.L), l = (int)(rc % a.L);
      float d[8];
      dw7_row<T>(X, (long)b * a.L, l, a.L, C, sub, wk, bv, d);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += d[i];
      const float mu = group_sum<LPR>(s) * (1.f / C);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { d[i] -= mu; q += d[i] * d[i]; }
      const float rs = 1.f / sqrtf(group_sum<LPR>(q) * (1.f / C) + kLnEps);
      float y[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = d[i] * rs * wv[i] + lb[i];
      store8f(Xs + p * LDX + sub * 8, y);
      if (valid) {
        store8f(reinterpret_cast<T*>(a.LN) + row * C + sub * 8, y);
        if (sub == 0) { a.mean[row] = mu; a.rstd[row] = rs; }
      }
    }
  }
  __syncthreads();

  // ---- stage 1: u = LN . W1^T + b1 ; h = gelu(u)
  {
    using S = ChainSplit<H2, TT>;
    const int wf = wave % S::WF, wt = wave / S::WF;
    const int m0 = wf * S::RB * 16, t0 = wt * S::NT * 16;
    f32x4 acc[S::RB][S::NT];
    chain_zero(acc);
    chain_gemm<T, S::RB, S::NT, C>(acc, reinterpret_cast<const T*>(a.W1), C, m0, Xs, LDX, t0, lane);
#pragma unroll
    for (int rb = 0; rb < S::RB; ++rb) {
      const int m = m0 + 16 * rb + 4 * g;
      float bb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) bb[i] = a.b1[m + i];
#pragma unroll
      for (int ti = 0; ti < S::NT; ++ti) {
        const int p = t0 + 16 * ti + r;
        const long row = n0 + p;
        float u[4], h[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u[i] = to_f(from_f<T>(acc[rb][ti][i] + bb[i]));
          h[i] = gelu_tanh(u[i]);
        }
        store4f(Hs + p * LDH + m, h);
        if (row < N) {
          store4f(reinterpret_cast<T*>(a.U) + row * H2 + m, u);
          store4f(reinterpret_cast<T*>(a.Hh) + row * H2 + m, h);
        }
      }
    }
  }
  __syncthreads();

  // ---- stage 2: z = h . W2^T + b2 ; y = x + gamma * z
  {
    using S = ChainSplit<C, TT>;
    const int wf = wave % S::WF, wt = wave / S::WF;
    const int m0 = wf * S::RB * 16, t0 = wt * S::NT * 16;
    f32x4 acc[S::RB][S::NT];
    chain_zero(acc);
    chain_gemm<T, S::RB, S::NT, H2>(acc, reinterpret_cast<const T*>(a.W2), H2, m0, Hs, LDH, t0, lane);
#pragma unroll
    for (int rb = 0; rb < S::RB; ++rb) {
      const int m = m0 + 16 * rb + 4 * g;
      float bb[4], gm[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { bb[i] = a.b2[m + i]; gm[i] = a.gamma[m + i]; }
#pragma unroll
      for (int ti = 0; ti < S::NT; ++ti) {
        const long row = n0 + t0 + 16 * ti + r;
        if (row < N) {
          float x[4], z[4], y[4];
          load4f(X + row * C + m, x);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            z[i] = to_f(from_f<T>(acc[rb][ti][i] + bb[i]));
            y[i] = x[i] + gm[i] * z[i];
          }
          store4f(reinterpret_cast<T*>(a.Z) + row * C + m, z);
          store4f(reinterpret_cast<T*>(a.Y) + row * C + m, y);
        }
      }
    }
  }
}

template <typename T, int C>
static void block_fwd_fused_launch(hipStream_t s, const BlockFwdArgs& a) {
  constexpr int TT = BlockTile<C>::TT;
  constexpr size_t lds = sizeof(T) * (size_t)TT * ((C + 8) + (2 * C + 8));
  static_assert(lds <= 64 * 1024, "LDS budget");
  const long N = (long)a.B * a.L;
  const int grid = (int)((N + TT - 1) / TT);
  hipLaunchKernelGGL((block_fwd_fused_kernel<T, C, TT>), dim3(grid), dim3(256), lds, s, a);
}

void block_fwd_fused(hipStream_t s, DType dt, const BlockFwdArgs& a) {
  const double N = (double)a.B * a.L, es = dt == DT_F32 ? 4 : 2;
  ProfScope ps(s, "block_fwd_fused", 2.0 * N * (7.0 * a.C + 4.0 * a.C * a.C), (8.0 * N * a.C + 4.0 * a.C * a.C) * es);
#define AMT_BF_CASE(CC_)                                                  \
  case CC_:                                                               \
    if (dt == DT_F32) block_fwd_fused_launch<float, CC_>(s, a);           \
    else block_fwd_fused_launch<bf16, CC_>(s, a);                         \
    break
  switch (a.C) {
    AMT_BF_CASE(16);
    AMT_BF_CASE(32);
    AMT_BF_CASE(64);
    AMT_BF_CASE(128);
    AMT_BF_CASE(256);
    default: break;
  }
#undef AMT_BF_CASE
}

}  // namespace amt
