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

// Phase timestamps for the micro-benchmark (scratch/chain_bench.hip): compiled out of the library.
#ifdef AMT_CHAIN_TIMING
__device__ unsigned long long amt_stamps[64][256][8];   // [launch slot][workgroup][phase]
#define AMT_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 256) amt_stamps[a.dbg_slot & 63][blockIdx.x][k] = __builtin_readcyclecounter(); } while (0)
#define AMT_STAMP_RT(k) do { if (threadIdx.x == 0 && blockIdx.x < 256) amt_stamps[a.dbg_slot & 63][blockIdx.x][k] = wall_clock64(); } while (0)
extern "C" int amt_debug_read_stamps(unsigned long long* out) {   // [64][256][8]; slots 6, 7 = 100 MHz real-time clock
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(amt_stamps), sizeof(amt_stamps));
}
#else
#define AMT_STAMP(k) do { } while (0)
#define AMT_STAMP_RT(k) do { } while (0)
#endif

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
  using S1 = ChainSplit<H2, TT>;
  using S2 = ChainSplit<C, TT>;
  constexpr int KC1 = ChainChunk<T, S1::RB, C>::KC, KC2 = ChainChunk<T, S2::RB, H2>::KC;
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  T* Xs = reinterpret_cast<T*>(chain_smem);            // [TT][LDX]  LN output
  T* Rs = Xs + TT * LDX;                               // [TT][LDX]  block input (residual)
  T* Hs = Rs + TT * LDX;                               // [TT][LDH]  gelu(u)
  T* Us = Hs + TT * LDH;                               // [TT][LDH]  u (staged for a coalesced copy-out)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const long N = (long)a.B * a.L;
  const long n0 = (long)blockIdx.x * TT;
  const T* X = reinterpret_cast<const T*>(a.X);
  const T* W1 = reinterpret_cast<const T*>(a.W1);
  const T* W2 = reinterpret_cast<const T*>(a.W2);
  const int m1 = (wave % S1::WF) * S1::RB * 16, t1 = (wave / S1::WF) * S1::NT * 16;
  const int m2 = (wave % S2::WF) * S2::RB * 16, t2 = (wave / S2::WF) * S2::NT * 16;

  AMT_STAMP_RT(6);
  AMT_STAMP(0);
  // ---- what depends on nothing is requested first: the stage-1 weights (chunk 0), biases, gamma.
  // Their round trip overlaps the activation loads of the row phase.  The stage-2 weights follow
  // after the row phase (in flight during stage 1): holding both sets across the row phase pushed the
  // kernel past 256 registers = one wave per SIMD, and such launches paid ~6 us more between
  // dependent kernels in the training step (profiles/r01_notes.md).
  Frag<T> wa[KC1][S1::RB], wb[KC2][S2::RB];
  chain_load_w<T, S1::RB, KC1, C>(wa, W1, m1, 0, lane);
  float bb1[S1::RB][4], bb2[S2::RB][4], gm[S2::RB][4];
#pragma unroll
  for (int rb = 0; rb < S1::RB; ++rb)
#pragma unroll
    for (int i = 0; i < 4; ++i) bb1[rb][i] = a.b1[m1 + 16 * rb + 4 * g + i];
#pragma unroll
  for (int rb = 0; rb < S2::RB; ++rb)
#pragma unroll
    for (int i = 0; i < 4; ++i) { bb2[rb][i] = a.b2[m2 + 16 * rb + 4 * g + i]; gm[rb][i] = a.gamma[m2 + 16 * rb + 4 * g + i]; }

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
      const long rowbase = (long)b * a.L;
      float x[7][8], d[8];
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        int ll = l + k - 3;
        int lc = ll < 0 ? 0 : (ll >= a.L ? a.L - 1 : ll);
        load8f(X + (rowbase + lc) * C + sub * 8, x[k]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) d[i] = bv[i];
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        int ll = l + k - 3;
        float m = (ll >= 0 && ll < a.L) ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] += wk[k][i] * (x[k][i] * m);
      }
      store8f(Rs + p * LDX + sub * 8, x[3]);
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
#ifdef AMT_CHAIN_TIMING
      if (!(a.dbg_slot & 1024))
#endif
      if (valid) {
        store8f(reinterpret_cast<T*>(a.LN) + row * C + sub * 8, y);
        if (sub == 0) { a.mean[row] = mu; a.rstd[row] = rs; }
      }
    }
  }
  chain_load_w<T, S2::RB, KC2, H2>(wb, W2, m2, 0, lane);
  __syncthreads();
  AMT_STAMP(1);

  // ---- stage 1: u = LN . W1^T + b1 ; h = gelu(u)
  {
    f32x4 acc[S1::RB][S1::NT];
    chain_zero(acc);
    chain_stage<T, S1::RB, S1::NT, KC1, C>(acc, wa, W1, m1, Xs, LDX, t1, lane);
    AMT_STAMP(2);
#pragma unroll
    for (int rb = 0; rb < S1::RB; ++rb) {
      const int m = m1 + 16 * rb + 4 * g;
#pragma unroll
      for (int ti = 0; ti < S1::NT; ++ti) {
        const int p = t1 + 16 * ti + r;
        float u[4], h[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u[i] = to_f(from_f<T>(acc[rb][ti][i] + bb1[rb][i]));
          h[i] = gelu_tanh(u[i]);
        }
        store4f(Hs + p * LDH + m, h);
        store4f(Us + p * LDH + m, u);
      }
    }
  }
  __syncthreads();
  AMT_STAMP(3);
  // u and h leave as whole rows, 16 bytes per lane (the MFMA accumulator layout would give 8-byte
  // pieces scattered over 16 rows per store instruction)
#ifdef AMT_CHAIN_TIMING
  if (!(a.dbg_slot & 256))
#endif
  {
  chain_copy_out<T, TT, H2, LDH>(Us, reinterpret_cast<T*>(a.U), n0, N, tid);
  chain_copy_out<T, TT, H2, LDH>(Hs, reinterpret_cast<T*>(a.Hh), n0, N, tid);
  }

  // ---- stage 2: z = h . W2^T + b2 ; y = x + gamma * z
  {
    f32x4 acc[S2::RB][S2::NT];
    chain_zero(acc);
    chain_stage<T, S2::RB, S2::NT, KC2, H2>(acc, wb, W2, m2, Hs, LDH, t2, lane);
    AMT_STAMP(4);
#pragma unroll
    for (int rb = 0; rb < S2::RB; ++rb) {
      const int m = m2 + 16 * rb + 4 * g;
#pragma unroll
      for (int ti = 0; ti < S2::NT; ++ti) {
        const int p = t2 + 16 * ti + r;
        float x[4], z[4], y[4];
        load4f(Rs + p * LDX + m, x);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          z[i] = to_f(from_f<T>(acc[rb][ti][i] + bb2[rb][i]));
          y[i] = x[i] + gm[rb][i] * z[i];
        }
        store4f(Xs + p * LDX + m, z);     // the LN panel is dead: z and y are staged in place
        store4f(Rs + p * LDX + m, y);
      }
    }
  }
  __syncthreads();
#ifdef AMT_CHAIN_TIMING
  if (!(a.dbg_slot & 512))
#endif
  chain_copy_out<T, TT, C, LDX>(Xs, reinterpret_cast<T*>(a.Z), n0, N, tid);
  chain_copy_out<T, TT, C, LDX>(Rs, reinterpret_cast<T*>(a.Y), n0, N, tid);
  AMT_STAMP(5);
  AMT_STAMP_RT(7);
}

template <typename T, int C>
static void block_fwd_fused_launch(hipStream_t s, const BlockFwdArgs& a) {
  constexpr int TT = BlockTile<C>::TT;
  constexpr size_t lds = sizeof(T) * (size_t)TT * (2 * (C + 8) + 2 * (2 * C + 8));
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool once = (lds > 64 * 1024
                          ? (void)hipFuncSetAttribute(reinterpret_cast<const void*>(block_fwd_fused_kernel<T, C, TT>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                          : (void)0,
                      true);
  (void)once;
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
