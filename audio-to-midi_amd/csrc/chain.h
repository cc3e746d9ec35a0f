// Building blocks of the token-tile CHAIN kernels (chain.hip).
//
// At 8 clips per GPU every dependent launch of the step sits on a ~5-8 us floor (launch + one
// memory round trip + drain; profiles/r01_notes.md), so the per-token operator chains of the model
// (Block: dwconv -> LN -> 1x1 -> GELU -> 1x1 -> gamma -> +x;  FFN: LN -> Linear -> GLU -> Linear ->
// dropout -> +x; ...) are evaluated by ONE workgroup per tile of TT tokens: the activations of the
// tile stay in LDS between the stages, the weights are streamed from L2 straight into MFMA A
// fragments (every weight element is used once per workgroup, so staging it in LDS buys nothing),
// and only the tensors the backward pass needs are written to HBM (fire-and-forget stores).
//
// Orientation is the one of gemm.hip: FEATURES on the MFMA row axis, TOKENS on the column axis;
// acc[rb][ti][i] = D[feature m0 + 16 rb + 4 g + i][token t0 + 16 ti + r],  r = lane & 15, g = lane >> 4.
#pragma once
#include "common.h"

namespace amt {

// acc[rb][ti] += W[m0 + 16 rb + .][0..K) . Xs[t0 + 16 ti + .][0..K)^T
//   W : row-major [..][ldw] in global memory (T compute copy), rows m0 + 16 rb + r
//   Xs: LDS panel [..][ldx] (T), rows t0 + 16 ti + r; columns >= K are never read as data
// K % 8 == 0.  The accumulation order over K equals gemm_nt's (one 32-deep step after the other into
// a single accumulator), so a chain kernel reproduces the unfused path bit for bit.
template <typename T, int RB, int NT, int K>
AMT_DEV void chain_gemm(f32x4 (&acc)[RB][NT], const T* __restrict__ W, int ldw, int m0, const T* Xs, int ldx, int t0,
                        int lane) {
  const int r = lane & 15, g = lane >> 4;
  constexpr int KS = (K + 31) / 32;
  const T* wp = W + (size_t)(m0 + r) * ldw + 8 * g;
  const T* xp = Xs + (t0 + r) * ldx + 8 * g;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    Frag<T> wf[RB], xf[NT];
    const bool ok = (K % 32 == 0) || (32 * ks + 8 * g < K);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) wf[rb] = ok ? frag_row(wp + (size_t)rb * 16 * ldw + 32 * ks) : frag_zero<T>();
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) xf[ti] = ok ? frag_row(xp + ti * 16 * ldx + 32 * ks) : frag_zero<T>();
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) mma(acc[rb][ti], wf[rb], xf[ti]);
  }
}

template <int RB, int NT> AMT_DEV void chain_zero(f32x4 (&acc)[RB][NT]) {
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) acc[rb][ti] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// How the 4 waves of a workgroup share an [M features] x [TT tokens] stage:
// WF waves along the features (each RB consecutive 16-row blocks), WT = 4 / WF along the tokens.
template <int M, int TT> struct ChainSplit {
  static constexpr int RBT = M / 16;
  static constexpr int WF = RBT >= 4 ? 4 : RBT;
  static constexpr int WT = 4 / WF;
  static constexpr int RB = RBT / WF;
  static constexpr int NT = (TT / 16) / WT;
  static_assert(M % 16 == 0 && TT % 16 == 0, "tile shape");
  static_assert(RBT % WF == 0 && (TT / 16) % WT == 0 && NT >= 1, "stage does not split over 4 waves");
};

}  // namespace amt
