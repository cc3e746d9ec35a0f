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
#include "rowops.h"

namespace amt {

// One stage:  acc[rb][ti] += W[m0 + 16 rb + .][0..K) . Xs[t0 + 16 ti + .][0..K)^T
//   W : fragment-packed T compute copy in global memory (rows m0 + 16 rb + r)
//   Xs: LDS panel [..][ldx] (T), rows t0 + 16 ti + r; columns >= K are never read as data
// K % 8 == 0.  The K axis is cut into chunks of KC 32-deep steps.  The caller loads chunk 0 with
// chain_load_w as EARLY as it can (the weights depend on nothing, so their L2/HBM round trip overlaps
// the activation loads of the row phase); chain_stage prefetches chunk j + 1 into the other register
// set while chunk j is multiplied.  The accumulation order over K equals gemm_nt's (one 32-deep step
// after the other into a single accumulator), so a chain kernel reproduces the unfused path bit
// for bit.
// W is the fragment-PACKED copy (kernels.h: chain_packed_elems): [M/16][KS][64 lanes][8]; m0 % 16 == 0.
template <typename T, int RB, int KC, int K>
AMT_DEV void chain_load_w(Frag<T> (&wf)[KC][RB], const T* __restrict__ W, int m0, int chunk, int lane) {
  constexpr int KS = (K + 31) / 32;
  const T* wp = W + ((size_t)(m0 >> 4) * KS) * 512 + lane * 8;
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const int ks = chunk * KC + kc;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
      wf[kc][rb] = ks < KS ? frag_row(wp + ((size_t)rb * KS + ks) * 512) : frag_zero<T>();
  }
}

template <typename T, int RB, int NT, int KC, int K>
AMT_DEV void chain_mma_chunk(f32x4 (&acc)[RB][NT], const Frag<T> (&wf)[KC][RB], const T* Xs, int ldx, int t0, int chunk,
                             int lane) {
  const int r = lane & 15, g = lane >> 4;
  constexpr int KS = (K + 31) / 32;
  const T* xp = Xs + (t0 + r) * ldx + 8 * g;
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const int ks = chunk * KC + kc;
    if (ks < KS) {
      const bool ok = (K % 32 == 0) || (32 * ks + 8 * g < K);
      Frag<T> xf[NT];
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) xf[ti] = ok ? frag_row(xp + ti * 16 * ldx + 32 * ks) : frag_zero<T>();
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) mma(acc[rb][ti], wf[kc][rb], xf[ti]);
    }
  }
}

template <typename T, int RB, int NT, int KC, int K>
AMT_DEV void chain_stage(f32x4 (&acc)[RB][NT], Frag<T> (&w0)[KC][RB], const T* __restrict__ W, int m0, const T* Xs,
                         int ldx, int t0, int lane) {
  constexpr int KS = (K + 31) / 32, NCH = (KS + KC - 1) / KC;
  Frag<T> w1[KC][RB];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (ch + 1 < NCH) {
      if (ch & 1) chain_load_w<T, RB, KC, K>(w0, W, m0, ch + 1, lane);
      else chain_load_w<T, RB, KC, K>(w1, W, m0, ch + 1, lane);
    }
    if (ch & 1) chain_mma_chunk<T, RB, NT, KC, K>(acc, w1, Xs, ldx, t0, ch, lane);
    else chain_mma_chunk<T, RB, NT, KC, K>(acc, w0, Xs, ldx, t0, ch, lane);
  }
}

// chunk depth: keep at most ~16 (bf16) / ~8 (f32) fragments of a stage in flight per register set
template <typename T, int RB, int K, int BF16_BUDGET = 16> struct ChainChunk {
  static constexpr int KS = (K + 31) / 32;
  static constexpr int BUDGET = sizeof(T) == 2 ? BF16_BUDGET : 8;
  static constexpr int KC = (BUDGET / RB) < 1 ? 1 : ((BUDGET / RB) > KS ? KS : (BUDGET / RB));
};

// LDS panel [TT][ld] -> global rows n0 .. n0 + TT (row-major, W features), 16 bytes (bf16: 8
// elements) per lane, consecutive lanes on consecutive chunks of a row: whole-line stores.
template <typename T, int TT, int W, int LD>
AMT_DEV void chain_copy_out(const T* Ps, T* __restrict__ G, long n0, long N, int tid) {
  constexpr int CPR = W / 8;                 // chunks per row
  constexpr int TOTAL = TT * CPR;
#pragma unroll
  for (int i = 0; i < (TOTAL + 255) / 256; ++i) {
    const int id = tid + i * 256;
    const int p = id / CPR, c = id % CPR;
    if ((TOTAL % 256 == 0 || id < TOTAL) && n0 + p < N)
      chunk_store(G + (n0 + p) * W + 8 * c, chunk_load(Ps + p * LD + 8 * c));
  }
}

// Column sums of per-lane accumulators: v[j], j = 8 * slot + i, of the lane that owns feature chunk
// sub = lane % LPR (features 8 sub + i) are summed over every lane of the WAVE with the same sub and
// stored to part[slot * C + 8 sub + i] (the wave's own LDS slice of NV / 8 * C floats: plain stores, the
// caller sums the four slices after a barrier).  LDS float atomics were the bound of the previous
// form: ~3.6 cycles per lane-add, 20 k cycles per workgroup at C = 128.
// Strides below 16 lanes are DPP adds; the 16- and 32-lane strides use the two-register permlane swaps
// so that ONE swap reduces two accumulators and the 4 (LPR <= 16) or 2 (LPR = 32) row groups of a wave
// end up owning the totals of 4 / 2 different accumulators.
template <int LPR, int NV>
AMT_DEV void fold_columns(float (&v)[NV], float* part, int C, int lane) {
  static_assert(NV % 4 == 0, "accumulators are folded four at a time");
  const int sub = lane % LPR;
  if constexpr (LPR < 16) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      float x = v[j];
      if constexpr (LPR <= 1) x += dpp_mov<DPP_XOR1>(x);
      if constexpr (LPR <= 2) x += dpp_mov<DPP_XOR2>(x);
      if constexpr (LPR <= 4) x += dpp_mov<DPP_ROR4>(x);
      if constexpr (LPR <= 8) x += dpp_mov<DPP_ROR8>(x);
      v[j] = x;
    }
  }
  if constexpr (LPR <= 16) {
    float t[NV / 4];
#pragma unroll
    for (int q = 0; q < NV / 4; ++q) {
      float a = v[4 * q], b = v[4 * q + 1], c = v[4 * q + 2], d = v[4 * q + 3];
      swap16(a, b);
      float sab = a + b;          // rows: a01 b01 a23 b23
      swap16(c, d);
      float scd = c + d;          // rows: c01 d01 c23 d23
      swap32(sab, scd);
      t[q] = sab + scd;           // rows: a b c d
    }
    if ((lane & 15) < LPR) {
      const int rho = lane >> 4;
#pragma unroll
      for (int q = 0; q < NV / 4; ++q) {
        const int j = 4 * q + rho;
        part[(j >> 3) * C + sub * 8 + (j & 7)] = t[q];
      }
    }
  } else {
    const int eta = lane >> 5;
#pragma unroll
    for (int q = 0; q < NV / 2; ++q) {
      float a = v[2 * q], b = v[2 * q + 1];
      swap32(a, b);
      const int j = 2 * q + eta;
      part[(j >> 3) * C + sub * 8 + (j & 7)] = a + b;
    }
  }
}

// global rows n0 .. n0 + TT (row-major, W features) -> LDS panel [TT][ld]; rows >= N are zero
template <typename T, int TT, int W, int LD>
AMT_DEV void chain_copy_in(T* Ps, const T* __restrict__ G, long n0, long N, int tid) {
  constexpr int CPR = W / 8;
  constexpr int TOTAL = TT * CPR;
#pragma unroll
  for (int i = 0; i < (TOTAL + 255) / 256; ++i) {
    const int id = tid + i * 256;
    const int p = id / CPR, c = id % CPR;
    if (TOTAL % 256 == 0 || id < TOTAL) {
      const long row = n0 + p;
      Chunk8<T> v = chunk_load(G + (row < N ? row : N - 1) * W + 8 * c);
      if (row >= N) v = chunk_zero<T>();
      chunk_store(Ps + p * LD + 8 * c, v);
    }
  }
}

// chain_copy_in in two halves: the loads are issued (into registers) where the launch starts, the LDS stores happen where
// the panel is first needed - a panel that only an EPILOGUE reads is then on its way while the first products run.  At the
// start of a launch every workgroup of the chip waits for its first bytes at once: that round trip is the longest single
// wait of a tile (profiles/r02_notes.md), so only what the first stage multiplies should be waited for in front of it.
template <typename T, int TT, int W> struct ChainCopyRegs { static constexpr int NR = (TT * (W / 8) + 255) / 256; Chunk8<T> v[NR]; };
template <typename T, int TT, int W>
AMT_DEV void chain_copy_load(ChainCopyRegs<T, TT, W>& regs, const T* __restrict__ G, int ldg, long n0, long N, int tid) {
  constexpr int CPR = W / 8, TOTAL = TT * CPR;
#pragma unroll
  for (int i = 0; i < ChainCopyRegs<T, TT, W>::NR; ++i) {
    const int id = tid + i * 256;
    const int p = id / CPR, c = id % CPR;
    if (TOTAL % 256 == 0 || id < TOTAL) {
      const long row = n0 + p;
      regs.v[i] = chunk_load(G + (row < N ? row : N - 1) * ldg + 8 * c);
      if (row >= N) regs.v[i] = chunk_zero<T>();
    }
  }
}
template <typename T, int TT, int W, int LD>
AMT_DEV void chain_copy_store(T* Ps, const ChainCopyRegs<T, TT, W>& regs, int tid) {
  constexpr int CPR = W / 8, TOTAL = TT * CPR;
#pragma unroll
  for (int i = 0; i < ChainCopyRegs<T, TT, W>::NR; ++i) {
    const int id = tid + i * 256;
    if (TOTAL % 256 == 0 || id < TOTAL) chunk_store(Ps + (id / CPR) * LD + 8 * (id % CPR), regs.v[i]);
  }
}

// Row LayerNorm over C = 8 * LPR features of the rows of an LDS panel (fp32 statistics, eqx LayerNorm):
// Ys = LN(Xs) (may alias Xs row-for-row), mean / rstd to global for rows < N.
// The passes of a tile are independent, but Ys may alias Xs, so written one pass after the other the compiler keeps every
// pass's LDS load behind the previous pass's store: one LDS round trip plus a ~120-instruction dependent chain (two
// lane-group reductions) per pass, in series, with one wave per SIMD and nothing to fill the gaps (in-kernel stamps:
// 850 cycles per 8-row pass, profiles/r03_notes.md section 7).  Here up to LN_GROUP passes are loaded first, evaluated
// side by side (independent chains the scheduler interleaves) and stored together; every row's arithmetic is unchanged.
// TI / TO: element types of the input panel and of the output (panel or global rows): the rows a LayerNorm reads are
// residual-stream rows (Res<T>, common.h), what it writes is T - or Res<T> again for the final CNN norm.  TI and TO have
// the same size, so a panel is normalised in place whatever the two types are.
constexpr int LN_GROUP = 4;
template <typename TI, typename TO, int TT, int C, int LD, bool TO_GLOBAL>
AMT_DEV void chain_ln_rows_impl(const TI* Xs, TO* Ys, TO* __restrict__ G, const float* __restrict__ w, const float* __restrict__ b,
                                float* __restrict__ mean, float* __restrict__ rstd, long n0, long N, int tid) {
  static_assert(sizeof(TI) == sizeof(TO), "in-place panels");
  constexpr int LPR = C / 8, RPP = 256 / LPR, NP = TT / RPP, GR = NP < LN_GROUP ? NP : LN_GROUP;
  static_assert(TT % RPP == 0, "row phase covers the tile in whole passes");
  static_assert(NP % GR == 0, "passes come in whole groups");
  const int sub = tid % LPR, prow = tid / LPR;
  float wv[8], bv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { wv[i] = w[sub * 8 + i]; bv[i] = b[sub * 8 + i]; }
#pragma unroll
  for (int p0 = 0; p0 < NP; p0 += GR) {
    float x[GR][8], mu[GR], rs[GR];
#pragma unroll
    for (int j = 0; j < GR; ++j) load8f(Xs + ((p0 + j) * RPP + prow) * LD + sub * 8, x[j]);
#pragma unroll
    for (int j = 0; j < GR; ++j) ln_stats<8, LPR>(x[j], mu[j], rs[j]);
#pragma unroll
    for (int j = 0; j < GR; ++j) {
      const int p = (p0 + j) * RPP + prow;
      float y[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = ln_apply(x[j][i], rs[j], wv[i], bv[i]);
      if constexpr (TO_GLOBAL) {
        if (n0 + p < N) {
          store8f(G + (n0 + p) * C + sub * 8, y);
          if (sub == 0) { mean[n0 + p] = mu[j]; rstd[n0 + p] = rs[j]; }
        }
      } else {
        store8f(Ys + p * LD + sub * 8, y);
        if (sub == 0 && n0 + p < N) { mean[n0 + p] = mu[j]; rstd[n0 + p] = rs[j]; }
      }
    }
  }
}
template <typename T, int TT, int C, int LD>
AMT_DEV void chain_ln_rows(const Res<T>* Xs, T* Ys, const float* __restrict__ w, const float* __restrict__ b,
                           float* __restrict__ mean, float* __restrict__ rstd, long n0, long N, int tid) {
  chain_ln_rows_impl<Res<T>, T, TT, C, LD, false>(Xs, Ys, nullptr, w, b, mean, rstd, n0, N, tid);
}

// The same, output rows to global memory (row-major, C features; rows >= N are not written): the LayerNorm of rows a
// chain holds in LDS, for a consumer that is another launch (kernels.h: TailLn).
template <typename T, typename TO, int TT, int C, int LD>
AMT_DEV void chain_ln_rows_out(const Res<T>* Xs, TO* __restrict__ G, const float* __restrict__ w, const float* __restrict__ b,
                               float* __restrict__ mean, float* __restrict__ rstd, long n0, long N, int tid) {
  chain_ln_rows_impl<Res<T>, TO, TT, C, LD, true>(Xs, nullptr, G, w, b, mean, rstd, n0, N, tid);
}

// Cooperative L2 warm-up of a weight range.  Every workgroup of a chain kernel streams the SAME
// weights, and the workgroups of one XCD (blockIdx % 8) run in lockstep, so a cold weight stream
// advances at the miss rate of ONE compute unit's request window (~25 GB/s measured) while the other
// CUs of the XCD wait for the same lines.  Here workgroup q = blockIdx / 8 of its XCD touches slice q of
// the range (16 B per lane, results only folded into `sink`), so that the whole range is requested from
// HBM at once by `nslices` CUs and the later fragment loads hit the XCD's L2.
// Issue it FIRST: vmcnt completes in order, so the first wait for real data covers these loads too.
template <int MAXLD>
AMT_DEV void chain_l2_warm(const void* base, size_t bytes, int q, int nslices, int tid, uint32_t& sink) {
  if (q >= nslices) return;
  const size_t per = ((bytes + nslices - 1) / nslices + 4095) & ~(size_t)4095;   // whole 4 KiB rounds
  const size_t beg = (size_t)q * per;
  const char* p = static_cast<const char*>(base) + beg + (size_t)tid * 16;
#pragma unroll
  for (int i = 0; i < MAXLD; ++i) {
    const size_t off = (size_t)i * 4096;
    if (off < per && beg + off + (size_t)tid * 16 + 16 <= bytes) {
      const uint4 v = *reinterpret_cast<const uint4*>(p + off);
      sink ^= v.x ^ v.y ^ v.z ^ v.w;
    }
  }
}

// The same warm-up, issued back to back: UNCONDITIONAL loads (addresses clamped into the range: a repeated line costs
// nothing) into registers, consumed by `fold` behind the first barrier of the tile.  chain_l2_warm's guarded loads are folded
// into the sink one by one and the compiler puts an `s_waitcnt vmcnt(0)` behind each: 6-12 round trips in series in front of
// the tile's first operand.  Measured (profiles/r02_notes.md): converting EVERY chain made the step slower (the serial form
// throttles the burst at the start of a launch); this form is used where the serial chain is longest (FFN forward: 12 rounds).
template <int MAXLD> struct L2Warm {
  uint4 v[MAXLD];
  AMT_DEV void issue(const void* base, size_t bytes, int q, int nslices, int tid) {
#pragma unroll
    for (int i = 0; i < MAXLD; ++i) v[i] = uint4{0u, 0u, 0u, 0u};
    if (q >= nslices || bytes < 16) return;                                          // uniform over the workgroup
    const size_t per = ((bytes + nslices - 1) / nslices + 4095) & ~(size_t)4095;   // whole 4 KiB rounds
    const size_t last = (bytes - 16) & ~(size_t)15;
    const char* p = static_cast<const char*>(base);
    const size_t beg = (size_t)q * per + (size_t)tid * 16;
#pragma unroll
    for (int i = 0; i < MAXLD; ++i) {
      size_t off = beg + (size_t)i * 4096;
      if ((size_t)i * 4096 >= per) off = beg;          // rounds past the slice repeat its first one
      off = off < last ? off : last;
      v[i] = *reinterpret_cast<const uint4*>(p + off);
    }
  }
  // an empty volatile asm that takes the registers as inputs: the loads stay alive, nothing is computed, and the compiler
  // cannot move this use (and with it the wait for the loads) in front of the barrier it is placed behind
  AMT_DEV void fold() const {
#pragma unroll
    for (int i = 0; i < MAXLD; ++i) asm volatile("" :: "v"(v[i].x), "v"(v[i].y), "v"(v[i].z), "v"(v[i].w));
  }
};

// LayerNorm backward over C = 256 features of TT rows (eqx LayerNorm, fp32 statistics):
//   dX = dRes + LN'(dY),  aw += dY * xhat,  ab += dY   (column accumulators of this lane, cs[0] / cs[1])
// dY rows come from an LDS panel (as stored, T), x (saved forward stream: Res<T>) / dRes rows from global memory; dX goes to an LDS
// panel (Os, may alias DYs row-for-row) and, if DX != nullptr, to global rows (16 B per lane).
// If DXD != nullptr a second copy with the FFN-output dropout mask of the consuming layer applied
// (model.py:237) is written as well: the mask depends only on (seed, stream, row, column).
template <typename T, int TT, int LD>
AMT_DEV void chain_ln_bwd_rows(const T* DYs, T* Os, const Res<T>* __restrict__ X, const float* __restrict__ mean,
                               const float* __restrict__ rstd, const float* __restrict__ w, const T* __restrict__ dRes,
                               T* __restrict__ DX, T* __restrict__ DXD, uint64_t seed, uint32_t stream, uint32_t thr,
                               float inv_keep, float (&cs)[2][8], long n0, long N, int tid) {
  constexpr int C = 256, LPR = 32, RPP = 8;
  static_assert(TT % RPP == 0, "row phase covers the tile in whole passes");
  const int sub = tid % LPR, prow = tid / LPR;
  float wv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) wv[i] = w[sub * 8 + i];
#pragma unroll
  for (int pp = 0; pp < TT / RPP; ++pp) {
    const int p = pp * RPP + prow;
    const long row = n0 + p;
    const bool valid = row < N;
    const long rc = valid ? row : N - 1;
    float x[8], dy[8], rr[8];
    load8f(X + rc * C + sub * 8, x);
    load8f(dRes + rc * C + sub * 8, rr);
    load8f(DYs + p * LD + sub * 8, dy);
    const float mu = mean[rc], rs = rstd[rc];
    const float fv = valid ? 1.f : 0.f;
    // explicit roundings (no compiler-chosen fma contraction): layernorm_bwd (norm.hip) uses the same
    // sequence, so the chain and the unfused launch agree bit for bit
    float s1 = 0.f, s2 = 0.f, gy[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      x[i] = __fmul_rn(__fsub_rn(x[i], mu), rs);
      gy[i] = __fmul_rn(dy[i], wv[i]);
      s1 = __fadd_rn(s1, gy[i]);
      s2 = __fmaf_rn(gy[i], x[i], s2);
      cs[0][i] += fv * dy[i] * x[i];
      cs[1][i] += fv * dy[i];
    }
    s1 = group_sum<LPR>(s1) * (1.f / C);
    s2 = group_sum<LPR>(s2) * (1.f / C);
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      o[i] = __fadd_rn(__fmul_rn(rs, __fmaf_rn(-x[i], s2, __fsub_rn(gy[i], s1))), rr[i]);
    store8f(Os + p * LD + sub * 8, o);
    if (valid && DX != nullptr) store8f(DX + row * C + sub * 8, o);
    if (valid && DXD != nullptr) {
      float od[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = to_f(from_f<T>(o[i]));   // the unfused path applies the mask to the stored value
        const bool keep = rng_keep(seed, stream, (uint32_t)row * (uint32_t)C + (uint32_t)(sub * 8 + i), thr);
        od[i] = keep ? v * inv_keep : 0.f;
      }
      store8f(DXD + row * C + sub * 8, od);
    }
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
