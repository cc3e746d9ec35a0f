// Weight-gradient GEMM body shared by gemm.hip (stand-alone launches) and chain.hip (weight-gradient
// workgroups that ride on a chain kernel's launch).
#pragma once
#include <type_traits>
#include "common.h"
#include "kernels.h"

#ifndef AMT_WGRAD_DEPTH
#define AMT_WGRAD_DEPTH 1
#endif
#ifndef AMT_WG_STAMP
#define AMT_WG_STAMP(k) do { } while (0)
#define AMT_WG_NOTE(k, v) do { } while (0)
#endif

namespace amt {

// dW[m][k] += sum_n G[n][m] A[n][k].  Tokens are the MFMA K axis, so both operands are read as
// COLUMNS of row-major LDS tiles.  Workgroup tile 64 (m) x 64 (k); the token range is split over
// blockIdx.y and partial tiles are combined with fp32 atomics (unordered: last-bit differences
// between runs are expected).
constexpr int WLD = 72;   // LDS row stride (elements) for the [32 tokens][64 features] tiles

template <typename T>
AMT_DEV Chunk8<T> load_f_chunk(const T* rowp, int f, int F) {
  if (f + 8 <= F) return chunk_load(rowp + f);
  if (f + 4 <= F) return chunk_load4(rowp + f);
  return chunk_zero<T>();
}

template <typename T> struct TnLds {
  static constexpr int TOK = sizeof(T) == 2 ? 128 : 64;
  static constexpr size_t BYTES = 2 * sizeof(T) * TOK * WLD + sizeof(float) * 32 * 65;
};

template <typename T>
AMT_DEV void tn_body(const WgradArgs& a, int tiles_k, int rows_per_split, int bx, int by, unsigned char* smem) {
  // TOK tokens per LDS stage (4 MFMA k-steps for bf16): every thread keeps 8 sixteen-byte loads in
  // flight per stage.  One LDS buffer + register prefetch of the next AMT_WGRAD_DEPTH - 1 stages.
  constexpr int TOK = sizeof(T) == 2 ? 128 : 64;
  constexpr int NCH = TOK * 8 / 256;          // chunks per thread per operand per stage
  T (*Gs)[WLD] = reinterpret_cast<T (*)[WLD]>(smem);
  T (*As)[WLD] = reinterpret_cast<T (*)[WLD]>(smem + sizeof(T) * TOK * WLD);
  float (*red)[65] = reinterpret_cast<float (*)[65]>(smem + 2 * sizeof(T) * TOK * WLD);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wk = wave & 1;
  const int m0 = (bx / tiles_k) * 64, k0 = (bx % tiles_k) * 64;
  const int nbeg = by * rows_per_split;
  const int nend = min(a.N, nbeg + rows_per_split);
  const T* G = reinterpret_cast<const T*>(a.G);
  const T* A = reinterpret_cast<const T*>(a.A);
  const bool do_bias = (a.dbias != nullptr) && (bx % tiles_k == 0);
  const int Mst = a.M_store > 0 ? a.M_store : a.M;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bsum[i] = 0.f;
  // row scales (layer scale folded into the pw2 gradients) of this lane's 8 output rows and of the bias row it reduces:
  // requested here, used in the epilogue (read there, each was a dependent L2 round trip in front of its atomic)
  float rsc[2][4], rsb = 1.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) rsc[i][e] = 1.f;
  if (a.row_scale) {   // uniform
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) rsc[i][e] = a.row_scale[min(m0 + wm * 32 + i * 16 + 4 * g + e, Mst - 1)];
    rsb = a.row_scale[min(m0 + (tid & 63), Mst - 1)];
  }

  const int lrow = tid >> 3, lc = tid & 7;   // thread covers rows lrow + 32 i, chunk lc
  // DEPTH register sets: the loads of stages st + 1 .. st + DEPTH - 1 are in flight while stage st is multiplied.  With
  // one set (round 1) a rider's stage took one full memory round trip under load (4.4 us measured in the Block-backward
  // launches at cfg2, profiles/r02_notes.md): its 5 stages ran 22 us and every carrier launch ended with its riders.
  constexpr int DEPTH = AMT_WGRAD_DEPTH;
  Chunk8<T> gr[DEPTH][NCH], ar[DEPTH][NCH];
  // Whole 16-byte chunks everywhere (the usual case): UNCONDITIONAL loads from a clamped row / column and a select
  // afterwards.  The guarded form below compiles into nested exec-mask branches with an `s_waitcnt vmcnt(0)` behind the
  // first load of every chunk (the register shuffles of the partial-chunk arms), i.e. the 8 loads of a stage went out one
  // round trip after the other: that, not the depth of the prefetch, was what a rider's stage waited for.
  const bool whole = ((a.M | a.Kp | a.ldg | a.lda) & 7) == 0;      // uniform
  const int gcol = m0 + 8 * lc, acol = k0 + 8 * lc;
  const bool gin = gcol < a.M, ain = acol < a.Kp;
  const T* Gc = G + (gin ? gcol : 0);
  const T* Ac = A + (ain ? acol : 0);
  auto load_tiles = [&](Chunk8<T> (&g_)[NCH], Chunk8<T> (&a_)[NCH], int nb) {
    if (whole) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int n = nb + lrow + 32 * i;
        const bool ok = n < nend;
        const size_t nc = (size_t)(ok ? n : nend - 1);
        g_[i] = chunk_load(Gc + nc * a.ldg);
        a_[i] = chunk_load(Ac + nc * a.lda);
        if (!(ok && gin)) g_[i] = chunk_zero<T>();
        if (!(ok && ain)) a_[i] = chunk_zero<T>();
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int n = nb + lrow + 32 * i;
      bool ok = n < nend;
      g_[i] = ok ? load_f_chunk(G + (size_t)n * a.ldg, m0 + 8 * lc, a.M) : chunk_zero<T>();
      a_[i] = ok ? load_f_chunk(A + (size_t)n * a.lda, k0 + 8 * lc, a.Kp) : chunk_zero<T>();
    }
  };
  auto store_tiles = [&](const Chunk8<T> (&g_)[NCH], const Chunk8<T> (&a_)[NCH]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      chunk_store(&Gs[lrow + 32 * i][8 * lc], g_[i]);
      chunk_store(&As[lrow + 32 * i][8 * lc], a_[i]);
      if (do_bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[e] += chunk_get(g_[i], e);
      }
    }
  };

  const int nsteps = (nend - nbeg + TOK - 1) / TOK;
  AMT_WG_STAMP(0);
  AMT_WG_NOTE(6, nsteps);
  AMT_WG_NOTE(7, a.M * 1000 + a.Kp);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < nsteps) load_tiles(gr[d], ar[d], nbeg + d * TOK);
  for (int st0 = 0; st0 < nsteps; st0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int st = st0 + d;
      if (st < nsteps) {               // uniform over the workgroup
        __syncthreads();               // previous stage fully consumed
        store_tiles(gr[d], ar[d]);
        __syncthreads();
        if (st == 0) AMT_WG_STAMP(1);
        if (st + DEPTH < nsteps) load_tiles(gr[d], ar[d], nbeg + (st + DEPTH) * TOK);
#pragma unroll
        for (int kk = 0; kk < TOK / 32; ++kk) {
          Frag<T> gf[2], af[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) gf[i] = frag_col16(&Gs[32 * kk + 8 * g][wm * 32 + i * 16], WLD, r);
#pragma unroll
          for (int j = 0; j < 2; ++j) af[j] = frag_col16(&As[32 * kk + 8 * g][wk * 32 + j * 16], WLD, r);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma(acc[i][j], gf[i], af[j]);
        }
      }
    }
  }

  AMT_WG_STAMP(2);
  // acc[i][j][e] = dW[m0 + wm*32 + i*16 + 4g + e][k0 + wk*32 + j*16 + r]
  {
    // destination and the two uniform questions (record or atomic, whole tile or ragged) are settled once: asked per
    // element they were three scalar-load waits and three branches in front of each of the 16 stores / atomics
    const bool rec = a.part != nullptr;
    float* const dst = rec ? a.part + (size_t)by * a.part_stride : a.dW;
    const size_t ldd = rec ? (size_t)a.Kp : (size_t)a.ldw;
    const bool full = (m0 + 64 <= Mst) && (k0 + 64 <= a.Kp);
    const int Kp = a.Kp;
    auto emit = [&](auto FULL, auto REC) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int k = k0 + wk * 32 + j * 16 + r;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int m = m0 + wm * 32 + i * 16 + 4 * g + e;
            if (decltype(FULL)::value || (m < Mst && k < Kp)) {
              const float v = acc[i][j][e] * rsc[i][e];
              float* q = dst + (size_t)m * ldd + k;
              if (decltype(REC)::value) *q = v;
              else atomicAdd(q, v);
            }
          }
        }
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    if (full) { if (rec) emit(Yes{}, Yes{}); else emit(Yes{}, No{}); }
    else { if (rec) emit(No{}, Yes{}); else emit(No{}, No{}); }
  }

  AMT_WG_STAMP(3);
#ifdef AMT_CHAIN_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  AMT_WG_STAMP(4);
#endif
  if (do_bias) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) red[lrow][8 * lc + i] = bsum[i];
    __syncthreads();
    if (tid < 64) {
      float s = 0.f;
#pragma unroll 8
      for (int q = 0; q < 32; ++q) s += red[q][tid];
      int m = m0 + tid;
      if (m < Mst) {
        s *= rsb;
        if (a.part) a.part[(size_t)by * a.part_stride + (size_t)Mst * a.Kp + m] = s;
        else atomicAdd(&a.dbias[m], s);
      }
    }
  }
}

// (Round 4 built a large-N form of this body - 128 x 128 and 128 x 64 output tiles, each wave a 64 x 64 quadrant, two MFMAs
// per column-fragment read - and measured it SLOWER at every shape of the 32-clip step: W1 (16 000 x 1024 x 256) 36.9 against
// 35.4 us, the C = 128 Block pair 24.5 against 15.5 us, the grouped launches 57 against 39 us; with few, large tiles a launch
// needs more token splits to fill the chip, every split adds its whole tile with atomics, and a workgroup is a serial chain
// of ~35 stage latencies either way.  Removed again; numbers in profiles/r04_notes.md.)
// several independent weight-gradient problems in one grid: b -> (problem, tile, token split)
template <typename T, typename GROUP = WgradGroup>
AMT_DEV void wgrad_group_dispatch(const GROUP& g, int b, unsigned char* smem) {
  // Workgroups go to the XCDs round robin (b, b + 8, ... share one), work items are numbered problem by problem, split by
  // split, tile by tile: give every XCD a CONTIGUOUS range of items, i.e. whole token splits, whose tiles then find each
  // other's operand panels in their own L2 (the L2s do not share: a panel read on 8 XCDs crosses the fabric 8 times).
  if (g.xcd == 1) {
    const int R = g.prefix[g.n], per = R >> 3, rem = R & 7, c = b & 7;
    b = c * per + (c < rem ? c : rem) + (b >> 3);
  }
  int i = 0;
  while (i + 1 < g.n && b >= g.prefix[i + 1]) ++i;
  int local = b - g.prefix[i];
  if (g.xcd == 2) {   // the same, problem by problem: every XCD keeps its eighth of every problem
    const int R = g.prefix[i + 1] - g.prefix[i], per = R >> 3, rem = R & 7, c = local & 7;
    local = c * per + (c < rem ? c : rem) + (local >> 3);
  }
  tn_body<T>(g.p[i], g.tiles_k[i], g.rps[i], local % g.tiles[i], local / g.tiles[i], smem);
}
}  // namespace amt
