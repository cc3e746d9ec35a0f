// MFMA GEMMs for gfx950:  gemm_nt  (Y = X . W^T with fused epilogues)  and  gemm_tn_wgrad.
//
// Reference ops served (all of them plain matmuls in XLA):
//   eqx.nn.Linear / 1x1 Conv1d forward   model.py:141-153,163-165,192,232-236,300-334,369
//   their data gradients (dY . W)        and weight gradients (dY^T . X)
//
// gemm_nt computes D = W_tile . X_tile^T on the matrix cores with FEATURES on the MFMA row axis
// and TOKENS on the column axis, so that a lane ends up with 4 consecutive output features of one
// token: bias / gamma / GLU pairing are per-register and stores are 8-byte (bf16) vectors.
#include <cstdio>
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "prof.h"
#include "wgrad.h"
#include <string>
#include <cstring>
#include <cstdlib>

namespace amt {

constexpr int GBF = 64;       // features per workgroup

template <typename T>
AMT_DEV Chunk8<T> load_k_chunk(const T* rowp, int k, int K) {
  if (k + 8 <= K) return chunk_load(rowp + k);
  if (k + 4 <= K) return chunk_load4(rowp + k);
  return chunk_zero<T>();
}

// ------------------------------------------------------------------ epilogues
template <typename T, int EPI>
AMT_DEV void epi_apply(const GemmArgs& a, int n, int m, float (&v)[4]) {
  if (a.bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += a.bias[m + i];
  }
  if constexpr (EPI == EPI_STORE) {
    store4f(reinterpret_cast<T*>(a.Y) + (size_t)n * a.ldy + m, v);
  } else if constexpr (EPI == EPI_STORE_F32) {
    store4f(reinterpret_cast<float*>(a.Y) + (size_t)n * a.ldy + m, v);
  } else if constexpr (EPI == EPI_STORE_RES) {
    store4f(reinterpret_cast<Res<T>*>(a.Y) + (size_t)n * a.ldy + m, v);
  } else if constexpr (EPI == EPI_RESID) {
    float r[4];
    load4f(reinterpret_cast<const Res<T>*>(a.R) + (size_t)n * a.ldr + m, r);
    if (a.drop_thresh24) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bool keep = rng_keep(a.seed, a.stream, (uint32_t)n * (uint32_t)a.M + (uint32_t)(m + i), a.drop_thresh24);
        v[i] = keep ? v[i] * a.inv_keep : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += r[i];
    store4f(reinterpret_cast<Res<T>*>(a.Y) + (size_t)n * a.ldy + m, v);
  } else if constexpr (EPI == EPI_GELU) {
    float h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = gelu_tanh(v[i]);      // from the unrounded u (chain.hip: block_fwd_tile)
    store4f(reinterpret_cast<T*>(a.Y) + (size_t)n * a.ldy + m, v);
    store4f(reinterpret_cast<T*>(a.Y2) + (size_t)n * a.ldy2 + m, h);
  } else if constexpr (EPI == EPI_SCALE_RESID) {
    float r[4], y[4];
    load4f(reinterpret_cast<const Res<T>*>(a.R) + (size_t)n * a.ldr + m, r);
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = r[i] + a.gamma[m + i] * v[i];      // z unrounded (chain.hip: block_fwd_tile)
    if (a.Y2 != nullptr) store4f(reinterpret_cast<T*>(a.Y2) + (size_t)n * a.ldy2 + m, v);
    store4f(reinterpret_cast<Res<T>*>(a.Y) + (size_t)n * a.ldy + m, y);
  } else if constexpr (EPI == EPI_GLU_BWD) {
    // m indexes dg columns j (0..M-1); u lives in the permuted [.. 16 x u1 | 16 x u2 ..] layout
    int pm = (m >> 4) * 32 + (m & 15);
    const T* up = reinterpret_cast<const T*>(a.U) + (size_t)n * a.ldu + pm;
    float u1[4], u2[4], d1[4], d2[4];
    load4f(up, u1);
    load4f(up + 16, u2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      d1[i] = v[i] * u2[i] * gelu_tanh_grad(u1[i]);
      d2[i] = v[i] * gelu_tanh(u1[i]);
    }
    T* yp = reinterpret_cast<T*>(a.Y) + (size_t)n * a.ldy + pm;
    store4f(yp, d1);
    store4f(yp + 16, d2);
  } else if constexpr (EPI == EPI_GELU_BWD) {
    float u[4];
    load4f(reinterpret_cast<const T*>(a.U) + (size_t)n * a.ldu + m, u);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] *= gelu_tanh_grad(u[i]);
    store4f(reinterpret_cast<T*>(a.Y) + (size_t)n * a.ldy + m, v);
  }
}

// paired epilogue (GLU forward): u1 at permuted feature m (m % 32 < 16), u2 at m + 16
template <typename T>
AMT_DEV void epi_glu(const GemmArgs& a, int n, int m, float (&u1)[4], float (&u2)[4]) {
  float gl[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (a.bias) { u1[i] += a.bias[m + i]; u2[i] += a.bias[m + 16 + i]; }
    u1[i] = to_f(from_f<T>(u1[i]));
    u2[i] = to_f(from_f<T>(u2[i]));
    gl[i] = gelu_tanh(u1[i]) * u2[i];
  }
  T* up = reinterpret_cast<T*>(a.Y2) + (size_t)n * a.ldy2 + m;
  store4f(up, u1);
  store4f(up + 16, u2);
  int j = (m >> 5) * 16 + (m & 15);
  store4f(reinterpret_cast<T*>(a.Y) + (size_t)n * a.ldy + j, gl);
}

// ------------------------------------------------------------------ gemm_nt kernel
// workgroup: 256 threads = 4 waves arranged 2 (token halves) x 2 (feature halves);
// tile BT tokens x 64 features; per wave (BT/2) x 32 = TI x 2 MFMA 16x16 sub-tiles.
template <typename T, int EPI, int BT, int BK>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs a) {
  a.seed = eff_seed(a.seed, a.seed_dev);   // kernel argument + the context's device seed word (common.h)
  constexpr int TI = BT / 32;
  constexpr int FI = 2;
  // BK = K panel per LDS stage, chosen by the launcher as the smallest of {64, 128, 256} (bf16) that
  // covers K: every load of a panel is issued up front (up to 16 sixteen-byte loads in flight per
  // thread), so K <= 256 costs ONE memory round trip instead of one per 64 columns.  One LDS
  // buffer; for K > BK the next panel is prefetched into registers while this one is multiplied.
  constexpr int LDK = BK + 8;
  constexpr int CPR = BK / 8;                // 8-element chunks per tile row
  constexpr int XCH = BT * CPR / 256;        // X chunks per thread per stage
  constexpr int WCH = GBF * CPR / 256;       // W chunks per thread per stage
  extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
  T (*Xs)[LDK] = reinterpret_cast<T (*)[LDK]>(gemm_smem);
  T (*Ws)[LDK] = reinterpret_cast<T (*)[LDK]>(gemm_smem + sizeof(T) * BT * LDK);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int wt = wave >> 1, wf = wave & 1;
  // XCD-aware tile order (1-D grid).  Workgroups are dealt round-robin over the 8 XCDs (id % 8), each
  // with a private L2: all feature tiles of one token tile get the same id % 8, so the X panel they
  // share is fetched into ONE L2 instead of up to eight (measured 2.8x the algorithmic bytes before).
  const int ftiles = (a.M + GBF - 1) / GBF;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int ttile = (jj / ftiles) * 8 + xcd;
  const int n0 = ttile * BT, m0 = (jj % ftiles) * GBF;
  if (n0 >= a.N) return;
  const T* X = reinterpret_cast<const T*>(a.X);
  const T* W = reinterpret_cast<const T*>(a.W);
  const int K = a.K;

  f32x4 acc[FI][TI];
#pragma unroll
  for (int fi = 0; fi < FI; ++fi)
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) acc[fi][ti] = f32x4{0.f, 0.f, 0.f, 0.f};

  Chunk8<T> xr[XCH], wr[WCH];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      int id = tid + i * 256, row = id / CPR, c = id % CPR;
      int n = n0 + row;
      xr[i] = (n < a.N) ? load_k_chunk(X + (size_t)n * a.ldx, k0 + 8 * c, K) : chunk_zero<T>();
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      int id = tid + i * 256, row = id / CPR, c = id % CPR;
      int m = m0 + row;
      wr[i] = (m < a.M) ? load_k_chunk(W + (size_t)m * a.ldw, k0 + 8 * c, K) : chunk_zero<T>();
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      int id = tid + i * 256, row = id / CPR, c = id % CPR;
      chunk_store(&Xs[row][8 * c], xr[i]);
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      int id = tid + i * 256, row = id / CPR, c = id % CPR;
      chunk_store(&Ws[row][8 * c], wr[i]);
    }
  };

  const int nk = (K + BK - 1) / BK;
  load_tiles(0);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt > 0) __syncthreads();   // previous panel fully consumed
    store_tiles();
    __syncthreads();
    if (kt + 1 < nk) load_tiles((kt + 1) * BK);
    const int kk_end = min(BK, K - kt * BK);
    for (int kk = 0; kk < kk_end; kk += 32) {
      Frag<T> wfr[FI], xfr[TI];
#pragma unroll
      for (int fi = 0; fi < FI; ++fi) wfr[fi] = frag_row(&Ws[wf * 32 + fi * 16 + r][kk + 8 * g]);
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) xfr[ti] = frag_row(&Xs[wt * (BT / 2) + ti * 16 + r][kk + 8 * g]);
#pragma unroll
      for (int fi = 0; fi < FI; ++fi)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) mma(acc[fi][ti], wfr[fi], xfr[ti]);
    }
  }

  // epilogue: acc[fi][ti][i] = Y[n0 + wt*BT/2 + ti*16 + r][m0 + wf*32 + fi*16 + 4g + i]
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) {
    int n = n0 + wt * (BT / 2) + ti * 16 + r;
    if (n >= a.N) continue;
    if constexpr (EPI == EPI_GLU) {
      int m = m0 + wf * 32 + 4 * g;
      if (m < a.M) {
        float u1[4], u2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { u1[i] = acc[0][ti][i]; u2[i] = acc[1][ti][i]; }
        epi_glu<T>(a, n, m, u1, u2);
      }
    } else {
#pragma unroll
      for (int fi = 0; fi < FI; ++fi) {
        int m = m0 + wf * 32 + fi * 16 + 4 * g;
        if (m < a.M) {
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[fi][ti][i];
          epi_apply<T, EPI>(a, n, m, v);
        }
      }
    }
  }
}

template <typename T, int EPI, int BT, int BK>
static void launch_nt_cfg(hipStream_t s, const GemmArgs& a, const char* name, double fl, double by) {
  constexpr size_t lds = sizeof(T) * (size_t)(BT + GBF) * (BK + 8);
  static bool once = (lds > 64 * 1024
                          ? (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<T, EPI, BT, BK>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                          : (void)0,
                      true);
  (void)once;
  ProfScope ps(s, name, fl, by);
  const int ttiles8 = ((a.N + BT - 1) / BT + 7) / 8 * 8;   // token tiles padded to a multiple of 8 (XCD remap)
  dim3 grid(ttiles8 * ((a.M + GBF - 1) / GBF));
  hipLaunchKernelGGL((gemm_nt_kernel<T, EPI, BT, BK>), grid, dim3(256), lds, s, a);
}

template <typename T, int EPI>
static void launch_nt(hipStream_t s, const GemmArgs& a) {
  static const char* const epi_names[EPI_COUNT] = {"store", "store_f32", "resid", "gelu", "scale_resid", "glu",
                                                   "glu_bwd", "gelu_bwd", "store_res"};
  constexpr size_t es = sizeof(T);
  constexpr int B0 = es == 2 ? 64 : 32;     // panel sizes: {64,128,256} for bf16, {32,64,128} for f32
  const double fl = 2.0 * a.N * (double)a.M * a.K;
  const double by = ((double)a.N * a.K + (double)a.M * a.K + (double)a.N * a.M) * es;
  // 128-token tiles when that still yields >= 2 workgroups per CU, else 64-token tiles
  long tiles128 = (long)((a.N + 127) / 128) * ((a.M + GBF - 1) / GBF);
  static const std::string nm128 = std::string("gemm_nt<") + (es == 4 ? "f32," : "bf16,") + epi_names[EPI] + ",128>";
  static const std::string nm64 = std::string("gemm_nt<") + (es == 4 ? "f32," : "bf16,") + epi_names[EPI] + ",64>";
  if (tiles128 >= 512) {
    if (a.K <= B0) launch_nt_cfg<T, EPI, 128, B0>(s, a, nm128.c_str(), fl, by);
    else launch_nt_cfg<T, EPI, 128, 2 * B0>(s, a, nm128.c_str(), fl, by);
  } else {
    if (a.K <= B0) launch_nt_cfg<T, EPI, 64, B0>(s, a, nm64.c_str(), fl, by);
    else if (a.K <= 2 * B0) launch_nt_cfg<T, EPI, 64, 2 * B0>(s, a, nm64.c_str(), fl, by);
    else launch_nt_cfg<T, EPI, 64, 4 * B0>(s, a, nm64.c_str(), fl, by);
  }
}

template <typename T>
static void dispatch_nt(hipStream_t s, EpiKind epi, const GemmArgs& a) {
  switch (epi) {
    case EPI_STORE: launch_nt<T, EPI_STORE>(s, a); break;
    case EPI_STORE_F32: launch_nt<T, EPI_STORE_F32>(s, a); break;
    case EPI_RESID: launch_nt<T, EPI_RESID>(s, a); break;
    case EPI_GELU: launch_nt<T, EPI_GELU>(s, a); break;
    case EPI_SCALE_RESID: launch_nt<T, EPI_SCALE_RESID>(s, a); break;
    case EPI_GLU: launch_nt<T, EPI_GLU>(s, a); break;
    case EPI_GLU_BWD: launch_nt<T, EPI_GLU_BWD>(s, a); break;
    case EPI_GELU_BWD: launch_nt<T, EPI_GELU_BWD>(s, a); break;
    case EPI_STORE_RES: launch_nt<T, EPI_STORE_RES>(s, a); break;
    default: break;
  }
}

void gemm_nt(hipStream_t s, DType dt, EpiKind epi, const GemmArgs& a) {
  if (a.N <= 0 || a.M <= 0) return;
  if (dt == DT_F32) dispatch_nt<float>(s, epi, a); else dispatch_nt<bf16>(s, epi, a);
}

// ------------------------------------------------------------------ gemm_tn_wgrad (body: wgrad.h)
template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_kernel(WgradArgs a, int tiles_k, int rows_per_split) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tn_smem[];
  tn_body<T>(a, tiles_k, rows_per_split, blockIdx.x, blockIdx.y, tn_smem);
}

// several independent weight-gradient problems in ONE launch (all wgrads of a transformer layer or
// of a CNN block)
template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_group_kernel(WgradGroupWide g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tn_smem[];
  wgrad_group_dispatch<T, WgradGroupWide>(g, blockIdx.x, tn_smem);
}

void gemm_tn_wgrad(hipStream_t s, DType dt, const WgradArgs& a_in, PartialSink* sink) {
  WgradArgs a = a_in;
  if (a.N <= 0) return;
  int tiles_m = (a.M + 63) / 64, tiles_k = (a.Kp + 63) / 64;
  int tiles = tiles_m * tiles_k;
  const int Mst = a.M_store > 0 ? a.M_store : a.M;
  if (sink && tiles == 1 && a.N >= 16384 && a.ldw == a.Kp) {
    // skinny weight gradient (one 64x64 tile, very long token axis): many short splits, each
    // writing one record; reduce_partials sums them (no same-address atomics)
    int rps = 256;
    int splits = (a.N + rps - 1) / rps;
    if (splits > 1024) { splits = 1024; rps = ((a.N + splits - 1) / splits + 127) / 128 * 128; splits = (a.N + rps - 1) / rps; }
    const size_t rec = (size_t)Mst * a.Kp + (a.dbias ? Mst : 0);
    size_t o = sink->take((size_t)splits * rec);
    sink->add(o, splits, (uint32_t)rec, (uint32_t)(Mst * a.Kp), a.dW);
    if (a.dbias) sink->add(o + (size_t)Mst * a.Kp, splits, (uint32_t)rec, (uint32_t)Mst, a.dbias);
    if (sink->dry) return;
    a.part = sink->ws + o;
    a.part_stride = (int)rec;
    ProfScope ps(s, dt == DT_F32 ? "gemm_tn_wgrad_skinny<f32>" : "gemm_tn_wgrad_skinny<bf16>",
                 2.0 * a.N * (double)a.M * a.Kp, ((double)a.N * a.M + (double)a.N * a.Kp) * (dt == DT_F32 ? 4 : 2));
    dim3 grid(1, splits);
    if (dt == DT_F32) hipLaunchKernelGGL((gemm_tn_kernel<float>), grid, dim3(256), TnLds<float>::BYTES, s, a, 1, rps);
    else hipLaunchKernelGGL((gemm_tn_kernel<bf16>), grid, dim3(256), TnLds<bf16>::BYTES, s, a, 1, rps);
    return;
  }
  if (sink && sink->dry) return;
  // aim for ~1024 workgroups with at least 128 tokens per split, but never more than 48 splits:
  // every split adds its partial tile into the same dW addresses and contended float atomics are
  // an order of magnitude slower than spread ones (MI355X_MICROARCH.md, Global float atomics)
  int splits = (1024 + tiles - 1) / tiles;
  int max_splits = (a.N + 127) / 128;
  if (splits > max_splits) splits = max_splits;
  if (splits > 48) splits = 48;
  if (splits < 1) splits = 1;
  int rps = ((a.N + splits - 1) / splits + 127) / 128 * 128;
  splits = (a.N + rps - 1) / rps;
  dim3 grid(tiles, splits);
  const double es = dt == DT_F32 ? 4 : 2;
  ProfScope ps(s, dt == DT_F32 ? "gemm_tn_wgrad<f32>" : "gemm_tn_wgrad<bf16>", 2.0 * a.N * (double)a.M * a.Kp,
               ((double)a.N * a.M + (double)a.N * a.Kp) * es + (double)a.M * a.Kp * 4);
  if (dt == DT_F32)
    hipLaunchKernelGGL((gemm_tn_kernel<float>), grid, dim3(256), TnLds<float>::BYTES, s, a, tiles_k, rps);
  else
    hipLaunchKernelGGL((gemm_tn_kernel<bf16>), grid, dim3(256), TnLds<bf16>::BYTES, s, a, tiles_k, rps);
}

// Fills `g` for up to 8 problems and returns the number of workgroups (0: nothing to launch, e.g.
// while planning).  One-tile problems with a very long token axis ("skinny", C <= 32 blocks) get many
// short splits that write per-split records into the sink instead of atomics.
template <int CAP>
static int wgrad_group_prepare_t(DType dt, const WgradArgs* probs, int n, PartialSink* sink, WgradGroupN<CAP>* gp, double* flops,
                                 double* bytes, int target_override) {
  WgradGroupN<CAP>& g = *gp;
  memset(&g, 0, sizeof(g));
  if (n <= 0) return 0;
  if (n > CAP) {             // a caller error (every caller chunks by CAP): losing weight gradients silently is not an option
    fprintf(stderr, "amt: wgrad_group_prepare: %d problems handed to a group of %d - aborting\n", n, CAP);
    abort();
  }
  g.n = n;
  // XCD-aware item order (kernels.h).  At cfg2 (riders inside latency-bound chain launches) it measured slower: off
  // (profiles/r02_notes.md).  The stand-alone grouped launches of the large-N schedule (target_override >= 512: no riders from
  // 8192 tokens on) run at 5.6 TB/s of fabric reads - 208 MB per launch for 119 MB of operands (profiles/r04_pmc_traffic_b32.json)
  // - and there "every XCD keeps its eighth of every problem" pays: 32 clips 8.21 -> 8.06 ms (0 / 1 / 2: 8.214 / 8.154 / 8.055;
  // with more token splits so that a split never straddles two XCDs, targets 768 / 1120 / 1536: 8.20 / 8.13 / 8.21).
  static const int xcd_order = getenv("AMT_WGRAD_XCD") ? atoi(getenv("AMT_WGRAD_XCD")) : -1;
  g.xcd = xcd_order >= 0 ? xcd_order : (target_override >= 512 ? 2 : 0);
  int total_tiles = 0;
  bool skinny[CAP];
  for (int i = 0; i < n; ++i) {
    g.p[i] = probs[i];
    g.tiles_k[i] = (probs[i].Kp + 63) / 64;
    g.tiles[i] = ((probs[i].M + 63) / 64) * g.tiles_k[i];
    skinny[i] = sink && g.tiles[i] == 1 && probs[i].N >= 16384 && probs[i].ldw == probs[i].Kp;
    if (!skinny[i]) total_tiles += g.tiles[i];
  }
  double fl = 0, by = 0;
  const double es = dt == DT_F32 ? 4 : 2;
  for (int i = 0; i < n; ++i) {
    WgradArgs& a = g.p[i];
    fl += 2.0 * a.N * (double)a.M * a.Kp;
    by += ((double)a.N * a.M + (double)a.N * a.Kp) * es + (double)a.M * a.Kp * 4;
    if (skinny[i]) {
      const int Mst = a.M_store > 0 ? a.M_store : a.M;
      static const int skinny_rps = getenv("AMT_SKINNY_RPS") ? atoi(getenv("AMT_SKINNY_RPS")) : 512;   // 256 / 512 / 768 / 1024: 3.202 / 3.194 / 3.211 / 3.227 ms/step (round 2, after the rider fixes; 256 won before them)
      int rps = skinny_rps;
      int splits = (a.N + rps - 1) / rps;
      if (splits > 1024) { splits = 1024; rps = ((a.N + splits - 1) / splits + 127) / 128 * 128; splits = (a.N + rps - 1) / rps; }
      const size_t rec = (size_t)Mst * a.Kp + (a.dbias ? Mst : 0);
      size_t o = sink->take((size_t)splits * rec);
      sink->add(o, splits, (uint32_t)rec, (uint32_t)(Mst * a.Kp), a.dW);
      if (a.dbias) sink->add(o + (size_t)Mst * a.Kp, splits, (uint32_t)rec, (uint32_t)Mst, a.dbias);
      a.part = sink->dry ? nullptr : sink->ws + o;
      a.part_stride = (int)rec;
      g.rps[i] = rps;
      g.prefix[i + 1] = g.prefix[i] + splits;
      continue;
    }
    // few token splits: every split adds M*Kp*4 bytes of fp32 atomics and the chip-wide atomic rate
    // (~1.3 TB/s) was the bound at 1536 workgroups (38 us per layer group vs 24 us at ~224)
    static const int target_wgs = getenv("AMT_WGRAD_TARGET_WGS") ? atoi(getenv("AMT_WGRAD_TARGET_WGS")) : 336;
    const int tgt = target_override > 0 ? target_override : target_wgs;
    int splits = (tgt + total_tiles - 1) / total_tiles;
    int max_splits = (a.N + 127) / 128;
    if (splits > max_splits) splits = max_splits;
    if (splits > 48) splits = 48;
    if (splits < 1) splits = 1;
    int rps = ((a.N + splits - 1) / splits + 127) / 128 * 128;
    splits = (a.N + rps - 1) / rps;
    g.rps[i] = rps;
    g.prefix[i + 1] = g.prefix[i] + g.tiles[i] * splits;
    // Opt-in (AMT_WGRAD_REC_MIN_SPLITS=n): from n token splits on, every split writes its tile into a record of its own with
    // plain stores instead of adding into the same 64 x 64 floats as the other splits, and reduce_partials sums the records
    // once per step.  Measured at cfg2 with n = 3: Block-backward launches 27.95 -> 27.37 us, reduce_partials 31 -> 99 us
    // (170 MB of records): slower overall, so the contended atomics are not what a rider waits for (profiles/r02_notes.md).
    static const int rec_min = getenv("AMT_WGRAD_REC_MIN_SPLITS") ? atoi(getenv("AMT_WGRAD_REC_MIN_SPLITS")) : 0;   // measured: off (below)
    if (sink && rec_min > 0 && splits >= rec_min && a.ldw == a.Kp) {
      const int Mst = a.M_store > 0 ? a.M_store : a.M;
      const size_t rec = (size_t)Mst * a.Kp + (a.dbias ? Mst : 0);
      size_t o = sink->take((size_t)splits * rec);
      sink->add(o, splits, (uint32_t)rec, (uint32_t)(Mst * a.Kp), a.dW);
      if (a.dbias) sink->add(o + (size_t)Mst * a.Kp, splits, (uint32_t)rec, (uint32_t)Mst, a.dbias);
      a.part = sink->dry ? nullptr : sink->ws + o;
      a.part_stride = (int)rec;
    }
  }
  if (flops) *flops = fl;
  if (bytes) *bytes = by;
  if (sink && sink->dry) return 0;
  return g.prefix[n];
}

int wgrad_group_prepare(DType dt, const WgradArgs* probs, int n, PartialSink* sink, WgradGroup* g, double* flops, double* bytes,
                        int target_override) {
  return wgrad_group_prepare_t<8>(dt, probs, n, sink, g, flops, bytes, target_override);
}
int wgrad_group_prepare(DType dt, const WgradArgs* probs, int n, PartialSink* sink, WgradGroupWide* g, double* flops, double* bytes,
                        int target_override) {
  return wgrad_group_prepare_t<16>(dt, probs, n, sink, g, flops, bytes, target_override);
}

void gemm_tn_wgrad_group(hipStream_t s, DType dt, const WgradArgs* probs, int n, PartialSink* sink, int target_wgs) {
  if (n <= 0) return;
  if (n == 1 && target_wgs <= 0) { gemm_tn_wgrad(s, dt, probs[0], sink); return; }
  WgradGroupWide g;
  double fl = 0, by = 0;
  const int nwg = wgrad_group_prepare(dt, probs, n, sink, &g, &fl, &by, target_wgs);
  if (nwg <= 0) return;
  ProfScope ps(s, dt == DT_F32 ? "gemm_tn_wgrad_group<f32>" : "gemm_tn_wgrad_group<bf16>", fl, by);
  if (dt == DT_F32) hipLaunchKernelGGL((gemm_tn_group_kernel<float>), dim3(nwg), dim3(256), TnLds<float>::BYTES, s, g);
  else hipLaunchKernelGGL((gemm_tn_group_kernel<bf16>), dim3(nwg), dim3(256), TnLds<bf16>::BYTES, s, g);
}

}  // namespace amt
