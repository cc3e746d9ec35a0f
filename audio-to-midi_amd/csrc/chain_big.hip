// Large-N forms of the forward chains (round 3): 128-token tiles, hidden dimension in SLICES.
//
// A chain workgroup streams the weights of its operator chain from L2 once per TILE, so the flops it gets out of every
// weight byte equal the tile length TT: the MFMA pipes of a compute unit want 4096 / TT bytes per clock at full rate,
// and one compute unit pulls ~28 B/clk from L2 (profiles/r01_notes.md).  The 16/32-token tiles of chain.hip are sized
// to fill the chip at 8 clips per GPU (4 000 tokens); from ~8 000 tokens on (streaming inference: 112 000 tokens,
// 32 clips per GPU: 16 000) they leave the kernels weight-stream-bound at 6-13 % of the MFMA peak.
//
// Here a tile is 128 tokens (bf16; 64 in f32).  What makes that fit:
//   * the wide hidden tensor (u / g of the FFN: 1024 / 512 per token; u / h of a C = 256 Block: 512) never exists for
//     the whole tile.  The hidden dimension is cut into SLICES of 64 activation columns: stage 1 computes the slice
//     (K = 256), the activation (GLU / GELU) writes it to a small LDS panel, stage 2 immediately multiplies it into the
//     output accumulators y[256][TT], which stay in registers across the slices (128 VGPRs per lane at one wave per
//     SIMD).  K order of every product is the unfused order (slice after slice = k-step after k-step into ONE
//     accumulator), so the results are bit-identical to the short-tile chains and to the unfused launches;
//   * the residual operand (x / h) lives in registers in the accumulator's lane layout instead of an LDS panel;
//   * inference variants (SAVE = false) write nothing but y.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "chain.h"
#include "kernels.h"
#include "prof.h"
#include "rowops.h"

// Phase stamps for scratch/big_bench.hip (compiled out of the library): stamps[workgroup][slot] += / = shader cycles.
#ifdef AMT_BIG_TIMING
__device__ unsigned long long amt_big_stamps[4096][16];
#define BIG_T0() unsigned long long big_t_ = __builtin_readcyclecounter()
#define BIG_ACC(k) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long n_ = __builtin_readcyclecounter(); \
                        if (threadIdx.x == 0 && blockIdx.x < 4096) amt_big_stamps[blockIdx.x][k] += n_ - big_t_; big_t_ = n_; } while (0)
#else
#define BIG_T0() do { } while (0)
#define BIG_ACC(k) do { } while (0)
#endif

namespace amt {

namespace {

template <typename T> struct Vec4T;
template <> struct Vec4T<bf16> { typedef bf16x4 type; };
template <> struct Vec4T<float> { typedef f32x4 type; };
template <> struct Vec4T<f16> { typedef f16x4 type; };

template <typename T> AMT_DEV typename Vec4T<T>::type ld4(const T* p) { return *reinterpret_cast<const typename Vec4T<T>::type*>(p); }

// One stage: acc[rb][ti] += sum over KS 32-deep k-steps of W-fragment[ks][rb] . X[token block ti], X fragments from an LDS
// panel (xp = panel + r * ld + 8 g; k-step ks starts 32 ks elements further).  The token fragments of k-step ks + 1 are
// requested before the products of k-step ks and nothing moves across a k-step: left to itself the scheduler hoists
// the LDS reads of ALL k-steps (8 x NT fragments = 256 registers at NT = 8) above the first product and spills.
template <typename T, int RB, int NT, int KS>
AMT_DEV void big_stage(f32x4 (&acc)[RB][NT], const Frag<T> (&wf)[KS][RB], const T* xp, int ld) {
  Frag<T> xf[2][NT];
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) xf[0][ti] = frag_row(xp + ti * 16 * ld);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 1 < KS) {
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) xf[(ks + 1) & 1][ti] = frag_row(xp + ti * 16 * ld + 32 * (ks + 1));
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) mma(acc[rb][ti], wf[ks][rb], xf[ks & 1][ti]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// fragment (row block rbk, k-step ks) of a fragment-packed weight copy [M/16][KS][64 lanes][8]
template <typename T>
AMT_DEV Frag<T> wfrag(const T* __restrict__ W, int rbk, int KS, int ks, int lane) {
  return frag_row(W + ((size_t)rbk * KS + ks) * 512 + lane * 8);
}

// LDS panel [TT][W] (row stride LD) -> columns [c0, c0 + W) of global rows n0 .. (row-major, ldg elements per row)
template <typename T, int TT, int W, int LD>
AMT_DEV void big_copy_out(const T* Ps, T* __restrict__ G, int ldg, int c0, long n0, long N, int tid) {
  constexpr int CPR = W / 8, TOTAL = TT * CPR;
#pragma unroll
  for (int i = 0; i < (TOTAL + 255) / 256; ++i) {
    const int id = tid + i * 256;
    const int p = id / CPR, c = id % CPR;
    if ((TOTAL % 256 == 0 || id < TOTAL) && n0 + p < N)
      chunk_store(G + (n0 + p) * ldg + c0 + 8 * c, chunk_load(Ps + p * LD + 8 * c));
  }
}

}  // namespace

// ------------------------------------------------------------------ transformer layer, second half, forward (large N)
//   h = x + ao . Wo^T ; hn = LN2(h) ; per slice s of 64 GLU columns: u = hn . W1[slice]^T + b1 ; g = gelu(u1) * u2 ;
//   y_acc += g . W2[:, slice]^T ; y = h + dropout(y_acc + b2)                                    model.py:369, 545-556
// Four waves, one per SIMD, each owns 64 of the 256 output features of the 256-wide stages (Wo, W2) and one (u1 | u2)
// row-block pair of every W1 slice, over ALL tokens of the tile: every weight fragment is loaded by exactly one wave and
// feeds NT = TT / 16 MFMAs.
// WT = 2: eight waves, two per SIMD - wave (wf = wave & 3, wt = wave >> 2) owns feature group wf over token half wt.  The two
// waves of a feature group request the same weight fragments within a few hundred cycles of each other (they meet at the
// slice barrier), so the second request finds the line in, or on its way into, the compute unit's L1; what the second
// wave buys is issue: one wave alone issues a vector instruction every 4 cycles, two alternate at 2, and one wave's GLU /
// LayerNorm epilogues run under the other's products (in-kernel stamps of the WT = 1 form: 45 of 107 thousand cycles per
// tile were epilogues with idle MFMA pipes, profiles/r03_notes.md).
template <typename T, int TT, bool SAVE, int WT, bool LEAN = false>
AMT_DEV void ffn_fwd_big_body(FfnFwdArgs& a) {
  a.seed = eff_seed(a.seed, a.seed_dev);
  constexpr int H = 256, I = 512, I2 = 1024, LDX = H + 8, SW = 64, LDG = SW + 8, NT = TT / 16 / WT, NS = I / SW;
  constexpr int KS1 = H / 32, KS2 = I / 32, TH = TT / WT;   // TH: tokens per wave / per 256-thread row-phase group
  extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
  using RT = Res<T>;                        // x, h, y: residual stream (common.h); same size as T, so the token panel holds either
  static_assert(sizeof(RT) == sizeof(T), "panels keep their size");
  typedef typename Vec4T<RT>::type RV4;
  T* As = reinterpret_cast<T*>(big_smem);   // [TT][LDX]     ao (T), then h (RT), then hn = LN2(h) (T), then y (RT)
  T* Gs = As + TT * LDX;                    // [2][TT][LDG]  g slices (double-buffered)
  __shared__ __attribute__((aligned(16))) float s_b1[I2], s_b2[H];

  const int tid_all = threadIdx.x, tid = tid_all & 255, lane = tid & 63;
  const int wave = (tid >> 6), wt = WT > 1 ? __builtin_amdgcn_readfirstlane(tid_all >> 8) : 0;   // wave: feature group
  const int r = lane & 15, g = lane >> 4;
  const long N = a.N, n0 = (long)blockIdx.x * TT;
  const T* Wo = reinterpret_cast<const T*>(a.Wo);
  const T* W1 = reinterpret_cast<const T*>(a.W1);
  const T* W2 = reinterpret_cast<const T*>(a.W2);
  const int mO = wave * 64, t0 = wt * TH;
  T* Ah = As + t0 * LDX;                    // this wave's / this row-phase group's half of the token panel
  const long nh = n0 + t0;

  // ---- requests that depend on nothing but the inputs: Wo fragments (first chunk), x in the accumulator layout, ao rows
  constexpr int KCO = ChainChunk<T, 4, H>::KC;
  Frag<T> wo[KCO][4];
  chain_load_w<T, 4, KCO, H>(wo, Wo, mO, 0, lane);
  RV4 hx[4][NT];                                  // x, then h = x + ao . Wo^T (as stored: RT), until the last epilogue
  {
    const RT* X = reinterpret_cast<const RT*>(a.X);
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {             // token block outer: the four loads of a block share its 128-byte lines
      const long row = nh + 16 * ti + r;
      const RT* xp = X + (row < N ? row : N - 1) * H + mO + 4 * g;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) hx[rb][ti] = ld4(xp + 16 * rb);
    }
  }
  chain_copy_in<T, TH, H, LDX>(Ah, reinterpret_cast<const T*>(a.AO), nh, N, tid);
  if (wt == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) s_b1[tid + 256 * j] = a.b1[tid + 256 * j];
    s_b2[tid] = a.b2[tid];
  }
  BIG_T0();
  __syncthreads();
  BIG_ACC(0);

  // ---- h = x + ao . Wo^T
  {
    f32x4 acc[4][NT];
    chain_zero(acc);
    chain_stage<T, 4, NT, KCO, H>(acc, wo, Wo, mO, As, LDX, t0, lane);
    __syncthreads();                              // every wave has read the ao panel: h goes in its place
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) {
        RV4 hv;                                   // h as stored (rounded to RT): what the residual adds at the end
#pragma unroll
        for (int i = 0; i < 4; ++i) hv[i] = from_f<RT>(acc[rb][ti][i] + to_f(hx[rb][ti][i]));
        hx[rb][ti] = hv;
        *reinterpret_cast<RV4*>(reinterpret_cast<RT*>(Ah) + (16 * ti + r) * LDX + mO + 16 * rb + 4 * g) = hv;
      }
  }
  BIG_ACC(1);
  // first W1 slice: in flight during the LayerNorm pass
  Frag<T> w1[KS1][2];
#pragma unroll
  for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
    for (int q = 0; q < 2; ++q) w1[ks][q] = wfrag(W1, 2 * wave + q, KS1, ks, lane);
  __syncthreads();
  if constexpr (SAVE) {
    chain_copy_out<RT, TH, H, LDX>(reinterpret_cast<RT*>(Ah), reinterpret_cast<RT*>(a.Hout), nh, N, tid);
    __syncthreads();
  }
  chain_ln_rows<T, TH, H, LDX>(reinterpret_cast<const RT*>(Ah), Ah, a.ln_w, a.ln_b, a.mean, a.rstd, nh, N, tid);
  __syncthreads();
  if constexpr (SAVE) chain_copy_out<T, TH, H, LDX>(Ah, reinterpret_cast<T*>(a.HN), nh, N, tid);
  BIG_ACC(2);

  // ---- slices
  f32x4 yacc[4][NT];
  chain_zero(yacc);
  // (a rolled loop on purpose: unrolled, the scheduler hoists the later slices' weight loads over the earlier slices'
  // products and the kernel spills - 256 + 256 registers and 200 bytes of scratch per lane)
#pragma unroll 1
  for (int s = 0; s < NS; ++s) {
    T* Gb = Gs + (s & 1) * TT * LDG;
    // the slice's W2 fragments: requested now, multiplied after the GLU epilogue
    // (LEAN, two workgroups per compute unit: requested behind stage 1 - 32 registers fewer are live across it, and the
    // other workgroup's instructions cover the shorter lead)
    Frag<T> w2[2][4];
    if constexpr (!LEAN) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) w2[kk][rb] = wfrag(W2, 4 * wave + rb, KS2, 2 * s + kk, lane);
    }
    // stage 1: (u1 | u2) rows 128 s + 32 wave .. + 32 of the permuted W1, K = 256
    f32x4 uacc[2][NT];
    chain_zero(uacc);
    BIG_ACC(3);
    big_stage<T, 2, NT, KS1>(uacc, w1, Ah + r * LDX + 8 * g, LDX);
    BIG_ACC(4);
    if constexpr (LEAN) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) w2[kk][rb] = wfrag(W2, 4 * wave + rb, KS2, 2 * s + kk, lane);
    }
    // next slice's W1 fragments: on their way during the epilogue and stage 2
    if (s + 1 < NS) {
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
        for (int q = 0; q < 2; ++q) w1[ks][q] = wfrag(W1, 8 * (s + 1) + 2 * wave + q, KS1, ks, lane);
    }
    // GLU epilogue -> g slice columns 16 wave + 4 g + i
    {
      const int m = 128 * s + 32 * wave + 4 * g;          // permuted row of u1; u2 sits 16 rows further
      float b1a[4], b1b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { b1a[i] = s_b1[m + i]; b1b[i] = s_b1[m + 16 + i]; }
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) {
        const int p = t0 + 16 * ti + r;
        float u1[4], u2[4], gl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u1[i] = to_f(from_f<T>(uacc[0][ti][i] + b1a[i]));
          u2[i] = to_f(from_f<T>(uacc[1][ti][i] + b1b[i]));
          gl[i] = gelu_tanh(u1[i]) * u2[i];
        }
        store4f(Gb + p * LDG + 16 * wave + 4 * g, gl);
        if constexpr (SAVE) {
          if (n0 + p < N) {
            T* up = reinterpret_cast<T*>(a.U) + (n0 + p) * I2 + m;
            store4f(up, u1);
            store4f(up + 16, u2);
          }
        }
      }
    }
    BIG_ACC(5);
    __syncthreads();      // the slice is complete; the OTHER buffer is free again (its readers passed this barrier's predecessor)
    if constexpr (SAVE) big_copy_out<T, TH, SW, LDG>(Gb + t0 * LDG, reinterpret_cast<T*>(a.G), I, SW * s, nh, N, tid);
    // stage 2: y += W2[:, 64 s .. 64 s + 64) . g slice
    BIG_ACC(6);
    big_stage<T, 4, NT, 2>(yacc, w2, Gb + (t0 + r) * LDG + 8 * g, LDG);
    BIG_ACC(7);
  }

  // ---- y = h + dropout(y_acc + b2) -> the hn panel (dead: every stage 1 is done behind the last slice barrier)
  auto y_epilogue = [&](auto DROP) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int m = mO + 16 * rb + 4 * g;
      float bb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) bb[i] = s_b2[m + i];
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) {
        const int p = t0 + 16 * ti + r;
        float y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = yacc[rb][ti][i] + bb[i];
          if (decltype(DROP)::value) {
            const bool keep = rng_keep(a.seed, a.stream, (uint32_t)(n0 + p) * (uint32_t)H + (uint32_t)(m + i), a.drop_thresh24);
            v = keep ? v * a.inv_keep : 0.f;
          }
          y[i] = v + to_f(hx[rb][ti][i]);
        }
        store4f(reinterpret_cast<RT*>(As) + p * LDX + m, y);
      }
    }
  };
  if (a.drop_thresh24) y_epilogue(std::true_type{}); else y_epilogue(std::false_type{});   // one uniform branch, not one per element
  BIG_ACC(8);
  __syncthreads();
  chain_copy_out<RT, TH, H, LDX>(reinterpret_cast<RT*>(Ah), reinterpret_cast<RT*>(a.Y), nh, N, tid);
  BIG_ACC(9);
}

template <typename T, int TT, bool SAVE, int WT>
__global__ __launch_bounds__(256 * WT, WT) void ffn_fwd_big_kernel(FfnFwdArgs a) { ffn_fwd_big_body<T, TT, SAVE, WT>(a); }
// two independent four-wave workgroups per compute unit (256 registers each)
template <typename T, int TT, bool SAVE>
__global__ __launch_bounds__(256, 2) void ffn_fwd_big_occ2_kernel(FfnFwdArgs a) { ffn_fwd_big_body<T, TT, SAVE, 1, true>(a); }

template <typename T, bool SAVE, int TT>
static void ffn_fwd_big_occ2_launch(hipStream_t s, const FfnFwdArgs& a) {
  constexpr size_t lds = sizeof(T) * (size_t)TT * ((256 + 8) + 2 * (64 + 8));
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_fwd_big_occ2_kernel<T, TT, SAVE>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
  (void)once;
  hipLaunchKernelGGL((ffn_fwd_big_occ2_kernel<T, TT, SAVE>), dim3((a.N + TT - 1) / TT), dim3(256), lds, s, a);
}

template <typename T, bool SAVE, int TT, int WT>
static void ffn_fwd_big_launch(hipStream_t s, const FfnFwdArgs& a) {
  constexpr size_t lds = sizeof(T) * (size_t)TT * ((256 + 8) + 2 * (64 + 8));
  static_assert(lds + 5 * 1024 + 256 <= 160 * 1024, "LDS budget (dynamic panels + the two bias arrays)");
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_fwd_big_kernel<T, TT, SAVE, WT>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
  (void)once;
  hipLaunchKernelGGL((ffn_fwd_big_kernel<T, TT, SAVE, WT>), dim3((a.N + TT - 1) / TT), dim3(256 * WT), lds, s, a);
}

bool ffn_fwd_big(hipStream_t s, DType dt, const FfnFwdArgs& a, bool save) {
  const char* e = getenv("AMT_BIG");      // read on every call: the equality test switches it between contexts
  // 0: off; 1: four waves per tile; 2: eight; default: eight on 128-token tiles (measured at 112 000 tokens: 188 against
  // 196 us), four on 64-token tiles (16 000 tokens, training form: 43.8 against 51.3 us; the short-tile chain: 52 us)
  // 5: two independent four-wave workgroups per compute unit on 64-token tiles (bf16, from 49 152 tokens on: 175 against 187 us;
  // at 16 000 tokens there is one tile per compute unit and the leaner register budget only costs: 8.99 against 8.83 ms/step;
  // 32-token tiles, two per compute unit, at 16 000 / 32 000 tokens: 41.4 / 76 us against 38.4 / 72 us for mode 1)
  const int mode = e ? atoi(e) : (a.N >= 49152 ? (dt == DT_F32 ? 2 : 5) : 1);
  if (mode == 0 || a.W1 == nullptr || a.N < 8192) return false;
  const double N = a.N, es = dt == DT_F32 ? 4 : 2;
  ProfScope ps(s, "ffn_fwd_big", 2.0 * N * (256.0 * 256 + 1024.0 * 256 + 256.0 * 512),
               (N * 256.0 * (save ? 6 : 3) + (save ? N * (1024.0 + 512) : 0.0) + 256.0 * 256 + 1024.0 * 256 + 256.0 * 512) * es);
  // 128-token tiles from ~2 tiles per compute unit on, 64-token tiles below (16 000 tokens = 32 clips per GPU: 250 tiles)
  const bool long_tile = a.N >= 49152;
  if (mode == 5 && dt != DT_F32) {
    if (save) ffn_fwd_big_occ2_launch<bf16, true, 64>(s, a); else ffn_fwd_big_occ2_launch<bf16, false, 64>(s, a);
    return true;
  }

#define AMT_FFN_BIG(T_, TL_, TS_)                                                                                   \
  do {                                                                                                               \
    if (mode == 1) {                                                                                                 \
      if (long_tile) { if (save) ffn_fwd_big_launch<T_, true, TL_, 1>(s, a); else ffn_fwd_big_launch<T_, false, TL_, 1>(s, a); } \
      else { if (save) ffn_fwd_big_launch<T_, true, TS_, 1>(s, a); else ffn_fwd_big_launch<T_, false, TS_, 1>(s, a); }           \
    } else {                                                                                                         \
      if (long_tile) { if (save) ffn_fwd_big_launch<T_, true, TL_, 2>(s, a); else ffn_fwd_big_launch<T_, false, TL_, 2>(s, a); } \
      else { if (save) ffn_fwd_big_launch<T_, true, TS_, 2>(s, a); else ffn_fwd_big_launch<T_, false, TS_, 2>(s, a); }           \
    }                                                                                                                \
  } while (0)
  if (dt == DT_F32) AMT_FFN_BIG(float, 64, 32);
  else AMT_FFN_BIG(bf16, 128, 64);
#undef AMT_FFN_BIG
  return true;
}

}  // namespace amt
