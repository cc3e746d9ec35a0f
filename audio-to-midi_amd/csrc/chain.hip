// Token-tile chain kernels: one launch per operator chain (see chain.h for the rationale).
//
//   block_fwd_fused   ConvNeXt Block forward, C >= 16                       model.py:160-167
//                     dwconv k7 -> LN -> 1x1 (C -> 2C) -> GELU -> 1x1 (2C -> C) -> gamma -> + x
//
// Every tensor the backward pass reads is written exactly as the unfused launch sequence
// (dwln_fwd, gemm_nt<gelu>, gemm_nt<scale_resid>) writes it, with the same roundings and the same
// accumulation order, so both paths agree bit for bit (tests/test_model_gpu.py).
#include <hip/hip_runtime.h>
// Phase timestamps for the micro-benchmark (scratch/chain_bench.hip): compiled out of the library.
#ifdef AMT_CHAIN_TIMING
__device__ unsigned long long amt_stamps[128][256][12];   // [launch slot][workgroup][phase]; 8 / 9: team barrier entered / left (100 MHz clock)
// rider workgroups (wgrad.h: tn_body) stamp cycle counts into slot 127, indexed by blockIdx.x % 256 (last launch wins)
// (AMT_WG_FILTER = M * 1000 + Kp restricts the stamps to the riders of one problem shape, e.g. 256128: pw1 of a C = 128 Block)
#ifndef AMT_WG_FILTER
#define AMT_WG_FILTER 0
#endif
#define AMT_WG_ON (threadIdx.x == 0 && (AMT_WG_FILTER == 0 || a.M * 1000 + a.Kp == AMT_WG_FILTER))
#define AMT_WG_STAMP(k) do { if (AMT_WG_ON) amt_stamps[127][blockIdx.x & 255][k] = __builtin_readcyclecounter(); } while (0)
#define AMT_WG_NOTE(k, v) do { if (AMT_WG_ON) amt_stamps[127][blockIdx.x & 255][k] = (unsigned long long)(v) + 1; } while (0)
#endif
#include "chain.h"
#include "kernels.h"
#include "prof.h"
#include "rowops.h"
#include "team.h"
#include "wgrad.h"

#ifdef AMT_CHAIN_TIMING
#define AMT_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 256) amt_stamps[a.dbg_slot & 127][blockIdx.x][k] = __builtin_readcyclecounter(); } while (0)
#define AMT_STAMP_RT(k) do { if (threadIdx.x == 0 && blockIdx.x < 256) amt_stamps[a.dbg_slot & 127][blockIdx.x][k] = wall_clock64(); } while (0)
extern "C" int amt_debug_read_stamps(unsigned long long* out) {   // [128][256][10]; slots 6 - 9 = 100 MHz real-time clock
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(amt_stamps), sizeof(amt_stamps));
}
#define AMT_DBG_IF(bit) if (!(a.dbg_slot & (bit)))   // timing experiments: switch a piece of the tile body off (wrong results)
#else
#define AMT_DBG_IF(bit)
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

// SAVE = false (inference): nothing but y leaves the workgroup and the tile is 2-4x longer, so the
// weights are streamed once per 64-512 tokens instead of once per 16-128.
// One tile of TT rows starting at global row n0; rows >= Nlim are padding (Nlim = B * L for the one-launch-per-Block
// form, the end of the tile's clip for the team form, whose tiles never straddle clips).  TEAM: X was written by
// another workgroup of this XCD team earlier in the same launch, so it is read with sc1 loads (team.h).
template <typename T, int C, int TT, bool SAVE, bool TEAM>
AMT_DEV void block_fwd_tile(const BlockFwdArgs& a_in, const long n0, const long Nlim, unsigned char* chain_smem) {
  // Every tensor is in global memory.  The one-launch-per-Block kernel knows that from its kernel arguments; in the team
  // form the pointers come out of LDS, and as generic pointers they would turn every access into a flat_* instruction
  // (which also counts in lgkmcnt: each LDS wait would then wait for the tile's fire-and-forget stores too).
  BlockFwdArgs a = a_in;
  a.X = team_global(a_in.X); a.wdw = team_global(a_in.wdw); a.bdw = team_global(a_in.bdw); a.lnw = team_global(a_in.lnw);
  a.lnb = team_global(a_in.lnb); a.W1 = team_global(a_in.W1); a.b1 = team_global(a_in.b1); a.W2 = team_global(a_in.W2);
  a.b2 = team_global(a_in.b2); a.gamma = team_global(a_in.gamma); a.LN = team_global(a_in.LN); a.mean = team_global(a_in.mean);
  a.rstd = team_global(a_in.rstd); a.U = team_global(a_in.U); a.Hh = team_global(a_in.Hh); a.Z = team_global(a_in.Z);
  a.Y = team_global(a_in.Y);
  constexpr int H2 = 2 * C, LDX = C + 8, LDH = H2 + 8;
  constexpr int LPR = C / 8, RPP = 256 / LPR;   // lanes per row, rows per pass of the row phase
  static_assert(TT % RPP == 0, "row phase covers the tile in whole passes");
  using S1 = ChainSplit<H2, TT>;
  using S2 = ChainSplit<C, TT>;
  constexpr int KC1 = ChainChunk<T, S1::RB, C>::KC, KC2 = ChainChunk<T, S2::RB, H2>::KC;
  using RT = Res<T>;                                   // the Block's input x and output y: residual stream (common.h)
  static_assert(sizeof(RT) == sizeof(T), "panels keep their size");
  T* Xs = reinterpret_cast<T*>(chain_smem);            // [TT][LDX]  LN output
  RT* Rs = reinterpret_cast<RT*>(Xs + TT * LDX);       // [TT][LDX]  block input (residual), then y
  T* Hs = reinterpret_cast<T*>(Rs + TT * LDX);         // [TT][LDH]  gelu(u)
  T* Us = Hs + TT * LDH;                               // [TT][LDH]  u (staged for a coalesced copy-out; SAVE only)

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;   // (& 255: the team form runs two tiles per workgroup, 256 threads each)
  const int r = lane & 15, g = lane >> 4;
  const long N = Nlim;
  const RT* X = reinterpret_cast<const RT*>(a.X);
  const __amdgpu_buffer_rsrc_t xrs = team_rsrc(a.X);
  const T* W1 = reinterpret_cast<const T*>(a.W1);
  const T* W2 = reinterpret_cast<const T*>(a.W2);
  const int m1 = (wave % S1::WF) * S1::RB * 16, t1 = (wave / S1::WF) * S1::NT * 16;
  const int m2 = (wave % S2::WF) * S2::RB * 16, t2 = (wave / S2::WF) * S2::NT * 16;

#ifdef AMT_CHAIN_TIMING
  // tile-start stamps are parked in LDS and written out at the end of the tile: no vector-memory instruction at the start
  __shared__ unsigned long long s_dbg_start[2];
  if (threadIdx.x == 0) { s_dbg_start[0] = wall_clock64(); s_dbg_start[1] = __builtin_readcyclecounter(); }
#endif

  // ---- what depends on nothing is requested first: the stage-1 weights (chunk 0), biases, gamma.
  // Their round trip overlaps the activation loads of the row phase.  The stage-2 weights follow
  // after the row phase (in flight during stage 1): holding both sets across the row phase pushed the
  // kernel past 256 registers = one wave per SIMD, and such launches paid ~6 us more between
  // dependent kernels in the training step (profiles/r01_notes.md).
  uint32_t warm = 0;   // cooperative L2 warm-up (chain.h); the team form requests the weights one Block ahead instead
  // (SAVE only: the long inference grids find the weights in L2 from their second round of tiles on, and the six
  // serial round trips of the warm-up stood in front of EVERY tile: ~9 of 32 thousand cycles per tile at 224 000 positions)
  if constexpr (C >= 128 && !TEAM && SAVE) {
    const int q = blockIdx.x >> 3, ns = max(1, (int)(gridDim.x >> 3));
    constexpr size_t WBYTES = chain_packed_elems(H2, C) * sizeof(T);
    constexpr int MAXLD = (int)((WBYTES + 8 * 4096 - 1) / (8 * 4096)) + 1;
    chain_l2_warm<MAXLD>(W1, WBYTES, q, ns, tid, warm);
    chain_l2_warm<MAXLD>(W2, WBYTES, q, ns, tid, warm);
  }
  Frag<T> wa[KC1][S1::RB], wb[KC2][S2::RB];
  AMT_DBG_IF(4096) chain_load_w<T, S1::RB, KC1, C>(wa, W1, m1, 0, lane);
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
    constexpr int NPASS = TT / RPP;
    // The seven tap rows of pass pp + 1 are requested before pass pp is evaluated (raw chunks, converted at their use):
    // one pass after the other, each pass waited for its own seven loads - eight L2 round trips in series per 128-row
    // inference tile (found in the ISA: `s_waitcnt vmcnt(0)` in front of every pass).
    Chunk8<RT> xa[7], xb[7];
    auto issue = [&](int pp, Chunk8<RT> (&xc)[7]) {
      const long row = n0 + pp * RPP + prow;
      const long rc = row < N ? row : N - 1;
      const int b = (int)(rc / a.L), l = (int)(rc % a.L);
      const long rowbase = (long)b * a.L;
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const int ll = l + k - 3;
        const int lc = ll < 0 ? 0 : (ll >= a.L ? a.L - 1 : ll);
        if constexpr (TEAM) {
          float t[8];
          team_load8f<TEAM>(xrs, X, (rowbase + lc) * C + sub * 8, t);
          store8f(reinterpret_cast<RT*>(&xc[k]), t);
        } else {
          xc[k] = chunk_load(X + (rowbase + lc) * C + sub * 8);
        }
      }
    };
    auto evaluate = [&](int pp, const Chunk8<RT> (&xc)[7]) {
      const int p = pp * RPP + prow;
      const long row = n0 + p;
      const bool valid = row < N;
      const long rc = valid ? row : N - 1;
      const int l = (int)(rc % a.L);
      float d[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) d[i] = bv[i];
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const int ll = l + k - 3;
        const float m = (ll >= 0 && ll < a.L) ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] += wk[k][i] * (chunk_get(xc[k], i) * m);
      }
      chunk_store(Rs + p * LDX + sub * 8, xc[3]);
      float mu, rs;
      ln_stats<8, LPR>(d, mu, rs);
      float y[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = ln_apply(d[i], rs, wv[i], lb[i]);
      store8f(Xs + p * LDX + sub * 8, y);
#ifdef AMT_CHAIN_TIMING
      if (!(a.dbg_slot & 1024))
#endif
      if (SAVE && valid) {
        store8f(reinterpret_cast<T*>(a.LN) + row * C + sub * 8, y);
        if (sub == 0) { a.mean[row] = mu; a.rstd[row] = rs; }
      }
    };
    issue(0, xa);
#pragma unroll
    for (int pp = 0; pp < NPASS; ++pp) {
      if (pp + 1 < NPASS) issue(pp + 1, (pp & 1) ? xa : xb);
      evaluate(pp, (pp & 1) ? xb : xa);
    }
  }
  AMT_DBG_IF(4096) chain_load_w<T, S2::RB, KC2, H2>(wb, W2, m2, 0, lane);
  __syncthreads();
  AMT_STAMP(1);

  // ---- stage 1: u = LN . W1^T + b1 ; h = gelu(u)
  {
    f32x4 acc[S1::RB][S1::NT];
    chain_zero(acc);
    AMT_DBG_IF(4096) chain_stage<T, S1::RB, S1::NT, KC1, C>(acc, wa, W1, m1, Xs, LDX, t1, lane);
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
          // h = gelu(u) from the UNROUNDED u (round 4); u is rounded only where it is stored for the backward pass
          // (gemm_nt<gelu> does the same: the two forms still agree bit for bit)
          u[i] = acc[rb][ti][i] + bb1[rb][i];
          h[i] = gelu_tanh(u[i]);
        }
        store4f(Hs + p * LDH + m, h);
        if constexpr (SAVE) store4f(Us + p * LDH + m, u);
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
  if constexpr (SAVE) {
    chain_copy_out<T, TT, H2, LDH>(Us, reinterpret_cast<T*>(a.U), n0, N, tid);
    chain_copy_out<T, TT, H2, LDH>(Hs, reinterpret_cast<T*>(a.Hh), n0, N, tid);
  }
  }

  // ---- stage 2: z = h . W2^T + b2 ; y = x + gamma * z
  {
    f32x4 acc[S2::RB][S2::NT];
    chain_zero(acc);
    AMT_DBG_IF(4096) chain_stage<T, S2::RB, S2::NT, KC2, H2>(acc, wb, W2, m2, Hs, LDH, t2, lane);
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
          // z enters y UNROUNDED since round 4: nothing stores z any more (dgamma comes from the pw2 weight gradient), so the
          // rounding to T "as stored" had become a pure loss (profiles/r04_notes.md: max |dlogit| 1.96e-2 -> 1.58e-2 in the
          // rounding study).  gemm_nt<scale_resid> and block_small_fwd do the same: the three forms still agree bit for bit.
          z[i] = acc[rb][ti][i] + bb2[rb][i];
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
  if constexpr (SAVE) {     // (the training path passes no Z: dgamma comes from the pw2 weight gradient, head.hip: cnn_gamma_finish)
    if (a.Z != nullptr) chain_copy_out<T, TT, C, LDX>(Xs, reinterpret_cast<T*>(a.Z), n0, N, tid);
  }
  chain_copy_out<RT, TT, C, LDX>(Rs, reinterpret_cast<RT*>(a.Y), n0, N, tid);
  if constexpr (!TEAM) {   // the LayerNorm the next operator opens with, on the y rows while they are here (kernels.h: TailLn)
    if (a.tail.w != nullptr) {
      bool res_out = false;      // the final norm (C = 256 only) writes the transformer's stream: Res
      if constexpr (C == 256) res_out = a.tail.y_res != 0;
      if (res_out) chain_ln_rows_out<T, RT, TT, C, LDX>(Rs, reinterpret_cast<RT*>(a.tail.Y), a.tail.w, a.tail.b, a.tail.mean, a.tail.rstd, n0, N, tid);
      else chain_ln_rows_out<T, T, TT, C, LDX>(Rs, reinterpret_cast<T*>(a.tail.Y), a.tail.w, a.tail.b, a.tail.mean, a.tail.rstd, n0, N, tid);
    }
  }
  if (warm == 0x9e3779b9u && a.L < 0) a.mean[0] = 0.f;   // keeps the warm-up loads alive; never true
  AMT_STAMP(5);
  AMT_STAMP_RT(7);
#ifdef AMT_CHAIN_TIMING
  if (threadIdx.x == 0 && blockIdx.x < 256) { amt_stamps[a.dbg_slot & 127][blockIdx.x][6] = s_dbg_start[0]; amt_stamps[a.dbg_slot & 127][blockIdx.x][0] = s_dbg_start[1]; }
#endif
}

template <typename T, int C, int TT, bool SAVE>
__global__ __launch_bounds__(256) void block_fwd_fused_kernel(BlockFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  block_fwd_tile<T, C, TT, SAVE, false>(a, (long)blockIdx.x * TT, (long)a.B * a.L, chain_smem);
}

// Inference, two independent workgroups per compute unit (256 registers each): one workgroup's row phase / GELU epilogue
// runs under the other's products without a barrier tying them together (profiles/r03_notes.md section 7).
template <typename T, int C, int TT>
__global__ __launch_bounds__(256, 2) void block_fwd_occ2_kernel(BlockFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  block_fwd_tile<T, C, TT, false, false>(a, (long)blockIdx.x * TT, (long)a.B * a.L, chain_smem);
}

// ------------------------------------------------------------------ Block forward, a whole run of Blocks per launch
// XCD-team form (team.h): the n Blocks of a CNN stage in ONE launch.  Clip b belongs to one XCD's team; Block j of
// every clip of the team is evaluated tile by tile (the same tile body as above), then the team meets at its barrier:
// the halo rows of Block j + 1 come from the neighbouring workgroups' stores, which are in the XCD's L2.
// NG wave groups of 256 threads per workgroup, each on a tile of its own (own LDS panels): with one workgroup per compute
// unit, two 16-row tiles side by side hide each other's latencies the way two co-resident workgroups of the
// one-launch-per-Block kernel do (one 32-row tile per 256 threads ran 8.2 us against 6.1 us, profiles/r02_notes.md).
template <typename T, int C, int TT, int NG>
__global__ __launch_bounds__(256 * NG) void block_run_fwd_kernel(BlockRunFwdArgs r) {
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem_all[];
  constexpr size_t LDS_TILE = sizeof(T) * (size_t)TT * (2 * (C + 8) + 2 * (2 * C + 8));
  const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  unsigned char* chain_smem = chain_smem_all + (size_t)grp * LDS_TILE;
  // The kernel-argument segment lives in HOST memory (HIP default): a scalar load from it is a PCIe round trip.  An
  // ordinary kernel pays that once, before its first instruction; here all items are copied into LDS up front.
  __shared__ BlockRunFwdItem s_it[BLOCK_RUN_MAX];
#if defined(__HIP_DEVICE_COMPILE__)
  {
    typedef const __attribute__((address_space(4))) unsigned long long* KernArg64;
    KernArg64 ka = (KernArg64)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int W0 = (int)(offsetof(BlockRunFwdArgs, it) / 8), WPI = (int)(sizeof(BlockRunFwdItem) / 8);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(s_it);
    for (int i = threadIdx.x; i < r.n * WPI; i += 256 * NG) dst[i] = ka[W0 + i];
  }
#endif
  __shared__ __attribute__((aligned(16))) unsigned char s_warm[1024 * 4 * NG];   // landing zone of team_warm, never read
  constexpr size_t WBYTES = chain_packed_elems(2 * C, C) * sizeof(T);    // == chain_packed_elems(C, 2 * C): W1 and W2
  const TeamTicket ticket = team_register(r.st);   // (its barrier orders the copy above before the first read below)
  {   // the first Block's weights, while the other workgroups are still arriving (assumes the usual 8 even teams: speed only)
    const int guess = (int)gridDim.x / TEAM_MAX > 0 ? (int)gridDim.x / TEAM_MAX : 1;
    team_warm(team_uniform(s_it[0].W1), WBYTES, ticket.rank % guess, guess, s_warm);
    team_warm(team_uniform(s_it[0].W2), WBYTES, ticket.rank % guess, guess, s_warm);
  }
  Team t = team_assemble(r.st, ticket);
  if (t.ok) {
    const int tiles_per_clip = (r.L + TT - 1) / TT;
#pragma unroll 1
    for (int j = 0; j < r.n; ++j) {
      BlockFwdArgs a;
      const BlockRunFwdItem& it = s_it[j];
      a.X = j == 0 ? r.X0 : team_uniform(s_it[j - 1].Y);
      a.wdw = team_uniform(it.wdw); a.bdw = team_uniform(it.bdw); a.lnw = team_uniform(it.lnw); a.lnb = team_uniform(it.lnb);
      a.W1 = team_uniform(it.W1); a.b1 = team_uniform(it.b1); a.W2 = team_uniform(it.W2); a.b2 = team_uniform(it.b2);
      a.gamma = team_uniform(it.gamma); a.LN = team_uniform(it.LN); a.mean = team_uniform(it.mean); a.rstd = team_uniform(it.rstd);
      a.U = team_uniform(it.U); a.Hh = team_uniform(it.Hh); a.Z = team_uniform(it.Z); a.Y = team_uniform(it.Y);
      a.B = r.B; a.L = r.L; a.C = C; a.dbg_slot = r.dbg_base + j;
      if (j + 1 < r.n) {   // the next Block's weights depend on nothing: into L2 now, each line requested by one member
        team_warm(team_uniform(s_it[j + 1].W1), WBYTES, t.rank, t.size, s_warm);
        team_warm(team_uniform(s_it[j + 1].W2), WBYTES, t.rank, t.size, s_warm);
      }
#pragma unroll 1
      for (int clip = t.index; clip < r.B; clip += t.n_teams)
#pragma unroll 1
        for (int base = t.rank * NG; base < tiles_per_clip; base += t.size * NG) {
          // opaque copies: without them the compiler strength-reduces every address of the tile body over these
          // two loops (dozens of 64-bit induction pointers in VGPRs) and hoists a Block's weight / bias loads out of
          // the tile loop, which keeps them live across the whole body
          int cl = clip, tl = base;
          BlockFwdArgs at = a;
          asm volatile("" : "+s"(cl), "+s"(tl), "+s"(at.X), "+s"(at.wdw), "+s"(at.bdw), "+s"(at.lnw), "+s"(at.lnb), "+s"(at.W1),
                       "+s"(at.b1), "+s"(at.W2), "+s"(at.b2), "+s"(at.gamma));
          tl += grp;
          // a wave group without a tile of its own (odd tile count) runs the body on an empty tile: every row is padding,
          // nothing is stored, and it keeps meeting the other group at the body's workgroup barriers
          const long nlim = (long)(cl + 1) * r.L;
          const long n0 = tl < tiles_per_clip ? (long)cl * r.L + (long)tl * TT : nlim;
          block_fwd_tile<T, C, TT, true, true>(at, n0, nlim, chain_smem);
          __syncthreads();   // the tile's LDS panels are rewritten by the next tile
        }
      if (j + 1 < r.n && !team_barrier(r.st, t)) break;
    }
  }
  team_leave(r.st, t);
}

// Inference tile length.  A chain workgroup streams the Block's 4 C^2 weight elements once per TILE, so the flops per
// weight byte it pulls from L2 equal the tile length: at ~28 B/clk per compute unit (profiles/r01_notes.md) the MFMA
// pipes are fed only from ~128 tokens per tile on (4096 / TT B/clk at the full MFMA rate).  The longest tile whose
// panels fit LDS: 128 tokens at C = 128 (137 KiB), 256 at C = 64; AMT_BLOCK_INF_LONG=0 restores round 2's 64 / 128.
template <int C, typename T> struct BlockInferTile {
  static constexpr int SHORT = BlockTile<C>::TT * (sizeof(T) == 4 || C == 256 ? 2 : 4);
  static constexpr int LONG = (C == 128 || C == 64 || (C == 256 && sizeof(T) == 2)) ? 2 * SHORT : SHORT;
};
template <typename T, int C, bool SAVE, int TT>
static void block_fwd_fused_launch_tt(hipStream_t s, const BlockFwdArgs& a) {
  constexpr size_t lds = sizeof(T) * (size_t)TT * (2 * (C + 8) + (SAVE ? 2 : 1) * (2 * C + 8));
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool once = (lds > 64 * 1024
                          ? (void)hipFuncSetAttribute(reinterpret_cast<const void*>(block_fwd_fused_kernel<T, C, TT, SAVE>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                          : (void)0,
                      true);
  (void)once;
  const long N = (long)a.B * a.L;
  const int grid = (int)((N + TT - 1) / TT);
  hipLaunchKernelGGL((block_fwd_fused_kernel<T, C, TT, SAVE>), dim3(grid), dim3(256), lds, s, a);
}
template <typename T, int C, int TT>
static void block_fwd_occ2_launch(hipStream_t s, const BlockFwdArgs& a) {
  constexpr size_t lds = sizeof(T) * (size_t)TT * (2 * (C + 8) + (2 * C + 8));
  static_assert(2 * lds <= 160 * 1024, "two workgroups per compute unit");
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(block_fwd_occ2_kernel<T, C, TT>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
  (void)once;
  const long N = (long)a.B * a.L;
  hipLaunchKernelGGL((block_fwd_occ2_kernel<T, C, TT>), dim3((int)((N + TT - 1) / TT)), dim3(256), lds, s, a);
}
template <typename T, int C, bool SAVE>
static void block_fwd_fused_launch(hipStream_t s, const BlockFwdArgs& a) {
  if constexpr (SAVE) {
    if constexpr (sizeof(T) == 2 && C == 128) {
      // 64-row training tiles for C = 128 from 16 000 rows on (one workgroup per compute unit, a quarter of the weight
      // streams): 32 clips per GPU 7.859 -> 7.822 ms, 30.3 -> 28.3 us per launch (32-row tiles: 7.900).  AMT_BLOCK_TRAIN_TT=16 | 32 | 64.
      // Rows are independent: the outputs are the 16-row form's bit for bit.
      static const int tt_env = getenv("AMT_BLOCK_TRAIN_TT") ? atoi(getenv("AMT_BLOCK_TRAIN_TT")) : 64;
      if (tt_env == 32 && (long)a.B * a.L >= 16000) { block_fwd_fused_launch_tt<T, C, true, 32>(s, a); return; }
      if (tt_env == 64 && (long)a.B * a.L >= 16000) { block_fwd_fused_launch_tt<T, C, true, 64>(s, a); return; }
    }
    block_fwd_fused_launch_tt<T, C, true, BlockTile<C>::TT>(s, a);
  } else {
    using IT = BlockInferTile<C, T>;
    // AMT_BLOCK_OCC2: bit mask of the widths whose inference form runs as two independent workgroups per compute unit on
    // half-length tiles (1: C = 16, 2: C = 64, 4: C = 128, 8: C = 256)
    if constexpr (sizeof(T) == 2 && (C == 16 || C == 64 || C == 128 || C == 256)) {
      // (default: all but C = 256, whose half-length form spills; measured at 448 windows x 5 s, profiles/r03_notes.md section 7:
      // C = 128 alone 9.62 -> 9.08 ms per pass, C = 16 / 64 -0.1 ms each, C = 256 +0.5 ms.  Read on every call: the equality test switches it)
      const char* oe = getenv("AMT_BLOCK_OCC2");
      const int occ2 = oe ? atoi(oe) : 7;
      constexpr int bit = C == 16 ? 1 : (C == 64 ? 2 : (C == 128 ? 4 : 8));
      constexpr int TT2 = IT::LONG / 2;
      if ((occ2 & bit) && (long)a.B * a.L >= 512L * TT2) { block_fwd_occ2_launch<T, C, TT2>(s, a); return; }
    }
    // AMT_BLOCK_INF_LONG: bit mask of the widths that take the long tile (1: C = 64, 2: C = 128, 4: C = 256; default all)
    static const int long_mask = getenv("AMT_BLOCK_INF_LONG") ? atoi(getenv("AMT_BLOCK_INF_LONG")) : 7;
    const bool long_tiles = (long_mask & (C == 64 ? 1 : (C == 128 ? 2 : 4))) != 0;
    // short clips (few tiles per compute unit) keep the short tile: the long one would leave compute units idle
    if (IT::LONG != IT::SHORT && long_tiles && (long)a.B * a.L >= 256L * IT::LONG) block_fwd_fused_launch_tt<T, C, false, IT::LONG>(s, a);
    else block_fwd_fused_launch_tt<T, C, false, IT::SHORT>(s, a);
  }
}

static int team_grid() {   // one workgroup per compute unit, on the full 8-XCD chip only
  static int cus = [] {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    if (const char* e = getenv("AMT_TEAM_GRID")) return atoi(e);   // experiments: fewer workgroups than compute units
    return p.multiProcessorCount;
  }();
  return cus;
}

TeamState* team_state_create() {
  TeamState* st = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&st), sizeof(TeamState)) != hipSuccess) return nullptr;
  if (hipMemset(st, 0, sizeof(TeamState)) != hipSuccess) { (void)hipFree(st); return nullptr; }
  return st;
}

int team_state_timeout(const TeamState* st) {
  unsigned v = 0;
  if (hipMemcpy(&v, reinterpret_cast<const char*>(st) + offsetof(TeamState, timeout), sizeof(v), hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  return (int)v;
}

template <typename T, int C, int TT, int NG>
static void block_run_fwd_launch_tt(hipStream_t s, const BlockRunFwdArgs& r, int grid) {
  constexpr size_t lds = NG * sizeof(T) * (size_t)TT * (2 * (C + 8) + 2 * (2 * C + 8));
  static_assert(lds <= 150 * 1024, "LDS budget");
  static bool once = (lds > 64 * 1024
                          ? (void)hipFuncSetAttribute(reinterpret_cast<const void*>(block_run_fwd_kernel<T, C, TT, NG>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                          : (void)0,
                      true);
  (void)once;
  hipLaunchKernelGGL((block_run_fwd_kernel<T, C, TT, NG>), dim3(grid), dim3(256 * NG), lds, s, r);
}

// How a clip's rows are shared by the ~grid / 8 workgroups of one XCD: two wave groups per workgroup, each on its own
// tile of the one-launch-per-Block length (a 10-s clip at C = 128: 63 tiles of 16 rows for 32 x 2 groups), as long as
// two tiles' LDS panels fit; otherwise one group.  Results do not depend on the choice (rows are independent).
template <typename T, int C>
static void block_run_fwd_launch(hipStream_t s, const BlockRunFwdArgs& r, int grid) {
  constexpr int TT = BlockTile<C>::TT;
  // (C = 256: two groups leave 256 registers per wave and the body spills)
  constexpr bool fits2 = C <= 128 && 2 * sizeof(T) * (size_t)TT * (2 * (C + 8) + 2 * (2 * C + 8)) <= 150 * 1024;
  static const int ng_env = getenv("AMT_TEAM_NG") ? atoi(getenv("AMT_TEAM_NG")) : 0;   // experiments: force 1 or 2
  if constexpr (fits2) {
    if (ng_env != 1) { block_run_fwd_launch_tt<T, C, TT, 2>(s, r, grid); return; }
  }
  block_run_fwd_launch_tt<T, C, TT, 1>(s, r, grid);
}

bool block_run_fwd(hipStream_t s, DType dt, const BlockRunFwdArgs& r) {
  const int grid = team_grid();
  // the team form needs every workgroup resident at once (one per CU) and 32-bit byte offsets into X
  if (grid < 8 || grid > TEAM_MAX * TEAM_MAX_MEMBERS || r.st == nullptr || r.n < 2 || r.n > BLOCK_RUN_MAX) return false;
  const double N = (double)r.B * r.L, es = dt == DT_F32 ? 4 : 2;
  if (N * r.C * es >= 2147483648.0) return false;
  static const char* const names[] = {"block_run_fwd<16>", "block_run_fwd<32>", "block_run_fwd<64>", "block_run_fwd<128>", "block_run_fwd<256>"};
  const char* name = r.C == 16 ? names[0] : r.C == 32 ? names[1] : r.C == 64 ? names[2] : r.C == 128 ? names[3] : names[4];
  ProfScope ps(s, name, r.n * 2.0 * N * (7.0 * r.C + 4.0 * r.C * r.C), r.n * (8.0 * N * r.C + 4.0 * r.C * r.C) * es);
#define AMT_BR_CASE(CC_)                                                                                       \
  case CC_:                                                                                                    \
    if (dt == DT_F32) block_run_fwd_launch<float, CC_>(s, r, grid); else block_run_fwd_launch<bf16, CC_>(s, r, grid);  \
    return true
  switch (r.C) {
    AMT_BR_CASE(16);
    AMT_BR_CASE(32);
    AMT_BR_CASE(64);
    AMT_BR_CASE(128);
    AMT_BR_CASE(256);
    default: break;
  }
#undef AMT_BR_CASE
  return false;
}

void block_fwd_fused(hipStream_t s, DType dt, const BlockFwdArgs& a) {
  const double N = (double)a.B * a.L, es = dt == DT_F32 ? 4 : 2;
  const bool save = a.U != nullptr;   // inference passes no save tensors
  ProfScope ps(s, "block_fwd_fused", 2.0 * N * (7.0 * a.C + 4.0 * a.C * a.C), ((save ? (a.Z ? 8.0 : 7.0) : 2.0) * N * a.C + 4.0 * a.C * a.C) * es);
#define AMT_BF_CASE(CC_)                                                  \
  case CC_:                                                               \
    if (dt == DT_F32) { if (save) block_fwd_fused_launch<float, CC_, true>(s, a); else block_fwd_fused_launch<float, CC_, false>(s, a); }  \
    else { if (save) block_fwd_fused_launch<bf16, CC_, true>(s, a); else block_fwd_fused_launch<bf16, CC_, false>(s, a); }                  \
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

// ------------------------------------------------------------------ Block backward
// One launch for the data-gradient chain of a Block (replaces gemm_nt<gelu_bwd> + gemm_nt<store> +
// dwblock_bwd):
//   du  = (dy . (gamma * W2)) * gelu'(u)            stage A   (layer scale folded into the weight copy)
//   dln = du . W1                                    stage B
//   dD  = LN'(dln) w.r.t. the depthwise-conv output (recomputed from x), rows +-3 around the tile
//   dx  = dy + conv^T(dD);  dlnw, dlnb, dwdw[7][C], dbdw, dgamma = sum dy * z   -> per-workgroup records
// A tile is TT consecutive positions of ONE clip: TL = TT - 6 interior rows + 3 halo rows each side
// (the two GEMM stages are evaluated for the halo rows too: 6 / TT extra work instead of an exchange).
// du is written to HBM for the weight-gradient launch; everything else stays in LDS.
#ifndef AMT_BB16_TT
#define AMT_BB16_TT 256
#endif
#ifndef AMT_BB256_TT
#define AMT_BB256_TT 16
#endif
#ifndef AMT_BB128_TT
#define AMT_BB128_TT 32
#endif
#ifndef AMT_BB64_TT
#define AMT_BB64_TT 64
#endif
template <typename T, int C> struct BlockBwdTile {   // == DwTile<C>::ROWS of the unfused kernel
  static constexpr int TT = (C == 16 && sizeof(T) == 2) ? AMT_BB16_TT
                            : (C <= 32 ? 128
                               : (C <= 64 ? (sizeof(T) == 2 ? AMT_BB64_TT : 64)
                                  : (C == 256 ? ((sizeof(T) == 4 || AMT_BB256_TT == 16) ? 16 : 32) : (sizeof(T) == 2 ? AMT_BB128_TT : 32))));
};

template <typename T, int C, int TT>
__global__ __launch_bounds__(256) void block_bwd_fused_kernel(BlockBwdArgs a, WgradGroup riders) {
  constexpr int H2 = 2 * C, LDX = C + 8, LDH = H2 + 8, TL = TT - 6;
  constexpr int LPR = C / 8, RPP = 256 / LPR;
  static_assert(TT % RPP == 0, "row phases cover the tile in whole passes");
  using SA = ChainSplit<H2, TT>;
  using SB = ChainSplit<C, TT>;
  constexpr int KCA = ChainChunk<T, SA::RB, C>::KC, KCB = ChainChunk<T, SB::RB, H2>::KC;
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  if ((int)blockIdx.x >= a.n_chain) {   // weight gradients of the previous scope ride on this launch
#ifdef AMT_CHAIN_TIMING
    if (threadIdx.x == 0) amt_stamps[a.dbg_slot & 127][((int)blockIdx.x - a.n_chain) & 255][10] = wall_clock64();
#endif
    wgrad_group_dispatch<T>(riders, (int)blockIdx.x - a.n_chain, chain_smem);
#ifdef AMT_CHAIN_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0) amt_stamps[a.dbg_slot & 127][((int)blockIdx.x - a.n_chain) & 255][11] = wall_clock64();
#endif
    return;
  }
  T* DYs = reinterpret_cast<T*>(chain_smem);           // [TT][LDX]      dy (zero outside the clip)
  T* Us = DYs + TT * LDX;                              // [TT][LDH]      u, then du in place
  T* DLs = Us + TT * LDH;                              // [TT][LDX]      dln, then dD in place
  using RT = Res<T>;                                   // the saved Block input: residual stream (common.h)
  RT* Xt = reinterpret_cast<RT*>(DLs + TT * LDX);      // [TT + 6][LDX]  block input, rows l0 - 6 .. (zero outside)
  float* part = reinterpret_cast<float*>(chain_smem);   // [4 waves][11 * C], over DYs/Us/DLs once they are dead
  static_assert(sizeof(T) * (TT * (2 * LDX + LDH) + (TT + 6) * LDX) >= sizeof(float) * 44 * C, "column-sum slices fit over the dead panels (Xt included: dead after row phase 2)");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / a.tiles_per_clip, l0 = (blockIdx.x % a.tiles_per_clip) * TL;
  const int L = a.L;
  const long rb = (long)b * L;
  const T* dY = reinterpret_cast<const T*>(a.dY);
  const RT* X = reinterpret_cast<const RT*>(a.X);
  const T* U = reinterpret_cast<const T*>(a.U);
  const T* WA = reinterpret_cast<const T*>(a.W2tg);
  const T* WB = reinterpret_cast<const T*>(a.W1t);
  const int mA = (wave % SA::WF) * SA::RB * 16, tA = (wave / SA::WF) * SA::NT * 16;
  const int mB = (wave % SB::WF) * SB::RB * 16, tB = (wave / SB::WF) * SB::NT * 16;
  const int sub = tid % LPR, prow = tid / LPR;

  AMT_STAMP_RT(6);
  AMT_STAMP(0);
  uint32_t warm = 0;   // cooperative L2 warm-up of the two 1x1 weights (chain.h) where they are big enough to matter
  if constexpr (C >= 128) {
    const int q = blockIdx.x >> 3, ns = max(1, a.n_chain >> 3);
    constexpr size_t WBYTES = chain_packed_elems(H2, C) * sizeof(T);
    constexpr int MAXLD = (int)((WBYTES + 8 * 4096 - 1) / (8 * 4096)) + 1;   // enough 4-KiB rounds for >= 8 slices
    chain_l2_warm<MAXLD>(WA, WBYTES, q, ns, tid, warm);
    chain_l2_warm<MAXLD>(WB, WBYTES, q, ns, tid, warm);
  }
  Frag<T> wa[KCA][SA::RB], wb[KCB][SB::RB];
  chain_load_w<T, SA::RB, KCA, C>(wa, WA, mA, 0, lane);
  chain_load_w<T, SB::RB, KCB, H2>(wb, WB, mB, 0, lane);   // registers are idle until the row phases
  // The row phases' parameters (7 depthwise taps, conv bias, LN weight: 9 C floats) are requested here and parked in LDS;
  // requested where the row phases start, their L2 round trip stood between stage B and the first row (ISA: 18 loads
  // and a descending chain of waits in front of row phase 1).  Unconditional loads from clamped indices (cnn.hip).
  __shared__ __attribute__((aligned(16))) float s_par[9 * C];
  {
    constexpr int NPAR = (9 * C + 255) / 256;
    float pv[NPAR];
#pragma unroll
    for (int j = 0; j < NPAR; ++j) {
      const int t = min(tid + 256 * j, 9 * C - 1);
      const float* src = t < 7 * C ? a.wdw + t : (t < 8 * C ? a.bdw + (t - 7 * C) : a.lnw + (t - 8 * C));
      pv[j] = *src;
    }
#pragma unroll
    for (int j = 0; j < NPAR; ++j)
      if (tid + 256 * j < 9 * C) s_par[tid + 256 * j] = pv[j];
  }

  // ---- staging: dy and u rows l0 - 3 .. l0 + TL + 2, x rows l0 - 6 .. l0 + TL + 5; rows outside the
  // clip are loaded from a clamped row and zeroed (unconditional loads, see cnn.hip).  Every load is issued here, but
  // only dy is waited for before stage A: u (needed by its epilogue) and x (needed by the row phases) land in registers
  // while the stage-A products run and go to LDS after them - at the start of a launch every workgroup of the chip is
  // waiting for its first bytes at once and that round trip is the longest phase of the tile (profiles/r02_notes.md).
  constexpr int NP = TT / RPP, NPX = (TT + 6 + RPP - 1) / RPP;
  Chunk8<T> dyc[NP], uc0[NP], uc1[NP];
  Chunk8<RT> xcr[NPX];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) {
    const int l = l0 - 3 + pp * RPP + prow;
    const bool in = l >= 0 && l < L;
    const int lc = l < 0 ? 0 : (l >= L ? L - 1 : l);
    dyc[pp] = chunk_load(dY + (rb + lc) * C + sub * 8);
    if (!in) dyc[pp] = chunk_zero<T>();
  }
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) {
    const int l = l0 - 3 + pp * RPP + prow;
    const int lc = l < 0 ? 0 : (l >= L ? L - 1 : l);
    uc0[pp] = chunk_load(U + (rb + lc) * H2 + sub * 8);
    uc1[pp] = chunk_load(U + (rb + lc) * H2 + C + sub * 8);
  }
#pragma unroll
  for (int pp = 0; pp < NPX; ++pp) {
    const int l = l0 - 6 + pp * RPP + prow;
    const bool in = l >= 0 && l < L;
    const int lc = l < 0 ? 0 : (l >= L ? L - 1 : l);
    xcr[pp] = chunk_load(X + (rb + lc) * C + sub * 8);
    if (!in) xcr[pp] = chunk_zero<RT>();
  }
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) chunk_store(DYs + (pp * RPP + prow) * LDX + sub * 8, dyc[pp]);
  __syncthreads();
  AMT_STAMP(1);

  // ---- stage A: du = (W2tg . dy^T) * gelu'(u), in place over u
  {
    f32x4 acc[SA::RB][SA::NT];
    chain_zero(acc);
    chain_stage<T, SA::RB, SA::NT, KCA, C>(acc, wa, WA, mA, DYs, LDX, tA, lane);
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      chunk_store(Us + (pp * RPP + prow) * LDH + sub * 8, uc0[pp]);
      chunk_store(Us + (pp * RPP + prow) * LDH + C + sub * 8, uc1[pp]);
    }
#pragma unroll
    for (int pp = 0; pp < NPX; ++pp)
      if (pp * RPP + prow < TT + 6) chunk_store(Xt + (pp * RPP + prow) * LDX + sub * 8, xcr[pp]);
    __syncthreads();
#pragma unroll
    for (int rbk = 0; rbk < SA::RB; ++rbk) {
      const int m = mA + 16 * rbk + 4 * g;
#pragma unroll
      for (int ti = 0; ti < SA::NT; ++ti) {
        const int p = tA + 16 * ti + r;
        float u[4], du[4];
        load4f(Us + p * LDH + m, u);
#pragma unroll
        for (int i = 0; i < 4; ++i) du[i] = acc[rbk][ti][i] * gelu_tanh_grad(u[i]);
        store4f(Us + p * LDH + m, du);
      }
    }
  }
  __syncthreads();
  // du of the interior rows leaves for the weight-gradient launch (whole rows, 16 B per lane)
  {
    constexpr int CPR = H2 / 8, TOTAL = TL * CPR;
#pragma unroll
    for (int i = 0; i < (TOTAL + 255) / 256; ++i) {
      const int id = tid + i * 256;
      const int q = id / CPR, cc = id % CPR;
      if (id < TOTAL && l0 + q < L)
        chunk_store(reinterpret_cast<T*>(a.dU) + (rb + l0 + q) * H2 + 8 * cc, chunk_load(Us + (q + 3) * LDH + 8 * cc));
    }
  }

  // ---- stage B: dln = W1t . du^T
  {
    f32x4 acc[SB::RB][SB::NT];
    chain_zero(acc);
    chain_stage<T, SB::RB, SB::NT, KCB, H2>(acc, wb, WB, mB, Us, LDH, tB, lane);
#pragma unroll
    for (int rbk = 0; rbk < SB::RB; ++rbk) {
      const int m = mB + 16 * rbk + 4 * g;
#pragma unroll
      for (int ti = 0; ti < SB::NT; ++ti) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[rbk][ti][i];
        store4f(DLs + (tB + 16 * ti + r) * LDX + m, v);
      }
    }
  }
  __syncthreads();
  AMT_STAMP(2);

  // ---- row phase 1: dD = LN'(dln) for every tile row (zero outside the clip), in place over dln
  // column-sum accumulators in record order: [0] lnw, [1] lnb, [2..8] the seven taps, [9] conv bias, [10] gamma
  float wk[7][8], cs[11][8];
  float (&alw)[8] = cs[0], (&alb)[8] = cs[1], (&ab)[8] = cs[9], (&ag)[8] = cs[10];
  float (&aw)[7][8] = *reinterpret_cast<float (*)[7][8]>(&cs[2]);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int q = 0; q < 11; ++q) cs[q][i] = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) wk[k][i] = s_par[k * C + sub * 8 + i];
  }
  // z rows of row phase 2 (layer-scale gradient): on their way during row phase 1
  constexpr int NP2 = (TL + RPP - 1) / RPP;
  Chunk8<T> zc[NP2];
  if (a.Z != nullptr) {
#pragma unroll
    for (int pp = 0; pp < NP2; ++pp) {
      const int l = l0 + pp * RPP + prow;
      zc[pp] = chunk_load(reinterpret_cast<const T*>(a.Z) + (rb + (l < L ? l : L - 1)) * C + sub * 8);
    }
  }
  {
    float bv[8], wv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { bv[i] = s_par[7 * C + sub * 8 + i]; wv[i] = s_par[8 * C + sub * 8 + i]; }
#pragma unroll
    for (int pp = 0; pp < TT / RPP; ++pp) {
      const int p = pp * RPP + prow;
      const int l = l0 - 3 + p;
      const bool valid = l >= 0 && l < L;
      const int lc = l < 0 ? 0 : (l >= L ? L - 1 : l);
      const float mu = a.mean[rb + lc], rs = a.rstd[rb + lc];
      float d[8], dl[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) d[i] = bv[i];
#pragma unroll
      for (int k = 0; k < 7; ++k) {     // x[l + k - 3] = Xt row p + k (zero outside the clip)
        float x[8];
        load8f(Xt + (p + k) * LDX + sub * 8, x);
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] += wk[k][i] * x[i];
      }
      load8f(DLs + p * LDX + sub * 8, dl);
      const float fin = (valid && p >= 3 && p < TL + 3) ? 1.f : 0.f;
      float s1 = 0.f, s2 = 0.f, gy[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
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
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = fv * rs * (gy[i] - s1 - d[i] * s2);
      store8f(DLs + p * LDX + sub * 8, o);
    }
  }
  __syncthreads();
  AMT_STAMP(3);

  // ---- row phase 2 (interior rows): dx = dy + conv^T(dD); depthwise weight / bias / gamma gradients
#pragma unroll
  for (int pp = 0; pp < (TL + RPP - 1) / RPP; ++pp) {
    const int i0 = pp * RPP + prow;
    const int l = l0 + i0;
    const bool valid = (i0 < TL) && (l < L);
    const int i0c = i0 < TL ? i0 : TL - 1;
    const int lcl = l < L ? l : L - 1;
    const float fv = valid ? 1.f : 0.f;
    float o[8], dd[8];
    load8f(DYs + (i0c + 3) * LDX + sub * 8, o);
    load8f(DLs + (i0c + 3) * LDX + sub * 8, dd);
    if (a.Z != nullptr) {
#pragma unroll
      for (int i = 0; i < 8; ++i) ag[i] += fv * o[i] * chunk_get(zc[pp], i);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { dd[i] *= fv; ab[i] += dd[i]; }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      float tq[8], x[8];
      load8f(DLs + (i0c + 6 - k) * LDX + sub * 8, tq);   // dD[l - k + 3]
      load8f(Xt + (i0c + 3 + k) * LDX + sub * 8, x);     // x[l + k - 3]
#pragma unroll
      for (int i = 0; i < 8; ++i) { o[i] += wk[k][i] * tq[i]; aw[k][i] += dd[i] * x[i]; }
    }
    if (valid) store8f(reinterpret_cast<T*>(a.dX) + (rb + l) * C + sub * 8, o);
  }

  AMT_STAMP(4);
  // ---- the eleven column sums of this workgroup -> its record [lnw | lnb | taps 7C | bias | gamma]
  __syncthreads();            // every wave is done with the panels the slices alias
  fold_columns<LPR, 88>(*reinterpret_cast<float (*)[88]>(&cs[0][0]), part + wave * 11 * C, C, lane);
  __syncthreads();
  float* rec = a.rec + (size_t)blockIdx.x * a.rec_stride;
  const int nrec = a.Z != nullptr ? 11 * C : 10 * C;
  for (int t = tid; t < nrec; t += 256)
    rec[t] = (part[t] + part[11 * C + t]) + (part[22 * C + t] + part[33 * C + t]);
  if (warm == 0x9e3779b9u && a.L < 0) rec[0] = 0.f;   // keeps the warm-up loads alive; never true
  AMT_STAMP(5);
  AMT_STAMP_RT(7);
}

template <typename T, int C>
static void block_bwd_fused_launch(hipStream_t s, const BlockBwdArgs& a, const WgradGroup& riders, int n_rider_wgs) {
  constexpr int TT = BlockBwdTile<T, C>::TT;
  constexpr size_t lds_chain = sizeof(T) * ((size_t)TT * (2 * (C + 8) + (2 * C + 8)) + (size_t)(TT + 6) * (C + 8));
  constexpr size_t lds = lds_chain > TnLds<T>::BYTES ? lds_chain : TnLds<T>::BYTES;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool once = (lds > 64 * 1024
                          ? (void)hipFuncSetAttribute(reinterpret_cast<const void*>(block_bwd_fused_kernel<T, C, TT>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                          : (void)0,
                      true);
  (void)once;
  hipLaunchKernelGGL((block_bwd_fused_kernel<T, C, TT>), dim3(a.n_chain + n_rider_wgs), dim3(256), lds, s, a, riders);
}

int block_bwd_fused_tile_len(DType dt, int C) {
  if (dt == DT_F32) return (C <= 32 ? 128 : (C <= 64 ? 64 : (C == 256 ? 16 : 32))) - 6;
  return (C == 16 ? AMT_BB16_TT : (C <= 32 ? 128 : (C <= 64 ? AMT_BB64_TT : (C == 256 ? AMT_BB256_TT : AMT_BB128_TT)))) - 6;
}

void block_bwd_fused(hipStream_t s, DType dt, const BlockBwdArgs& a_in, float* dlnw, float* dlnb, float* dwdw,
                     float* dbdw, float* dgamma, PartialSink* sink, const WgradArgs* riders, int n_riders) {
  BlockBwdArgs a = a_in;
  const int C = a.C, TL = block_bwd_fused_tile_len(dt, C);
  a.tiles_per_clip = (a.L + TL - 1) / TL;
  const int nb = a.B * a.tiles_per_clip;
  const int rec = 11 * C;
  size_t o = sink->take((size_t)nb * rec);
  sink->add(o, nb, rec, C, dlnw);
  sink->add(o + C, nb, rec, C, dlnb);
  sink->add(o + 2 * C, nb, rec, 7 * C, dwdw);
  sink->add(o + 9 * C, nb, rec, C, dbdw);
  if (dgamma) sink->add(o + 10 * C, nb, rec, C, dgamma);
  WgradGroup rg;
  double rfl = 0, rby = 0;
  // Rider workgroups per Block-chain launch.  Measured (ms/step at cfg2): 48: 4.70, 96: 4.35, 144: 4.21, 224: 4.18,
  // 336: 4.23, 448: 4.25, 640: 4.43; sizing them to the compute units the chain leaves free (256 - nb) gave 4.52: a
  // rider workgroup is bound by its own load -> LDS -> MFMA stage latency, so fewer, longer riders end later.
  // (re-swept after the riders' loads and epilogue were fixed, round 2: 176: 3.284, 192: 3.284, 200: 3.300, 224: 3.306 -
  // with 192 the chain and rider workgroups of a C = 128 launch fit the 512 resident slots, with 208 eight riders start late)
  static const int rider_target = getenv("AMT_RIDER_TARGET") ? atoi(getenv("AMT_RIDER_TARGET")) : 192;
  // C = 256 (256 registers per lane: ONE workgroup per compute unit, 160 chain workgroups at cfg2): riders beyond the
  // free compute units wait for a chain workgroup to finish and the launch ends late (54 us per launch with 224)
  static const int rider_target_256 = getenv("AMT_RIDER_TARGET_256") ? atoi(getenv("AMT_RIDER_TARGET_256")) : 288;
  const int n_rider_wgs = wgrad_group_prepare(dt, riders, n_riders, sink, &rg, &rfl, &rby, C == 256 ? rider_target_256 : rider_target);
  if (sink->dry) return;
  a.rec = sink->ws + o;
  a.rec_stride = rec;
  a.n_chain = nb;
  const double N = (double)a.B * a.L, es = dt == DT_F32 ? 4 : 2;
  ProfScope ps(s, n_rider_wgs > 0 ? "block_bwd_fused+wgrad" : "block_bwd_fused",
               2.0 * N * (21.0 * C + 4.0 * C * C) + rfl, ((a.Z ? 8.0 : 7.0) * N * C + 4.0 * C * C) * es + rby);
#define AMT_BB_CASE(CC_)                                                  \
  case CC_:                                                               \
    if (dt == DT_F32) block_bwd_fused_launch<float, CC_>(s, a, rg, n_rider_wgs);   \
    else block_bwd_fused_launch<bf16, CC_>(s, a, rg, n_rider_wgs);                 \
    break
  switch (C) {
    AMT_BB_CASE(16);
    AMT_BB_CASE(32);
    AMT_BB_CASE(64);
    AMT_BB_CASE(128);
    AMT_BB_CASE(256);
    default: break;
  }
#undef AMT_BB_CASE
}


// ------------------------------------------------------------------ transformer layer, second half, forward
// One workgroup per tile of TT tokens; all 256 hidden features of a token stay in the workgroup, so
// the LayerNorm between the output projection and the FFN is a row pass over an LDS panel.
// (backward chains: round 1 saw wrong bf16 gradients with 16-token tiles; round 2: the kernels are fine with 16 - the
// record count in chain_ln_records still assumed 32 - but with their riders they are slower: qkv_bwd_fused+wgrad 29.8
// against 25.2 us, 3.84 against 3.77 ms/step)
template <typename T> struct FfnTile { static constexpr int TT = sizeof(T) == 2 ? 32 : 16; };
// forward chains: 16-token tiles for both dtypes (250 instead of 125 workgroups at cfg2: qkv_fwd_fused 11.4 -> 10.1 us,
// attn_out_fused 8.8 -> 8.0 us); the backward chains keep 32 (they are bound by their riders, profiles/r01_notes.md)
template <typename T> struct FfnTileFwd { static constexpr int TT = 16; };

// FULL = false stops after the LayerNorm (h, hn, mean, rstd written): the production form.  The full
// chain streams 896 KiB of weights through every compute unit (~28 B/clk L2-hit rate per CU, less for
// the cold first touch) and measured 27 us against ~31 us for the four separate launches; with only
// the 128 KiB output projection in the chain the kernel takes ~6 us and W1 / W2 stay tiled GEMMs,
// whose 64 x 64 tiles move 3.5x fewer weight bytes per CU (profiles/r01_notes.md).
// wx0 / wx1: two more weight ranges to warm with this body's own (the q/kv weights of the body that follows it in the
// layer-boundary launch: warmed there, in the middle of the launch, they were six round trips in series)
template <typename T, int TT, bool FULL>
AMT_DEV void ffn_fwd_body(FfnFwdArgs& a, unsigned char* chain_smem, const void* wx0 = nullptr, size_t bx0 = 0,
                          const void* wx1 = nullptr, size_t bx1 = 0) {
  a.seed = eff_seed(a.seed, a.seed_dev);   // kernel argument + the context's device seed word (common.h)
  constexpr int H = 256, I = 512, I2 = 1024, LDX = H + 8, LDG = I + 8;
  using SO = ChainSplit<H, TT>;        // Wo and W2 stages: 256 output features
  using S1 = ChainSplit<I2 / 2, TT>;   // W1 in two halves of 512 permuted rows
  // one wave per SIMD (125 workgroups): the register file is the prefetch buffer.  32 fragments
  // (32 KiB per wave, 128 KiB per CU) per chunk and two chunks in flight; with 16 the cold weight
  // stream ran at ~21 GB/s per CU (one HBM round trip per 64 KiB).
  constexpr int KCO = ChainChunk<T, SO::RB, H, 32>::KC, KC1 = ChainChunk<T, S1::RB, H, 32>::KC,
                KC2 = ChainChunk<T, SO::RB, I, 32>::KC;
  static_assert(S1::RB % 2 == 0, "a wave owns whole (u1 | u2) row-block pairs");
  using RT = Res<T>;                          // x, h, y: residual stream (common.h)
  static_assert(sizeof(RT) == sizeof(T), "panels keep their size");
  T* As = reinterpret_cast<T*>(chain_smem);   // [TT][LDX]  attention output, then LN2(h)
  RT* Xs = reinterpret_cast<RT*>(As + TT * LDX);   // [TT][LDX]  x, then y (staged for the copy-out)
  RT* Hs = Xs + TT * LDX;                     // [TT][LDX]  h
  T* Gs = reinterpret_cast<T*>(Hs + TT * LDX);     // [TT][LDG]  g

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const long N = a.N, n0 = (long)blockIdx.x * TT;
  const T* Wo = reinterpret_cast<const T*>(a.Wo);
  const T* W1 = reinterpret_cast<const T*>(a.W1);
  const T* W2 = reinterpret_cast<const T*>(a.W2);
  const int mO = (wave % SO::WF) * SO::RB * 16, tO = (wave / SO::WF) * SO::NT * 16;
  const int m1 = (wave % S1::WF) * S1::RB * 16, t1 = (wave / S1::WF) * S1::NT * 16;

  AMT_STAMP_RT(6);
  AMT_STAMP(0);
  uint32_t warm = 0;
  L2Warm<FULL ? 6 : 1> lw1;
  L2Warm<FULL ? 3 : 1> lw2;
  L2Warm<FULL ? 2 : 1> lx0;
  L2Warm<1> lx1;
  if constexpr (FULL) {
    const int q = blockIdx.x >> 3, ns = max(1, (int)(gridDim.x >> 3));
    lw1.issue(W1, sizeof(T) * (size_t)I2 * H, q, ns, tid);
    lw2.issue(W2, sizeof(T) * (size_t)H * I, q, ns, tid);
    lx0.issue(wx0, bx0, q, ns, tid);
    lx1.issue(wx1, bx1, q, ns, tid);
  }
  Frag<T> wo[KCO][SO::RB];
  chain_load_w<T, SO::RB, KCO, H>(wo, Wo, mO, 0, lane);
  chain_copy_in<T, TT, H, LDX>(As, reinterpret_cast<const T*>(a.AO), n0, N, tid);
  ChainCopyRegs<RT, TT, H> xreg;    // x is read by the residual epilogue only: it lands while the products run
  chain_copy_load<RT, TT, H>(xreg, reinterpret_cast<const RT*>(a.X), H, n0, N, tid);
  // The two FFN biases are requested here and parked in LDS.  Read from global memory where the epilogues use them (two
  // 16-byte loads per row-block pair, waited for on the spot), each was an L2 round trip between a stage's products and
  // its stores: ~8 per tile in the ISA.
  __shared__ __attribute__((aligned(16))) float s_b1[FULL ? 1024 : 4], s_b2[FULL ? 256 : 4];
  float pb1[4], pb2 = 0.f;
  if constexpr (FULL) {
    static_assert(I2 == 1024 && H == 256, "one bias element per thread and 256-stride");
#pragma unroll
    for (int j = 0; j < 4; ++j) pb1[j] = a.b1[tid + 256 * j];
    pb2 = a.b2[tid];
  }
  __syncthreads();
  AMT_STAMP(1);
  if constexpr (FULL) { lw1.fold(); lw2.fold(); lx0.fold(); lx1.fold(); }
  if constexpr (FULL) {
#pragma unroll
    for (int j = 0; j < 4; ++j) s_b1[tid + 256 * j] = pb1[j];
    s_b2[tid] = pb2;
  }

  // ---- h = x + ao . Wo^T
  Frag<T> w1a[KC1][S1::RB];
  {
    f32x4 acc[SO::RB][SO::NT];
    chain_zero(acc);
    chain_stage<T, SO::RB, SO::NT, KCO, H>(acc, wo, Wo, mO, As, LDX, tO, lane);
    chain_copy_store<RT, TT, H, LDX>(Xs, xreg, tid);
    __syncthreads();
    if constexpr (FULL) chain_load_w<T, S1::RB, KC1, H>(w1a, W1, m1, 0, lane);   // in flight during the LayerNorm pass
#pragma unroll
    for (int rb = 0; rb < SO::RB; ++rb) {
      const int m = mO + 16 * rb + 4 * g;
#pragma unroll
      for (int ti = 0; ti < SO::NT; ++ti) {
        const int p = tO + 16 * ti + r;
        float x[4], h[4];
        load4f(Xs + p * LDX + m, x);
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = acc[rb][ti][i] + x[i];
        store4f(Hs + p * LDX + m, h);
      }
    }
  }
  __syncthreads();
  AMT_STAMP(2);
  // ---- hn = LN2(h) over the panel rows (As is dead: it receives hn)
  chain_ln_rows<T, TT, H, LDX>(Hs, As, a.ln_w, a.ln_b, a.mean, a.rstd, n0, N, tid);
  chain_copy_out<RT, TT, H, LDX>(Hs, reinterpret_cast<RT*>(a.Hout), n0, N, tid);
  __syncthreads();
  chain_copy_out<T, TT, H, LDX>(As, reinterpret_cast<T*>(a.HN), n0, N, tid);
  AMT_STAMP(3);
  if constexpr (!FULL) {
    AMT_STAMP(4);
    AMT_STAMP(5);
    AMT_STAMP_RT(7);
    return;
  }

  // ---- u = hn . W1^T + b1 ; g = gelu(u1) * u2, in two halves of 512 permuted rows
  Frag<T> w2f[KC2][SO::RB], w1b[KC1][S1::RB];
  auto glu_half = [&](int mh, Frag<T> (&wpre)[KC1][S1::RB], bool last) {
    f32x4 acc[S1::RB][S1::NT];
    chain_zero(acc);
    chain_stage<T, S1::RB, S1::NT, KC1, H>(acc, wpre, W1, mh, As, LDX, t1, lane);
    // the next stage's first weights are requested before the GLU epilogue (exp, stores) runs
    if (!last) chain_load_w<T, S1::RB, KC1, H>(w1b, W1, mh + I2 / 2, 0, lane);
    else chain_load_w<T, SO::RB, KC2, I>(w2f, W2, mO, 0, lane);
#pragma unroll
    for (int rb = 0; rb < S1::RB; rb += 2) {
      const int m = mh + 16 * rb + 4 * g;          // permuted row of u1; u2 sits 16 rows further
      float b1a[4], b1b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { b1a[i] = s_b1[m + i]; b1b[i] = s_b1[m + 16 + i]; }
      const int j = (m >> 5) * 16 + (m & 15);      // g column
#pragma unroll
      for (int ti = 0; ti < S1::NT; ++ti) {
        const int p = t1 + 16 * ti + r;
        const long row = n0 + p;
        float u1[4], u2[4], gl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u1[i] = to_f(from_f<T>(acc[rb][ti][i] + b1a[i]));
          u2[i] = to_f(from_f<T>(acc[rb + 1][ti][i] + b1b[i]));
          gl[i] = gelu_tanh(u1[i]) * u2[i];
        }
        store4f(Gs + p * LDG + j, gl);
        if (row < N) {
          T* up = reinterpret_cast<T*>(a.U) + row * I2 + m;
          store4f(up, u1);
          store4f(up + 16, u2);
        }
      }
    }
  };
  glu_half(m1, w1a, false);
  glu_half(I2 / 2 + m1, w1b, true);
  __syncthreads();
  AMT_STAMP(4);
  chain_copy_out<T, TT, I, LDG>(Gs, reinterpret_cast<T*>(a.G), n0, N, tid);

  // ---- y = h + dropout(g . W2^T + b2)
  {
    f32x4 acc[SO::RB][SO::NT];
    chain_zero(acc);
    chain_stage<T, SO::RB, SO::NT, KC2, I>(acc, w2f, W2, mO, Gs, LDG, tO, lane);
#pragma unroll
    for (int rb = 0; rb < SO::RB; ++rb) {
      const int m = mO + 16 * rb + 4 * g;
      float bb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) bb[i] = s_b2[m + i];
#pragma unroll
      for (int ti = 0; ti < SO::NT; ++ti) {
        const int p = tO + 16 * ti + r;
        const long row = n0 + p;
        float h[4], y[4];
        load4f(Hs + p * LDX + m, h);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = acc[rb][ti][i] + bb[i];
          if (a.drop_thresh24) {
            const bool keep = rng_keep(a.seed, a.stream, (uint32_t)row * (uint32_t)H + (uint32_t)(m + i), a.drop_thresh24);
            v = keep ? v * a.inv_keep : 0.f;
          }
          y[i] = v + h[i];
        }
        store4f(Xs + p * LDX + m, y);
      }
    }
  }
  __syncthreads();
  chain_copy_out<RT, TT, H, LDX>(Xs, reinterpret_cast<RT*>(a.Y), n0, N, tid);
  if (warm == 0x9e3779b9u && a.N < 0) a.mean[0] = 0.f;   // keeps the warm-up loads alive; never true
  AMT_STAMP(5);
  AMT_STAMP_RT(7);
}

template <typename T, int TT, bool FULL>
__global__ __launch_bounds__(256) void ffn_fwd_fused_kernel(FfnFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  ffn_fwd_body<T, TT, FULL>(a, chain_smem);
}

template <typename T, bool FULL, int TT>
static void ffn_fwd_fused_launch_tt(hipStream_t s, const FfnFwdArgs& a) {
  constexpr size_t lds = sizeof(T) * (size_t)TT * (3 * (256 + 8) + (FULL ? 512 + 8 : 0));
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool once = (lds > 64 * 1024
                          ? (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_fwd_fused_kernel<T, TT, FULL>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                          : (void)0,
                      true);
  (void)once;
  hipLaunchKernelGGL((ffn_fwd_fused_kernel<T, TT, FULL>), dim3((a.N + TT - 1) / TT), dim3(256), lds, s, a);
}
// 16-token tiles while they are what fills the chip (N < 8192 tokens: cfg2), the longer tile of the backward chains
// above that (weights are streamed once per tile)
template <typename T, bool FULL>
static void ffn_fwd_fused_launch(hipStream_t s, const FfnFwdArgs& a) {
  if (a.N < 8192) ffn_fwd_fused_launch_tt<T, FULL, FfnTileFwd<T>::TT>(s, a);
  else ffn_fwd_fused_launch_tt<T, FULL, FfnTile<T>::TT>(s, a);
}

void ffn_fwd_fused(hipStream_t s, DType dt, const FfnFwdArgs& a) {
  const double N = a.N, es = dt == DT_F32 ? 4 : 2;
  if (a.W1 == nullptr) {   // output projection + residual + LayerNorm only
    ProfScope ps(s, "attn_out_fused", 2.0 * N * 256.0 * 256, (N * 256.0 * 4 + 256.0 * 256) * es);
    if (dt == DT_F32) ffn_fwd_fused_launch<float, false>(s, a);
    else ffn_fwd_fused_launch<bf16, false>(s, a);
    return;
  }
  ProfScope ps(s, "ffn_fwd_fused", 2.0 * N * (256.0 * 256 + 1024.0 * 256 + 256.0 * 512),
               (N * (256.0 * 6 + 1024 + 512) + 256.0 * 256 + 1024.0 * 256 + 256.0 * 512) * es);
  if (dt == DT_F32) ffn_fwd_fused_launch<float, true>(s, a);
  else ffn_fwd_fused_launch<bf16, true>(s, a);
}


// ------------------------------------------------------------------ transformer layer, first half, forward
// XLDS != nullptr: the tile's x rows are already in LDS (panel [TT][H + 8] left by the chain that produced them in the
// same launch: ffn_qkv_fwd_fused) and are copied from there instead of from a.X.
// RoPE table entries of the tile for the rotated form of q / k (QkvFwdArgs::rope_cos): the (cos, sin) pairs of this lane's
// feature quads, requested where the launch starts (the boundary launch: in front of the FFN body) so that the epilogues
// that rotate find them in registers.
// MODE 0: no rotation (local layers; an empty struct: the instantiation carries no registers for it); 1: entries requested
// at the start of the launch and held in registers (16-token tiles: one workgroup per compute unit, the latency would be
// exposed in front of the epilogues); 2: entries loaded where the epilogues use them (longer tiles: two workgroups per compute
// unit, and 100 more registers would leave only one; the tile's table rows are staged in LDS - TT x 64 floats behind the three
// panels - and read from there: loaded from L2 at every use, the rotated 32-token kernel ran 88 against 65 us at 112 000 tokens).
template <int TT, int MODE> struct QkvRope {
  using SQ = ChainSplit<320, TT>;
  using SK = ChainSplit<512, TT>;
  static constexpr int NQ = MODE == 1 ? SQ::RB : 1, NQT = MODE == 1 ? SQ::NT : 1, NK = MODE == 1 ? SK::RB : 1, NKT = MODE == 1 ? SK::NT : 1;
  float2 qc[NQ][NQT], qs[NQ][NQT], kc[NK][NKT], ks[NK][NKT];
  static AMT_DEV int pos_of(const QkvFwdArgs& a, long row) { return (int)((row < a.N ? row : (long)a.N - 1) % a.rope_S); }
  AMT_DEV void load(const QkvFwdArgs& a) {
    if constexpr (MODE == 1) {
      const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
      const long n0 = (long)blockIdx.x * TT;
      const int mQ = (wave % SQ::WF) * SQ::RB * 16, tQ = (wave / SQ::WF) * SQ::NT * 16;
      const int mK = (wave % SK::WF) * SK::RB * 16, tK = (wave / SK::WF) * SK::NT * 16;
#pragma unroll
      for (int ti = 0; ti < SQ::NT; ++ti) {
        const int pos = pos_of(a, n0 + tQ + 16 * ti + r);
#pragma unroll
        for (int rb = 0; rb < SQ::RB; ++rb) {
          const int m = mQ + 16 * rb + 4 * g;
          if (m < 256) {                                  // uniform over the wave (16-row blocks)
            qc[rb][ti] = *reinterpret_cast<const float2*>(a.rope_cos + pos * 32 + ((m & 63) >> 1));
            qs[rb][ti] = *reinterpret_cast<const float2*>(a.rope_sin + pos * 32 + ((m & 63) >> 1));
          }
        }
      }
      if (mK < 256) {                                     // uniform over the wave: the k half of kv
#pragma unroll
        for (int ti = 0; ti < SK::NT; ++ti) {
          const int pos = pos_of(a, n0 + tK + 16 * ti + r);
#pragma unroll
          for (int rb = 0; rb < SK::RB; ++rb) {
            const int m = mK + 16 * rb + 4 * g;
            kc[rb][ti] = *reinterpret_cast<const float2*>(a.rope_cos + pos * 32 + ((m & 63) >> 1));
            ks[rb][ti] = *reinterpret_cast<const float2*>(a.rope_sin + pos * 32 + ((m & 63) >> 1));
          }
        }
      }
    }
  }
  // (cos, sin) pairs of feature quad m .. m + 3 of global row `row`; K selects the k table of MODE 1
  template <bool K> AMT_DEV void get(const QkvFwdArgs& a, int rb, int ti, long row, int m, float2& c, float2& s) const {
    if constexpr (MODE == 1) {
      c = K ? kc[rb][ti] : qc[rb][ti];
      s = K ? ks[rb][ti] : qs[rb][ti];
    } else {
      const float* rp = lds_rows + (int)(row - (long)blockIdx.x * TT) * 64 + ((m & 63) >> 1);
      c = *reinterpret_cast<const float2*>(rp);
      s = *reinterpret_cast<const float2*>(rp + 32);
    }
  }
  // MODE 2: [TT][cos 32 | sin 32] of the tile's rows, staged by stage_rows (visible behind the body's first barrier)
  const float* lds_rows = nullptr;
  AMT_DEV void stage_rows(const QkvFwdArgs& a, float* dst) {
    if constexpr (MODE == 2) {
      lds_rows = dst;
      const long n0 = (long)blockIdx.x * TT;
      for (int id = threadIdx.x; id < TT * 16; id += 256) {
        const int p = id >> 4, q = id & 15;                 // row, 16-byte granule: 0..7 cos, 8..15 sin
        const int pos = pos_of(a, n0 + p);
        const float* src = (q < 8 ? a.rope_cos : a.rope_sin) + pos * 32 + 4 * (q & 7);
        *reinterpret_cast<f32x4*>(dst + p * 64 + 4 * q) = *reinterpret_cast<const f32x4*>(src);
      }
    }
  }
};
// rotate the two interleaved pairs of a feature quad as the attention kernels do while staging (attn.hip: rope_pair):
// the value as stored (rounded to T), the explicit rounding sequence, the result rounded to T by the store that follows
template <typename T>
AMT_DEV void rope_quad(float (&v)[4], float2 c, float2 s) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float x1 = to_f(from_f<T>(v[2 * j])), x2 = to_f(from_f<T>(v[2 * j + 1]));
    const float cj = j == 0 ? c.x : c.y, sj = j == 0 ? s.x : s.y;
    v[2 * j] = __fmaf_rn(x1, cj, -__fmul_rn(x2, sj));          // (rope_pair's scale is 1 here: the product by it is exact and dropped)
    v[2 * j + 1] = __fmaf_rn(x1, sj, __fmul_rn(x2, cj));
  }
}

template <typename T, int TT, int RM>
AMT_DEV void qkv_fwd_body(const QkvFwdArgs& a, unsigned char* chain_smem, const Res<T>* XLDS, QkvRope<TT, RM>& rope) {
  constexpr int H = 256, QC = 320, CK = 64, KV = 512, LDX = H + 8, LDQ = QC + 8, LDK = KV + 8;
  using SQ = ChainSplit<QC, TT>;
  using SK = ChainSplit<KV, TT>;
  constexpr int KCQ = ChainChunk<T, SQ::RB, H, 32>::KC, KCK = ChainChunk<T, SK::RB, CK>::KC;
  using RT = Res<T>;                          // x: residual stream (common.h), same size as T: LN1 runs in place
  T* Xs = reinterpret_cast<T*>(chain_smem);   // [TT][LDX]  x (as RT), then LN1(x) in place (as T)
  RT* Xr = reinterpret_cast<RT*>(chain_smem);
  T* Qs = Xs + TT * LDX;                      // [TT][LDQ]  qc
  T* Ks = Qs + TT * LDQ;                      // [TT][LDK]  kv
  rope.stage_rows(a, reinterpret_cast<float*>(Ks + TT * LDK));   // (rotated long-tile form: [TT][64] floats behind the panels)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const long N = a.N, n0 = (long)blockIdx.x * TT;
  const T* Wq = reinterpret_cast<const T*>(a.Wqc);
  const T* Wk = reinterpret_cast<const T*>(a.Wkv);
  const int mQ = (wave % SQ::WF) * SQ::RB * 16, tQ = (wave / SQ::WF) * SQ::NT * 16;
  const int mK = (wave % SK::WF) * SK::RB * 16, tK = (wave / SK::WF) * SK::NT * 16;

  AMT_STAMP_RT(6);
  AMT_STAMP(0);
  uint32_t warm = 0;   // cooperative L2 warm-up (chain.h); behind the FFN body of a boundary launch that body has done it
  if (XLDS == nullptr) {
    const int q = blockIdx.x >> 3, ns = max(1, (int)(gridDim.x >> 3));
    chain_l2_warm<4>(Wq, sizeof(T) * (size_t)QC * H, q, ns, tid, warm);
    chain_l2_warm<2>(Wk, sizeof(T) * (size_t)KV * CK, q, ns, tid, warm);
  }
  Frag<T> wq[KCQ][SQ::RB], wk[KCK][SK::RB];
  chain_load_w<T, SQ::RB, KCQ, H>(wq, Wq, mQ, 0, lane);
  if (XLDS == nullptr) {
    chain_copy_in<RT, TT, H, LDX>(Xr, reinterpret_cast<const RT*>(a.X), n0, N, tid);
  } else {                                   // rows past N hold zeros there as well (masked when they were computed)
    constexpr int CPR = H / 8;
    for (int id = tid; id < TT * CPR; id += 256) {
      const int p = id / CPR, c = id % CPR;
      chunk_store(Xr + p * LDX + 8 * c, chunk_load(XLDS + p * LDX + 8 * c));
    }
  }
  __syncthreads();
  AMT_STAMP(1);
  chain_ln_rows<T, TT, H, LDX>(Xr, Xs, a.ln_w, a.ln_b, a.mean, a.rstd, n0, N, tid);
  chain_load_w<T, SK::RB, KCK, CK>(wk, Wk, mK, 0, lane);
  __syncthreads();
  AMT_STAMP(2);
  if (a.XN != nullptr) chain_copy_out<T, TT, H, LDX>(Xs, reinterpret_cast<T*>(a.XN), n0, N, tid);   // (inference passes none)

  // ---- qc = xn . Wqc^T
  {
    f32x4 acc[SQ::RB][SQ::NT];
    chain_zero(acc);
    chain_stage<T, SQ::RB, SQ::NT, KCQ, H>(acc, wq, Wq, mQ, Xs, LDX, tQ, lane);
#pragma unroll
    for (int rb = 0; rb < SQ::RB; ++rb)
#pragma unroll
      for (int ti = 0; ti < SQ::NT; ++ti) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[rb][ti][i];
        if constexpr (RM != 0) {
          if (mQ + 16 * rb < 256) {
            float2 rc, rs;
            rope.template get<false>(a, rb, ti, n0 + tQ + 16 * ti + r, mQ + 16 * rb + 4 * g, rc, rs);
            rope_quad<T>(v, rc, rs);
          }
        }
        store4f(Qs + (tQ + 16 * ti + r) * LDQ + mQ + 16 * rb + 4 * g, v);
      }
  }
  __syncthreads();
  AMT_STAMP(3);
  chain_copy_out<T, TT, QC, LDQ>(Qs, reinterpret_cast<T*>(a.QC), n0, N, tid);

  // ---- kv = c . Wkv^T, c = qc[:, 256:320] as stored
  {
    f32x4 acc[SK::RB][SK::NT];
    chain_zero(acc);
    chain_stage<T, SK::RB, SK::NT, KCK, CK>(acc, wk, Wk, mK, Qs + H, LDQ, tK, lane);
#pragma unroll
    for (int rb = 0; rb < SK::RB; ++rb)
#pragma unroll
      for (int ti = 0; ti < SK::NT; ++ti) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[rb][ti][i];
        if constexpr (RM != 0) {
          if (mK < 256) {
            float2 rc, rs;
            rope.template get<true>(a, rb, ti, n0 + tK + 16 * ti + r, mK + 16 * rb + 4 * g, rc, rs);
            rope_quad<T>(v, rc, rs);
          }
        }
        store4f(Ks + (tK + 16 * ti + r) * LDK + mK + 16 * rb + 4 * g, v);
      }
  }
  __syncthreads();
  AMT_STAMP(4);
  chain_copy_out<T, TT, KV, LDK>(Ks, reinterpret_cast<T*>(a.KV), n0, N, tid);
  if (warm == 0x9e3779b9u && a.N < 0) a.mean[0] = 0.f;   // keeps the warm-up loads alive; never true
  AMT_STAMP(5);
  AMT_STAMP_RT(7);
}

template <typename T, int TT, int RM>
__global__ __launch_bounds__(256, (RM == 2 && TT == 32 && sizeof(T) == 2) ? 2 : 1) void qkv_fwd_fused_kernel(QkvFwdArgs a) {   // (the rotated long-tile form keeps the raw form's two workgroups per compute unit)
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  QkvRope<TT, RM> rope;
  rope.load(a);
  qkv_fwd_body<T, TT, RM>(a, chain_smem, nullptr, rope);
}

// Layer boundary in one launch: the second half of layer l (ffn_fwd_body, FULL) followed, for the same token tile, by the
// first half of layer l + 1 up to its attention core (qkv_fwd_body).  Both are per-token chains, the tile's y rows go
// from one to the other through LDS, and one kernel boundary per layer - with the write-back of what the first kernel
// stored (profiles/r02_notes.md) - leaves the dependent chain.
template <typename T, int TT, int RM>
__global__ __launch_bounds__(256) void ffn_qkv_fwd_fused_kernel(FfnFwdArgs fa, QkvFwdArgs qa) {
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  constexpr int LDX = 256 + 8;
  QkvRope<TT, RM> rope;
  rope.load(qa);
  ffn_fwd_body<T, TT, true>(fa, chain_smem, qa.Wqc, sizeof(T) * (size_t)320 * 256, qa.Wkv, sizeof(T) * (size_t)512 * 64);
  __syncthreads();                                   // the copy-out of y has read its panel; every panel is free
  // y sits in the SECOND [TT][LDX] panel of the FFN body; the q/kv body keeps x in the first: copy, then go on
  const Res<T>* ys = reinterpret_cast<const Res<T>*>(chain_smem) + TT * LDX;
  qkv_fwd_body<T, TT, RM>(qa, chain_smem, ys, rope);
}

template <typename T, int TT>
static void qkv_fwd_fused_launch_tt(hipStream_t s, const QkvFwdArgs& a) {
  constexpr int RM = TT <= 16 ? 1 : 2;     // rotated form: table entries in registers on the short tile, staged in LDS on longer ones
  constexpr size_t lds = sizeof(T) * (size_t)TT * ((256 + 8) + (320 + 8) + (512 + 8)) + (RM == 2 ? (size_t)TT * 64 * sizeof(float) : 0);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static_assert(TT != 32 || sizeof(T) != 2 || 2 * lds <= 160 * 1024, "two workgroups per compute unit on the 32-token bf16 tile");
  static bool once = (lds > 64 * 1024
                          ? ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(qkv_fwd_fused_kernel<T, TT, 0>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                             (void)hipFuncSetAttribute(reinterpret_cast<const void*>(qkv_fwd_fused_kernel<T, TT, RM>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
                          : (void)0,
                      true);
  (void)once;
  if (a.rope_cos != nullptr) hipLaunchKernelGGL((qkv_fwd_fused_kernel<T, TT, RM>), dim3((a.N + TT - 1) / TT), dim3(256), lds, s, a);
  else hipLaunchKernelGGL((qkv_fwd_fused_kernel<T, TT, 0>), dim3((a.N + TT - 1) / TT), dim3(256), lds, s, a);
}
template <typename T>
static void qkv_fwd_fused_launch(hipStream_t s, const QkvFwdArgs& a) {
  // tile length: 16 tokens while that is what fills the chip, 32 from 8192 tokens on (AMT_QKV_FWD_TT=64: 64-token tiles
  // from 49152 tokens on - three 128-KiB panels per workgroup, one workgroup per compute unit: slower)
  static const int tt_env = getenv("AMT_QKV_FWD_TT") ? atoi(getenv("AMT_QKV_FWD_TT")) : 0;
  if (a.N < 8192) qkv_fwd_fused_launch_tt<T, FfnTileFwd<T>::TT>(s, a);
  else if (sizeof(T) == 2 && a.N >= 49152 && tt_env == 64) qkv_fwd_fused_launch_tt<T, 2 * FfnTile<T>::TT>(s, a);   // opt-in: measured 79 against 63 us at 112 000 tokens
  else qkv_fwd_fused_launch_tt<T, FfnTile<T>::TT>(s, a);
}

template <typename T>
static void ffn_qkv_fwd_fused_launch(hipStream_t s, const FfnFwdArgs& fa, const QkvFwdArgs& qa) {
  constexpr int TT = FfnTileFwd<T>::TT;
  constexpr size_t lds_f = sizeof(T) * (size_t)TT * (3 * (256 + 8) + 512 + 8);
  constexpr size_t lds_q = sizeof(T) * (size_t)TT * ((256 + 8) + (320 + 8) + (512 + 8));
  constexpr size_t lds = lds_f > lds_q ? lds_f : lds_q;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool once = (lds > 64 * 1024
                          ? ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_qkv_fwd_fused_kernel<T, TT, 0>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                             (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_qkv_fwd_fused_kernel<T, TT, 1>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
                          : (void)0,
                      true);
  (void)once;
  if (qa.rope_cos != nullptr) hipLaunchKernelGGL((ffn_qkv_fwd_fused_kernel<T, TT, 1>), dim3((fa.N + TT - 1) / TT), dim3(256), lds, s, fa, qa);
  else hipLaunchKernelGGL((ffn_qkv_fwd_fused_kernel<T, TT, 0>), dim3((fa.N + TT - 1) / TT), dim3(256), lds, s, fa, qa);
}

bool ffn_qkv_fwd_fused(hipStream_t s, DType dt, const FfnFwdArgs& fa, const QkvFwdArgs& qa) {
  if (fa.N != qa.N || fa.N >= 8192 || fa.W1 == nullptr) return false;   // the 16-token forward tile only (training sizes)
  const double N = fa.N, es = dt == DT_F32 ? 4 : 2;
  ProfScope ps(s, "ffn_qkv_fwd_fused", 2.0 * N * (256.0 * 256 + 1024.0 * 256 + 256.0 * 512 + 320.0 * 256 + 512.0 * 64),
               (N * (256.0 * 7 + 1024 + 512 + 320 + 512) + 256.0 * 256 + 1024.0 * 256 + 256.0 * 512 + 320.0 * 256 + 512.0 * 64) * es);
  if (dt == DT_F32) ffn_qkv_fwd_fused_launch<float>(s, fa, qa);
  else ffn_qkv_fwd_fused_launch<bf16>(s, fa, qa);
  return true;
}

void qkv_fwd_fused(hipStream_t s, DType dt, const QkvFwdArgs& a) {
  const double N = a.N, es = dt == DT_F32 ? 4 : 2;
  ProfScope ps(s, "qkv_fwd_fused", 2.0 * N * (320.0 * 256 + 512.0 * 64), (N * (256.0 * 2 + 320 + 512) + 320.0 * 256 + 512.0 * 64) * es);
  if (dt == DT_F32) qkv_fwd_fused_launch<float>(s, a);
  else qkv_fwd_fused_launch<bf16>(s, a);
}


// ------------------------------------------------------------------ transformer layer backward chains
template <typename T, int TT>
__global__ __launch_bounds__(256, 2) void qkv_bwd_fused_kernel(QkvBwdArgs a, WgradGroup riders) {   // two waves per SIMD: chain + rider workgroups share compute units
  a.seed = eff_seed(a.seed, a.seed_dev);   // kernel argument + the context's device seed word (common.h)
  constexpr int H = 256, QC = 320, CK = 64, KV = 512, LDX = H + 8, LDQ = QC + 8, LDK = KV + 8;
  using SC = ChainSplit<CK, TT>;     // dc: 64 features
  using SX = ChainSplit<H, TT>;      // dxn: 256 features
  constexpr int KCC = ChainChunk<T, SC::RB, KV>::KC, KCX = ChainChunk<T, SX::RB, QC, 32>::KC;
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  if ((int)blockIdx.x >= a.n_chain) {   // weight gradients of the previous scope ride on this launch
    wgrad_group_dispatch<T>(riders, (int)blockIdx.x - a.n_chain, chain_smem);
    return;
  }
  T* Ks = reinterpret_cast<T*>(chain_smem);   // [TT][LDK]  dkv
  T* Qs = Ks + TT * LDK;                      // [TT][LDQ]  dqc (dq | dc)
  T* Xs = Qs + TT * LDQ;                      // [TT][LDX]  dxn, then dx
  float* part = reinterpret_cast<float*>(chain_smem);   // [4][2 * 256] over Ks once it is dead

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const long N = a.N, n0 = (long)blockIdx.x * TT;
  const T* Wk = reinterpret_cast<const T*>(a.Wkvt);
  const T* Wq = reinterpret_cast<const T*>(a.Wqct);
  const int mC = (wave % SC::WF) * SC::RB * 16, tC = (wave / SC::WF) * SC::NT * 16;
  const int mX = (wave % SX::WF) * SX::RB * 16, tX = (wave / SX::WF) * SX::NT * 16;

  AMT_STAMP_RT(6);
  AMT_STAMP(0);
  uint32_t warm = 0;   // cooperative L2 warm-up (chain.h), as in the FFN chains
  {
    const int q = blockIdx.x >> 3, ns = max(1, a.n_chain >> 3);
    chain_l2_warm<2>(Wk, sizeof(T) * (size_t)CK * KV, q, ns, tid, warm);
    chain_l2_warm<4>(Wq, sizeof(T) * chain_packed_elems(H, QC), q, ns, tid, warm);
  }
  Frag<T> wk[KCC][SC::RB], wq[KCX][SX::RB];
  chain_load_w<T, SC::RB, KCC, KV>(wk, Wk, mC, 0, lane);
  chain_load_w<T, SX::RB, KCX, QC>(wq, Wq, mX, 0, lane);
  chain_copy_in<T, TT, KV, LDK>(Ks, reinterpret_cast<const T*>(a.dKV), n0, N, tid);
  // dq part of dqc (320-wide rows, the first 256 columns): multiplied by the SECOND stage only, so it lands while the
  // first one runs and goes to LDS behind its products (the barrier that follows the stage orders it)
  ChainCopyRegs<T, TT, H> dqreg;
  chain_copy_load<T, TT, H>(dqreg, reinterpret_cast<const T*>(a.dQC), QC, n0, N, tid);
  __syncthreads();
  AMT_STAMP(1);

  // ---- dc = dkv . Wkv  -> columns 256..319 of the dqc panel and of dQC
  {
    f32x4 acc[SC::RB][SC::NT];
    chain_zero(acc);
    chain_stage<T, SC::RB, SC::NT, KCC, KV>(acc, wk, Wk, mC, Ks, LDK, tC, lane);
    chain_copy_store<T, TT, H, LDQ>(Qs, dqreg, tid);
#pragma unroll
    for (int rb = 0; rb < SC::RB; ++rb)
#pragma unroll
      for (int ti = 0; ti < SC::NT; ++ti) {
        const int p = tC + 16 * ti + r, m = mC + 16 * rb + 4 * g;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[rb][ti][i];
        store4f(Qs + p * LDQ + H + m, v);
        if (n0 + p < N) store4f(reinterpret_cast<T*>(a.dQC) + (n0 + p) * QC + H + m, v);
      }
  }
  __syncthreads();
  AMT_STAMP(2);

  // ---- dxn = dqc . Wqc
  {
    f32x4 acc[SX::RB][SX::NT];
    chain_zero(acc);
    chain_stage<T, SX::RB, SX::NT, KCX, QC>(acc, wq, Wq, mX, Qs, LDQ, tX, lane);
#pragma unroll
    for (int rb = 0; rb < SX::RB; ++rb)
#pragma unroll
      for (int ti = 0; ti < SX::NT; ++ti) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[rb][ti][i];
        store4f(Xs + (tX + 16 * ti + r) * LDX + mX + 16 * rb + 4 * g, v);
      }
  }
  __syncthreads();
  AMT_STAMP(3);

  // ---- dx = dh + LN1'(dxn) (+ masked copy), column sums
  float cs[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { cs[0][i] = 0.f; cs[1][i] = 0.f; }
  chain_ln_bwd_rows<T, TT, LDX>(Xs, Xs, reinterpret_cast<const Res<T>*>(a.X), a.mean, a.rstd, a.ln_w,
                                reinterpret_cast<const T*>(a.dH), reinterpret_cast<T*>(a.dX),
                                a.drop_thresh24 ? reinterpret_cast<T*>(a.dXD) : nullptr, a.seed, a.stream,
                                a.drop_thresh24, a.inv_keep, cs, n0, N, tid);
  AMT_STAMP(4);
  fold_columns<32, 16>(*reinterpret_cast<float (*)[16]>(&cs[0][0]), part + wave * 2 * H, H, lane);
  __syncthreads();
  float* rec = a.rec + (size_t)blockIdx.x * a.rec_stride;
  for (int t = tid; t < 2 * H; t += 256)
    rec[t] = (part[t] + part[2 * H + t]) + (part[4 * H + t] + part[6 * H + t]);
  if (warm == 0x9e3779b9u && a.N < 0) rec[0] = 0.f;   // keeps the warm-up loads alive; never true
  AMT_STAMP(5);
  AMT_STAMP_RT(7);
}

template <typename T, int TT>
__global__ __launch_bounds__(256) void lnout_bwd_fused_kernel(LnOutBwdArgs a, WgradGroup riders) {
  constexpr int H = 256, LDX = H + 8;
  using SO = ChainSplit<H, TT>;
  constexpr int KCO = ChainChunk<T, SO::RB, H>::KC;   // 16 fragments per set: two waves per SIMD with the riders
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  if ((int)blockIdx.x >= a.n_chain) {
    wgrad_group_dispatch<T>(riders, (int)blockIdx.x - a.n_chain, chain_smem);
    return;
  }
  T* Ds = reinterpret_cast<T*>(chain_smem);   // [TT][LDX]  dhn, then dh
  T* Os = Ds + TT * LDX;                      // [TT][LDX]  dao (staged for the copy-out)
  float* part = reinterpret_cast<float*>(Os); // [4][2 * 256] before Os is used

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const long N = a.N, n0 = (long)blockIdx.x * TT;
  const T* Wo = reinterpret_cast<const T*>(a.Wot);
  const int mO = (wave % SO::WF) * SO::RB * 16, tO = (wave / SO::WF) * SO::NT * 16;

  AMT_STAMP_RT(6);
  AMT_STAMP(0);
  Frag<T> wo[KCO][SO::RB];
  chain_load_w<T, SO::RB, KCO, H>(wo, Wo, mO, 0, lane);
  chain_copy_in<T, TT, H, LDX>(Ds, reinterpret_cast<const T*>(a.dHN), n0, N, tid);
  __syncthreads();
  AMT_STAMP(1);
  float cs[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { cs[0][i] = 0.f; cs[1][i] = 0.f; }
  chain_ln_bwd_rows<T, TT, LDX>(Ds, Ds, reinterpret_cast<const Res<T>*>(a.Hres), a.mean, a.rstd, a.ln_w,
                                reinterpret_cast<const T*>(a.dY), reinterpret_cast<T*>(a.dH), nullptr, 0, 0, 0, 1.f, cs,
                                n0, N, tid);
  fold_columns<32, 16>(*reinterpret_cast<float (*)[16]>(&cs[0][0]), part + wave * 2 * H, H, lane);
  __syncthreads();
  AMT_STAMP(2);
  {
    float* rec = a.rec + (size_t)blockIdx.x * a.rec_stride;
    for (int t = tid; t < 2 * H; t += 256)
      rec[t] = (part[t] + part[2 * H + t]) + (part[4 * H + t] + part[6 * H + t]);
  }
  // ---- dao = dh . Wo
  f32x4 acc[SO::RB][SO::NT];
  chain_zero(acc);
  chain_stage<T, SO::RB, SO::NT, KCO, H>(acc, wo, Wo, mO, Ds, LDX, tO, lane);
  __syncthreads();                                   // every thread has read its record values from `part`
  AMT_STAMP(3);
#pragma unroll
  for (int rb = 0; rb < SO::RB; ++rb)
#pragma unroll
    for (int ti = 0; ti < SO::NT; ++ti) {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = acc[rb][ti][i];
      store4f(Os + (tO + 16 * ti + r) * LDX + mO + 16 * rb + 4 * g, v);
    }
  __syncthreads();
  AMT_STAMP(4);
  chain_copy_out<T, TT, H, LDX>(Os, reinterpret_cast<T*>(a.dAO), n0, N, tid);
  AMT_STAMP(5);
  AMT_STAMP_RT(7);
}

// ------------------------------------------------------------------ transformer layer, second half, backward (whole)
// One launch for gemm_nt<glu_bwd> + gemm_nt<store> + lnout_bwd_fused (kernels.h: FfnBwdArgs).  Every stage repeats its
// unfused launch: same K order into one accumulator, du and dhn rounded to T where the unfused path stores them.
template <typename T, int TT>
__global__ __launch_bounds__(256) void ffn_bwd_fused_kernel(FfnBwdArgs a, WgradGroup riders) {
  constexpr int H = 256, I = 512, I2 = 1024, LDX = H + 8, LDU = I2 + 8;
  using SO = ChainSplit<H, TT>;                        // every stage is cut into 256 output features per pass
  constexpr int KCA = ChainChunk<T, SO::RB, H>::KC;    // dg halves: K = 256
  constexpr int KCB = ChainChunk<T, SO::RB, I2>::KC;   // dhn:       K = 1024
  constexpr int KCO = ChainChunk<T, SO::RB, H>::KC;    // dao:       K = 256
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  if ((int)blockIdx.x >= a.n_chain) {
    wgrad_group_dispatch<T>(riders, (int)blockIdx.x - a.n_chain, chain_smem);
    return;
  }
  T* Ds = reinterpret_cast<T*>(chain_smem);   // [TT][LDX]  dyd, then dhn, then dh
  T* Os = Ds + TT * LDX;                      // [TT][LDX]  dao (staged for the copy-out)
  float* part = reinterpret_cast<float*>(Os); // [4][2 * 256] before Os is used
  T* Us = Os + TT * LDX;                      // [TT][LDU]  u, overwritten in place by du

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const long N = a.N, n0 = (long)blockIdx.x * TT;
  const T* W2t = reinterpret_cast<const T*>(a.W2t);
  const T* W1t = reinterpret_cast<const T*>(a.W1t);
  const T* Wo = reinterpret_cast<const T*>(a.Wot);
  const int mO = (wave % SO::WF) * SO::RB * 16, tO = (wave / SO::WF) * SO::NT * 16;

  // cooperative L2 warm-up of the two big weight matrices, as in ffn_fwd_fused: every workgroup of an XCD streams the
  // same 896 KiB and a cold stream advances at one compute unit's miss rate
  uint32_t warm = 0;
  L2Warm<3> lw2;   // back-to-back form (chain.h)
  L2Warm<6> lw1;
  {
    const int q = blockIdx.x >> 3, ns = max(1, a.n_chain >> 3);
    lw2.issue(W2t, sizeof(T) * (size_t)I * H, q, ns, tid);
    lw1.issue(W1t, sizeof(T) * (size_t)H * I2, q, ns, tid);
  }
  Frag<T> wa[KCA][SO::RB], wb[KCB][SO::RB];
  chain_load_w<T, SO::RB, KCA, H>(wa, W2t, mO, 0, lane);
  chain_copy_in<T, TT, H, LDX>(Ds, reinterpret_cast<const T*>(a.dYD), n0, N, tid);
  ChainCopyRegs<T, TT, I2> ureg;   // u is read by the GLU' epilogue only: it lands while the first products run
  chain_copy_load<T, TT, I2>(ureg, reinterpret_cast<const T*>(a.U), I2, n0, N, tid);
  __syncthreads();
  lw2.fold(); lw1.fold();

  // ---- dg = dyd . W2 (two halves of 256 columns) ; du = GLU'(dg, u), in place over u
  auto glu_bwd_half = [&](int half, Frag<T> (&wpre)[KCA][SO::RB]) {
    f32x4 acc[SO::RB][SO::NT];
    chain_zero(acc);
    chain_stage<T, SO::RB, SO::NT, KCA, H>(acc, wpre, W2t, 256 * half + mO, Ds, LDX, tO, lane);
    if (half == 0) {
      chain_copy_store<T, TT, I2, LDU>(Us, ureg, tid);
      __syncthreads();
    }
    // the next stage's first weights are requested before the epilogue (exp, LDS traffic) runs
    if (half == 0) chain_load_w<T, SO::RB, KCA, H>(wpre, W2t, 256 + mO, 0, lane);
    else chain_load_w<T, SO::RB, KCB, I2>(wb, W1t, mO, 0, lane);
#pragma unroll
    for (int rb = 0; rb < SO::RB; ++rb) {
      const int j = 256 * half + mO + 16 * rb + 4 * g;   // dg column; u lives in the permuted (16 x u1 | 16 x u2) layout
      const int pm = (j >> 4) * 32 + (j & 15);
#pragma unroll
      for (int ti = 0; ti < SO::NT; ++ti) {
        T* up = Us + (tO + 16 * ti + r) * LDU + pm;
        float u1[4], u2[4], d1[4], d2[4];
        load4f(up, u1);
        load4f(up + 16, u2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = acc[rb][ti][i];
          d1[i] = v * u2[i] * gelu_tanh_grad(u1[i]);
          d2[i] = v * gelu_tanh(u1[i]);
        }
        store4f(up, d1);
        store4f(up + 16, d2);
      }
    }
  };
  glu_bwd_half(0, wa);
  glu_bwd_half(1, wa);
  __syncthreads();
  chain_copy_out<T, TT, I2, LDU>(Us, reinterpret_cast<T*>(a.dU), n0, N, tid);   // for the W1 weight gradient

  // ---- dhn = du . W1 (K = 1024) -> Ds (dyd is dead)
  Frag<T> wo[KCO][SO::RB];
  {
    f32x4 acc[SO::RB][SO::NT];
    chain_zero(acc);
    chain_stage<T, SO::RB, SO::NT, KCB, I2>(acc, wb, W1t, mO, Us, LDU, tO, lane);
    chain_load_w<T, SO::RB, KCO, H>(wo, Wo, mO, 0, lane);
#pragma unroll
    for (int rb = 0; rb < SO::RB; ++rb)
#pragma unroll
      for (int ti = 0; ti < SO::NT; ++ti) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[rb][ti][i];
        store4f(Ds + (tO + 16 * ti + r) * LDX + mO + 16 * rb + 4 * g, v);
      }
  }
  __syncthreads();

  // ---- dh = dy + LN2'(dhn) ; dln2 w / b column sums (as lnout_bwd_fused)
  float cs[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { cs[0][i] = 0.f; cs[1][i] = 0.f; }
  chain_ln_bwd_rows<T, TT, LDX>(Ds, Ds, reinterpret_cast<const Res<T>*>(a.Hres), a.mean, a.rstd, a.ln_w,
                                reinterpret_cast<const T*>(a.dY), reinterpret_cast<T*>(a.dH), nullptr, 0, 0, 0, 1.f, cs,
                                n0, N, tid);
  fold_columns<32, 16>(*reinterpret_cast<float (*)[16]>(&cs[0][0]), part + wave * 2 * H, H, lane);
  __syncthreads();
  {
    float* rec = a.rec + (size_t)blockIdx.x * a.rec_stride;
    for (int t = tid; t < 2 * H; t += 256)
      rec[t] = (part[t] + part[2 * H + t]) + (part[4 * H + t] + part[6 * H + t]);
  }
  // ---- dao = dh . Wo
  f32x4 acc[SO::RB][SO::NT];
  chain_zero(acc);
  chain_stage<T, SO::RB, SO::NT, KCO, H>(acc, wo, Wo, mO, Ds, LDX, tO, lane);
  __syncthreads();                                   // every thread has read its record values from `part`
#pragma unroll
  for (int rb = 0; rb < SO::RB; ++rb)
#pragma unroll
    for (int ti = 0; ti < SO::NT; ++ti) {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = acc[rb][ti][i];
      store4f(Os + (tO + 16 * ti + r) * LDX + mO + 16 * rb + 4 * g, v);
    }
  __syncthreads();
  chain_copy_out<T, TT, H, LDX>(Os, reinterpret_cast<T*>(a.dAO), n0, N, tid);
  if (warm == 0x9e3779b9u && a.N < 0) a.rec[0] = 0.f;   // keeps the warm-up loads alive; never true
}

template <typename K> static void chain_set_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <typename T, int TT>
static void qkv_bwd_fused_launch(hipStream_t s, const QkvBwdArgs& a, const WgradGroup& rg, int n_rider_wgs) {
  constexpr size_t lds_chain = sizeof(T) * (size_t)TT * ((512 + 8) + (320 + 8) + (256 + 8));
  constexpr size_t lds = lds_chain > TnLds<T>::BYTES ? lds_chain : TnLds<T>::BYTES;
  static_assert(lds <= 160 * 1024 && sizeof(T) * TT * (512 + 8) >= sizeof(float) * 8 * 256, "LDS budget");
  static bool once = (chain_set_lds(qkv_bwd_fused_kernel<T, TT>, lds), true);
  (void)once;
  hipLaunchKernelGGL((qkv_bwd_fused_kernel<T, TT>), dim3(a.n_chain + n_rider_wgs), dim3(256), lds, s, a, rg);
}
template <typename T>
static void lnout_bwd_fused_launch(hipStream_t s, const LnOutBwdArgs& a, const WgradGroup& rg, int n_rider_wgs) {
  constexpr int TT = FfnTile<T>::TT;
  constexpr size_t lds_chain = sizeof(T) * (size_t)TT * 2 * (256 + 8);
  constexpr size_t lds = lds_chain > TnLds<T>::BYTES ? lds_chain : TnLds<T>::BYTES;
  static_assert(sizeof(T) * TT * (256 + 8) >= sizeof(float) * 8 * 256, "column-sum slices fit in one panel");
  static bool once = (chain_set_lds(lnout_bwd_fused_kernel<T, TT>, lds), true);
  (void)once;
  hipLaunchKernelGGL((lnout_bwd_fused_kernel<T, TT>), dim3(a.n_chain + n_rider_wgs), dim3(256), lds, s, a, rg);
}

// 16-token tiles (250 workgroups at cfg2, 50 KiB of LDS: riders share the compute units); 32-token tiles for bf16 ran
// 51 us per launch against 43 us for the three launches it replaces
constexpr int kFfnBwdTile = 16;
template <typename T, int TT>
static void ffn_bwd_fused_launch(hipStream_t s, const FfnBwdArgs& a, const WgradGroup& rg, int n_rider_wgs) {
  constexpr size_t lds_chain = sizeof(T) * (size_t)TT * (2 * (256 + 8) + (1024 + 8));
  constexpr size_t lds = lds_chain > TnLds<T>::BYTES ? lds_chain : TnLds<T>::BYTES;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static_assert(sizeof(T) * TT * (256 + 8) >= sizeof(float) * 8 * 256, "column-sum slices fit in one panel");
  static bool once = (chain_set_lds(ffn_bwd_fused_kernel<T, TT>, lds), true);
  (void)once;
  hipLaunchKernelGGL((ffn_bwd_fused_kernel<T, TT>), dim3(a.n_chain + n_rider_wgs), dim3(256), lds, s, a, rg);
}

static int chain_ln_records(DType dt, int N, float* dlnw, float* dlnb, PartialSink* sink, float** rec, int tile = 0) {
  const int TT = tile > 0 ? tile : (dt == DT_F32 ? 16 : 32), nb = (N + TT - 1) / TT;   // default == FfnTile<T>::TT
  size_t o = sink->take((size_t)nb * 512);
  sink->add(o, nb, 512, 256, dlnw);
  sink->add(o + 256, nb, 512, 256, dlnb);
  *rec = sink->dry ? nullptr : sink->ws + o;
  return nb;
}

void qkv_bwd_fused(hipStream_t s, DType dt, const QkvBwdArgs& a_in, float* dlnw, float* dlnb, PartialSink* sink,
                   const WgradArgs* riders, int n_riders) {
  QkvBwdArgs a = a_in;
  // Tile length: 32 tokens in bf16 (production), 16 in f32; AMT_QKV_BWD_TT = 16 | 32 forces either for BOTH dtypes (read on
  // every call, so a test can switch it between contexts): the f32 context with 32-token tiles runs the production tile
  // shape of the bf16 path and pins its indexing at the f32 tolerance (tests/test_parity_full_gpu.py).
  const char* tt_s = getenv("AMT_QKV_BWD_TT");
  const int tt_env = tt_s ? atoi(tt_s) : 0;
  const int tt = tt_env == 32 ? 32 : ((dt == DT_F32 || (tt_env == 16 && a.N < 8192)) ? 16 : 32);
  a.n_chain = chain_ln_records(dt, a.N, dlnw, dlnb, sink, &a.rec, tt);
  WgradGroup rg;
  double rfl = 0, rby = 0;
  static const int tgt_qkv = getenv("AMT_RIDER_TARGET_QKV") ? atoi(getenv("AMT_RIDER_TARGET_QKV")) : 0;   // 0: the weight-gradient default
  const int n_rider_wgs = wgrad_group_prepare(dt, riders, n_riders, sink, &rg, &rfl, &rby, tgt_qkv);
  if (sink->dry) return;
  a.rec_stride = 512;
  const double N = a.N, es = dt == DT_F32 ? 4 : 2;
  ProfScope ps(s, n_rider_wgs > 0 ? "qkv_bwd_fused+wgrad" : "qkv_bwd_fused", 2.0 * N * (64.0 * 512 + 256.0 * 320) + rfl,
               (N * (512 + 320 + 64 + 256 * 4.0) + 64.0 * 512 + 256.0 * 320) * es + rby);
  if (dt == DT_F32 && tt == 16) qkv_bwd_fused_launch<float, 16>(s, a, rg, n_rider_wgs);
  else if (dt == DT_F32) qkv_bwd_fused_launch<float, 32>(s, a, rg, n_rider_wgs);
  else if (tt == 16) qkv_bwd_fused_launch<bf16, 16>(s, a, rg, n_rider_wgs);
  else qkv_bwd_fused_launch<bf16, 32>(s, a, rg, n_rider_wgs);
}

void lnout_bwd_fused(hipStream_t s, DType dt, const LnOutBwdArgs& a_in, float* dlnw, float* dlnb, PartialSink* sink,
                     const WgradArgs* riders, int n_riders) {
  LnOutBwdArgs a = a_in;
  a.n_chain = chain_ln_records(dt, a.N, dlnw, dlnb, sink, &a.rec);
  WgradGroup rg;
  double rfl = 0, rby = 0;
  const int n_rider_wgs = wgrad_group_prepare(dt, riders, n_riders, sink, &rg, &rfl, &rby);
  if (sink->dry) return;
  a.rec_stride = 512;
  const double N = a.N, es = dt == DT_F32 ? 4 : 2;
  ProfScope ps(s, n_rider_wgs > 0 ? "lnout_bwd_fused+wgrad" : "lnout_bwd_fused", 2.0 * N * 256.0 * 256 + rfl,
               (N * 256 * 5.0 + 256.0 * 256) * es + rby);
  if (dt == DT_F32) lnout_bwd_fused_launch<float>(s, a, rg, n_rider_wgs);
  else lnout_bwd_fused_launch<bf16>(s, a, rg, n_rider_wgs);
}

void ffn_bwd_fused(hipStream_t s, DType dt, const FfnBwdArgs& a_in, float* dlnw, float* dlnb, PartialSink* sink,
                   const WgradArgs* riders, int n_riders) {
  FfnBwdArgs a = a_in;
  // 16-token tiles at every size; AMT_FFN_BWD_TT=32: 32-token tiles from 8192 tokens on (bf16) - half the weight
  // streams, but 100 KiB of LDS per workgroup leave one workgroup per compute unit
  static const int tt_env = getenv("AMT_FFN_BWD_TT") ? atoi(getenv("AMT_FFN_BWD_TT")) : 0;
  const int tt = (dt == DT_BF16 && a.N >= 8192 && tt_env == 32) ? 32 : kFfnBwdTile;   // opt-in: measured 110 against 82 us at 16 000 tokens
  a.n_chain = chain_ln_records(dt, a.N, dlnw, dlnb, sink, &a.rec, tt);
  WgradGroup rg;
  double rfl = 0, rby = 0;
  static const int tgt_ffn = getenv("AMT_RIDER_TARGET_FFN") ? atoi(getenv("AMT_RIDER_TARGET_FFN")) : 264;   // 250 chain + ~264 rider workgroups fill the 512 resident slots (0 / 176 / 220 / 264 / 308: 3.255 / 3.252 / 3.236 / 3.235 / 3.260 ms)
  const int n_rider_wgs = wgrad_group_prepare(dt, riders, n_riders, sink, &rg, &rfl, &rby, tgt_ffn);
  if (sink->dry) return;
  a.rec_stride = 512;
  const double N = a.N, es = dt == DT_F32 ? 4 : 2;
  ProfScope ps(s, n_rider_wgs > 0 ? "ffn_bwd_fused+wgrad" : "ffn_bwd_fused",
               2.0 * N * (512.0 * 256 + 256.0 * 1024 + 256.0 * 256) + rfl,
               (N * (256 * 5.0 + 2.0 * 1024) + 512.0 * 256 + 256.0 * 1024 + 256.0 * 256) * es + rby);
  if (dt == DT_F32) ffn_bwd_fused_launch<float, kFfnBwdTile>(s, a, rg, n_rider_wgs);
  else if (tt == 32) ffn_bwd_fused_launch<bf16, 32>(s, a, rg, n_rider_wgs);
  else ffn_bwd_fused_launch<bf16, kFfnBwdTile>(s, a, rg, n_rider_wgs);
}

}  // namespace amt
