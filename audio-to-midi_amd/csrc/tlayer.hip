// XCD-team forward of the transformer stack: every layer of model.py:559-670 in ONE launch (kernels.h: TLayerArgs).
//
// Why.  At 8 clips x 500 frames the launch-per-operator chains (chain.hip) give every one of 250 workgroups 16 tokens and
// make it stream the layer's whole weight set (1.1 MiB) from L2: 16 flop per weight byte, one wave per SIMD, and the
// transformer forward runs at ~8 % of the matrix peak.  Here a clip belongs to the 32 workgroups of ONE XCD and each
// per-token stage is cut in two dimensions, token groups of TG rows x 8 feature groups: a workgroup multiplies TG = 128
// tokens with 1/8 of the stage's weights (128 flop per weight byte, every A fragment feeds 2-8 MFMAs), 8 waves per
// compute unit, and what a stage produces goes to the next stage's consumers through that XCD's L2 (team.h: plain
// stores, drained; sc1 loads; one flag word per workgroup, waits on the 8 producers a consumer needs, never on the team).
// Nothing leaves the launch between layers: no kernel boundary, no write-back drain in the dependent chain.
//
// Numerics.  Every stage repeats its stand-alone form: K order of chain_stage (= gemm_nt), the LayerNorm row arithmetic of
// chain_ln_rows (32 lanes per row, the same reduction tree), roundings to T / Res<T> where qkv_fwd_fused / ffn_fwd_fused
// store, rope_quad, the attention tile bodies of attn_fwd.h, the dropout indices.  The saved tensors are bit-identical to
// the launch-per-operator path (tests/test_model_gpu.py).
#include "attn_fwd.h"
#include "chain.h"
#include "prof.h"
#include "team.h"

namespace amt {

namespace {

// every phase is a function of its own: its registers are allocated for that phase alone (inlined into one body the five
// phases kept each other's addresses and weight fragments alive: 256 registers and 70 spilled)
#define TL_PHASE __device__ __attribute__((noinline))
constexpr int TL_THREADS = 512;
constexpr int TL_NF = 4;            // feature groups per token group (= heads)
constexpr unsigned TL_SPIN_LIMIT = 1u << 21;

template <typename T> struct TlTile { static constexpr int TG = sizeof(T) == 2 ? 64 : 32; };   // rows per token group

struct TlEnv {
  int tid, lane, wave;
  int clip, S, tg, fg;
  long tok0;       // first global row of the token group
  int nvalid;      // rows of the group inside the clip
};

// Launch constants and the per-layer items live in LDS: the phases are functions of their own (TL_PHASE) and a non-kernel
// function receives its arguments in VECTOR registers - a pointer that arrives that way is divergent to the compiler (flat
// loads, waterfall loops around buffer resources).  Read from LDS and passed through readfirstlane it is scalar again.
struct TlConst {
  const float *rope_cos, *rope_sin;
  TeamState* st;
  const void* X0;
  unsigned long long seed;
  int B, S, n_layers, prerot, padL, nW;
  unsigned thr; float inv_keep;
  int xcc, rank, size;   // this workgroup's team (team.h)
  unsigned long long* stamps;   // diagnostic (TLayerArgs::stamps)
};
constexpr int TL_STAMP_SLOTS = 32;   // per (workgroup, layer): 0..5 phase boundaries, 8.. inside the phases
#define TL_SUB(k) do { unsigned long long* sp_ = tl_p(s_c.stamps); if (sp_ != nullptr && threadIdx.x == 0) sp_[((size_t)blockIdx.x * tl_u(s_c.n_layers) + l) * TL_STAMP_SLOTS + (k)] = wall_clock64(); } while (0)
__shared__ TlConst s_c;
__shared__ TLayerItem s_it[TLAYER_MAX];
__shared__ int s_wait_ok;

// a pointer read from LDS: wave-uniform (scalar registers) and known to point to global memory (global_* / buffer_*
// instructions, not flat_*)
template <typename P> AMT_DEV P* tl_p(P* p) {
  typedef __attribute__((address_space(1))) P* G;
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  // integer -> GLOBAL pointer -> generic: one address-space cast whose source is known to be global, which the compiler's
  // address-space inference propagates to every access (a generic -> global -> generic round trip is folded away first)
  return (P*)reinterpret_cast<G>(((unsigned long long)hi << 32) | lo);
}
AMT_DEV int tl_u(int v) { return __builtin_amdgcn_readfirstlane(v); }
AMT_DEV unsigned tl_u(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
AMT_DEV float tl_u(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
AMT_DEV unsigned long long tl_u(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

AMT_DEV TlEnv tl_env(int clip, int role, int TG) {
  TlEnv e;
  e.tid = threadIdx.x; e.lane = e.tid & 63; e.wave = tl_u(e.tid >> 6);
  e.clip = tl_u(clip); e.S = tl_u(s_c.S);
  role = tl_u(role);
  e.tg = role / TL_NF; e.fg = role % TL_NF;
  e.tok0 = (long)e.clip * e.S + (long)e.tg * TG;
  e.nvalid = min(TG, e.S - e.tg * TG);
  return e;
}

// ---- exchange: publish "this workgroup has finished `epoch` phases", wait for the producers a role reads from
AMT_DEV void tl_publish(unsigned epoch) {
  team_drain();                 // every storing wave: its stores have reached the XCD's L2
  __syncthreads();
  if (threadIdx.x == 0) {
    TeamState* st = tl_p(s_c.st);
    __hip_atomic_store(&st->flags[tl_u(s_c.xcc)][tl_u(s_c.rank)], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    team_drain();
  }
}
// lanes j < n of wave 0 poll the flag of the workgroup that runs role_of(j); also a workgroup barrier.  false: timed out
// (sticky in TeamState::timeout: every later wait of the context falls through)
template <typename F>
AMT_DEV bool tl_wait(unsigned need, int n, F role_of) {
  if (threadIdx.x < 64) {
    TeamState* st = tl_p(s_c.st);
    const int xcc = tl_u(s_c.xcc), size = tl_u(s_c.size);
    const int lane = threadIdx.x;
    bool ok = true;
    unsigned spins = 0;
    const int rk = lane < n ? role_of(lane) % size : 0;
    while (ok) {
      const unsigned v = lane < n ? team_ld_u32(&st->flags[xcc][rk]) : need;
      if (__all((int)(v - need) >= 0)) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > TL_SPIN_LIMIT || ((spins & 1023u) == 0u && team_ld_u32(&st->timeout[0]) != 0u)) {
        ok = false;
        if (lane == 0) __hip_atomic_store(&st->timeout[0], 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (lane == 0) s_wait_ok = ok ? 1 : 0;
  }
  __syncthreads();
  return tl_u(s_wait_ok) != 0;
}

// ---- panels
// rows tok0 .. tok0 + TG of G (row stride ld, W columns from col0) -> LDS panel [TG][LD]; rows >= nvalid are zero.
// SC1 loads: the rows were written by other workgroups of this launch.  All loads of a thread are issued before its first store.
template <typename E, int TG, int W, int LD>
AMT_DEV void tl_copy_in(E* Ps, const E* __restrict__ G, int ld, long tok0, int nvalid, int col0, int tid) {
  constexpr int CPR = W / 8, TOTAL = TG * CPR, NR = (TOTAL + TL_THREADS - 1) / TL_THREADS;
  Chunk8<E> v[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int id = tid + i * TL_THREADS;
    const int p = id / CPR, c = id % CPR;
    if (TOTAL % TL_THREADS == 0 || id < TOTAL) {
      const int pc = p < nvalid ? p : nvalid - 1;
      v[i] = team_chunk_load<true>(G, (size_t)(tok0 + pc) * ld + col0 + 8 * c);
    }
  }
  // the scheduler must not sink the loads to their stores (it did: one load, one s_waitcnt vmcnt(0), one ds_write - sixteen L2
  // round trips in series per panel, 10 us of a 12-us phase)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int id = tid + i * TL_THREADS;
    const int p = id / CPR, c = id % CPR;
    if (TOTAL % TL_THREADS == 0 || id < TOTAL) {
      if (p >= nvalid) v[i] = chunk_zero<E>();
      chunk_store(Ps + p * LD + 8 * c, v[i]);
    }
  }
}
// LDS panel rows p0 .. p1 -> global rows (W columns at col0), 16 bytes per lane
template <typename E, int W, int LD>
AMT_DEV void tl_copy_out(const E* Ps, E* __restrict__ G, int ld, long tok0, int nvalid, int col0, int p0, int p1, int tid) {
  constexpr int CPR = W / 8;
  const int total = (p1 - p0) * CPR;
  for (int id = tid; id < total; id += TL_THREADS) {
    const int p = p0 + id / CPR, c = id % CPR;
    if (p < nvalid) chunk_store(G + (size_t)(tok0 + p) * ld + col0 + 8 * c, chunk_load(Ps + p * LD + 8 * c));
  }
}

// Row LayerNorm over 256 features of the TG rows of a panel, in place (Res<T> in, T out): the row arithmetic of
// chain_ln_rows (chain.h) on 512 threads - 32 lanes per row, 16 rows per pass, up to LN_GROUP passes side by side.
template <typename T, int TG, int LD>
AMT_DEV void tl_ln_rows(const Res<T>* Xs, T* Ys, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ mean,
                        float* __restrict__ rstd, long tok0, int nvalid, int tid) {
  constexpr int LPR = 32, RPP = TL_THREADS / LPR, NP = TG / RPP, GR = NP < LN_GROUP ? NP : LN_GROUP;
  static_assert(TG % RPP == 0 && NP % GR == 0, "row phase covers the group in whole pass groups");
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
      store8f(Ys + p * LD + sub * 8, y);
      if (mean != nullptr && sub == 0 && p < nvalid) { mean[tok0 + p] = mu[j]; rstd[tok0 + p] = rs[j]; }
    }
  }
}

// rotate the two interleaved pairs of a feature quad (chain.hip: rope_quad; the same rounding sequence)
template <typename T>
AMT_DEV void tl_rope_quad(float (&v)[4], float2 c, float2 s) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float x1 = to_f(from_f<T>(v[2 * j])), x2 = to_f(from_f<T>(v[2 * j + 1]));
    const float cj = j == 0 ? c.x : c.y, sj = j == 0 ? s.x : s.y;
    v[2 * j] = __fmaf_rn(x1, cj, -__fmul_rn(x2, sj));
    v[2 * j + 1] = __fmaf_rn(x1, sj, __fmul_rn(x2, cj));
  }
}

template <typename T, int TG> struct TlDims {
  static constexpr int H = 256, QC = 320, CK = 64, KV = 512, I = 512, I2 = 1024;
  static constexpr int NTB = TG / 16, NTH = NTB / 2;         // token blocks of a group; of half a group
  static constexpr int LDX = H + 8, LDC = CK + 8, LDQ = 64 + 8, LDGS = 128 + 8, LDG = I + 8;
  static_assert(NTH >= 1 && TG % TL_NF == 0, "token group of at least 32 rows");
  static constexpr size_t ES = sizeof(T);
  static constexpr size_t LDS_A = ES * TG * (LDX + LDC + 3 * LDQ);
  static constexpr size_t LDS_BG = ES * 2 * 3 * 64 * ALD;                                      // global attention, two key-tile groups
  static constexpr size_t LDS_BL = ES * (size_t)(TG / 8 < 8 ? TG / 8 : 8) * LOCAL_FWD_LDS;    // local attention, one stripe per wave
  static constexpr size_t LDS_D = ES * TG * (3 * LDX + LDGS);
  static constexpr size_t LDS_E = ES * TG * (LDG + LDQ);
  static constexpr size_t mx(size_t a, size_t b) { return a > b ? a : b; }
  static constexpr size_t LDS_ALL = mx(mx(LDS_A, LDS_BG), mx(LDS_BL, mx(LDS_D, LDS_E)));
};

// ------------------------------------------------------------------ phase A: LN1, q / c / k / v projections
// role (tg, fg): rows of token group tg; q features 64 fg .. + 64 (= head fg), k and v features of the same head; c whole
template <typename T, int TG>
TL_PHASE bool tl_phase_a(int l, int clip, int role, unsigned need) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tl_smem[];
  using D = TlDims<T, TG>;
  using RT = Res<T>;
  constexpr int H = D::H, QC = D::QC, CK = D::CK, KV = D::KV, LDX = D::LDX, LDC = D::LDC, LDQ = D::LDQ, NTB = D::NTB, NTH = D::NTH;
  l = tl_u(l);
  const TlEnv e = tl_env(clip, role, TG);
  const TLayerItem& it = s_it[l];
  T* Xs = reinterpret_cast<T*>(tl_smem);
  RT* Xr = reinterpret_cast<RT*>(tl_smem);
  T* Cs = Xs + TG * LDX;
  T* Qs = Cs + TG * LDC;
  T* Ks = Qs + TG * LDQ;
  T* Vs = Ks + TG * LDQ;
  const int tid = e.tid, lane = e.lane, w = e.wave, r = lane & 15, g = lane >> 4, fg = e.fg;
  const T* Wqc = reinterpret_cast<const T*>(tl_p(it.Wqc));
  const T* Wkv = reinterpret_cast<const T*>(tl_p(it.Wkv));
  const bool rope = (l & 1) != 0 && tl_u(s_c.prerot) != 0;      // global layers: q / k leave rotated
  constexpr int KCX = ChainChunk<T, 1, H>::KC, KCK = ChainChunk<T, 1, CK>::KC;

  // what depends on nothing is requested before the wait: the c stage's weights, the RoPE table entries of this lane's rows
  const int fbh = w & 3, th = (w >> 2) * NTH * 16;       // (feature block, token half) of the c and q stages
  const int mC = 256 + 16 * fbh, mQ = 64 * fg + 16 * fbh;
  const bool isk = w < 4;                                // k / v stage: wave w multiplies feature block w of [k (4) | v (4)] for all rows
  const int mK = (isk ? 0 : 256) + 64 * fg + 16 * (w & 3);
  Frag<T> wc[KCX][1];
  chain_load_w<T, 1, KCX, H>(wc, Wqc, mC, 0, lane);
  float2 qcs[NTH], qsn[NTH], kcs[NTB], ksn[NTB];
  if (rope) {
    const float* rc = tl_p(s_c.rope_cos);
    const float* rsn = tl_p(s_c.rope_sin);
#pragma unroll
    for (int ti = 0; ti < NTH; ++ti) {
      int p = th + 16 * ti + r;
      p = p < e.nvalid ? p : e.nvalid - 1;
      const int pos = e.tg * TG + p;
      qcs[ti] = *reinterpret_cast<const float2*>(rc + pos * 32 + (((mQ + 4 * g) & 63) >> 1));
      qsn[ti] = *reinterpret_cast<const float2*>(rsn + pos * 32 + (((mQ + 4 * g) & 63) >> 1));
    }
    if (isk) {
#pragma unroll
      for (int ti = 0; ti < NTB; ++ti) {
        int p = 16 * ti + r;
        p = p < e.nvalid ? p : e.nvalid - 1;
        const int pos = e.tg * TG + p;
        kcs[ti] = *reinterpret_cast<const float2*>(rc + pos * 32 + (((mK + 4 * g) & 63) >> 1));
        ksn[ti] = *reinterpret_cast<const float2*>(rsn + pos * 32 + (((mK + 4 * g) & 63) >> 1));
      }
    }
  }
  if (l > 0 && !tl_wait(need, TL_NF, [&](int j) { return e.tg * TL_NF + j; })) return false;
  const RT* xin = reinterpret_cast<const RT*>(l == 0 ? tl_p(s_c.X0) : tl_p(s_it[l > 0 ? l - 1 : 0].y));
  tl_copy_in<RT, TG, H, LDX>(Xr, xin, H, e.tok0, e.nvalid, 0, tid);
  __syncthreads();
  tl_ln_rows<T, TG, LDX>(Xr, Xs, tl_p(it.ln1w), tl_p(it.ln1b), fg == 0 ? tl_p(it.m1) : nullptr, tl_p(it.r1), e.tok0, e.nvalid, tid);
  __syncthreads();
  tl_copy_out<T, H, LDX>(Xs, reinterpret_cast<T*>(tl_p(it.xn)), H, e.tok0, e.nvalid, 0, fg * (TG / TL_NF), (fg + 1) * (TG / TL_NF), tid);

  // ---- c = xn . Wc^T (all 64 features: every feature group needs them for its k / v slice)
  Frag<T> wq[KCX][1];
  {
    f32x4 acc[1][NTH];
    chain_zero(acc);
    chain_stage<T, 1, NTH, KCX, H>(acc, wc, Wqc, mC, Xs, LDX, th, lane);
    chain_load_w<T, 1, KCX, H>(wq, Wqc, mQ, 0, lane);
#pragma unroll
    for (int ti = 0; ti < NTH; ++ti) {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = acc[0][ti][i];
      store4f(Cs + (th + 16 * ti + r) * LDC + 16 * fbh + 4 * g, v);
    }
  }
  Frag<T> wk[KCK][1];
  chain_load_w<T, 1, KCK, CK>(wk, Wkv, mK, 0, lane);
  __syncthreads();   // c panel complete
  T* qc_out = reinterpret_cast<T*>(tl_p(it.qc));
  T* kv_out = reinterpret_cast<T*>(tl_p(it.kv));
  if (fg == 0) tl_copy_out<T, CK, LDC>(Cs, qc_out, QC, e.tok0, e.nvalid, 256, 0, TG, tid);

  // ---- q slice = xn . Wq[64 fg .. + 64)^T
  {
    f32x4 acc[1][NTH];
    chain_zero(acc);
    chain_stage<T, 1, NTH, KCX, H>(acc, wq, Wqc, mQ, Xs, LDX, th, lane);
#pragma unroll
    for (int ti = 0; ti < NTH; ++ti) {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = acc[0][ti][i];
      if (rope) tl_rope_quad<T>(v, qcs[ti], qsn[ti]);
      store4f(Qs + (th + 16 * ti + r) * LDQ + 16 * fbh + 4 * g, v);
    }
  }
  // ---- k / v slices = c . Wkv[..]^T, c as stored
  {
    f32x4 acc[1][NTB];
    chain_zero(acc);
    chain_stage<T, 1, NTB, KCK, CK>(acc, wk, Wkv, mK, Cs, LDC, 0, lane);
#pragma unroll
    for (int ti = 0; ti < NTB; ++ti) {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = acc[0][ti][i];
      if (rope && isk) tl_rope_quad<T>(v, kcs[ti], ksn[ti]);
      store4f((isk ? Ks : Vs) + (16 * ti + r) * LDQ + 16 * (w & 3) + 4 * g, v);
    }
  }
  __syncthreads();
  tl_copy_out<T, 64, LDQ>(Qs, qc_out, QC, e.tok0, e.nvalid, 64 * fg, 0, TG, tid);
  tl_copy_out<T, 64, LDQ>(Ks, kv_out, KV, e.tok0, e.nvalid, 64 * fg, 0, TG, tid);
  tl_copy_out<T, 64, LDQ>(Vs, kv_out, KV, e.tok0, e.nvalid, 256 + 64 * fg, 0, TG, tid);
  __syncthreads();   // the panels are rewritten by the workgroup's next role / phase
  return true;
}

// ------------------------------------------------------------------ phase B: attention core of head fg
// global layers: the 64-query tile that holds the group's rows (TG = 32: computed by the even group of the pair);
// local layers: the group's TG / 8 stripes, one per wave
template <typename T, int TG, bool DROP>
TL_PHASE bool tl_phase_b(int l, int clip, int role, unsigned need) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tl_smem[];
  static_assert(TG == 64 || TG == 32, "query tiles of 64 rows");
  l = tl_u(l);
  const TlEnv e = tl_env(clip, role, TG);
  const TLayerItem& it = s_it[l];
  const int head = e.fg;
  const int n_tg = (e.S + TG - 1) / TG;
  if (!tl_wait(need, n_tg, [&](int q) { return q * TL_NF + head; })) return false;
  const bool local = (l & 1) == 0;       // model.py:609-611: local layer first, then global
  AttnArgs aa;
  aa.QC = tl_p(it.qc); aa.ldqc = 320; aa.KV = tl_p(it.kv); aa.ldkv = 512; aa.O = tl_p(it.ao); aa.ldo = 256;
  aa.LSE = tl_p(it.lse);
  aa.cos = tl_p(s_c.rope_cos); aa.sin = tl_p(s_c.rope_sin); aa.B = tl_u(s_c.B); aa.S = e.S; aa.H = 4;
  aa.seed = tl_u(s_c.seed); aa.stream = 2u * (uint32_t)l; aa.drop_thresh24 = tl_u(s_c.thr); aa.inv_keep = tl_u(s_c.inv_keep);
  aa.seed_dev = nullptr; aa.keepbits = tl_p(it.keepbits); aa.prerot = local ? 0 : tl_u(s_c.prerot);
  if (local) {
    constexpr int NSTR = TG / 8;                // stripes of 8 rows of the group: one per wave
    if (e.wave < NSTR) {
      const int s = e.tg * NSTR + e.wave;
      if (8 * s < e.S)
        attn_local_fwd_wave<T, true, LocalWaveSync>(aa, tl_u(s_c.padL), tl_u(s_c.nW), s, head, e.clip,
                                                     reinterpret_cast<T*>(tl_smem) + e.wave * LOCAL_FWD_LDS, e.lane);
    }
  } else {
    constexpr int GPQ = 64 / TG;                // token groups per 64-query tile
    if (e.tg % GPQ == 0) attn_global_fwd_tile<T, 2, DROP, true>(aa, e.clip, head, e.tg / GPQ, tl_smem);
  }
  __syncthreads();
  return true;
}

// ------------------------------------------------------------------ phase D: h = x + ao . Wo^T (whole rows, every feature
// group of the token group computes them: no exchange between the output projection and LN2); hn = LN2(h); u, g slices
template <typename T, int TG>
TL_PHASE bool tl_phase_d(int l, int clip, int role, unsigned need) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tl_smem[];
  using D = TlDims<T, TG>;
  using RT = Res<T>;
  constexpr int H = D::H, I = D::I, I2 = D::I2, LDX = D::LDX, LDGS = D::LDGS, NTB = D::NTB;
  l = tl_u(l);
  const TlEnv e = tl_env(clip, role, TG);
  const TLayerItem& it = s_it[l];
  T* As = reinterpret_cast<T*>(tl_smem);                 // ao, later the u slice (staging)
  RT* Hs = reinterpret_cast<RT*>(As + TG * LDX);         // x, then h
  T* Ns = reinterpret_cast<T*>(Hs + TG * LDX);           // hn
  T* Gs = Ns + TG * LDX;                                 // g slice (staging)
  T* Us = As;
  const int tid = e.tid, lane = e.lane, w = e.wave, r = lane & 15, g = lane >> 4, fg = e.fg;
  const T* Wo = reinterpret_cast<const T*>(tl_p(it.Wo));
  const T* W1 = reinterpret_cast<const T*>(tl_p(it.W1));
  constexpr int KCO = ChainChunk<T, 2, H>::KC, KC1 = ChainChunk<T, 2, H>::KC;
  const int mO = 32 * w;                                 // output projection: wave w owns features 32 w .. + 32 of every row
  const int m1 = 256 * fg + 32 * w;                      // W1: (u1 | u2) row-block pair 8 fg + w
  Frag<T> wo[KCO][2];
  chain_load_w<T, 2, KCO, H>(wo, Wo, mO, 0, lane);
  float b1a[4], b1b[4];
  {
    const float* b1 = tl_p(it.b1);
#pragma unroll
    for (int i = 0; i < 4; ++i) { b1a[i] = b1[m1 + 4 * g + i]; b1b[i] = b1[m1 + 16 + 4 * g + i]; }
  }
  TL_SUB(20);
  // ao rows of the group: written by the four heads' roles of this group (global, TG = 32: of the group that ran the tile)
  const bool local = (l & 1) == 0;
  const int tgp = local ? e.tg : e.tg - e.tg % (64 / TG);
  if (!tl_wait(need, TL_NF, [&](int j) { return tgp * TL_NF + j; })) return false;
  TL_SUB(21);
  const RT* xin = reinterpret_cast<const RT*>(l == 0 ? tl_p(s_c.X0) : tl_p(s_it[l > 0 ? l - 1 : 0].y));
  tl_copy_in<T, TG, H, LDX>(As, reinterpret_cast<const T*>(tl_p(it.ao)), H, e.tok0, e.nvalid, 0, tid);
  tl_copy_in<RT, TG, H, LDX>(Hs, xin, H, e.tok0, e.nvalid, 0, tid);
  __syncthreads();
  TL_SUB(22);
  Frag<T> w1[KC1][2];
  {
    f32x4 acc[2][NTB];
    chain_zero(acc);
    chain_stage<T, 2, NTB, KCO, H>(acc, wo, Wo, mO, As, LDX, 0, lane);
    chain_load_w<T, 2, KC1, H>(w1, W1, m1, 0, lane);      // in flight during the LayerNorm pass
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int ti = 0; ti < NTB; ++ti) {
        RT* hp = Hs + (16 * ti + r) * LDX + mO + 16 * rb + 4 * g;
        float x[4], h[4];
        load4f(hp, x);
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = acc[rb][ti][i] + x[i];
        store4f(hp, h);
      }
  }
  __syncthreads();
  TL_SUB(23);
  tl_copy_out<RT, H, LDX>(Hs, reinterpret_cast<RT*>(tl_p(it.h)), H, e.tok0, e.nvalid, 0, fg * (TG / TL_NF), (fg + 1) * (TG / TL_NF), tid);
  tl_ln_rows<T, TG, LDX>(Hs, Ns, tl_p(it.ln2w), tl_p(it.ln2b), fg == 0 ? tl_p(it.m2) : nullptr, tl_p(it.r2), e.tok0, e.nvalid, tid);
  __syncthreads();
  TL_SUB(24);
  tl_copy_out<T, H, LDX>(Ns, reinterpret_cast<T*>(tl_p(it.hn)), H, e.tok0, e.nvalid, 0, fg * (TG / TL_NF), (fg + 1) * (TG / TL_NF), tid);
  // ---- u = hn . W1^T + b1 (permuted rows: 16 x u1 | 16 x u2); g = gelu(u1) * u2
  {
    f32x4 acc[2][NTB];
    chain_zero(acc);
    chain_stage<T, 2, NTB, KC1, H>(acc, w1, W1, m1, Ns, LDX, 0, lane);
    TL_SUB(25);
#pragma unroll
    for (int ti = 0; ti < NTB; ++ti) {
      const int p = 16 * ti + r;
      float u1[4], u2[4], gl[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u1[i] = to_f(from_f<T>(acc[0][ti][i] + b1a[i]));
        u2[i] = to_f(from_f<T>(acc[1][ti][i] + b1b[i]));
        gl[i] = gelu_tanh(u1[i]) * u2[i];
      }
      store4f(Gs + p * LDGS + 16 * w + 4 * g, gl);
      store4f(Us + p * LDX + 32 * w + 4 * g, u1);          // (As is dead: every wave has passed the barrier behind the projection)
      store4f(Us + p * LDX + 32 * w + 16 + 4 * g, u2);
    }
  }
  __syncthreads();
  tl_copy_out<T, 256, LDX>(Us, reinterpret_cast<T*>(tl_p(it.u)), I2, e.tok0, e.nvalid, 256 * fg, 0, TG, tid);
  tl_copy_out<T, 128, LDGS>(Gs, reinterpret_cast<T*>(tl_p(it.g)), I, e.tok0, e.nvalid, 128 * fg, 0, TG, tid);
  __syncthreads();
  TL_SUB(26);
  return true;
}

// ------------------------------------------------------------------ phase E: y slice = h + dropout(g . W2^T + b2)
template <typename T, int TG>
TL_PHASE bool tl_phase_e(int l, int clip, int role, unsigned need) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tl_smem[];
  using D = TlDims<T, TG>;
  using RT = Res<T>;
  constexpr int H = D::H, I = D::I, LDG = D::LDG, LDQ = D::LDQ, NTH = D::NTH;
  l = tl_u(l);
  const TlEnv e = tl_env(clip, role, TG);
  const TLayerItem& it = s_it[l];
  T* Gs = reinterpret_cast<T*>(tl_smem);
  RT* Hs = reinterpret_cast<RT*>(Gs + TG * LDG);
  const int tid = e.tid, lane = e.lane, w = e.wave, r = lane & 15, g = lane >> 4, fg = e.fg;
  const T* W2 = reinterpret_cast<const T*>(tl_p(it.W2));
  constexpr int KC2 = ChainChunk<T, 1, I>::KC;
  const int fbh = w & 3, th = (w >> 2) * NTH * 16;
  const int mO = 64 * fg + 16 * fbh;
  Frag<T> w2[KC2][1];
  chain_load_w<T, 1, KC2, I>(w2, W2, mO, 0, lane);
  float bb[4];
  {
    const float* b2 = tl_p(it.b2);
#pragma unroll
    for (int i = 0; i < 4; ++i) bb[i] = b2[mO + 4 * g + i];
  }
  TL_SUB(27);
  if (!tl_wait(need, TL_NF, [&](int j) { return e.tg * TL_NF + j; })) return false;
  TL_SUB(28);
  tl_copy_in<T, TG, I, LDG>(Gs, reinterpret_cast<const T*>(tl_p(it.g)), I, e.tok0, e.nvalid, 0, tid);
  tl_copy_in<RT, TG, 64, LDQ>(Hs, reinterpret_cast<const RT*>(tl_p(it.h)), H, e.tok0, e.nvalid, 64 * fg, tid);
  __syncthreads();
  TL_SUB(29);
  f32x4 acc[1][NTH];
  chain_zero(acc);
  chain_stage<T, 1, NTH, KC2, I>(acc, w2, W2, mO, Gs, LDG, th, lane);
  TL_SUB(30);
  const uint32_t stream = 2u * (uint32_t)l + 1u, thr = tl_u(s_c.thr);
  const uint64_t seed = tl_u(s_c.seed);
  const float inv_keep = tl_u(s_c.inv_keep);
#pragma unroll
  for (int ti = 0; ti < NTH; ++ti) {
    const int p = th + 16 * ti + r;
    const long row = e.tok0 + p;
    RT* hp = Hs + p * LDQ + 16 * fbh + 4 * g;
    float h[4], y[4];
    load4f(hp, h);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = acc[0][ti][i] + bb[i];
      if (thr) {
        const bool keep = rng_keep(seed, stream, (uint32_t)row * (uint32_t)H + (uint32_t)(mO + 4 * g + i), thr);
        v = keep ? v * inv_keep : 0.f;
      }
      y[i] = v + h[i];
    }
    store4f(hp, y);
  }
  __syncthreads();
  tl_copy_out<RT, 64, LDQ>(Hs, reinterpret_cast<RT*>(tl_p(it.y)), H, e.tok0, e.nvalid, 64 * fg, 0, TG, tid);
  __syncthreads();
  TL_SUB(31);
  return true;
}

#define TL_STAMP(k) do { if (a.stamps != nullptr && threadIdx.x == 0) a.stamps[((size_t)blockIdx.x * a.n_layers + l) * TL_STAMP_SLOTS + (k)] = wall_clock64(); } while (0)

template <typename T, int TG, bool DROP>
__global__ __launch_bounds__(TL_THREADS) void tlayer_fwd_kernel(TLayerArgs a) {
  // The kernel-argument segment lives in HOST memory (HIP default): the items are copied into LDS up front (chain.hip: block_run_fwd)
#if defined(__HIP_DEVICE_COMPILE__)
  {
    typedef const __attribute__((address_space(4))) unsigned long long* KernArg64;
    KernArg64 ka = (KernArg64)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int W0 = (int)(offsetof(TLayerArgs, it) / 8), WPI = (int)(sizeof(TLayerItem) / 8);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(s_it);
    for (int i = threadIdx.x; i < a.n_layers * WPI; i += TL_THREADS) dst[i] = ka[W0 + i];
  }
#endif
  Team t = team_form(a.st);            // (its barriers order the copy above before the first read below)
  if (threadIdx.x == 0) {
    s_c.rope_cos = a.rope_cos; s_c.rope_sin = a.rope_sin; s_c.st = a.st; s_c.X0 = a.X0;
    s_c.seed = eff_seed(a.seed, a.seed_dev);
    s_c.B = a.B; s_c.S = a.S; s_c.n_layers = a.n_layers; s_c.prerot = a.prerot; s_c.padL = a.padL; s_c.nW = a.nW;
    s_c.thr = a.drop_thresh24; s_c.inv_keep = a.inv_keep;
    s_c.xcc = t.xcc; s_c.rank = t.rank; s_c.size = t.size; s_c.stamps = a.stamps;
  }
  __syncthreads();
  bool ok = t.ok;
  if (ok) {
    const int n_tg = (a.S + TG - 1) / TG, n_roles = n_tg * TL_NF;
    unsigned epoch = 0;                // phases this workgroup has published
#pragma unroll 1
    for (int clip = t.index; clip < a.B && ok; clip += t.n_teams) {
#pragma unroll 1
      for (int l = 0; l < a.n_layers && ok; ++l) {
        TL_STAMP(0);
#pragma unroll 1
        for (int role = t.rank; role < n_roles && ok; role += t.size) ok = tl_phase_a<T, TG>(l, clip, role, epoch);
        tl_publish(++epoch);
        TL_STAMP(1);
#pragma unroll 1
        for (int role = t.rank; role < n_roles && ok; role += t.size) ok = tl_phase_b<T, TG, DROP>(l, clip, role, epoch);
        tl_publish(++epoch);
        TL_STAMP(2);
#pragma unroll 1
        for (int role = t.rank; role < n_roles && ok; role += t.size) ok = tl_phase_d<T, TG>(l, clip, role, epoch);
        tl_publish(++epoch);
        TL_STAMP(3);
#pragma unroll 1
        for (int role = t.rank; role < n_roles && ok; role += t.size) ok = tl_phase_e<T, TG>(l, clip, role, epoch);
        tl_publish(++epoch);
        TL_STAMP(4);
      }
    }
  }
  team_leave(a.st, t);
}

int tl_grid() {   // one workgroup per compute unit, on the full 8-XCD chip only
  static int cus = [] {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    return p.multiProcessorCount;
  }();
  return cus;
}

template <typename T, bool DROP>
void tlayer_fwd_launch(hipStream_t s, const TLayerArgs& a, int grid) {
  constexpr int TG = TlTile<T>::TG;
  constexpr size_t lds = TlDims<T, TG>::LDS_ALL;
  static_assert(lds + sizeof(TLayerItem) * TLAYER_MAX + 512 <= 160 * 1024, "LDS budget");
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(tlayer_fwd_kernel<T, TG, DROP>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
  (void)once;
  hipLaunchKernelGGL((tlayer_fwd_kernel<T, TG, DROP>), dim3(grid), dim3(TL_THREADS), lds, s, a);
}

}  // namespace

bool tlayer_fwd_device_ok() { return tl_grid() == TEAM_MAX * 32; }

bool tlayer_fwd(hipStream_t s, DType dt, const TLayerArgs& a) {
  const int grid = tl_grid();
  // every workgroup resident at once (one per compute unit), 8 XCDs x 32, 32-bit byte offsets into every tensor
  if (grid != TEAM_MAX * 32 || a.st == nullptr || a.n_layers < 1 || a.n_layers > TLAYER_MAX) return false;
  if ((double)a.B * a.S * 1024 * 4 >= 2147483648.0) return false;
  // bf16 at S <= 256 takes the one-group attention forward (attn.hip: attn_global_fwd); this launch has the two-group tile only
  if (dt != DT_F32 && a.S <= 256) return false;
  const double N = (double)a.B * a.S, es = dt == DT_F32 ? 4 : 2;
  const double wl = 320.0 * 256 + 512.0 * 64 + 256.0 * 256 + 1024.0 * 256 + 256.0 * 512;
  ProfScope ps(s, "tlayer_fwd", a.n_layers * (2.0 * N * wl + 0.5 * 4.0 * a.B * 4 * (double)a.S * a.S * 64),
               a.n_layers * (N * (256.0 * 8 + 1024 + 512 + 320 + 512) + wl) * es);
  const bool drop = a.drop_thresh24 != 0;
  if (dt == DT_F32) { if (drop) tlayer_fwd_launch<float, true>(s, a, grid); else tlayer_fwd_launch<float, false>(s, a, grid); }
  else { if (drop) tlayer_fwd_launch<bf16, true>(s, a, grid); else tlayer_fwd_launch<bf16, false>(s, a, grid); }
  return true;
}

}  // namespace amt
