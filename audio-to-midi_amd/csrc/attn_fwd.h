// Forward attention bodies (global flash-style tile, local stripe), shared by the stand-alone kernels (attn.hip) and the
// XCD-team layer kernel (tlayer.hip).  See attn.hip for the conventions.
#pragma once
#include "common.h"
#include "kernels.h"
#include "team.h"

namespace amt {

constexpr int ALD = 72;   // LDS row stride (elements) of the [64][64] tiles

template <typename T> AMT_DEV Frag<T> make_frag(const float (&y)[8]);
template <> AMT_DEV Frag<bf16> make_frag<bf16>(const float (&y)[8]) {
  Frag<bf16> f;
#pragma unroll
  for (int i = 0; i < 8; ++i) f.v[i] = (bf16)y[i];
  return f;
}
template <> AMT_DEV Frag<float> make_frag<float>(const float (&y)[8]) {
  Frag<float> f;
#pragma unroll
  for (int i = 0; i < 8; ++i) f.v[i] = y[i];
  return f;
}

// one interleaved pair, explicit rounding sequence (every site that rotates q / k must produce the same
// bits: forward and backward stage the same tensors independently)
AMT_DEV void rope_pair(float& x1, float& x2, float c, float s, float scale) {
  const float a = __fmul_rn(__fmaf_rn(x1, c, -__fmul_rn(x2, s)), scale);
  const float b = __fmul_rn(__fmaf_rn(x1, s, __fmul_rn(x2, c)), scale);
  x1 = a; x2 = b;
}
// rotate 4 interleaved pairs starting at pair index pi for position pos, then scale
AMT_DEV void rope8(float (&x)[8], const float* __restrict__ cs, const float* __restrict__ sn, int pos, int pi,
                   float scale) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    rope_pair(x[2 * j], x[2 * j + 1], cs[pos * 32 + pi + j], sn[pos * 32 + pi + j], scale);
  }
}
// transpose of the rotation (gradient w.r.t. the un-rotated input) for 2 pairs (4 features)
AMT_DEV void rope4_bwd(float (&d)[4], const float* __restrict__ cs, const float* __restrict__ sn, int pos, int pi,
                       float scale) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float c = cs[pos * 32 + pi + j], s = sn[pos * 32 + pi + j];
    float d1 = d[2 * j], d2 = d[2 * j + 1];
    d[2 * j] = (d1 * c + d2 * s) * scale;
    d[2 * j + 1] = (-d1 * s + d2 * c) * scale;
  }
}

template <typename T>
AMT_DEV void stage_row_chunk(T* dst, const T* src, bool valid, const float* cs, const float* sn, int pos, int pi,
                             bool do_rope, float scale) {
  float x[8];
  if (valid) {
    load8f(src, x);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.f;
  }
  if (do_rope && valid) rope8(x, cs, sn, pos, pi, scale);
  store8f(dst, x);
}


// K/V tile staging split in two so that the global loads of tile t+1 are in flight while tile t is
// being multiplied: kv_load issues this thread's four 16-byte loads (rows past the clip are read
// from a clamped row), kv_store applies RoPE to K, zeroes invalid rows and writes LDS.
// The RoPE table entries of the tile travel with it (RopeRegs): loaded inside kv_store they cost one more
// dependent global round trip per key tile, after the barrier.
struct RopeRegs { float c[2][4], s[2][4]; };

template <typename T, bool SC1 = false>
AMT_DEV void kv_load(Chunk8<T> (&kc)[2], Chunk8<T> (&vc)[2], RopeRegs& rr, const T* KV, int ldkv, int b, int h, int S,
                     int kt, int tid, const float* __restrict__ cs, const float* __restrict__ sn, bool rot = true) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    int id = tid + 256 * it, row = id >> 3, c = id & 7;
    int key = kt * 64 + row;
    int keyc = key < S ? key : S - 1;
    const size_t ke = (size_t)(b * S + keyc) * ldkv + h * 64 + 8 * c;
    kc[it] = team_chunk_load<SC1>(KV, ke);
    vc[it] = team_chunk_load<SC1>(KV, ke + 256);
    if (rot) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { rr.c[it][j] = cs[keyc * 32 + 4 * c + j]; rr.s[it][j] = sn[keyc * 32 + 4 * c + j]; }
    }
  }
}
// rot = false: k arrives rotated and rounded (AttnArgs::prerot): the tile is copied
template <typename T>
AMT_DEV void kv_store(T* Ks, T* Vs, const Chunk8<T> (&kc)[2], const Chunk8<T> (&vc)[2], const RopeRegs& rr, int S, int kt,
                      int tid, bool rot = true) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    int id = tid + 256 * it, row = id >> 3, c = id & 7;
    int key = kt * 64 + row;
    float kx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) kx[i] = chunk_get(kc[it], i);
    if (key < S) {
      if (rot) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rope_pair(kx[2 * j], kx[2 * j + 1], rr.c[it][j], rr.s[it][j], 1.f);   // = rope8
        store8f(Ks + row * ALD + 8 * c, kx);
      } else {
        chunk_store(Ks + row * ALD + 8 * c, kc[it]);
      }
      chunk_store(Vs + row * ALD + 8 * c, vc[it]);
    } else {
      chunk_store(Ks + row * ALD + 8 * c, chunk_zero<T>());
      chunk_store(Vs + row * ALD + 8 * c, chunk_zero<T>());
    }
  }
}

AMT_DEV uint32_t gdrop_idx(int b, int H, int h, int S, int q, int key) {
  return (uint32_t)(((b * H + h) * S + q)) * (uint32_t)S + (uint32_t)key;
}
// OR over the four lanes that share (lane & 15): the bits move through the float swap helpers untouched
AMT_DEV uint32_t or_over_g(uint32_t v) {
  float a, b;
  xrow16(__builtin_bit_cast(float, v), a, b);
  v = __builtin_bit_cast(uint32_t, a) | __builtin_bit_cast(uint32_t, b);
  xrow32(__builtin_bit_cast(float, v), a, b);
  return __builtin_bit_cast(uint32_t, a) | __builtin_bit_cast(uint32_t, b);
}
// keep-bit word of (clip b, head h, key tile kt, query q): see AttnArgs::keepbits
AMT_DEV size_t keepbits_index(int b, int H, int h, int ntile, int kt, int q) {
  return ((size_t)(b * H + h) * ntile + kt) * (size_t)(64 * ntile) + q;
}

// ------------------------------------------------------------------ global forward
// NG = 2 groups of four waves share the 64 queries of the workgroup and split the key tiles
// round-robin, each with its own K/V/P tiles in LDS, and merge their (max, sum, O) triples once at the end.  With one
// group (= one wave per SIMD at 8 clips: 256 workgroups) the loop was bound by its own dependency chain
// (MFMA -> softmax -> LDS round trip -> MFMA, three barriers per tile): 2.1 us per 64-key tile for ~1600
// cycles of VALU issue.
// DROP: attention dropout compiled in or out (a run-time test per score quad splits the softmax into 16 basic blocks per
// key tile, which the scheduler cannot move work across)
#ifndef AMT_ATTN_FWD1_OCC
#define AMT_ATTN_FWD1_OCC 1
#endif
// The tile body: every thread of the 256 * NG-thread workgroup calls it for query tile `qtile` of (clip b, head h); a.seed is
// the EFFECTIVE seed.  SC1: q, k, v were written by other workgroups of THIS launch (XCD team, team.h): L1-bypassing loads.
template <typename T, int NG, bool DROP, bool SC1 = false>
AMT_DEV void attn_global_fwd_tile(const AttnArgs& a, int b, int h, int qtile, unsigned char* smem_raw) {
  const int tid_all = threadIdx.x, half = tid_all >> 8, tid = tid_all & 255;   // half = key-tile group 0 .. NG-1
  T* Ks = reinterpret_cast<T*>(smem_raw) + half * (3 * 64 * ALD);
  T* Vs = Ks + 64 * ALD;
  T* Ps = Vs + 64 * ALD;
  const int lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const int S = a.S, ntile = (S + 63) / 64;
  const int qrow = qtile * 64 + wave * 16 + r;
  const T* QC = reinterpret_cast<const T*>(a.QC);
  const T* KV = reinterpret_cast<const T*>(a.KV);
  T* Pw = Ps + wave * 16 * ALD;
  const bool rot = a.prerot == 0;          // q / k arrive raw: rotate while staging

  Frag<T> qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.f;
    if (qrow < S) {
      team_load8f<SC1>(team_rsrc(QC), QC, (long)((size_t)(b * S + qrow) * a.ldqc + h * 64 + 32 * ks + 8 * g), x);
      if (rot) rope8(x, a.cos, a.sin, qrow, 16 * ks + 4 * g, 1.f);
      // reference: rope output is cast to the activation dtype, then divided by sqrt(64) (exact)
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = to_f(from_f<T>(x[i])) * 0.125f;
    }
    qf[ks] = make_frag<T>(x);
  }

  float m_run = -1e30f, l_run = 0.f;
  f32x4 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = (S + 63) / 64, niter = (nkt + NG - 1) / NG;
  Chunk8<T> kc[2], vc[2];
  RopeRegs rr;
  if (half < nkt) kv_load<T, SC1>(kc, vc, rr, KV, a.ldkv, b, h, S, half, tid, a.cos, a.sin, rot);
  for (int it = 0; it < niter; ++it) {
    const int kt = NG * it + half;
    const bool active = kt < nkt;            // uniform over the group; barriers are taken by all 512 threads
    __syncthreads();
    if (active) kv_store<T>(Ks, Vs, kc, vc, rr, S, kt, tid, rot);
    __syncthreads();
#ifndef AMT_ABL_NOLOAD
    if (kt + NG < nkt) kv_load<T, SC1>(kc, vc, rr, KV, a.ldkv, b, h, S, kt + NG, tid, a.cos, a.sin, rot);
#endif

    if (active) {
      f32x4 sacc[4];
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        sacc[st] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) mma(sacc[st], frag_row(Ks + (16 * st + r) * ALD + 32 * ks + 8 * g), qf[ks]);
      }
      // sacc[st][e] = score[key = kt*64 + 16 st + 4 g + e][query = this lane's]
      float mx = -1e30f;
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int key = kt * 64 + 16 * st + 4 * g + e;
          if (key >= S) sacc[st][e] = -1e30f;
          mx = fmaxf(mx, sacc[st][e]);
        }
      mx = max_xor32(max_xor16(mx));
      float m_new = fmaxf(m_run, mx);
      float alpha = __expf(m_run - m_new);
      float psum = 0.f;
      uint32_t kb[2] = {0u, 0u};       // keep bits of this lane's 16 scores: word st >> 1, bit 16 (st & 1) + e (+ 4 g below)
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        float pv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float p = __expf(sacc[st][e] - m_new);
          psum += p;
          if (DROP) {
            int key = kt * 64 + 16 * st + 4 * g + e;
            bool keep = rng_keep(a.seed, a.stream, gdrop_idx(b, a.H, h, S, qrow, key), a.drop_thresh24);
            p = keep ? p * a.inv_keep : 0.f;
            kb[st >> 1] |= (keep ? 1u : 0u) << (16 * (st & 1) + e);
          }
          pv[e] = p;
        }
        store4f(Pw + r * ALD + 16 * st + 4 * g, pv);
      }
      if (DROP) {
        if (a.keepbits != nullptr) {   // uniform
          const uint32_t lo = or_over_g(kb[0] << (4 * g)), hi = or_over_g(kb[1] << (4 * g));
          if (g == 0) a.keepbits[keepbits_index(b, a.H, h, ntile, kt, qrow)] = ((unsigned long long)hi << 32) | lo;
        }
      }
      psum = sum_xor32(sum_xor16(psum));
      l_run = l_run * alpha + psum;
      m_run = m_new;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oacc[dt] *= alpha;
    }
    __syncthreads();
    if (active) {
      // O^T[d][q] += V^T[d][key] P^T[key][q]
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        Frag<T> pf = frag_row(Pw + r * ALD + 32 * ks + 8 * g);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) mma(oacc[dt], frag_col16(Vs + (32 * ks + 8 * g) * ALD + 16 * dt, ALD, r), pf);
      }
    }
  }
  // merge: groups 1 .. NG-1 hand their (m, l, O) to group 0 through LDS
  __syncthreads();
  float* mg = reinterpret_cast<float*>(smem_raw) + (size_t)(wave * 64 + lane) * 18;
  constexpr int MGS = 256 * 18;     // floats per group
  if (half > 0) {
    float* w = mg + (half - 1) * MGS;
    w[0] = m_run; w[1] = l_run;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int e = 0; e < 4; ++e) w[2 + 4 * dt + e] = oacc[dt][e];
  }
  __syncthreads();
  if (half == 0 && qrow < S) {
    float m = m_run;
#pragma unroll
    for (int k = 0; k < NG - 1; ++k) m = fmaxf(m, mg[k * MGS]);
    const float a0 = __expf(m_run - m);
    float l = l_run * a0;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] *= a0;
#pragma unroll
    for (int k = 0; k < NG - 1; ++k) {
      const float* w = mg + k * MGS;
      const float ak = __expf(w[0] - m);
      l += w[1] * ak;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int e = 0; e < 4; ++e) oacc[dt][e] += w[2 + 4 * dt + e] * ak;
    }
    const float inv = 1.f / l;
    T* op = reinterpret_cast<T*>(a.O) + (size_t)(b * S + qrow) * a.ldo + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = oacc[dt][e] * inv;
      store4f(op + 16 * dt + 4 * g, o);
    }
    if (g == 0 && a.LSE != nullptr) a.LSE[(size_t)(b * a.H + h) * S + qrow] = m + logf(l);
  }
}
template <typename T, int NG, bool DROP>
__global__ __launch_bounds__(256 * NG, NG == 1 ? AMT_ATTN_FWD1_OCC : 1) void attn_global_fwd_kernel(AttnArgs a) {
  a.seed = eff_seed(a.seed, a.seed_dev);   // kernel argument + the context's device seed word (common.h)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // XCD-aware order: the query tiles of one (clip, head) share K/V, so they get the same id % 8
  const int ntile = (a.S + 63) / 64;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int bh = (jj / ntile) * 8 + xcd, qtile = jj % ntile;
  if (bh >= a.B * a.H) return;
  attn_global_fwd_tile<T, NG, DROP, false>(a, bh / a.H, bh % a.H, qtile, smem_raw);
}

// ------------------------------------------------------------------ local (windowed) attention
// One wave per (stripe of 8 output positions, head, clip).  Output position j = 8 s + i is covered
// by window A = s - 1 (its rows 8..15) and window B = s (its rows 0..7); both windows are
// evaluated by the same wave and averaged, which reproduces model.py:451-469 without atomics.
// A 16 x 16 x 64 window is exactly MFMA-shaped: the 24 tokens the stripe touches are staged once in
// LDS (q / k rotated per window, as the reference rotates inside every window), scores and dP come out
// of MFMAs with the query on the lane axis (softmax = a reduction over 4 registers and the 4 lane
// groups), and the second GEMMs take both windows in ONE 32-deep step: their operand tiles hold
// window A in k = 0..15 and window B in k = 16..31 (rows 0..7 / 8..15 where the two results must stay
// apart because their rotations differ).  (The first version of these kernels did the same sums as
// fp32 VALU dot products over LDS: 13.9 / 37.8 us per launch at 8 clips.)
constexpr int OLD = 40;   // row stride (elements) of the 16 x 32 operand tiles

struct LocalGeom { int padL, nW, Lp; };
static LocalGeom local_geom(int S, int W) {
  int stride = W / 2;
  int req = stride - (S - W) % stride;
  int pl = 0, pr = 0;
  if (req != stride) { pl = req / 2; pr = req / 2 + (req % 2); }
  LocalGeom gm;
  gm.padL = pl;
  gm.Lp = S + pl + pr;
  gm.nW = (gm.Lp - W) / stride + 1;
  return gm;
}

AMT_DEV uint32_t ldrop_idx(int b, int H, int h, int nW, int win, int iw, int j) {
  return (uint32_t)((((b * H + h) * nW + win) * 16 + iw) * 16 + j);
}

// load one 8-feature chunk of the token at padded position p.  The load is unconditional (clamped
// token) so that a thread's loads can all be in flight together; `mask8` zeroes pad tokens later.
template <typename T, bool SC1 = false>
AMT_DEV bool load_tok_chunk(const T* base, int ld, int b, int S, int p, int padL, int off, float (&x)[8]) {
  int t = p - padL;
  bool ok = (t >= 0) && (t < S);
  int tc = t < 0 ? 0 : (t >= S ? S - 1 : t);
  team_load8f<SC1>(team_rsrc(base), base, (long)((size_t)(b * S + tc) * ld + off), x);
  return ok;
}
AMT_DEV void mask8(float (&x)[8], bool ok) {
  if (!ok) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.f;
  }
}

// rotate chunk c of a token for window-relative position pos, round to the activation dtype (as the
// reference does), scale (a power of two: exact), store as 8 elements of an LDS tile row
template <typename T>
AMT_DEV void store_roped(T* row, int c, const float (&x)[8], const float* cs, const float* sn, int pos, float scale) {
  float y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) y[i] = x[i];
  rope8(y, cs, sn, pos, 4 * c, 1.f);
#pragma unroll
  for (int i = 0; i < 8; ++i) y[i] = to_f(from_f<T>(y[i])) * scale;
  store8f(row + 8 * c, y);
}

// LDS of one wave's stripe (elements of T)
constexpr int LOCAL_FWD_LDS = (16 + 16 + 16 + 24) * ALD + 8 * OLD;
// barrier between the stripe's phases: the stripe is ONE wave's work.  BlockSync: the wave is the whole workgroup (the
// stand-alone kernel); WaveSync: other waves of the workgroup run other stripes (team kernel, tlayer.hip) - LDS operations of
// one wave execute in order, so its own writes are visible to its later reads once the compiler keeps the order
struct LocalBlockSync { static AMT_DEV void sync() { __syncthreads(); } };
struct LocalWaveSync {
  static AMT_DEV void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
};
// One wave: stripe s of (head h, clip b); a.seed is the EFFECTIVE seed; lds: LOCAL_FWD_LDS elements, 16-byte aligned.
template <typename T, bool SC1, typename SYNC>
AMT_DEV void attn_local_fwd_wave(const AttnArgs& a, int padL, int nW, int s, int h, int b, T* lds, int lane) {
  // qs: the stripe's 8 queries twice: row i rotated as row i of window B, row 8 + i as row 8 + i of window A
  typedef T (*Rows)[ALD];
  typedef T (*RowsO)[OLD];
  Rows qs = reinterpret_cast<Rows>(lds), kA = reinterpret_cast<Rows>(lds + 16 * ALD), kB = reinterpret_cast<Rows>(lds + 32 * ALD),
       vs = reinterpret_cast<Rows>(lds + 48 * ALD);
  RowsO Pt = reinterpret_cast<RowsO>(lds + 72 * ALD);
  const int r = lane & 15, g = lane >> 4;
  const int S = a.S;
  const bool validA = (s >= 1) && (s - 1 < nW), validB = (s < nW);
  const T* QC = reinterpret_cast<const T*>(a.QC);
  const T* KV = reinterpret_cast<const T*>(a.KV);
  {
    // issue all seven loads of this lane, then rotate / store
    float qx[8], kx[3][8], vx[3][8];
    bool okq, okk[3];
    const int qi = lane >> 3, qc = lane & 7;
    okq = load_tok_chunk<T, SC1>(QC, a.ldqc, b, S, 8 * s + qi, padL, h * 64 + 8 * qc, qx);
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      int id = lane + 64 * it, jp = id >> 3, c = id & 7;
      int p = 8 * s - 8 + jp;
      okk[it] = load_tok_chunk<T, SC1>(KV, a.ldkv, b, S, p, padL, h * 64 + 8 * c, kx[it]);
      load_tok_chunk<T, SC1>(KV, a.ldkv, b, S, p, padL, 256 + h * 64 + 8 * c, vx[it]);
    }
    mask8(qx, okq);
    store_roped<T>(qs[8 + qi], qc, qx, a.cos, a.sin, 8 + qi, 0.125f);
    store_roped<T>(qs[qi], qc, qx, a.cos, a.sin, qi, 0.125f);
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      int id = lane + 64 * it, jp = id >> 3, c = id & 7;
      mask8(kx[it], okk[it]);
      mask8(vx[it], okk[it]);
      if (jp < 16) store_roped<T>(kA[jp], c, kx[it], a.cos, a.sin, jp, 1.f);
      if (jp >= 8) store_roped<T>(kB[jp - 8], c, kx[it], a.cos, a.sin, jp - 8, 1.f);
      store8f(&vs[jp][8 * c], vx[it]);
    }
  }
  SYNC::sync();
  // scores, transposed: lane (r, g) holds keys 4 g + e of query row r; rows >= 8 belong to window A
  f32x4 sA = f32x4{0.f, 0.f, 0.f, 0.f}, sB = sA;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const Frag<T> qf = frag_row(&qs[r][32 * ks + 8 * g]);
    mma(sA, frag_row(&kA[r][32 * ks + 8 * g]), qf);
    mma(sB, frag_row(&kB[r][32 * ks + 8 * g]), qf);
  }
  const bool isA = r >= 8;
  float sc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) sc[e] = isA ? sA[e] : sB[e];
  float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
  mx = max_xor32(max_xor16(mx));
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { sc[e] = __expf(sc[e] - mx); sum += sc[e]; }
  sum = sum_xor32(sum_xor16(sum));
  const float fa = validA ? 1.f : 0.f, fb = validB ? 1.f : 0.f;
  const float wgt = (isA ? fa : fb) / (fa + fb) / sum;      // window average folded into P (0, 1/2 or 1: exact)
  const int win = isA ? s - 1 : s;
  float pv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float p = sc[e] * wgt;
    if (a.drop_thresh24) {
      bool keep = rng_keep(a.seed, a.stream, ldrop_idx(b, a.H, h, nW, win, r, 4 * g + e), a.drop_thresh24);
      p = keep ? p * a.inv_keep : 0.f;
    }
    pv[e] = p;
  }
  // Pt[i][0..15] = P_A[8 + i][keys], Pt[i][16..31] = P_B[i][keys]
  store4f(&Pt[r & 7][(isA ? 0 : 16) + 4 * g], pv);
  SYNC::sync();
  // O^T[d][i] = sum_k V^T[d][k] Pt[i][k]; k = 8 g + j is token row {0, 8, 8, 16}[g] + j of the stripe's 24
  const Frag<T> pf = frag_row(&Pt[r & 7][8 * g]);
  const int vrow = g == 0 ? 0 : (g == 3 ? 16 : 8);
  T* O = reinterpret_cast<T*>(a.O);
  const int jo = 8 * s + r;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
    mma(o, frag_col16(&vs[vrow][16 * dt], ALD, r), pf);
    if (r < 8 && jo < S) {
      float ov[4] = {o[0], o[1], o[2], o[3]};
      store4f(O + (size_t)(b * S + jo) * a.ldo + h * 64 + 16 * dt + 4 * g, ov);
    }
  }
}
template <typename T>
__global__ __launch_bounds__(64) void attn_local_fwd_kernel(AttnArgs a, int padL, int nW) {
  a.seed = eff_seed(a.seed, a.seed_dev);   // kernel argument + the context's device seed word (common.h)
  __shared__ __attribute__((aligned(16))) T lds[LOCAL_FWD_LDS];
  attn_local_fwd_wave<T, false, LocalBlockSync>(a, padL, nW, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, lds, (int)threadIdx.x);
}

}  // namespace amt
