// Attention kernels.
//   attn_global_*  : SelfAttention core (model.py:241-257, 353-367): RoPE on q,k (rope.py:25-53),
//                    q / sqrt(64), softmax in fp32, optional weight dropout, P.V — flash-style,
//                    QK^T and PV on the matrix cores (16x16 MFMA tiles), online softmax in registers.
//   attn_local_*   : LocalSelfAttention (model.py:409-471): 16-token windows, stride 8,
//                    window-relative RoPE, outputs averaged over the covering windows with the
//                    reference's padded-coordinate scatter (out[j] <- padded position j).
//
// Inputs are the raw projections: q = QC[:, h*64 ..], k = KV[:, h*64 ..], v = KV[:, 256 + h*64 ..].
// RoPE is applied while staging (and its transpose while writing dq/dk), so no separate RoPE pass
// ever touches HBM.
//
// MFMA orientation ("transposed" form): every product is computed as D^T so that a lane owns ONE
// query (or one key in the dK/dV kernel) = lane & 15, and its 4 accumulator registers are 4
// consecutive head-dim features: row statistics are two xor-shuffles, outputs are 8-byte vectors.
#include "common.h"
#include "kernels.h"
#include "prof.h"
#include "attn_fwd.h"

namespace amt {


// ------------------------------------------------------------------ global forward, fp8 (e4m3) QK^T and PV
// BASELINE configs[4] ("fp8 MFMA attention + bf16 everything else").  q (rotated, / sqrt(64)), k (rotated) and
// v are scaled per tensor and converted to OCP e4m3 while they are staged; S = QK^T and O = PV run on
// v_mfma_f32_16x16x32_fp8_fp8 with fp32 accumulation; the softmax stays fp32; P is scaled by 384 (dropout
// can raise a probability to 1.11) before its conversion.  Delayed scaling: the scales come from the
// per-tensor |max| observed in the PREVIOUS call of this layer (amax_prev; 0 -> a default of 16), this call
// records its own into amax_cur (atomic max).  V is written to LDS transposed so that both PV operands
// are plain 8-byte row reads.  The backward pass is the bf16 one (it recomputes S from the saved bf16
// q / k and the log-sum-exp of this kernel).
constexpr int F8LD = 72;                     // bytes per LDS row of the 64-wide fp8 tiles
constexpr float F8_MAX = 448.f, F8_P_SCALE = 384.f;

AMT_DEV long pack_fp8x8(const float (&x)[8], float scale) {
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(x[0] * scale, x[1] * scale, 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(x[2] * scale, x[3] * scale, lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(x[4] * scale, x[5] * scale, 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(x[6] * scale, x[7] * scale, hi, true);
  return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
}
AMT_DEV float fp8_scale_from_amax(float amax) { return (F8_MAX * 0.9f) / (amax > 0.f ? amax : 16.f); }

// key tile in flight: K as two 8-element chunks per thread (+ their RoPE entries), V as 2 x 4 keys x one
// feature pair per thread (the transposed LDS tile takes four keys of one feature as one dword)
struct Fp8Stage { Chunk8<bf16> kc[2]; RopeRegs rr; uint32_t v[2][4]; };

AMT_DEV void fp8_kv_load(Fp8Stage& st, const bf16* KV, int ldkv, int b, int h, int S, int kt, int tid,
                         const float* __restrict__ cs, const float* __restrict__ sn) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int id = tid + 256 * it, row = id >> 3, c = id & 7;
    const int key = kt * 64 + row, keyc = key < S ? key : S - 1;
    st.kc[it] = chunk_load(KV + (size_t)(b * S + keyc) * ldkv + h * 64 + 8 * c);
#pragma unroll
    for (int j = 0; j < 4; ++j) { st.rr.c[it][j] = cs[keyc * 32 + 4 * c + j]; st.rr.s[it][j] = sn[keyc * 32 + 4 * c + j]; }
    const int dp = tid & 31, kg = (tid >> 5) + 8 * it;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int vk = kt * 64 + 4 * kg + j, vkc = vk < S ? vk : S - 1;
      st.v[it][j] = *reinterpret_cast<const uint32_t*>(KV + (size_t)(b * S + vkc) * ldkv + 256 + h * 64 + 2 * dp);
    }
  }
}

AMT_DEV void fp8_kv_store(unsigned char* Ks8, unsigned char* Vt8, const Fp8Stage& st, int S, int kt, int tid, float sk,
                          float sv, float& mk, float& mv) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int id = tid + 256 * it, row = id >> 3, c = id & 7;
    const int key = kt * 64 + row;
    float kx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) kx[i] = chunk_get(st.kc[it], i);
#pragma unroll
    for (int j = 0; j < 4; ++j) rope_pair(kx[2 * j], kx[2 * j + 1], st.rr.c[it][j], st.rr.s[it][j], 1.f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      kx[i] = key < S ? to_f(from_f<bf16>(kx[i])) : 0.f;
      mk = fmaxf(mk, fabsf(kx[i]));
    }
    *reinterpret_cast<long*>(Ks8 + row * F8LD + 8 * c) = pack_fp8x8(kx, sk);
    const int dp = tid & 31, kg = (tid >> 5) + 8 * it;
    float lo[4], hi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = kt * 64 + 4 * kg + j < S;
      lo[j] = ok ? __uint_as_float(st.v[it][j] << 16) : 0.f;
      hi[j] = ok ? __uint_as_float(st.v[it][j] & 0xffff0000u) : 0.f;
      mv = fmaxf(mv, fmaxf(fabsf(lo[j]), fabsf(hi[j])));
    }
    int p0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0] * sv, lo[1] * sv, 0, false);
    p0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2] * sv, lo[3] * sv, p0, true);
    int p1 = __builtin_amdgcn_cvt_pk_fp8_f32(hi[0] * sv, hi[1] * sv, 0, false);
    p1 = __builtin_amdgcn_cvt_pk_fp8_f32(hi[2] * sv, hi[3] * sv, p1, true);
    *reinterpret_cast<int*>(Vt8 + (2 * dp) * F8LD + 4 * kg) = p0;
    *reinterpret_cast<int*>(Vt8 + (2 * dp + 1) * F8LD + 4 * kg) = p1;
  }
}

__global__ __launch_bounds__(256) void attn_global_fwd_fp8_kernel(AttnArgs a, const float* __restrict__ amax_prev,
                                                                  float* __restrict__ amax_cur) {
  a.seed = eff_seed(a.seed, a.seed_dev);   // kernel argument + the context's device seed word (common.h)
  typedef bf16 T;
  __shared__ __attribute__((aligned(16))) unsigned char Ks8[64 * F8LD];   // [key][d]
  __shared__ __attribute__((aligned(16))) unsigned char Vt8[64 * F8LD];   // [d][key]
  __shared__ __attribute__((aligned(16))) unsigned char Ps8[4 * 16 * F8LD];
  __shared__ float wmax[4][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const int S = a.S, ntile = (S + 63) / 64;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int bh = (jj / ntile) * 8 + xcd, qtile = jj % ntile;
  if (bh >= a.B * a.H) return;
  const int h = bh % a.H, b = bh / a.H;
  const int qrow = qtile * 64 + wave * 16 + r;
  const T* QC = reinterpret_cast<const T*>(a.QC);
  const T* KV = reinterpret_cast<const T*>(a.KV);
  unsigned char* Pw = Ps8 + wave * 16 * F8LD;
  Fp8Stage stg;
  fp8_kv_load(stg, KV, a.ldkv, b, h, S, 0, tid, a.cos, a.sin);
  const float sq = fp8_scale_from_amax(amax_prev[0]), sk = fp8_scale_from_amax(amax_prev[1]),
              sv = fp8_scale_from_amax(amax_prev[2]);
  const float s_deq = 1.f / (sq * sk), o_deq = 1.f / (F8_P_SCALE * sv);
  float mq = 0.f, mk = 0.f, mv = 0.f;

  long qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.f;
    if (qrow < S) {
      load8f(QC + (size_t)(b * S + qrow) * a.ldqc + h * 64 + 32 * ks + 8 * g, x);
      rope8(x, a.cos, a.sin, qrow, 16 * ks + 4 * g, 1.f);
#pragma unroll
      for (int i = 0; i < 8; ++i) { x[i] = to_f(from_f<T>(x[i])) * 0.125f; mq = fmaxf(mq, fabsf(x[i])); }
    }
    qf[ks] = pack_fp8x8(x, sq);
  }

  float m_run = -1e30f, l_run = 0.f;
  f32x4 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = (S + 63) / 64;
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    fp8_kv_store(Ks8, Vt8, stg, S, kt, tid, sk, sv, mk, mv);
    __syncthreads();
    if (kt + 1 < nkt) fp8_kv_load(stg, KV, a.ldkv, b, h, S, kt + 1, tid, a.cos, a.sin);

    f32x4 sacc[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      sacc[st] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const long kf = *reinterpret_cast<const long*>(Ks8 + (16 * st + r) * F8LD + 32 * ks + 8 * g);
        sacc[st] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(kf, qf[ks], sacc[st], 0, 0, 0);
      }
    }
    float mx = -1e30f;
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int key = kt * 64 + 16 * st + 4 * g + e;
        sacc[st][e] = key < S ? sacc[st][e] * s_deq : -1e30f;
        mx = fmaxf(mx, sacc[st][e]);
      }
    mx = max_xor32(max_xor16(mx));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      float pv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float p = __expf(sacc[st][e] - m_new);
        psum += p;
        if (a.drop_thresh24) {
          const int key = kt * 64 + 16 * st + 4 * g + e;
          const bool keep = rng_keep(a.seed, a.stream, gdrop_idx(b, a.H, h, S, qrow, key), a.drop_thresh24);
          p = keep ? p * a.inv_keep : 0.f;
        }
        pv[e] = p * F8_P_SCALE;
      }
      int pk = __builtin_amdgcn_cvt_pk_fp8_f32(pv[0], pv[1], 0, false);
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(pv[2], pv[3], pk, true);
      *reinterpret_cast<int*>(Pw + r * F8LD + 16 * st + 4 * g) = pk;
    }
    psum = sum_xor32(sum_xor16(psum));
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] *= alpha;
    __syncthreads();
    // O^T[d][q] += V^T[d][key] P^T[key][q]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const long pf = *reinterpret_cast<const long*>(Pw + r * F8LD + 32 * ks + 8 * g);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const long vf = *reinterpret_cast<const long*>(Vt8 + (16 * dt + r) * F8LD + 32 * ks + 8 * g);
        oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(vf, pf, oacc[dt], 0, 0, 0);
      }
    }
  }
  if (qrow < S) {
    const float inv = o_deq / l_run;
    T* op = reinterpret_cast<T*>(a.O) + (size_t)(b * S + qrow) * a.ldo + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = oacc[dt][e] * inv;
      store4f(op + 16 * dt + 4 * g, o);
    }
    if (g == 0 && a.LSE != nullptr) a.LSE[(size_t)(b * a.H + h) * S + qrow] = m_run + logf(l_run);
  }
  // per-tensor |max| of this call: one atomic per workgroup and tensor, and only if it raises the value
  // (non-negative floats order like their bit patterns)
  mq = group_max<64>(mq); mk = group_max<64>(mk); mv = group_max<64>(mv);
  if (lane == 0) { wmax[wave][0] = mq; wmax[wave][1] = mk; wmax[wave][2] = mv; }
  __syncthreads();
  if (tid < 3) {
    const float m = fmaxf(fmaxf(wmax[0][tid], wmax[1][tid]), fmaxf(wmax[2][tid], wmax[3][tid]));
    if (m > __builtin_nontemporal_load(amax_cur + tid))
      atomicMax(reinterpret_cast<unsigned int*>(amax_cur) + tid, __float_as_uint(m));
  }
}

__global__ void attn_fp8_roll_kernel(float* amax, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * 4) {
    const int l = i >> 2, j = i & 3;
    const float cur = amax[8 * l + 4 + j];
    if (cur > 0.f) amax[8 * l + j] = cur;     // a layer that did not run keeps its last scale
    amax[8 * l + 4 + j] = 0.f;
  }
}
void attn_fp8_roll(hipStream_t s, float* amax, int n_layers) {
  hipLaunchKernelGGL(attn_fp8_roll_kernel, dim3((n_layers * 4 + 63) / 64), dim3(64), 0, s, amax, n_layers);
}

void attn_global_fwd_fp8(hipStream_t s, const AttnArgs& a, const float* amax_prev, float* amax_cur) {
  dim3 grid(((a.B * a.H + 7) / 8 * 8) * ((a.S + 63) / 64));
  ProfScope ps(s, "attn_global_fwd_fp8", 4.0 * a.B * a.H * (double)a.S * a.S * 64, 4.0 * a.B * a.S * a.H * 64 * 2);
  hipLaunchKernelGGL(attn_global_fwd_fp8_kernel, grid, dim3(256), 0, s, a, amax_prev, amax_cur);
}

// ------------------------------------------------------------------ global backward
// One launch, two independent roles: workgroups [0, ntile) of a (clip, head) compute dQ (and delta) for a 64-query
// tile, workgroups [ntile, 2 ntile) dK / dV for a 64-key tile; S and dP are recomputed in both.  NG groups of four
// waves share the workgroup's tile and split the inner tile loop (key tiles in the dQ role, query
// tiles in the dK/dV role) round-robin, each group with its own LDS tiles; their accumulators are summed through
// LDS at the end (no rescaling: both roles use the forward's log-sum-exp).  Launched with NG = 1 (see attn_global_bwd).
template <typename T> struct GBwdLds {
  static constexpr size_t GROUP_BYTES = (size_t)(64 * 2 + 128) * ALD * sizeof(T) + 256 * sizeof(float);   // + lse, delta, keep words
};

// BITS: the dropout keep decisions come from the forward's bit mask (AttnArgs::keepbits) instead of the hash: ~3 instead of
// ~14 vector instructions per score in a loop that is bound by vector issue (~720 per 64 x 64 tile, profiles/r01_notes.md).
template <typename T, int NG, bool DROP, bool BITS>
AMT_DEV void attn_global_bwd_dq_body(const AttnBwdArgs& ba, int tile, int b, int h, unsigned char* smem_all) {
  const AttnArgs& a = ba.f;
  const int tid_all = threadIdx.x, half = NG > 1 ? (tid_all >> 8) : 0, tid = tid_all & 255;
  unsigned char* smem_raw = smem_all + half * GBwdLds<T>::GROUP_BYTES;
  T* Ks = reinterpret_cast<T*>(smem_raw);
  T* Vs = Ks + 64 * ALD;
  T* Ps = Vs + 64 * ALD;
  const int lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const int S = a.S;
  const int qrow = tile * 64 + wave * 16 + r;
  const T* QC = reinterpret_cast<const T*>(a.QC);
  const T* KV = reinterpret_cast<const T*>(a.KV);
  const T* O = reinterpret_cast<const T*>(a.O);
  const T* dO = reinterpret_cast<const T*>(ba.dO);
  T* Pw = Ps + wave * 16 * ALD;
  const bool rot = a.prerot == 0;          // q / k arrive raw: rotate while staging

  Frag<T> qf[2], dof[2];
  float dl = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    float x[8], d[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = 0.f; d[i] = 0.f; o[i] = 0.f; }
    if (qrow < S) {
      int off = h * 64 + 32 * ks + 8 * g;
      load8f(QC + (size_t)(b * S + qrow) * a.ldqc + off, x);
      if (rot) rope8(x, a.cos, a.sin, qrow, 16 * ks + 4 * g, 1.f);
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = to_f(from_f<T>(x[i])) * 0.125f;
      load8f(dO + (size_t)(b * S + qrow) * ba.lddo + off, d);
      load8f(O + (size_t)(b * S + qrow) * a.ldo + off, o);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) dl += d[i] * o[i];
    qf[ks] = make_frag<T>(x);
    dof[ks] = make_frag<T>(d);
  }
  dl = sum_xor32(sum_xor16(dl));
  float lse = 0.f;
  if (qrow < S) {
    lse = a.LSE[(size_t)(b * a.H + h) * S + qrow];
    if (g == 0 && half == 0) ba.delta[(size_t)(b * a.H + h) * S + qrow] = dl;
  }

  f32x4 dqacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dqacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = (S + 63) / 64, niter = (nkt + NG - 1) / NG;
  Chunk8<T> kc[2], vc[2];
  RopeRegs rr;
  if (half < nkt) kv_load<T>(kc, vc, rr, KV, a.ldkv, b, h, S, half, tid, a.cos, a.sin, rot);
  for (int it = 0; it < niter; ++it) {
    const int kt = NG * it + half;
    const bool active = kt < nkt;          // uniform over the group; every thread of the workgroup takes the barriers
    __syncthreads();
    if (active) kv_store<T>(Ks, Vs, kc, vc, rr, S, kt, tid, rot);
    __syncthreads();
#ifndef AMT_ABL_NOLOAD
    if (kt + NG < nkt) kv_load<T>(kc, vc, rr, KV, a.ldkv, b, h, S, kt + NG, tid, a.cos, a.sin, rot);
#endif
    uint32_t kw[2] = {0u, 0u};             // this query's keep bits for the tile, shifted so that bit 16 (st & 1) + e is key (st, e)
    if (DROP && BITS && active) {
      const unsigned long long w = a.keepbits[keepbits_index(b, a.H, h, nkt, kt, qrow)];
      kw[0] = (uint32_t)w >> (4 * g);
      kw[1] = (uint32_t)(w >> 32) >> (4 * g);
    }
    if (active) {
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f}, dpacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          mma(sacc, frag_row(Ks + (16 * st + r) * ALD + 32 * ks + 8 * g), qf[ks]);
          mma(dpacc, frag_row(Vs + (16 * st + r) * ALD + 32 * ks + 8 * g), dof[ks]);
        }
        float ds[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int key = kt * 64 + 16 * st + 4 * g + e;
          float p = (key < S) ? __expf(sacc[e] - lse) : 0.f;
          float dp = dpacc[e];
          if (DROP) {
            const bool keep = BITS ? ((kw[st >> 1] >> (16 * (st & 1) + e)) & 1u) != 0u
                                   : rng_keep(a.seed, a.stream, gdrop_idx(b, a.H, h, S, qrow, key), a.drop_thresh24);
            dp = keep ? dp * a.inv_keep : 0.f;
          }
          ds[e] = p * (dp - dl);
        }
        store4f(Pw + r * ALD + 16 * st + 4 * g, ds);
      }
    }
    __syncthreads();
    if (active) {
      // dQs^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        Frag<T> dsf = frag_row(Pw + r * ALD + 32 * ks + 8 * g);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) mma(dqacc[dt], frag_col16(Ks + (32 * ks + 8 * g) * ALD + 16 * dt, ALD, r), dsf);
      }
    }
  }
  if constexpr (NG > 1) {
    __syncthreads();
    float* mg = reinterpret_cast<float*>(smem_all) + (size_t)(wave * 64 + lane) * 16;
    if (half == 1) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int e = 0; e < 4; ++e) mg[4 * dt + e] = dqacc[dt][e];
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int e = 0; e < 4; ++e) dqacc[dt][e] += mg[4 * dt + e];
    }
  }
  if (half == 0 && qrow < S) {
    T* dq = reinterpret_cast<T*>(ba.dQC) + (size_t)(b * S + qrow) * ba.lddqc + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      float d[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = dqacc[dt][e];
      rope4_bwd(d, a.cos, a.sin, qrow, 8 * dt + 2 * g, 0.125f);
      store4f(dq + 16 * dt + 4 * g, d);
    }
  }
}

template <typename T, int NG, bool DROP, bool BITS>
AMT_DEV void attn_global_bwd_dkv_body(const AttnBwdArgs& ba, int tile, int b, int h, unsigned char* smem_all) {
  const AttnArgs& a = ba.f;
  const int tid_all = threadIdx.x, half = NG > 1 ? (tid_all >> 8) : 0, tid = tid_all & 255;
  unsigned char* smem_raw = smem_all + half * GBwdLds<T>::GROUP_BYTES;
  T* Qs = reinterpret_cast<T*>(smem_raw);
  T* dOs = Qs + 64 * ALD;
  T* PT = dOs + 64 * ALD;          // [4 waves][16][ALD]
  T* dST = PT + 4 * 16 * ALD;      // [4 waves][16][ALD]
  float* lse_s = reinterpret_cast<float*>(dST + 4 * 16 * ALD);
  float* delta_s = lse_s + 64;
  uint32_t* kb_s = reinterpret_cast<uint32_t*>(delta_s + 64);   // [2][64]: low / high halves of the 64 queries' keep words
  const int lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const int S = a.S;
  const int key = tile * 64 + wave * 16 + r;
  const T* QC = reinterpret_cast<const T*>(a.QC);
  const T* KV = reinterpret_cast<const T*>(a.KV);
  const T* dO = reinterpret_cast<const T*>(ba.dO);
  const T* O = reinterpret_cast<const T*>(a.O);
  T* PTw = PT + wave * 16 * ALD;
  T* dSTw = dST + wave * 16 * ALD;
  const bool rot = a.prerot == 0;          // q / k arrive raw: rotate while staging

  Frag<T> kf[2], vf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    float x[8], v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = 0.f; v[i] = 0.f; }
    if (key < S) {
      const T* kp = KV + (size_t)(b * S + key) * a.ldkv + h * 64 + 32 * ks + 8 * g;
      load8f(kp, x);
      if (rot) rope8(x, a.cos, a.sin, key, 16 * ks + 4 * g, 1.f);
      load8f(kp + 256, v);
    }
    kf[ks] = make_frag<T>(x);
    vf[ks] = make_frag<T>(v);
  }
  f32x4 dkacc[4], dvacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { dkacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nqt = (S + 63) / 64, niter = (nqt + NG - 1) / NG;
  // The query tile of iteration it + 1 (q, dO, O rows, their RoPE entries, lse, keep words) is requested while tile it is
  // multiplied and goes to LDS at the top of the next iteration (as the dQ role does with its K / V tiles): requested at
  // the top of its own iteration, every one of the 8 tiles exposed a full memory round trip behind a barrier.
  Chunk8<T> qc8[2], dc8[2], oc8[2];
  RopeRegs qrr;
  float lse_r = 0.f;
  unsigned long long kw_r = 0ull;
  auto q_load = [&](int qt) {
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
      const int id = tid + 256 * i2, row = id >> 3, c = id & 7;
      const int q = qt * 64 + row, qcl = q < S ? q : S - 1;
      qc8[i2] = chunk_load(QC + (size_t)(b * S + qcl) * a.ldqc + h * 64 + 8 * c);
      dc8[i2] = chunk_load(dO + (size_t)(b * S + qcl) * ba.lddo + h * 64 + 8 * c);
      oc8[i2] = chunk_load(O + (size_t)(b * S + qcl) * a.ldo + h * 64 + 8 * c);
      if (rot) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { qrr.c[i2][j] = a.cos[qcl * 32 + 4 * c + j]; qrr.s[i2][j] = a.sin[qcl * 32 + 4 * c + j]; }
      }
    }
    if (tid < 64) {
      const int q = qt * 64 + tid;
      lse_r = a.LSE[(size_t)(b * a.H + h) * S + (q < S ? q : S - 1)];
      if (DROP && BITS) kw_r = a.keepbits[keepbits_index(b, a.H, h, nqt, tile, q)];   // this key tile's keep words of the 64 queries: 512 contiguous bytes
    }
  };
  auto q_store = [&](int qt) {
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
      const int id = tid + 256 * i2, row = id >> 3, c = id & 7;
      const int q = qt * 64 + row;
      float qx[8], dch[8], och[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { qx[i] = chunk_get(qc8[i2], i); dch[i] = chunk_get(dc8[i2], i); och[i] = chunk_get(oc8[i2], i); }
      if (q < S) {
        if (rot) {
#pragma unroll
          for (int j = 0; j < 4; ++j) rope_pair(qx[2 * j], qx[2 * j + 1], qrr.c[i2][j], qrr.s[i2][j], 1.f);   // = rope8
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) qx[i] = to_f(from_f<T>(qx[i])) * 0.125f;
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { qx[i] = 0.f; dch[i] = 0.f; och[i] = 0.f; }
      }
      store8f(Qs + row * ALD + 8 * c, qx);
      store8f(dOs + row * ALD + 8 * c, dch);
      // delta[q] = sum_d dO[q][d] O[q][d] (8 lanes share a row), so this role does not depend on
      // the dQ role and both run in one launch
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) part += dch[i] * och[i];
      part = group_sum<8>(part);
      if (c == 0) delta_s[row] = part;
    }
    if (tid < 64) {
      lse_s[tid] = (qt * 64 + tid < S) ? lse_r : 0.f;
      if (DROP && BITS) { kb_s[tid] = (uint32_t)kw_r; kb_s[64 + tid] = (uint32_t)(kw_r >> 32); }
    }
  };
  if (half < nqt) q_load(half);
  for (int it = 0; it < niter; ++it) {
    const int qt = NG * it + half;
    const bool active = qt < nqt;
    __syncthreads();
    if (active) q_store(qt);
    __syncthreads();
#ifndef AMT_ABL_NOLOAD
    if (qt + NG < nqt) q_load(qt + NG);
#endif
    if (active) {
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f}, dpacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          mma(sacc, frag_row(Qs + (16 * st + r) * ALD + 32 * ks + 8 * g), kf[ks]);     // S[q][key]
          mma(dpacc, frag_row(dOs + (16 * st + r) * ALD + 32 * ks + 8 * g), vf[ks]);   // dP[q][key]
        }
        float pd[4], ds[4];
        const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + 16 * st + 4 * g);
        const f32x4 del4 = *reinterpret_cast<const f32x4*>(delta_s + 16 * st + 4 * g);
        // this lane's key is bit (16 wave + r) of a query's word: the half is uniform over the wave
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
        u32x4 kq = u32x4{0u, 0u, 0u, 0u};
        if (DROP && BITS) kq = *reinterpret_cast<const u32x4*>(kb_s + 64 * (wave >> 1) + 16 * st + 4 * g);
        const int kbit = 16 * (wave & 1) + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int ql = 16 * st + 4 * g + e, q = qt * 64 + ql;
          bool valid = (q < S) && (key < S);
          float p = valid ? __expf(sacc[e] - lse4[e]) : 0.f;
          float dp = dpacc[e], pdrop = p;
          if (DROP) {
            const bool keep = BITS ? ((kq[e] >> kbit) & 1u) != 0u
                                   : rng_keep(a.seed, a.stream, gdrop_idx(b, a.H, h, S, q, key), a.drop_thresh24);
            dp = keep ? dp * a.inv_keep : 0.f;
            pdrop = keep ? p * a.inv_keep : 0.f;
          }
          pd[e] = pdrop;
          ds[e] = p * (dp - del4[e]);
        }
        store4f(PTw + r * ALD + 16 * st + 4 * g, pd);
        store4f(dSTw + r * ALD + 16 * st + 4 * g, ds);
      }
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        Frag<T> pf = frag_row(PTw + r * ALD + 32 * ks + 8 * g);
        Frag<T> dsf = frag_row(dSTw + r * ALD + 32 * ks + 8 * g);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          mma(dvacc[dt], frag_col16(dOs + (32 * ks + 8 * g) * ALD + 16 * dt, ALD, r), pf);   // dV^T[d][key]
          mma(dkacc[dt], frag_col16(Qs + (32 * ks + 8 * g) * ALD + 16 * dt, ALD, r), dsf);   // dK^T[d][key]
        }
      }
    }
  }
  if constexpr (NG > 1) {
    __syncthreads();
    float* mg = reinterpret_cast<float*>(smem_all) + (size_t)(wave * 64 + lane) * 32;
    if (half == 1) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int e = 0; e < 4; ++e) { mg[4 * dt + e] = dkacc[dt][e]; mg[16 + 4 * dt + e] = dvacc[dt][e]; }
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int e = 0; e < 4; ++e) { dkacc[dt][e] += mg[4 * dt + e]; dvacc[dt][e] += mg[16 + 4 * dt + e]; }
    }
  }
  if (half == 0 && key < S) {
    T* dk = reinterpret_cast<T*>(ba.dKV) + (size_t)(b * S + key) * ba.lddkv + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      float d[4], v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { d[e] = dkacc[dt][e]; v[e] = dvacc[dt][e]; }
      rope4_bwd(d, a.cos, a.sin, key, 8 * dt + 2 * g, 1.f);
      store4f(dk + 16 * dt + 4 * g, d);
      store4f(dk + 256 + 16 * dt + 4 * g, v);
    }
  }
}

// blocks [0, ntile) of a (clip, head) compute dQ, blocks [ntile, 2 ntile) dK/dV
template <typename T, int NG, bool DROP, bool BITS>
#ifndef AMT_ATTN_BWD_OCC
#define AMT_ATTN_BWD_OCC 2
#endif
__global__ __launch_bounds__(256 * NG, (sizeof(T) == 2 ? AMT_ATTN_BWD_OCC : 2) * NG) void attn_global_bwd_kernel(AttnBwdArgs ba, int ntile) {
  ba.f.seed = eff_seed(ba.f.seed, ba.f.seed_dev);   // kernel argument + the context's device seed word (common.h)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // XCD-aware order: the 2 * ntile workgroups of one (clip, head) share its Q/K/V/dO: same id % 8
  const int per = 2 * ntile;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int bh = (jj / per) * 8 + xcd, role = jj % per;
  if (bh >= ba.f.B * ba.f.H) return;
  const int h = bh % ba.f.H, b = bh / ba.f.H;
  if (role < ntile) attn_global_bwd_dq_body<T, NG, DROP, BITS>(ba, role, b, h, smem_raw);
  else attn_global_bwd_dkv_body<T, NG, DROP, BITS>(ba, role - ntile, b, h, smem_raw);
}

template <typename K> static void set_max_lds(K kernel, size_t bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes);
}

void attn_global_fwd(hipStream_t s, DType dt, const AttnArgs& a) {
  dim3 grid(((a.B * a.H + 7) / 8 * 8) * ((a.S + 63) / 64));
  ProfScope ps(s, "attn_global_fwd", 4.0 * a.B * a.H * (double)a.S * a.S * 64,
               4.0 * a.B * a.S * a.H * 64 * (dt == DT_F32 ? 4 : 2));
  if (dt == DT_F32) {
    size_t lds = (size_t)2 * (64 * 2 + 64) * ALD * sizeof(float);
    static bool once = (set_max_lds(attn_global_fwd_kernel<float, 2, true>, 128 * 1024),
                        set_max_lds(attn_global_fwd_kernel<float, 2, false>, 128 * 1024), true);
    (void)once;
    if (a.drop_thresh24) hipLaunchKernelGGL((attn_global_fwd_kernel<float, 2, true>), grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL((attn_global_fwd_kernel<float, 2, false>), grid, dim3(512), lds, s, a);
  } else {
    // (four groups of two tiles each measured the same 13.9 us as two groups of four: what is left is the
    // first-tile latency, the merge and the launch itself)
    size_t lds = (size_t)2 * (64 * 2 + 64) * ALD * sizeof(bf16);
    // short sequences (the reference's 5-s windows: 250 frames = 4 key tiles): one wave group per workgroup - two groups
    // would take two key tiles each and then merge.  Chosen by S alone, so that a window's logits do not depend on how
    // many windows share the call (AMT_ATTN_FWD_NG=2 keeps two groups).
    static const int ng_env = getenv("AMT_ATTN_FWD_NG") ? atoi(getenv("AMT_ATTN_FWD_NG")) : 0;
    if (ng_env != 2 && a.S <= 256) {
      if (a.drop_thresh24) hipLaunchKernelGGL((attn_global_fwd_kernel<bf16, 1, true>), grid, dim3(256), lds / 2, s, a);
      else hipLaunchKernelGGL((attn_global_fwd_kernel<bf16, 1, false>), grid, dim3(256), lds / 2, s, a);
      return;
    }
    if (a.drop_thresh24) hipLaunchKernelGGL((attn_global_fwd_kernel<bf16, 2, true>), grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL((attn_global_fwd_kernel<bf16, 2, false>), grid, dim3(512), lds, s, a);
  }
}

void attn_global_bwd(hipStream_t s, DType dt, const AttnBwdArgs& a) {
  const int ntile = (a.f.S + 63) / 64;
  dim3 grid(((a.f.B * a.f.H + 7) / 8 * 8) * 2 * ntile);
  ProfScope ps(s, "attn_global_bwd", 8.0 * a.f.B * a.f.H * (double)a.f.S * a.f.S * 64,
               8.0 * a.f.B * a.f.S * a.f.H * 64 * (dt == DT_F32 ? 4 : 2));
  const bool drop = a.f.drop_thresh24 != 0, bits = drop && a.f.keepbits != nullptr;
  if (dt == DT_F32) {
    static bool once = (set_max_lds(attn_global_bwd_kernel<float, 1, true, true>, 96 * 1024),
                        set_max_lds(attn_global_bwd_kernel<float, 1, true, false>, 96 * 1024),
                        set_max_lds(attn_global_bwd_kernel<float, 1, false, false>, 96 * 1024), true);
    (void)once;
    const size_t lds = GBwdLds<float>::GROUP_BYTES;
    if (bits) hipLaunchKernelGGL((attn_global_bwd_kernel<float, 1, true, true>), grid, dim3(256), lds, s, a, ntile);
    else if (drop) hipLaunchKernelGGL((attn_global_bwd_kernel<float, 1, true, false>), grid, dim3(256), lds, s, a, ntile);
    else hipLaunchKernelGGL((attn_global_bwd_kernel<float, 1, false, false>), grid, dim3(256), lds, s, a, ntile);
  } else {
    // NG = 2 (the forward's split) measured 47.5 us against 38.2 us for NG = 1 stand-alone: 128 VGPRs with 37
    // spilled, and this kernel is bound by VALU / LDS issue (~720 VALU + 95 LDS instructions per tile), not latency
    const size_t lds = GBwdLds<bf16>::GROUP_BYTES;
    if (bits) hipLaunchKernelGGL((attn_global_bwd_kernel<bf16, 1, true, true>), grid, dim3(256), lds, s, a, ntile);
    else if (drop) hipLaunchKernelGGL((attn_global_bwd_kernel<bf16, 1, true, false>), grid, dim3(256), lds, s, a, ntile);
    else hipLaunchKernelGGL((attn_global_bwd_kernel<bf16, 1, false, false>), grid, dim3(256), lds, s, a, ntile);
  }
}


// number of valid windows covering padded/output position p
AMT_DEV float local_cnt(int p, int nW) {
  int st = p >> 3;
  float c = 0.f;
  if (st - 1 >= 0 && st - 1 < nW) c += 1.f;
  if (st < nW) c += 1.f;
  return c;
}

template <typename T>
__global__ __launch_bounds__(64) void attn_local_bwd_kernel(AttnBwdArgs ba, int padL, int nW) {
  ba.f.seed = eff_seed(ba.f.seed, ba.f.seed_dev);   // kernel argument + the context's device seed word (common.h)
  const AttnArgs& a = ba.f;
  __shared__ __attribute__((aligned(16))) T qA[16][ALD], qB[16][ALD], kA[16][ALD], kB[16][ALD], vs[24][ALD], dOs[24][ALD];
  // operand tiles of the second GEMMs, k = 0..15 window A, k = 16..31 window B:
  //   opq rows 0..7  = dS_A[8 + m][key k]       rows 8..15 = dS_B[m - 8][key k - 16]    (other halves zero)
  //   opk rows 0..7  = dS_A[query k][8 + m]      rows 8..15 = dS_B[query k - 16][m - 8]
  //   opv rows 0..7  = Pd_A[query k][8 + m] | Pd_B[query k - 16][m]                      rows 8..15 zero
  __shared__ __attribute__((aligned(16))) T ops[3][16][OLD];
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  const int s = blockIdx.x, h = blockIdx.y, b = blockIdx.z, S = a.S;
  const bool validA = (s >= 1) && (s - 1 < nW), validB = (s < nW);
  const float fa = validA ? 1.f : 0.f, fb = validB ? 1.f : 0.f;
  const T* QC = reinterpret_cast<const T*>(a.QC);
  const T* KV = reinterpret_cast<const T*>(a.KV);
  const T* dO = reinterpret_cast<const T*>(ba.dO);
  {
    // issue all twelve loads of this lane, then rotate / store
    float qx[3][8], kx[3][8], vx[3][8], dx[3][8];
    bool okt[3];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      int id = lane + 64 * it, jp = id >> 3, c = id & 7;
      int p = 8 * s - 8 + jp;
      okt[it] = load_tok_chunk(QC, a.ldqc, b, S, p, padL, h * 64 + 8 * c, qx[it]);
      load_tok_chunk(KV, a.ldkv, b, S, p, padL, h * 64 + 8 * c, kx[it]);
      load_tok_chunk(KV, a.ldkv, b, S, p, padL, 256 + h * 64 + 8 * c, vx[it]);
      int pc = p < 0 ? 0 : (p >= S ? S - 1 : p);   // dO lives at OUTPUT coordinate p (no pad shift)
      load8f(dO + (size_t)(b * S + pc) * ba.lddo + h * 64 + 8 * c, dx[it]);
    }
    T* opz = &ops[0][0][0];
    for (int i = lane; i < 3 * 16 * OLD / 8; i += 64) chunk_store(opz + 8 * i, chunk_zero<T>());
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      int id = lane + 64 * it, jp = id >> 3, c = id & 7;
      int p = 8 * s - 8 + jp;
      mask8(qx[it], okt[it]);
      mask8(kx[it], okt[it]);
      mask8(vx[it], okt[it]);
      if (jp < 16) store_roped<T>(qA[jp], c, qx[it], a.cos, a.sin, jp, 0.125f);
      if (jp >= 8) store_roped<T>(qB[jp - 8], c, qx[it], a.cos, a.sin, jp - 8, 0.125f);
      if (jp < 16) store_roped<T>(kA[jp], c, kx[it], a.cos, a.sin, jp, 1.f);
      if (jp >= 8) store_roped<T>(kB[jp - 8], c, kx[it], a.cos, a.sin, jp - 8, 1.f);
      // window-output gradient divided by the cover count (1 or 2: exact); zero outside [0, S)
      const bool okp = (p >= 0) && (p < S);
      const float ic = okp ? 1.f / local_cnt(okp ? p : 0, nW) : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) dx[it][i] *= ic;
      store8f(&vs[jp][8 * c], vx[it]);
      store8f(&dOs[jp][8 * c], dx[it]);
    }
  }
  // rotation entries of the output phase: this lane un-rotates rows 4 g + e of the result tiles, i.e.
  // window position 8 + 4 g + e (window A, lane groups 0-1) or 4 (g - 2) + e (window B), feature 16 dt + r
  float ucs[4][4], usn[4][4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int pos = g < 2 ? 8 + 4 * g + e : 4 * (g - 2) + e;
      ucs[dt][e] = a.cos[pos * 32 + 8 * dt + (r >> 1)];
      usn[dt][e] = a.sin[pos * 32 + 8 * dt + (r >> 1)];
    }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    T (*qw)[ALD] = w ? qB : qA;
    T (*kw)[ALD] = w ? kB : kA;
    const int off = 8 * w;
    // transposed scores and dP: lane (r, g) holds keys 4 g + e of query r
    f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f}, dp = sc;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      mma(sc, frag_row(&kw[r][32 * ks + 8 * g]), frag_row(&qw[r][32 * ks + 8 * g]));
      mma(dp, frag_row(&vs[off + r][32 * ks + 8 * g]), frag_row(&dOs[off + r][32 * ks + 8 * g]));
    }
    float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
    mx = max_xor32(max_xor16(mx));
    float p[4], sum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { p[e] = __expf(sc[e] - mx); sum += p[e]; }
    sum = sum_xor32(sum_xor16(sum));
    const float inv = 1.f / sum;
    const int win = s - 1 + w;
    float dl = 0.f, pd[4], dpm[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      p[e] *= inv;
      pd[e] = p[e];
      dpm[e] = dp[e];
      if (a.drop_thresh24) {
        bool keep = rng_keep(a.seed, a.stream, ldrop_idx(b, a.H, h, nW, win, r, 4 * g + e), a.drop_thresh24);
        pd[e] = keep ? p[e] * a.inv_keep : 0.f;
        dpm[e] = keep ? dpm[e] * a.inv_keep : 0.f;
      }
      dl += p[e] * dpm[e];
    }
    dl = sum_xor32(sum_xor16(dl));
    float ds[4];
    const float fw = w ? fb : fa;
#pragma unroll
    for (int e = 0; e < 4; ++e) { ds[e] = p[e] * (dpm[e] - dl); pd[e] *= fw; }
    if (w == 0) {
      if (r >= 8) store4f(&ops[0][r - 8][4 * g], ds);
      if (g >= 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ops[1][4 * g + e - 8][r] = from_f<T>(ds[e]);
          ops[2][4 * g + e - 8][r] = from_f<T>(pd[e]);
        }
      }
    } else {
      if (r < 8) store4f(&ops[0][8 + r][16 + 4 * g], ds);
      if (g < 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ops[1][8 + 4 * g + e][16 + r] = from_f<T>(ds[e]);
          ops[2][4 * g + e][16 + r] = from_f<T>(pd[e]);
        }
      }
    }
  }
  __syncthreads();
  // vs is dead: its 24 rows become the three [8][64] output tiles (dq, dk, dv of the stripe)
  T (*outq)[ALD] = vs;
  T (*outk)[ALD] = vs + 8;
  T (*outv)[ALD] = vs + 16;
  {
    const Frag<T> fq = frag_row(&ops[0][r][8 * g]), fk = frag_row(&ops[1][r][8 * g]), fv = frag_row(&ops[2][r][8 * g]);
    const T* kp = g < 2 ? &kA[8 * g][0] : &kB[8 * (g - 2)][0];
    const T* qp = g < 2 ? &qA[8 * g][0] : &qB[8 * (g - 2)][0];
    const T* dop = &dOs[g == 0 ? 0 : (g == 3 ? 16 : 8)][0];
    const float fw = g < 2 ? fa : fb;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 aq = f32x4{0.f, 0.f, 0.f, 0.f}, ak = aq, av = aq;
      mma(aq, fq, frag_col16(kp + 16 * dt, ALD, r));
      mma(ak, fk, frag_col16(qp + 16 * dt, ALD, r));
      mma(av, fv, frag_col16(dop + 16 * dt, ALD, r));
      // acc[e] = result row 4 g + e (rows 0..7: window A part of position i = row, rows 8..15: window B part
      // of position i = row - 8), feature d = 16 dt + r.  Transpose of the rotation: the partner feature
      // sits in the neighbouring lane.
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float c = ucs[dt][e], sn = usn[dt][e];
        const float oq = dpp_mov<DPP_XOR1>(aq[e]), ok = dpp_mov<DPP_XOR1>(ak[e]);
        float x = ((r & 1) ? (-oq * sn + aq[e] * c) : (aq[e] * c + oq * sn)) * (fw * 0.125f);
        float y = ((r & 1) ? (-ok * sn + ak[e] * c) : (ak[e] * c + ok * sn)) * fw;
        swap32(x, y);        // lanes 0..31: x = dq parts A, y = dq part B;  lanes 32..63: dk part A, dk part B
        const float tot = x + y;
        const int i = 4 * (g & 1) + e;
        if (g < 2) {
          outq[i][16 * dt + r] = from_f<T>(tot);
          outv[i][16 * dt + r] = from_f<T>(av[e]);
        } else {
          outk[i][16 * dt + r] = from_f<T>(tot);
        }
      }
    }
  }
  __syncthreads();
  {
    const int i = lane >> 3, c = lane & 7;
    const int t = 8 * s + i - padL;
    if (t >= 0 && t < S) {
      const size_t row = (size_t)(b * S + t);
      T* dQC = reinterpret_cast<T*>(ba.dQC);
      T* dKV = reinterpret_cast<T*>(ba.dKV);
      chunk_store(dQC + row * ba.lddqc + h * 64 + 8 * c, chunk_load(&outq[i][8 * c]));
      chunk_store(dKV + row * ba.lddkv + h * 64 + 8 * c, chunk_load(&outk[i][8 * c]));
      chunk_store(dKV + row * ba.lddkv + 256 + h * 64 + 8 * c, chunk_load(&outv[i][8 * c]));
    }
  }
}

void attn_local_fwd(hipStream_t s, DType dt, const AttnArgs& a) {
  LocalGeom gm = local_geom(a.S, 16);
  dim3 grid((a.S + 7) / 8, a.H, a.B);
  ProfScope ps(s, "attn_local_fwd", 4.0 * a.B * a.H * gm.nW * 16.0 * 16 * 64,
               4.0 * a.B * a.S * a.H * 64 * (dt == DT_F32 ? 4 : 2));
  if (dt == DT_F32) hipLaunchKernelGGL((attn_local_fwd_kernel<float>), grid, dim3(64), 0, s, a, gm.padL, gm.nW);
  else hipLaunchKernelGGL((attn_local_fwd_kernel<bf16>), grid, dim3(64), 0, s, a, gm.padL, gm.nW);
}

void attn_local_bwd(hipStream_t s, DType dt, const AttnBwdArgs& a) {
  LocalGeom gm = local_geom(a.f.S, 16);
  dim3 grid(gm.Lp / 8, a.f.H, a.f.B);
  ProfScope ps(s, "attn_local_bwd", 8.0 * a.f.B * a.f.H * gm.nW * 16.0 * 16 * 64,
               8.0 * a.f.B * a.f.S * a.f.H * 64 * (dt == DT_F32 ? 4 : 2));
  if (dt == DT_F32) hipLaunchKernelGGL((attn_local_bwd_kernel<float>), grid, dim3(64), 0, s, a, gm.padL, gm.nW);
  else hipLaunchKernelGGL((attn_local_bwd_kernel<bf16>), grid, dim3(64), 0, s, a, gm.padL, gm.nW);
}

}  // namespace amt
