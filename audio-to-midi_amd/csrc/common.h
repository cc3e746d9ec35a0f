// Common device helpers for the gfx950 (MI355X / CDNA4) kernels.
// wave = 64 lanes; MFMA 16x16x32 (bf16) or 8 x 16x16x4 (f32) per 32-deep K step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amt {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define AMT_DEV __device__ __forceinline__
constexpr int WAVE = 64;

// ------------------------------------------------------------------ the forward RESIDUAL STREAM's storage type
// Res<T>: how a context with activation type T stores the tensors of its forward residual stream - the CNN stream (Stem /
// Downsample / Block outputs: x and y = x + gamma z of model.py:160-167) and the transformer stream (x, h = x + attn,
// y = h + ffn of model.py:545-556, and the final CNN LayerNorm output that starts it).  float contexts: float.  bf16
// contexts: IEEE half - the reference's own activation type (train.py:35-37), 11 significand bits in the same two bytes:
// every LDS panel, copy loop and allocation keeps its size.  The stream is re-rounded at each of its 39 + 32 additions and
// every branch reads it through a LayerNorm; stored as bf16 (8 bits) those roundings alone put 2.4e-2 + 2.2e-2 of the
// 3.7e-2 maximum logit error of the bf16 path at the benchmark shape (profiles/r04_bf16_rounding_study.txt), against
// SURVEY 8(c)'s bound of 3e-2; as half they are below 2e-3.  Everything a GEMM multiplies (LayerNorm outputs, q, k, v,
// attention output, u, g, h = gelu(u)), every gradient (bf16's range: no loss scale) and the audio stay T.
template <typename T> struct ResOf { typedef T type; };
template <> struct ResOf<bf16> { typedef f16 type; };
template <typename T> using Res = typename ResOf<T>::type;

// ------------------------------------------------------------------ scalar conversions
template <typename T> AMT_DEV float to_f(T x) { return (float)x; }
template <typename T> AMT_DEV T from_f(float x) { return (T)x; }

// ------------------------------------------------------------------ vector load/store (n = 4 or 8 elements)
// 8 x bf16 = 16 B, 8 x f32 = 32 B (two 16-B accesses), 4 x bf16 = 8 B, 4 x f32 = 16 B.
template <typename T> struct V8;
template <> struct V8<bf16> { bf16x8 v; };
template <> struct V8<float> { f32x4 a, b; };
template <> struct V8<f16> { f16x8 v; };

AMT_DEV void load8f(const bf16* p, float (&o)[8]) {
  bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
}
AMT_DEV void load8f(const f16* p, float (&o)[8]) {
  f16x8 v = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
}
AMT_DEV void store8f(f16* p, const float (&o)[8]) {
  f16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (f16)o[i];
  *reinterpret_cast<f16x8*>(p) = v;
}
AMT_DEV void load4f(const f16* p, float (&o)[4]) {
  f16x4 v = *reinterpret_cast<const f16x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (float)v[i];
}
AMT_DEV void store4f(f16* p, const float (&o)[4]) {
  f16x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (f16)o[i];
  *reinterpret_cast<f16x4*>(p) = v;
}
AMT_DEV void load8f(const float* p, float (&o)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
}
AMT_DEV void store8f(bf16* p, const float (&o)[8]) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (bf16)o[i];
  *reinterpret_cast<bf16x8*>(p) = v;
}
AMT_DEV void store8f(float* p, const float (&o)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = o[i]; b[i] = o[4 + i]; }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
AMT_DEV void load4f(const bf16* p, float (&o)[4]) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (float)v[i];
}
AMT_DEV void load4f(const float* p, float (&o)[4]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = a[i];
}
AMT_DEV void store4f(bf16* p, const float (&o)[4]) {
  bf16x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (bf16)o[i];
  *reinterpret_cast<bf16x4*>(p) = v;
}
AMT_DEV void store4f(float* p, const float (&o)[4]) {
  f32x4 a;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = o[i];
  *reinterpret_cast<f32x4*>(p) = a;
}
// generic VEC (4 or 8) forms
template <int VEC, typename T> AMT_DEV void loadvf(const T* p, float (&o)[VEC]) {
  if constexpr (VEC == 8) load8f(p, o); else load4f(p, o);
}
template <int VEC, typename T> AMT_DEV void storevf(T* p, const float (&o)[VEC]) {
  if constexpr (VEC == 8) store8f(p, o); else store4f(p, o);
}

// raw (no conversion) 8-element chunk copy global -> regs -> LDS
template <typename T> struct Chunk8;
template <> struct Chunk8<bf16> { bf16x8 v; };
template <> struct Chunk8<float> { f32x4 a, b; };
template <> struct Chunk8<f16> { f16x8 v; };

template <typename T> AMT_DEV Chunk8<T> chunk_zero();
template <> AMT_DEV Chunk8<f16> chunk_zero<f16>() { Chunk8<f16> c; for (int i = 0; i < 8; ++i) c.v[i] = (f16)0.f; return c; }
AMT_DEV Chunk8<f16> chunk_load(const f16* p) { Chunk8<f16> c; c.v = *reinterpret_cast<const f16x8*>(p); return c; }
AMT_DEV void chunk_store(f16* p, const Chunk8<f16>& c) { *reinterpret_cast<f16x8*>(p) = c.v; }
AMT_DEV float chunk_get(const Chunk8<f16>& c, int i) { return (float)c.v[i]; }
template <> AMT_DEV Chunk8<bf16> chunk_zero<bf16>() { Chunk8<bf16> c; for (int i = 0; i < 8; ++i) c.v[i] = (bf16)0.f; return c; }
template <> AMT_DEV Chunk8<float> chunk_zero<float>() { Chunk8<float> c; c.a = f32x4{0, 0, 0, 0}; c.b = f32x4{0, 0, 0, 0}; return c; }

AMT_DEV Chunk8<bf16> chunk_load(const bf16* p) { Chunk8<bf16> c; c.v = *reinterpret_cast<const bf16x8*>(p); return c; }
AMT_DEV Chunk8<float> chunk_load(const float* p) {
  Chunk8<float> c; c.a = *reinterpret_cast<const f32x4*>(p); c.b = *reinterpret_cast<const f32x4*>(p + 4); return c;
}
// 4 valid elements followed by 4 zeros (K == 4 / M == 4 operands)
AMT_DEV Chunk8<bf16> chunk_load4(const bf16* p) {
  Chunk8<bf16> c = chunk_zero<bf16>();
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  for (int i = 0; i < 4; ++i) c.v[i] = v[i];
  return c;
}
AMT_DEV Chunk8<float> chunk_load4(const float* p) {
  Chunk8<float> c = chunk_zero<float>();
  c.a = *reinterpret_cast<const f32x4*>(p);
  return c;
}
AMT_DEV void chunk_store(bf16* p, const Chunk8<bf16>& c) { *reinterpret_cast<bf16x8*>(p) = c.v; }
AMT_DEV void chunk_store(float* p, const Chunk8<float>& c) {
  *reinterpret_cast<f32x4*>(p) = c.a; *reinterpret_cast<f32x4*>(p + 4) = c.b;
}
AMT_DEV float chunk_get(const Chunk8<bf16>& c, int i) { return (float)c.v[i]; }
AMT_DEV float chunk_get(const Chunk8<float>& c, int i) { return i < 4 ? c.a[i] : c.b[i - 4]; }

// ------------------------------------------------------------------ MFMA fragments
// One K step is 32 deep.  Lane l = (r = l & 15, g = l >> 4) holds operand elements
// k = 8 g + j, j = 0..7, of row r (A: A[r][k]; B: B[k][r]).  For bf16 that is exactly the
// v_mfma_f32_16x16x32_bf16 operand map; for f32 the same registers feed 8 x
// v_mfma_f32_16x16x4_f32 (element j of every lane forms k-slice {j, 8+j, 16+j, 24+j}), so both
// dtypes share every address computation.  C/D: acc[i] = D[row 4 g + i][col r].
template <typename T> struct Frag;
template <> struct Frag<bf16> { bf16x8 v; };
template <> struct Frag<float> { float v[8]; };

// 8 contiguous elements
AMT_DEV Frag<bf16> frag_row(const bf16* p) { Frag<bf16> f; f.v = *reinterpret_cast<const bf16x8*>(p); return f; }
AMT_DEV Frag<float> frag_row(const float* p) {
  Frag<float> f;
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { f.v[i] = a[i]; f.v[4 + i] = b[i]; }
  return f;
}
// 8 elements at stride `ld` (column of a row-major tile)
AMT_DEV Frag<bf16> frag_col(const bf16* p, int ld) {
  Frag<bf16> f;
#pragma unroll
  for (int i = 0; i < 8; ++i) f.v[i] = p[i * ld];
  return f;
}
AMT_DEV Frag<float> frag_col(const float* p, int ld) {
  Frag<float> f;
#pragma unroll
  for (int i = 0; i < 8; ++i) f.v[i] = p[i * ld];
  return f;
}
// Column fragment of a 16-column sub-tile: p0 = &tile[first of this lane group's 8 rows][first
// column of the sub-tile], r = lane & 15 selects the column.  bf16 uses the gfx950 transposing
// LDS read (ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block is delivered
// column-major; lane 4q+p supplies the address of row q, columns 4p..4p+3), two reads per
// fragment instead of eight 16-bit reads.  Needs all 64 lanes active and 8-byte aligned rows.
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;
AMT_DEV Frag<bf16> frag_col16(const bf16* p0, int ld, int r) {
  const bf16* a0 = p0 + (r >> 2) * ld + 4 * (r & 3);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + 4 * ld));
  Frag<bf16> f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { f.v[i] = lo[i]; f.v[4 + i] = hi[i]; }
  return f;
}
AMT_DEV Frag<float> frag_col16(const float* p0, int ld, int r) { return frag_col(p0 + r, ld); }

template <typename T> AMT_DEV Frag<T> frag_zero();
template <> AMT_DEV Frag<bf16> frag_zero<bf16>() { Frag<bf16> f; for (int i = 0; i < 8; ++i) f.v[i] = (bf16)0.f; return f; }
template <> AMT_DEV Frag<float> frag_zero<float>() { Frag<float> f; for (int i = 0; i < 8; ++i) f.v[i] = 0.f; return f; }

AMT_DEV void mma(f32x4& acc, const Frag<bf16>& a, const Frag<bf16>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
AMT_DEV void mma(f32x4& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
  for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[i], b.v[i], acc, 0, 0, 0);
}

// ------------------------------------------------------------------ math
// jax.nn.gelu default (approximate=True): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3), written as
// x * sigmoid(2u) = x / (1 + exp(-2u)) with exp(-2u) = exp2(x (c0 + c1 x^2)): SEVEN instructions (mul, fma, mul, v_exp_f32,
// add, v_rcp_f32, mul).  Round 2's form (separate constant multiplies, __expf, and __frcp_rn = a correctly rounded
// division: v_div_scale x 2, v_rcp, four fma, v_div_fmas, v_div_fixup) compiled to 22 and made every GELU / GLU epilogue
// VALU-bound: 57 million evaluations per launch at streaming-inference sizes, 891 of the ~1100 vector instructions
// between two MFMA stages of the FFN chain (in-kernel stamps, profiles/r03_notes.md).  v_exp_f32 and v_rcp_f32 are
// accurate to 1 ulp, the whole expression to ~4e-7 relative: inside the fp32 tolerance of the parity tests (2e-5 against float64).
constexpr float GELU_C0 = -2.f * 0.7978845608028654f * 1.4426950408889634f;   // -2 sqrt(2/pi) log2(e)
constexpr float GELU_C1 = GELU_C0 * 0.044715f;
AMT_DEV float gelu_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(x * x, GELU_C1, GELU_C0));   // exp(-2u); inf for x << 0
  return x * __builtin_amdgcn_rcpf(1.f + e);
}
AMT_DEV float gelu_tanh_grad(float x) {
  const float t = x * x;
  const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(t, GELU_C1, GELU_C0));
  const float s = __builtin_amdgcn_rcpf(1.f + e);                       // (1 + tanh u) / 2
  const float du2 = __builtin_fmaf(t, 6.f * 0.7978845608028654f * 0.044715f, 2.f * 0.7978845608028654f);   // 2 du/dx
  return __builtin_fmaf(x * s * (1.f - s), du2, s);                     // s + 2 x s (1 - s) du/dx
}
AMT_DEV float sigmoidf(float z) { return 1.f / (1.f + __expf(-z)); }

// ------------------------------------------------------------------ cross-lane reductions
// __shfl_xor lowers to ds_bpermute_b32 (an LDS-crossbar round trip, ~100+ cycles, serialised by
// its wait): 176-264 of them made the reduction tail of the backward row kernels cost ~12 us.
// These use DPP (one VALU instruction, fused into v_add_f32_dpp) inside a 16-lane row and the
// gfx950 v_permlane16_swap / v_permlane32_swap across rows.
template <int CTRL> AMT_DEV float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7 - i inside each group of 8
constexpr int DPP_MIRROR = 0x140;      // lane i <-> 15 - i inside each row of 16
constexpr int DPP_ROR4 = 0x124;        // rotate right by 4 inside each row of 16
constexpr int DPP_ROR8 = 0x128;
// value of lane (l ^ 16) / (l ^ 32) paired with the own value: both halves of the swap result
// Inline asm on purpose: with both operands holding the same value, hipcc (ROCm 7.2) folded the two
// results of __builtin_amdgcn_permlane{16,32}_swap into one register (a + a instead of a + b).
// Two read-write operands cannot share a register.  s_nop 1 covers the VALU-write -> permlane
// read hazard (cdna_hip_programming.md T21).
AMT_DEV void xrow16(float v, float& a, float& b) {
  float x = v, y = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(x), "+v"(y));
  a = x; b = y;
}
AMT_DEV void xrow32(float v, float& a, float& b) {
  float x = v, y = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(x), "+v"(y));
  a = x; b = y;
}
// two-register forms: x = [x.r0, y.r0, x.r2, y.r2], y = [x.r1, y.r1, x.r3, y.r3] (rows of 16 lanes) and
// x = [x.lo, y.lo], y = [x.hi, y.hi] (halves of 32 lanes).  x + y then holds the pairwise row / half sums
// of BOTH inputs: one swap reduces two values (fold_columns in chain.h).
AMT_DEV void swap16(float& x, float& y) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(x), "+v"(y));
}
AMT_DEV void swap32(float& x, float& y) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(x), "+v"(y));
}
AMT_DEV float sum_xor16(float v) { float a, b; xrow16(v, a, b); return a + b; }
AMT_DEV float sum_xor32(float v) { float a, b; xrow32(v, a, b); return a + b; }
AMT_DEV float max_xor16(float v) { float a, b; xrow16(v, a, b); return fmaxf(a, b); }
AMT_DEV float max_xor32(float v) { float a, b; xrow32(v, a, b); return fmaxf(a, b); }

// all-reduce over groups of WIDTH consecutive lanes (WIDTH power of two <= 64)
template <int WIDTH> AMT_DEV float group_sum(float v) {
  if constexpr (WIDTH >= 2) v += dpp_mov<DPP_XOR1>(v);
  if constexpr (WIDTH >= 4) v += dpp_mov<DPP_XOR2>(v);
  if constexpr (WIDTH >= 8) v += dpp_mov<DPP_HALF_MIRROR>(v);
  if constexpr (WIDTH >= 16) v += dpp_mov<DPP_MIRROR>(v);
  if constexpr (WIDTH >= 32) v = sum_xor16(v);
  if constexpr (WIDTH >= 64) v = sum_xor32(v);
  return v;
}
template <int WIDTH> AMT_DEV float group_max(float v) {
  if constexpr (WIDTH >= 2) v = fmaxf(v, dpp_mov<DPP_XOR1>(v));
  if constexpr (WIDTH >= 4) v = fmaxf(v, dpp_mov<DPP_XOR2>(v));
  if constexpr (WIDTH >= 8) v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR>(v));
  if constexpr (WIDTH >= 16) v = fmaxf(v, dpp_mov<DPP_MIRROR>(v));
  if constexpr (WIDTH >= 32) v = max_xor16(v);
  if constexpr (WIDTH >= 64) v = max_xor32(v);
  return v;
}
// all-reduce over the lanes that share (lane % STRIDE) (STRIDE power of two <= 32)
template <int STRIDE> AMT_DEV float strided_sum(float v) {
  if constexpr (STRIDE <= 1) v += dpp_mov<DPP_XOR1>(v);
  if constexpr (STRIDE <= 2) v += dpp_mov<DPP_XOR2>(v);
  if constexpr (STRIDE <= 4) v += dpp_mov<DPP_ROR4>(v);
  if constexpr (STRIDE <= 8) v += dpp_mov<DPP_ROR8>(v);
  if constexpr (STRIDE <= 16) v = sum_xor16(v);
  if constexpr (STRIDE <= 32) v = sum_xor32(v);
  return v;
}

// ------------------------------------------------------------------ counter-based dropout RNG
// keep(seed, stream, idx) is a pure function, so the backward pass regenerates the forward mask.
// (JAX's threefry streams cannot be reproduced; SURVEY.md §7 "Dropout parity".)
AMT_DEV uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// (seed, stream) -> 32-bit key: wave-uniform, so it is computed once on the scalar unit; per element
// one multiply + one mix32 (the three-round form cost 0.66 us per 64-key attention tile: 16 elements
// per lane, profiles/r01_notes.md).
AMT_DEV uint32_t rng_key(uint64_t seed, uint32_t stream) {
  uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
  return mix32(mix32(s0 ^ (stream * 0x85ebca6bU)) + s1 + 0x165667b1U);
}
AMT_DEV uint32_t rng_u32(uint64_t seed, uint32_t stream, uint32_t idx) {
  return mix32((idx * 0x9e3779b1U) ^ rng_key(seed, stream));
}
// The seed of a launch is `seed` (a kernel argument, frozen when a hipGraph is captured) plus an optional device word
// the host or a tiny kernel changes between replays (include/amt.h: amt_seed_word): replayed steps get fresh masks.
AMT_DEV uint64_t eff_seed(uint64_t seed, const uint64_t* seed_dev) { return seed_dev ? seed + *seed_dev : seed; }
// keep with probability (1 - p): threshold on the top 24 bits
AMT_DEV bool rng_keep(uint64_t seed, uint32_t stream, uint32_t idx, uint32_t drop_thresh24) {
  return (rng_u32(seed, stream, idx) >> 8) >= drop_thresh24;
}

}  // namespace amt
