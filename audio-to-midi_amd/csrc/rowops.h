// Shared helpers for the row-wise (per token / per position) kernels: a row of C features is
// spread over LPR = C / VEC consecutive lanes, VEC = 8 (4 for C = 4) contiguous features each,
// so every global access is a 16-byte (bf16: 8-byte for C = 4) vector and LayerNorm statistics
// are wavefront shuffles over LPR lanes.
#pragma once
#include "common.h"

namespace amt {

template <int C> struct RowCfg {
  static constexpr int VEC = (C >= 8) ? 8 : 4;
  static constexpr int LPR = C / VEC;       // lanes per row (1..32)
  static constexpr int RPW = 64 / LPR;      // rows per wave
  static constexpr int RPB = 4 * RPW;       // rows per 256-thread block
};

// Sum per-lane partial column sums over every lane of the block that owns the same feature
// chunk, then atomically add the C results to global memory.  All 256 threads must call it.
template <int VEC, int LPR>
// pstride == 0: atomicAdd into gdst[0..C).  pstride > 0: plain store into this block's record,
// gdst[blockIdx.x * pstride + 0..C) — the records are summed once per step by reduce_partials,
// which removes the same-address atomic contention of hundreds of blocks.
AMT_DEV void reduce_cols_to_global(const float (&acc)[VEC], float* gdst, float* smem /* >= 4*VEC*LPR floats */,
                                   int pstride = 0) {
  // every wave stores its column totals into its own slice and the four slices are summed after a
  // barrier: LDS float atomics cost ~3.6 cycles per lane-add (measured in chain.hip: 20 k cycles per
  // workgroup for 88 accumulators), plain stores are free by comparison
  constexpr int C = VEC * LPR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane % LPR, rw = lane / LPR;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    float v = strided_sum<LPR>(acc[i]);
    if (rw == 0) smem[wave * C + sub * VEC + i] = v;
  }
  __syncthreads();
  if (gdst != nullptr) {
    for (int t = tid; t < C; t += 256) {
      const float sum = (smem[t] + smem[C + t]) + (smem[2 * C + t] + smem[3 * C + t]);
      if (pstride > 0) gdst[(size_t)blockIdx.x * pstride + t] = sum;
      else atomicAdd(&gdst[t], sum);
    }
  }
}



static inline int rows_per_block(int C) {
  int vec = C >= 8 ? 8 : 4;
  return 4 * (64 / (C / vec));
}

static inline int row_grid(long N, int rows_per_block, int max_blocks = 2048) {
  long nb = (N + rows_per_block - 1) / rows_per_block;
  if (nb > max_blocks) nb = max_blocks;
  if (nb < 1) nb = 1;
  return (int)nb;
}

// Backward row kernels end with one atomic per feature per block into the SAME C addresses
// (parameter gradients).  Contended float atomics run ~14x below the uncontended rate
// (MI355X_MICROARCH.md, Global float atomics), so keep the number of adding blocks small and give
// every block at least 8 row groups of streaming work.
static inline int bwd_row_grid(long N, int rows_per_block, int max_blocks = 128, int min_groups = 8) {
  long nb = (N + (long)rows_per_block * min_groups - 1) / ((long)rows_per_block * min_groups);
  if (nb > max_blocks) nb = max_blocks;
  if (nb < 1) nb = 1;
  return (int)nb;
}

// grid of a backward row kernel: with per-block records (sink) the blocks do not contend, so use more
template <typename Sink> static inline int bwd_blocks(long N, int C, const Sink* sink) {
  return sink ? bwd_row_grid(N, rows_per_block(C), 256, 2) : bwd_row_grid(N, rows_per_block(C), 128, 8);
}

#define AMT_DISPATCH_C(C_, ...)                       \
  switch (C_) {                                         \
    case 4: { constexpr int CC = 4; __VA_ARGS__; } break;     \
    case 8: { constexpr int CC = 8; __VA_ARGS__; } break;     \
    case 16: { constexpr int CC = 16; __VA_ARGS__; } break;   \
    case 32: { constexpr int CC = 32; __VA_ARGS__; } break;   \
    case 64: { constexpr int CC = 64; __VA_ARGS__; } break;   \
    case 128: { constexpr int CC = 128; __VA_ARGS__; } break; \
    case 256: { constexpr int CC = 256; __VA_ARGS__; } break; \
    default: break;                                     \
  }


constexpr float kLnEps = 1e-5f;   // eqx.nn.LayerNorm default eps

// LayerNorm statistics of a row spread over LPR lanes (VEC features each) with an explicit rounding
// sequence: the saved mean / rstd feed the backward pass in fp32, so the unfused kernels and the chain
// kernels must produce the same bits (a compiler-chosen fma contraction differs between kernels).
// On return x holds x - mean.
template <int VEC, int LPR>
AMT_DEV void ln_stats(float (&x)[VEC], float& mu, float& rs) {
  constexpr int C = VEC * LPR;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) s = __fadd_rn(s, x[i]);
  mu = __fmul_rn(group_sum<LPR>(s), 1.f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) { x[i] = __fsub_rn(x[i], mu); q = __fmaf_rn(x[i], x[i], q); }
  // v_rsq_f32 (1 ulp) instead of 1.f / sqrtf(): a correctly rounded square root followed by a correctly rounded division
  // is ~25 instructions per row pass, in every lane; every kernel takes its rstd from here, so the bits stay consistent
  rs = __builtin_amdgcn_rsqf(__fadd_rn(__fmul_rn(group_sum<LPR>(q), 1.f / C), kLnEps));
}
// (x - mean) * rstd * w + b
AMT_DEV float ln_apply(float d, float rs, float w, float b) { return __fmaf_rn(__fmul_rn(d, rs), w, b); }

}  // namespace amt
