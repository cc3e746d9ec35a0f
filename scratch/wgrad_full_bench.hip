// Feasibility bench for a FULL-K weight-gradient GEMM: dW[m][k] = sum over ALL tokens n of G[n][m] A[n][k],
// one workgroup per 128 x TK output tile (TK = 128 or 64), no token split, no atomics, plain stores.  Four waves, each a
// 64 x TK/2 quadrant (4 x 4 or 4 x 2 MFMA 16x16x32 blocks per 32-token step); 64-token stages, two LDS buffers, one barrier per
// stage; both operands are read as COLUMNS of row-major [token][feature] LDS tiles (ds_read_b64_tr_b16).
//
//   hipcc --offload-arch=gfx950 -O3 -I audio-to-midi_amd/csrc -o /tmp/wgrad_full_bench scratch/wgrad_full_bench.hip
//   /tmp/wgrad_full_bench [tokens]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "common.h"

using namespace amt;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Prob { const bf16* G; const bf16* A; float* dW; int N, M, K, ldg, lda; };
struct Tile { int prob, m0, k0, tk; };

constexpr int TS = 64;            // tokens per stage
constexpr int LDT = 128 + 8;      // LDS row stride (elements): 272 bytes, rows shift by 4 banks

template <int TK>
__device__ __forceinline__ void tile_body(const Prob& p, int m0, int k0, unsigned char* smem) {
  constexpr int WK = TK / 2;                 // columns of A per wave
  constexpr int NB = WK / 16;                // B blocks per wave (4 or 2)
  constexpr int GCH = TS * 16 / 256;         // 16-byte chunks per thread per stage, G operand (128 columns): 4
  constexpr int ACH = TS * (TK / 8) / 256;   // A operand: 4 (TK = 128) or 2 (TK = 64)
  bf16* Gs = reinterpret_cast<bf16*>(smem);                    // [2][TS][LDT]
  bf16* As = Gs + 2 * TS * LDT;                                // [2][TS][LDT] (TK <= 128 columns used)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wk = wave & 1;
  f32x4 acc[4][NB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nst = (p.N + TS - 1) / TS;
  // two register sets: the loads of stages st + 1 and st + 2 are in flight while stage st is multiplied (with one set the loop
  // ran at one memory round trip per stage: 2.0 us per 64-token stage for 0.21 us of MFMA)
  Chunk8<bf16> gr[2][GCH], ar[2][ACH];
  auto load = [&](int st, Chunk8<bf16> (&g_)[GCH], Chunk8<bf16> (&a_)[ACH]) {
#pragma unroll
    for (int i = 0; i < GCH; ++i) {
      const int id = tid + 256 * i, row = id >> 4, c = id & 15;
      const int n = st * TS + row;
      const int nc = n < p.N ? n : p.N - 1;
      const int col = m0 + 8 * c;
      g_[i] = chunk_load(p.G + (size_t)nc * p.ldg + (col < p.M ? col : 0));
      if (n >= p.N || col >= p.M) g_[i] = chunk_zero<bf16>();
    }
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int id = tid + 256 * i, row = id / (TK / 8), c = id % (TK / 8);
      const int n = st * TS + row;
      const int nc = n < p.N ? n : p.N - 1;
      const int col = k0 + 8 * c;
      a_[i] = chunk_load(p.A + (size_t)nc * p.lda + (col < p.K ? col : 0));
      if (n >= p.N || col >= p.K) a_[i] = chunk_zero<bf16>();
    }
  };
  auto store = [&](int buf, const Chunk8<bf16> (&g_)[GCH], const Chunk8<bf16> (&a_)[ACH]) {
    bf16* gs = Gs + buf * TS * LDT;
    bf16* as = As + buf * TS * LDT;
#pragma unroll
    for (int i = 0; i < GCH; ++i) {
      const int id = tid + 256 * i, row = id >> 4, c = id & 15;
      chunk_store(gs + row * LDT + 8 * c, g_[i]);
    }
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int id = tid + 256 * i, row = id / (TK / 8), c = id % (TK / 8);
      chunk_store(as + row * LDT + 8 * c, a_[i]);
    }
  };
  auto compute = [&](int buf) {
    const bf16* gs = Gs + buf * TS * LDT;
    const bf16* as = As + buf * TS * LDT;
#pragma unroll
    for (int kk = 0; kk < TS / 32; ++kk) {
      Frag<bf16> gf[4], af[NB];
#pragma unroll
      for (int i = 0; i < 4; ++i) gf[i] = frag_col16(gs + (32 * kk + 8 * g) * LDT + wm * 64 + i * 16, LDT, r);
#pragma unroll
      for (int j = 0; j < NB; ++j) af[j] = frag_col16(as + (32 * kk + 8 * g) * LDT + wk * WK + j * 16, LDT, r);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) mma(acc[i][j], gf[i], af[j]);
    }
  };
  // LDS buffer st & 1 holds stage st; register set (st + 1) & 1 holds stage st + 1, set st & 1 stage st + 2
  load(0, gr[0], ar[0]);
  store(0, gr[0], ar[0]);
  if (nst > 1) load(1, gr[1], ar[1]);
  if (nst > 2) load(2, gr[0], ar[0]);
  __syncthreads();
  for (int st = 0; st < nst; st += 2) {
    // even stage st: set 1 holds st + 1, set 0 holds st + 2
    if (st + 1 < nst) store(1, gr[1], ar[1]);
    if (st + 3 < nst) load(st + 3, gr[1], ar[1]);
    compute(0);
    __syncthreads();
    if (st + 1 >= nst) break;
    // odd stage st + 1: set 0 holds st + 2, set 1 holds st + 3
    if (st + 2 < nst) store(0, gr[0], ar[0]);
    if (st + 4 < nst) load(st + 4, gr[0], ar[0]);
    compute(1);
    __syncthreads();
  }
  // acc[i][j][e] = dW[m0 + wm*64 + i*16 + 4g + e][k0 + wk*WK + j*16 + r]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int k = k0 + wk * WK + j * 16 + r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wm * 64 + i * 16 + 4 * g + e;
        if (m < p.M && k < p.K) p.dW[(size_t)m * p.K + k] += acc[i][j][e];
      }
    }
}

__global__ __launch_bounds__(256, 2) void wgrad_full_kernel(const Prob* probs, const Tile* tiles, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const Tile tl = tiles[t];
    if (tl.prob < 0) continue;
    const Prob p = probs[tl.prob];
    if (tl.tk == 128) tile_body<128>(p, tl.m0, tl.k0, smem);
    else tile_body<64>(p, tl.m0, tl.k0, smem);
    __syncthreads();
  }
}

__global__ void ref_kernel(Prob p, float* out) {      // one thread per output element (validation only)
  const int m = blockIdx.y * 16 + threadIdx.y, k = blockIdx.x * 16 + threadIdx.x;
  if (m >= p.M || k >= p.K) return;
  float s = 0.f;
  for (int n = 0; n < p.N; ++n) s += (float)p.G[(size_t)n * p.ldg + m] * (float)p.A[(size_t)n * p.lda + k];
  out[(size_t)m * p.K + k] = s;
}

__global__ void fill_kernel(bf16* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 0x9e3779b1u ^ seed;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    p[i] = (bf16)(((float)(x >> 8) / 16777216.f - 0.5f) * 0.25f);
  }
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 4000;
  const int layers = 16;
  // one transformer layer: (M, K) = G columns x A columns; + the C = 128 Block pair at twice the rows
  const int shapes[][3] = {{320, 256, 1}, {512, 64, 1}, {256, 256, 1}, {1024, 256, 1}, {256, 512, 1}};
  std::vector<Prob> probs;
  std::vector<Tile> tiles;
  double flops = 0;
  for (int l = 0; l < layers; ++l)
    for (auto& s : shapes) {
      Prob p;
      p.N = N * s[2]; p.M = s[0]; p.K = s[1]; p.ldg = s[0]; p.lda = s[1];
      bf16 *G, *A; float* dW;
      CHECK(hipMalloc(&G, (size_t)p.N * p.M * 2)); CHECK(hipMalloc(&A, (size_t)p.N * p.K * 2)); CHECK(hipMalloc(&dW, (size_t)p.M * p.K * 4));
      hipLaunchKernelGGL(fill_kernel, dim3(512), dim3(256), 0, 0, G, (size_t)p.N * p.M, 17u + (unsigned)probs.size());
      hipLaunchKernelGGL(fill_kernel, dim3(512), dim3(256), 0, 0, A, (size_t)p.N * p.K, 9001u + (unsigned)probs.size());
      CHECK(hipMemset(dW, 0, (size_t)p.M * p.K * 4));
      p.G = G; p.A = A; p.dW = dW;
      const int tk = p.K >= 128 ? 128 : 64;
      for (int m0 = 0; m0 < p.M; m0 += 128)
        for (int k0 = 0; k0 < p.K; k0 += tk) tiles.push_back(Tile{(int)probs.size(), m0, k0, tk});
      flops += 2.0 * p.N * p.M * p.K;
      probs.push_back(p);
    }
  // XCD-aware order: workgroup b runs on XCD b % 8 (round robin, observed), and the tiles of one problem re-read the same two
  // operand panels - a problem's tiles all go to ONE XCD (position % 8 == problem % 8), where they run side by side and find
  // each other's rows in that L2.  (In problem order over all XCDs the launch pulled 1.2 GB through the fabric: 4.5 TB/s.)
  if (!(argc > 2 && atoi(argv[2]) == 0)) {
    std::vector<std::vector<Tile>> per(8);
    for (auto& t : tiles) per[(t.prob / 5) % 8].push_back(t);      // (5 problems per layer: a layer's problems share an XCD)
    std::vector<Tile> out;
    for (size_t j = 0; out.size() < tiles.size(); ++j)
      for (int x = 0; x < 8; ++x)
        if (j < per[x].size()) out.push_back(per[x][j]);
        else out.push_back(Tile{-1, 0, 0, 0});                       // keep the positions aligned: an empty slot
    tiles.swap(out);
  }
  Prob* dprobs; Tile* dtiles;
  CHECK(hipMalloc(&dprobs, probs.size() * sizeof(Prob))); CHECK(hipMalloc(&dtiles, tiles.size() * sizeof(Tile)));
  CHECK(hipMemcpy(dprobs, probs.data(), probs.size() * sizeof(Prob), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dtiles, tiles.data(), tiles.size() * sizeof(Tile), hipMemcpyHostToDevice));
  const size_t lds = (size_t)4 * TS * LDT * 2;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_full_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int grid : {256, 512, (int)tiles.size()}) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(wgrad_full_kernel, dim3(grid), dim3(256), lds, 0, dprobs, dtiles, (int)tiles.size());
    CHECK(hipEventRecord(e0, 0));
    const int reps = 10;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(wgrad_full_kernel, dim3(grid), dim3(256), lds, 0, dprobs, dtiles, (int)tiles.size());
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("N = %d tokens, %zu problems, %zu tiles, grid %d: %.1f us per launch = %.0f TFLOP/s (%.2f GFLOP)\n", N, probs.size(), tiles.size(), grid,
           ms / reps * 1e3, flops / (ms / reps * 1e-3) / 1e12, flops / 1e9);
  }
  // validation on problem 0 (320 x 256: a partial M tile) and problem 1 (512 x 64: the 64-column tile)
  for (int pi : {0, 1}) {
    Prob p = probs[pi];
    CHECK(hipMemset(p.dW, 0, (size_t)p.M * p.K * 4));
    std::vector<Tile> one;
    for (auto& t : tiles) if (t.prob == pi) one.push_back(t);
    if (one.empty()) continue;
    CHECK(hipMemcpy(dtiles, one.data(), one.size() * sizeof(Tile), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(wgrad_full_kernel, dim3((int)one.size()), dim3(256), lds, 0, dprobs, dtiles, (int)one.size());
    float* ref; CHECK(hipMalloc(&ref, (size_t)p.M * p.K * 4));
    hipLaunchKernelGGL(ref_kernel, dim3((p.K + 15) / 16, (p.M + 15) / 16), dim3(16, 16), 0, 0, p, ref);
    std::vector<float> a((size_t)p.M * p.K), b((size_t)p.M * p.K);
    CHECK(hipMemcpy(a.data(), p.dW, a.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), ref, b.size() * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); ++i) { num += (a[i] - b[i]) * (a[i] - b[i]); den += b[i] * b[i]; }
    printf("problem %d (%d x %d): relative L2 error against the one-thread-per-element reference %.3e\n", pi, p.M, p.K, sqrt(num / den));
  }
  return 0;
}
