// Micro-benchmark of the chain kernels with in-kernel phase timestamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics scratch/chain_bench.hip -o scratch/chain_bench
// mode "hot": the same launch back to back.  mode "chain": NB blocks with distinct weights and
// buffers, Y of one is X of the next, caches thrashed (1 GiB memset) before every pass = the state
// a launch finds inside the training step.
#define AMT_CHAIN_TIMING 1
#include "../audio-to-midi_amd/csrc/chain.hip"
#include "../audio-to-midi_amd/csrc/gemm.hip"
#include "../audio-to-midi_amd/csrc/cnn.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace amt {
Profiler& profiler() { static Profiler p; return p; }
int Profiler::class_id(const char*) { return 0; }
hipEvent_t Profiler::event() { return nullptr; }
}
using namespace amt;

__global__ void fill_kernel(bf16* p, size_t n, unsigned seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (bf16)(scale * ((float)(mix32((unsigned)i ^ seed) >> 8) / 8388608.f - 1.f));
}
__global__ void fill_f_kernel(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = scale * ((float)(mix32((unsigned)i ^ seed) >> 8) / 8388608.f - 1.f);
}
static bf16* dev_bf(size_t n, unsigned seed, float scale) {
  bf16* p; (void)hipMalloc(&p, n * 2); hipLaunchKernelGGL(fill_kernel, dim3(512), dim3(256), 0, 0, p, n, seed, scale); return p;
}
static float* dev_f(size_t n, unsigned seed, float scale) {
  float* p; (void)hipMalloc(&p, n * 4); hipLaunchKernelGGL(fill_f_kernel, dim3(64), dim3(256), 0, 0, p, n, seed, scale); return p;
}

static void report(const char* what, int slot0, int slot1, int nwg, int nst, float us) {
  static std::vector<unsigned long long> st(64 * 256 * 8);
  (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(amt_stamps), st.size() * 8);
  double acc[8] = {0}, span = 0;
  if (nwg > 256) nwg = 256;
  for (int s = slot0; s < slot1; ++s) {
    unsigned long long first = ~0ull, last = 0;
    for (int w = 0; w < nwg; ++w) {
      const unsigned long long* q = &st[((size_t)s * 256 + w) * 8];
      for (int k = 1; k < nst; ++k) acc[k] += (double)(q[k] - q[k - 1]);
      if (q[0] < first) first = q[0];
      if (q[nst - 1] > last) last = q[nst - 1];
    }
    span += (double)(last - first);
  }
  const double n = (double)nwg * (slot1 - slot0);
  printf("%-30s %7.2f us/launch | cycles/phase:", what, us);
  for (int k = 1; k < nst; ++k) printf(" %6.0f", acc[k] / n);
  printf(" | first WG start -> last WG end %6.0f cyc (100 MHz ticks? see below)\n", span / (slot1 - slot0));
}

static void unfused(const BlockFwdArgs& a) {
  const int C = a.C, N = a.B * a.L;
  dwln_fwd(0, DT_BF16, a.X, a.wdw, a.bdw, a.lnw, a.lnb, a.LN, a.mean, a.rstd, a.B, a.L, C);
  GemmArgs g1; memset(&g1, 0, sizeof g1);
  g1.X = a.LN; g1.ldx = C; g1.W = a.W1; g1.ldw = C; g1.N = N; g1.M = 2 * C; g1.K = C; g1.bias = a.b1; g1.Y = a.U; g1.ldy = 2 * C;
  g1.Y2 = a.Hh; g1.ldy2 = 2 * C; g1.inv_keep = 1.f;
  gemm_nt(0, DT_BF16, EPI_GELU, g1);
  GemmArgs g2; memset(&g2, 0, sizeof g2);
  g2.X = a.Hh; g2.ldx = 2 * C; g2.W = a.W2; g2.ldw = 2 * C; g2.N = N; g2.M = C; g2.K = 2 * C; g2.bias = a.b2; g2.Y = a.Y; g2.ldy = C;
  g2.Y2 = a.Z; g2.ldy2 = C; g2.R = a.X; g2.ldr = C; g2.gamma = a.gamma; g2.inv_keep = 1.f;
  gemm_nt(0, DT_BF16, EPI_SCALE_RESID, g2);
}

static char* g_arena = nullptr; static size_t g_off = 0;
static void* carve(size_t bytes) { size_t o = (g_off + 255) & ~(size_t)255; g_off = o + bytes; return g_arena + o; }
static BlockFwdArgs make_block_arena(int B, int L, int C, unsigned seed, const void* X) {
  size_t N = (size_t)B * L;
  BlockFwdArgs a;
  a.X = X ? X : dev_bf(N * C, seed + 1, 1.f);
  a.wdw = dev_f(7 * C, seed + 2, 0.3f); a.bdw = dev_f(C, seed + 3, 0.1f); a.lnw = dev_f(C, seed + 4, 1.f);
  a.lnb = dev_f(C, seed + 5, 0.1f); a.W1 = dev_bf((size_t)2 * C * C, seed + 6, 0.1f); a.b1 = dev_f(2 * C, seed + 7, 0.1f);
  a.W2 = dev_bf((size_t)2 * C * C, seed + 8, 0.1f); a.b2 = dev_f(C, seed + 9, 0.1f); a.gamma = dev_f(C, seed + 10, 0.1f);
  a.LN = carve(N * C * 2); a.mean = (float*)carve(N * 4); a.rstd = (float*)carve(N * 4); a.U = carve(N * 2 * C * 2);
  a.Hh = carve(N * 2 * C * 2); a.Z = carve(N * C * 2); a.Y = carve(N * C * 2);
  a.B = B; a.L = L; a.C = C; a.dbg_slot = 0;
  return a;
}
static void gaps(const char* what, int s0, int s1) {
  static std::vector<unsigned long long> st(64 * 256 * 8);
  (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(amt_stamps), st.size() * 8);
  double g = 0, e = 0; int n = 0;
  unsigned long long prev_end = 0;
  for (int s = s0; s < s1; ++s) {
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < 250; ++w) { const unsigned long long* q = &st[((size_t)s * 256 + w) * 8]; if (q[6] < t0) t0 = q[6]; if (q[7] > t1) t1 = q[7]; }
    if (s > s0) { g += (double)(t0 - prev_end) / 100.0; ++n; }
    e += (double)(t1 - t0) / 100.0;
    prev_end = t1;
  }
  printf("%-30s exec %.2f us, gap %.2f us\n", what, e / (s1 - s0), g / n);
}
static BlockFwdArgs make_block(int B, int L, int C, unsigned seed, const void* X) {
  size_t N = (size_t)B * L;
  BlockFwdArgs a;
  a.X = X ? X : dev_bf(N * C, seed + 1, 1.f);
  a.wdw = dev_f(7 * C, seed + 2, 0.3f); a.bdw = dev_f(C, seed + 3, 0.1f); a.lnw = dev_f(C, seed + 4, 1.f);
  a.lnb = dev_f(C, seed + 5, 0.1f); a.W1 = dev_bf((size_t)2 * C * C, seed + 6, 0.1f); a.b1 = dev_f(2 * C, seed + 7, 0.1f);
  a.W2 = dev_bf((size_t)2 * C * C, seed + 8, 0.1f); a.b2 = dev_f(C, seed + 9, 0.1f); a.gamma = dev_f(C, seed + 10, 0.1f);
  a.LN = dev_bf(N * C, 0, 0); a.mean = dev_f(N, 0, 0); a.rstd = dev_f(N, 0, 0); a.U = dev_bf(N * 2 * C, 0, 0);
  a.Hh = dev_bf(N * 2 * C, 0, 0); a.Z = dev_bf(N * C, 0, 0); a.Y = dev_bf(N * C, 0, 0);
  a.B = B; a.L = L; a.C = C; a.dbg_slot = 0;
  return a;
}

extern "C" int bench_main(int argc, char** argv) {
  const int B = 8;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int Cs[5] = {16, 32, 64, 128, 256};
  for (int ci = 0; ci < 5; ++ci) {
    int C = Cs[ci], L = 64000 / C;
    BlockFwdArgs a = make_block(B, L, C, 100 * ci, nullptr);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < 5; ++i) block_fwd_fused(0, DT_BF16, a);
    (void)hipEventRecord(e0, 0);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) { a.dbg_slot = i; block_fwd_fused(0, DT_BF16, a); }
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    char nm[64]; snprintf(nm, sizeof nm, "hot   C=%d N=%d", C, B * L);
    int TT = C <= 16 ? 128 : (C <= 32 ? 64 : (C <= 64 ? 32 : 16));
    report(nm, 10, 50, (B * L + TT - 1) / TT, 6, ms * 1000.f / reps);
    for (int i = 0; i < 5; ++i) unfused(a);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) unfused(a);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("      unfused 3 launches      %7.2f us\n", ms * 1000.f / reps);
  }
  // chain of 21 C=128 blocks, cold caches
  {
    const int C = 128, L = 500, NB = 21;
    std::vector<BlockFwdArgs> blk;
    const void* x = nullptr;
    for (int i = 0; i < NB; ++i) { blk.push_back(make_block(B, L, C, 1000 + 50 * i, x)); x = blk.back().Y; }
    void* thrash; (void)hipMalloc(&thrash, (size_t)1 << 30);
    float tot = 0;
    const int reps = 5;
    for (int rep = 0; rep < reps; ++rep) {
      (void)hipMemsetAsync(thrash, rep, (size_t)1 << 30, 0);
      (void)hipEventRecord(e0, 0);
      for (int i = 0; i < NB; ++i) { blk[i].dbg_slot = i; block_fwd_fused(0, DT_BF16, blk[i]); }
      (void)hipEventRecord(e1, 0);
      (void)hipDeviceSynchronize();
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      tot += ms;
    }
    report("chain C=128 x21 cold", 1, NB, 250, 6, tot * 1000.f / (reps * NB));
    tot = 0;
    for (int rep = 0; rep < reps; ++rep) {
      (void)hipMemsetAsync(thrash, rep, (size_t)1 << 30, 0);
      (void)hipEventRecord(e0, 0);
      for (int i = 0; i < NB; ++i) unfused(blk[i]);
      (void)hipEventRecord(e1, 0);
      (void)hipDeviceSynchronize();
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      tot += ms;
    }
    printf("      unfused chain cold      %7.2f us per block\n", tot * 1000.f / (reps * NB));
    // same chain, stamps -> gaps; then with every output carved from one 1.5 GiB arena
    (void)hipMemsetAsync(thrash, 1, (size_t)1 << 30, 0);
    for (int i = 0; i < NB; ++i) { blk[i].dbg_slot = i; block_fwd_fused(0, DT_BF16, blk[i]); }
    (void)hipDeviceSynchronize();
    gaps("separate allocations", 1, NB);
    (void)hipMalloc(&g_arena, (size_t)3 << 29);
    g_off = (size_t)600 << 20;
    std::vector<BlockFwdArgs> ab;
    x = nullptr;
    for (int i = 0; i < NB; ++i) { ab.push_back(make_block_arena(B, L, C, 5000 + 50 * i, x)); x = ab.back().Y; }
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipMemsetAsync(thrash, 1, (size_t)1 << 30, 0);
      for (int i = 0; i < NB; ++i) { ab[i].dbg_slot = i; block_fwd_fused(0, DT_BF16, ab[i]); }
      (void)hipDeviceSynchronize();
      gaps("one arena", 1, NB);
    }
  }
  return 0;
}
int main(int argc, char** argv) { return bench_main(argc, argv); }
