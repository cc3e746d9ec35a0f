// Stand-alone timing of the large-N chain kernels (csrc/chain_big.hip) with in-kernel phase stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DAMT_BIG_TIMING scratch/big_bench.hip -o scratch/big_bench
#include "../audio-to-midi_amd/csrc/chain_big.hip"
#include <cstdio>
#include <cstring>
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
  bf16* p; (void)hipMalloc(&p, n * 2); hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, p, n, seed, scale); return p;
}
static float* dev_f(size_t n, unsigned seed, float scale) {
  float* p; (void)hipMalloc(&p, n * 4); hipLaunchKernelGGL(fill_f_kernel, dim3(64), dim3(256), 0, 0, p, n, seed, scale); return p;
}
__global__ void checksum_kernel(const unsigned short* p, size_t n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc += (unsigned long long)p[i] * (unsigned long long)((i % 65521) + 1);
  atomicAdd(out, acc);
}
static unsigned long long checksum(const void* p, size_t n) {
  static unsigned long long* d = nullptr;
  if (!d) (void)hipMalloc(&d, 8);
  (void)hipMemset(d, 0, 8);
  hipLaunchKernelGGL(checksum_kernel, dim3(1024), dim3(256), 0, 0, (const unsigned short*)p, n, d);
  unsigned long long h = 0; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); return h;
}
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 112000;
  const int save = argc > 2 ? atoi(argv[2]) : 0;
  FfnFwdArgs a; memset(&a, 0, sizeof a);
  a.AO = dev_bf((size_t)N * 256, 1, 1.f); a.X = dev_bf((size_t)N * 256, 2, 1.f);
  a.Wo = dev_bf(256 * 256, 3, 0.06f); a.W1 = dev_bf(1024 * 256, 4, 0.06f); a.W2 = dev_bf(256 * 512, 5, 0.04f);
  a.ln_w = dev_f(256, 6, 1.f); a.ln_b = dev_f(256, 7, 0.1f); a.b1 = dev_f(1024, 8, 0.1f); a.b2 = dev_f(256, 9, 0.1f);
  a.Hout = dev_bf((size_t)N * 256, 10, 0.f); a.HN = dev_bf((size_t)N * 256, 11, 0.f); a.mean = dev_f(N, 12, 0.f); a.rstd = dev_f(N, 13, 0.f);
  a.U = dev_bf((size_t)N * 1024, 14, 0.f); a.G = dev_bf((size_t)N * 512, 15, 0.f); a.Y = dev_bf((size_t)N * 256, 16, 0.f);
  a.N = N; a.inv_keep = 1.f;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const char* modes = argc > 3 ? argv[3] : "1";
  for (const char* mp = modes; *mp; ++mp) {
  char mode_s[2] = {*mp, 0};
  setenv("AMT_BIG", mode_s, 1);
  (void)hipMemset(a.Y, 0, (size_t)N * 512);
  for (int rep = 0; rep < 3; ++rep) {
#ifdef AMT_BIG_TIMING
    static std::vector<unsigned long long> z(4096 * 16, 0);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(amt_big_stamps), z.data(), z.size() * 8);
#endif
    (void)hipEventRecord(e0, 0);
    ffn_fwd_big(0, DT_BF16, a, save != 0);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st(4096 * 16);
#ifdef AMT_BIG_TIMING
    (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(amt_big_stamps), st.size() * 8);
#endif
    const int tile = (*mp == '6') ? 32 : (*mp == '3' || *mp == '5') ? 64 : (*mp == '4') ? 128 : (N >= 49152 ? 128 : 64);
    const int nwg = (N + tile - 1) / tile < 4096 ? (N + tile - 1) / tile : 4096;
    double acc[16] = {0};
    for (int w = 0; w < nwg; ++w) for (int k = 0; k < 16; ++k) acc[k] += (double)st[(size_t)w * 16 + k];
    printf("mode %s ffn_fwd_big N=%d save=%d: %.1f us | mean cycles per tile: wait-in %.0f  Wo+h %.0f  LN %.0f | slices: w2-issue %.0f stage1 %.0f w1-issue+GLU %.0f barrier %.0f stage2 %.0f | y-epi %.0f copy-out %.0f | sum %.0f\n",
           mode_s, N, save, ms * 1e3, acc[0] / nwg, acc[1] / nwg, acc[2] / nwg, acc[3] / nwg, acc[4] / nwg, acc[5] / nwg, acc[6] / nwg, acc[7] / nwg,
           acc[8] / nwg, acc[9] / nwg, (acc[0]+acc[1]+acc[2]+acc[3]+acc[4]+acc[5]+acc[6]+acc[7]+acc[8]+acc[9]) / nwg);
  }
  printf("mode %s checksums: Y %016llx  U %016llx  G %016llx  H %016llx  HN %016llx\n", mode_s, checksum(a.Y, (size_t)N * 256),
         save ? checksum(a.U, (size_t)N * 1024) : 0ull, save ? checksum(a.G, (size_t)N * 512) : 0ull,
         save ? checksum(a.Hout, (size_t)N * 256) : 0ull, save ? checksum(a.HN, (size_t)N * 256) : 0ull);
  }
  return 0;
}
