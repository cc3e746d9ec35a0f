// Micro-benchmark of the grouped weight-gradient kernel (transformer layer shapes) with phase stamps.
#include <hip/hip_runtime.h>
__device__ unsigned long long wg_stamps[2048][4];
#define TN_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 2048) wg_stamps[blockIdx.x][k] = wall_clock64(); } while (0)
#include "../audio-to-midi_amd/csrc/gemm.hip"
#include <cstdio>
#include <vector>
namespace amt {
Profiler& profiler() { static Profiler p; return p; }
int Profiler::class_id(const char*) { return 0; }
hipEvent_t Profiler::event() { return nullptr; }
}
using namespace amt;
__global__ void fill_kernel(bf16* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (bf16)(((float)(mix32((unsigned)i ^ seed) >> 8) / 8388608.f - 1.f));
}
static bf16* dev_bf(size_t n, unsigned seed) { bf16* p; (void)hipMalloc(&p, n * 2); hipLaunchKernelGGL(fill_kernel, dim3(512), dim3(256), 0, 0, p, n, seed); return p; }
int main() {
  const int N = 4000;
  struct P { int M, K; } ps[5] = {{256, 512}, {1024, 256}, {256, 256}, {512, 64}, {320, 256}};
  WgradArgs w[5];
  float* dW; (void)hipMalloc(&dW, 4 << 20); (void)hipMemset(dW, 0, 4 << 20);
  size_t off = 0;
  for (int i = 0; i < 5; ++i) {
    memset(&w[i], 0, sizeof(w[i]));
    w[i].G = dev_bf((size_t)N * ps[i].M, i); w[i].ldg = ps[i].M; w[i].A = dev_bf((size_t)N * ps[i].K, 10 + i); w[i].lda = ps[i].K;
    w[i].N = N; w[i].M = ps[i].M; w[i].Kp = ps[i].K; w[i].dW = dW + off; w[i].ldw = ps[i].K; off += (size_t)ps[i].M * ps[i].K;
  }
  void* thrash; (void)hipMalloc(&thrash, (size_t)1 << 30);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    float tot = 0;
    for (int rep = 0; rep < 6; ++rep) {
      if (mode) (void)hipMemsetAsync(thrash, rep, (size_t)1 << 30, 0);
      (void)hipEventRecord(e0, 0);
      gemm_tn_wgrad_group(0, DT_BF16, w, 5, nullptr);
      (void)hipEventRecord(e1, 0);
      (void)hipDeviceSynchronize();
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep) tot += ms;
    }
    std::vector<unsigned long long> st(2048 * 4);
    (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(wg_stamps), st.size() * 8);
    unsigned long long t0 = ~0ull, t1 = 0; double loop = 0, epi = 0; int n = 0;
    for (int b = 0; b < 2048; ++b) { const unsigned long long* q = &st[b * 4]; if (!q[0] || q[2] < q[0]) continue; ++n; if (q[0] < t0) t0 = q[0]; if (q[2] > t1) t1 = q[2]; loop += (q[1] - q[0]) / 100.0; epi += (q[2] - q[1]) / 100.0; }
    printf("%s: %.2f us/launch | %d WGs: loop %.2f us, epilogue %.2f us avg; first start -> last end %.2f us\n", mode ? "cold" : "hot ", tot * 1000 / 5, n, loop / n, epi / n, (t1 - t0) / 100.0);
    (void)hipMemset(HIP_SYMBOL(wg_stamps), 0, 0);
  }
  return 0;
}
