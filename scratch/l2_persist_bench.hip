// Does a line that kernel A brought into an XCD's L2 still hit there in kernel B (next launch on the same stream)?
// K_warm: every workgroup reads the whole buffer W (each XCD's L2 gets its copy).  K_time: lane 0 of every workgroup
// chases 64 dependent pointers through W and reports cycles per load.  Variants: W warmed by the previous launch; W never
// touched in this run (HBM); W touched long ago (a 512 MiB sweep of another buffer in between: Infinity Cache or HBM).
// build: hipcc --offload-arch=gfx950 -O3 -o scratch/l2_persist_bench scratch/l2_persist_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ __launch_bounds__(256) void k_warm(const uint4* __restrict__ w, size_t n16, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = threadIdx.x; i < n16; i += 256) { uint4 v = w[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_sweep(const uint4* __restrict__ w, size_t n16, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { uint4 v = w[i]; acc ^= v.x ^ v.y; }
  if (acc == 0x12345678u) sink[0] = acc;
}
// W holds a pointer chain: element i (one per 4 KiB page... per 256 B) holds the index of the next element
__global__ __launch_bounds__(64) void k_time(const unsigned* __restrict__ w, unsigned start_stride, unsigned* out, int nload) {
  if (threadIdx.x == 0) {
    unsigned idx = (blockIdx.x * start_stride) & 0xfffu;   // < NE = 4096 elements
    // first hop untimed (TLB etc. would otherwise dominate)
    idx = __builtin_nontemporal_load(w + (size_t)idx * 64);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < nload; ++i) idx = w[(size_t)idx * 64];
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 2] = (unsigned)((t1 - t0) / nload);
    out[blockIdx.x * 2 + 1] = idx;
  }
}

int main() {
  const size_t NE = 4096;                 // chain elements, one per 256 B -> 1 MiB buffer
  const size_t bytes = NE * 256;
  unsigned *w, *w2, *out, *sink; uint4* big;
  CHECK(hipMalloc(&w, bytes)); CHECK(hipMalloc(&w2, bytes)); CHECK(hipMalloc(&out, 4096 * 8)); CHECK(hipMalloc(&sink, 64));
  CHECK(hipMalloc(&big, (size_t)512 << 20));
  CHECK(hipMemset(big, 1, (size_t)512 << 20));
  std::vector<unsigned> h(bytes / 4, 0);
  // random permutation cycle over NE elements
  std::vector<unsigned> perm(NE); for (size_t i = 0; i < NE; ++i) perm[i] = (unsigned)i;
  srand(1); for (size_t i = NE - 1; i > 0; --i) std::swap(perm[i], perm[rand() % (i + 1)]);
  for (size_t i = 0; i < NE; ++i) h[(size_t)perm[i] * 64] = perm[(i + 1) % NE];
  CHECK(hipMemcpy(w, h.data(), bytes, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(w2, h.data(), bytes, hipMemcpyHostToDevice));
  hipStream_t s; CHECK(hipStreamCreate(&s));
  std::vector<unsigned> ho(512);
  auto report = [&](const char* what) {
    CHECK(hipStreamSynchronize(s));
    CHECK(hipMemcpy(ho.data(), out, 256 * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned> c; for (int i = 0; i < 256; ++i) c.push_back(ho[2 * i]);
    std::sort(c.begin(), c.end());
    printf("%-70s cycles per dependent load: min %u  median %u  p90 %u  max %u\n", what, c[0], c[128], c[230], c[255]);
  };
  const int NL = 48;
  for (int rep = 0; rep < 2; ++rep) {
    // (a) warmed by the previous launch
    hipLaunchKernelGGL(k_warm, dim3(256), dim3(256), 0, s, (const uint4*)w, bytes / 16, sink);
    hipLaunchKernelGGL(k_time, dim3(256), dim3(64), 0, s, w, 97u, out, NL);
    report("(a) W read by every workgroup of the PREVIOUS launch");
    // (b) same launch twice: the second k_time after the first (which touched only a few lines per workgroup)
    hipLaunchKernelGGL(k_time, dim3(256), dim3(64), 0, s, w, 97u, out, NL);
    report("(b) k_time again (same 48 lines per workgroup as the launch before)");
    // (c) after sweeping 512 MiB of another buffer (L2 and Infinity Cache flushed of W)
    hipLaunchKernelGGL(k_sweep, dim3(2048), dim3(256), 0, s, (const uint4*)big, ((size_t)512 << 20) / 16, sink);
    hipLaunchKernelGGL(k_time, dim3(256), dim3(64), 0, s, w, 97u, out, NL);
    report("(c) after a 512 MiB sweep of another buffer (HBM)");
    // (d) warm, then an unrelated small kernel in between, then time
    hipLaunchKernelGGL(k_warm, dim3(256), dim3(256), 0, s, (const uint4*)w, bytes / 16, sink);
    hipLaunchKernelGGL(k_sweep, dim3(256), dim3(256), 0, s, (const uint4*)big, ((size_t)8 << 20) / 16, sink);
    hipLaunchKernelGGL(k_time, dim3(256), dim3(64), 0, s, w, 97u, out, NL);
    report("(d) warmed two launches ago, 8 MiB of other reads in between");
    // (e) in-kernel reference: warm and time in the SAME launch is what the chains do today (not measured here)
    // (f) a 38 MiB sweep in between: out of L2, inside the Infinity Cache
    hipLaunchKernelGGL(k_warm, dim3(256), dim3(256), 0, s, (const uint4*)w, bytes / 16, sink);
    hipLaunchKernelGGL(k_sweep, dim3(1024), dim3(256), 0, s, (const uint4*)big, ((size_t)64 << 20) / 16, sink);
    hipLaunchKernelGGL(k_time, dim3(256), dim3(64), 0, s, w, 97u, out, NL);
    report("(f) warmed, then a 64 MiB sweep (beyond the L2s, inside the Infinity Cache)");
  }
  printf("done\n");
  return 0;
}
