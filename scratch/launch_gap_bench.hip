// What does a dependent kernel boundary cost on MI355X, and what does it depend on?  (round 2)
// A chain of 64 graph-captured launches of a kernel that does almost nothing; varied one at a time: grid size,
// dynamic LDS per workgroup, registers per lane, bytes stored per workgroup, kernel-argument bytes.
// build: hipcc --offload-arch=gfx950 -O3 -o scratch/launch_gap_bench scratch/launch_gap_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

struct Args { float* out; const float* in; int store_f4_per_thread; int pad[509]; };   // ~2 KiB
struct SmallArgs { float* out; const float* in; int store_f4_per_thread; };

template <int NREG, typename A, int STORE = 0>
__global__ __launch_bounds__(256) void k(A a) {
  extern __shared__ float lds[];
  float r[NREG];
  const float x = a.in[threadIdx.x & 63];
#pragma unroll
  for (int i = 0; i < NREG; ++i) r[i] = x * (float)(i + 1);
  // keep NREG values live across a barrier
  if (threadIdx.x == 0) lds[0] = x;
  __syncthreads();
  float s = lds[0];
#pragma unroll
  for (int i = 0; i < NREG; ++i) s += r[i] * r[(i * 7 + 3) % NREG];
  float4 v = {s, s, s, s};
  float4* o = reinterpret_cast<float4*>(a.out) + ((size_t)blockIdx.x * 256 + threadIdx.x) * (size_t)(a.store_f4_per_thread > 0 ? a.store_f4_per_thread : 1);
  if (STORE == 0) {
    for (int i = 0; i < a.store_f4_per_thread; ++i) o[i] = v;
  } else if (STORE == 1) {   // write-through (sc1): the bytes leave the L2 while the kernel runs
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)o, 0, 0x7fffffff, 0x00020000);
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    u4 w = {__builtin_bit_cast(unsigned, s), 1u, 2u, 3u};
    for (int i = 0; i < a.store_f4_per_thread; ++i) __builtin_amdgcn_raw_buffer_store_b128(w, rs, i * 16, 0, 16);
  } else if (STORE == 2) {   // non-temporal
    typedef __attribute__((ext_vector_type(4))) float f4v;
    f4v w = {s, s, s, s};
    for (int i = 0; i < a.store_f4_per_thread; ++i) __builtin_nontemporal_store(w, reinterpret_cast<f4v*>(o) + i);
  } else {                   // sc0 sc1 (system scope)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)o, 0, 0x7fffffff, 0x00020000);
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    u4 w = {__builtin_bit_cast(unsigned, s), 1u, 2u, 3u};
    for (int i = 0; i < a.store_f4_per_thread; ++i) __builtin_amdgcn_raw_buffer_store_b128(w, rs, i * 16, 0, 17);
  }
  if (a.store_f4_per_thread == 0 && s == 12345.678f) o[0] = v;
}

template <typename F>
static float chain_us(F launch, int n = 64, int reps = 20) {
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipGraph_t g;
  hipGraphExec_t ge;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < n; ++i) launch(s);
  CHECK(hipStreamEndCapture(s, &g));
  CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CHECK(hipGraphLaunch(ge, s));
  CHECK(hipStreamSynchronize(s));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) CHECK(hipGraphLaunch(ge, s));
  CHECK(hipEventRecord(e1, s));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g)); CHECK(hipStreamDestroy(s));
  return ms * 1e3f / reps / n;
}

template <int NREG, typename A, int STORE = 0>
static float run(int grid, size_t lds, int store_f4, float* out, const float* in) {
  if (lds > 64 * 1024) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<NREG, A, STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  A a{};
  a.out = out; a.in = in; a.store_f4_per_thread = store_f4;
  return chain_us([&](hipStream_t s) { hipLaunchKernelGGL((k<NREG, A, STORE>), dim3(grid), dim3(256), lds, s, a); });
}

int main() {
  float *out, *in;
  CHECK(hipMalloc(&out, (size_t)1024 * 256 * 64 * 16));   // 256 MiB
  CHECK(hipMalloc(&in, 4096));
  CHECK(hipMemset(in, 0, 4096));
  printf("dependent launch period, graph replay of 64 launches (us per launch)\n");
  printf("baseline: grid 256, 256 B LDS, 8 registers, no stores, small args: %.2f\n", run<8, SmallArgs>(256, 256, 0, out, in));
  for (int grid : {64, 128, 256, 512, 1024, 2048}) printf("grid %4d: %.2f\n", grid, run<8, SmallArgs>(grid, 256, 0, out, in));
  for (size_t lds : {(size_t)256, (size_t)16 << 10, (size_t)32 << 10, (size_t)64 << 10, (size_t)100 << 10, (size_t)150 << 10})
    printf("LDS %6zu B (grid 256): %.2f   (grid 512): %.2f\n", lds, run<8, SmallArgs>(256, lds, 0, out, in), run<8, SmallArgs>(512, lds, 0, out, in));
  printf("registers   8: %.2f\n", run<8, SmallArgs>(256, 256, 0, out, in));
  printf("registers  64: %.2f\n", run<64, SmallArgs>(256, 256, 0, out, in));
  printf("registers 120: %.2f\n", run<120, SmallArgs>(256, 256, 0, out, in));
  printf("registers 240: %.2f\n", run<240, SmallArgs>(256, 256, 0, out, in));
  printf("registers 240, grid 512: %.2f\n", run<240, SmallArgs>(512, 256, 0, out, in));
  for (int st : {0, 1, 4, 16, 64}) printf("stores %5d B per workgroup (%.1f MB per launch): %.2f\n", st * 16 * 256, st * 16.0 * 256 * 256 / 1e6, run<8, SmallArgs>(256, 256, st, out, in));
  for (int st : {4, 16, 64}) {
    printf("stores %.1f MB per launch: plain %.2f | sc1 (write-through) %.2f | nt %.2f | sc0 sc1 %.2f\n", st * 16.0 * 256 * 256 / 1e6,
           run<8, SmallArgs, 0>(256, 256, st, out, in), run<8, SmallArgs, 1>(256, 256, st, out, in), run<8, SmallArgs, 2>(256, 256, st, out, in),
           run<8, SmallArgs, 3>(256, 256, st, out, in));
  }
  for (int st : {2, 8, 32}) {
    printf("grid 512, stores %.1f MB per launch: plain %.2f | sc1 %.2f | nt %.2f\n", st * 16.0 * 256 * 512 / 1e6,
           run<8, SmallArgs, 0>(512, 256, st, out, in), run<8, SmallArgs, 1>(512, 256, st, out, in), run<8, SmallArgs, 2>(512, 256, st, out, in));
  }
  printf("2 KiB of kernel arguments: %.2f\n", run<8, Args>(256, 256, 0, out, in));
  printf("like a chain kernel (grid 512, 30 KiB LDS, 120 registers, 16 KiB stores per workgroup, 2 KiB args): %.2f\n", run<120, Args>(512, 30 << 10, 4, out, in));
  printf("done\n");
  return 0;
}
