// What ONE compute unit can pull from a warm L2 when every compute unit of the chip streams the SAME weight range
// (the access pattern of the token-tile chain kernels: every workgroup reads all of a layer's packed weight fragments).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/l2_stream_bench.hip -o scratch/l2_stream_bench
// Variants: waves per workgroup (4 = one per SIMD, 8 = two), loads in flight per wave (registers), register loads
// (global_load_dwordx4, 1 KiB per wave instruction) against LDS-DMA (global_load_lds_dwordx4).  VERDICT r2 item 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// every workgroup streams `bytes` (a multiple of waves * DEPTH KiB) from `w`, `reps` times; each wave its own 1/waves of it
template <int DEPTH>
__global__ void stream_regs(const u32x4* __restrict__ w, size_t bytes, int reps, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const size_t frags = bytes / 1024;                 // 1-KiB fragments
  unsigned acc = 0;
  for (int r = 0; r < reps; ++r) {
    for (size_t f = wave * DEPTH; f + DEPTH <= frags; f += (size_t)nw * DEPTH) {
      u32x4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) v[d] = w[(f + d) * 64 + lane];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc ^= v[d].x ^ v[d].w;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int DEPTH>
__global__ void stream_lds(const u32x4* __restrict__ w, size_t bytes, int reps, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef const __attribute__((address_space(1))) void* GlobalPtr;
  typedef __attribute__((address_space(3))) void* LdsPtr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const size_t frags = bytes / 1024;
  unsigned char* mine = smem + (size_t)wave * DEPTH * 1024;
  for (int r = 0; r < reps; ++r) {
    for (size_t f = wave * DEPTH; f + DEPTH <= frags; f += (size_t)nw * DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
        __builtin_amdgcn_global_load_lds((GlobalPtr)(w + (f + d) * 64 + lane), (LdsPtr)(mine + d * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  if (reps < 0) sink[0] = mine[lane];
}

template <typename K>
static void run(const char* name, K kernel, int threads, size_t lds, const u32x4* w, size_t bytes, int reps, unsigned* sink, int grid) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, 0, w, bytes, 2, sink);      // warm L2
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int it = 0; it < 5; ++it) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, 0, w, bytes, reps, sink);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double per_cu = (double)bytes * reps / (best * 1e-3);        // bytes per second per workgroup (= per CU at grid 256)
  printf("%-44s grid %4d  %7.1f us  %6.1f GB/s per workgroup  %6.2f TB/s chip  (%.1f B/clk at 2.4 GHz)\n", name, grid, best * 1e3,
         per_cu / 1e9, per_cu * grid / 1e12, per_cu / 2.4e9);
}

int main() {
  const size_t sizes[] = {256 * 1024, 896 * 1024};      // a Block's two 1x1 weights at C = 128 (bf16); the FFN chain's weights
  u32x4* w; unsigned* sink;
  (void)hipMalloc(&w, 1 << 20); (void)hipMemset(w, 1, 1 << 20); (void)hipMalloc(&sink, 64);
  for (size_t bytes : sizes) {
    const int reps = (int)((64u << 20) / bytes);       // ~64 MiB per workgroup
    printf("---- every workgroup streams the same %zu KiB, %d times\n", bytes / 1024, reps);
    for (int grid : {256, 512}) {
      run("regs, 4 waves, 4 KiB in flight per wave", stream_regs<4>, 256, 0, w, bytes, reps, sink, grid);
      run("regs, 4 waves, 16 KiB in flight per wave", stream_regs<16>, 256, 0, w, bytes, reps, sink, grid);
      run("regs, 4 waves, 32 KiB in flight per wave", stream_regs<32>, 256, 0, w, bytes, reps, sink, grid);
      run("regs, 8 waves, 8 KiB in flight per wave", stream_regs<8>, 512, 0, w, bytes, reps, sink, grid);
      run("regs, 8 waves, 16 KiB in flight per wave", stream_regs<16>, 512, 0, w, bytes, reps, sink, grid);
      run("regs, 16 waves, 8 KiB in flight per wave", stream_regs<8>, 1024, 0, w, bytes, reps, sink, grid);
      run("LDS-DMA, 4 waves, 16 KiB in flight per wave", stream_lds<16>, 256, 4 * 16 * 1024, w, bytes, reps, sink, grid);
      run("LDS-DMA, 8 waves, 8 KiB in flight per wave", stream_lds<8>, 512, 8 * 8 * 1024, w, bytes, reps, sink, grid);
    }
  }
  return 0;
}
