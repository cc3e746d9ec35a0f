// FETCH_SIZE / WRITE_SIZE calibration for the access shapes of the weight-gradient workgroups.
//
// MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports HALF the bytes of wide coalesced 16-byte-per-lane reads, WRITE_SIZE is
// exact for 16-byte-per-lane stores and float atomics; "other access widths are uncalibrated".  The rider / grouped
// weight-gradient workgroups (csrc/wgrad.h) read 128-byte row segments of token panels (64 bf16 columns of rows that are
// 256 .. 1024 elements apart) and add 64-byte fp32 segments with global_atomic_add_f32.  Each kernel below moves a KNOWN
// number of bytes in one of those shapes, from / into buffers larger than the 256 MiB Infinity Cache where the shape allows;
// scratch/fetch_calib.sh runs them under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` and prints counter / known.
//
//   hipcc --offload-arch=gfx950 -O3 -o scratch/fetch_calib scratch/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// (1) control: contiguous 16-byte-per-lane streaming read of `bytes`
__global__ void k_stream16(const uint4* __restrict__ p, size_t n16, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = p[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x9e3779b9u) *sink = acc;
}
// (2) the operand shape: 128-byte segments (8 lanes x 16 B) of rows `row_bytes` apart; rows * 128 bytes in all
__global__ void k_panel128(const char* __restrict__ p, size_t rows, size_t row_bytes, unsigned* sink) {
  unsigned acc = 0;
  const size_t lanes = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * 8; i += lanes) {
    const uint4 v = *reinterpret_cast<const uint4*>(p + (i >> 3) * row_bytes + (i & 7) * 16);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x9e3779b9u) *sink = acc;
}
// (3) the result shape: every wave-instruction adds four 64-byte fp32 segments (16 lanes x 4 B) in four rows `row_floats`
// apart (one register of a 16x16 accumulator tile); `adders` workgroups add into the same footprint of `rows` rows x 64 floats
__global__ void k_atomic64(float* __restrict__ out, int rows, int row_floats, int adders_per_row_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int nblk = rows / 64;                                  // 64-row blocks of the footprint
  const int blk = blockIdx.x % nblk;
  for (int t = 0; t < 4; ++t)                                  // 4 waves x 4 tiles of 16 rows = 64 rows; 4 column tiles of 16
    for (int c = 0; c < 4; ++c)
      atomicAdd(out + (size_t)(blk * 64 + wave * 16 + 4 * t + g) * row_floats + 16 * c + r, 1.0f);
}
// (4) 16-byte-per-lane streaming store of `bytes` (control for WRITE_SIZE)
__global__ void k_store16(uint4* __restrict__ p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    p[i] = uint4{1u, 2u, 3u, 4u};
}

int main() {
  const size_t big = (size_t)1 << 30;                          // 1 GiB: past the Infinity Cache
  char* buf;
  unsigned* sink;
  CHECK(hipMalloc(&buf, big));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(buf, 1, big));
  CHECK(hipDeviceSynchronize());
  // (1) 1 GiB contiguous
  hipLaunchKernelGGL(k_stream16, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const uint4*>(buf), big / 16, sink);
  // (2) 128-byte segments at row strides 512 B (256 bf16), 1024 B, 2048 B: 256 / 128 / 64 MiB of segments
  for (size_t rb : {(size_t)512, (size_t)1024, (size_t)2048})
    hipLaunchKernelGGL(k_panel128, dim3(4096), dim3(256), 0, 0, buf, big / rb, rb, sink);      // every row of the 1-GiB buffer once
  // (3) atomics: footprint 65536 rows x 64 floats of 256-float rows (64 MiB span, 16 MiB touched), 8 adders per address:
  //     8 x 65536 x 64 x 4 B = 128 MiB added
  CHECK(hipMemset(buf, 0, (size_t)65536 * 256 * 4));
  CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_atomic64, dim3(65536 / 64 * 8), dim3(256), 0, 0, reinterpret_cast<float*>(buf), 65536, 256, 8);
  // (4) 1 GiB of 16-byte stores
  hipLaunchKernelGGL(k_store16, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint4*>(buf), big / 16);
  CHECK(hipDeviceSynchronize());
  printf("known bytes: stream16 %zu; panel128 %zu / %zu / %zu; atomic64 added %zu (footprint touched %zu); store16 %zu\n", big,
         big / 512 * 128, big / 1024 * 128, big / 2048 * 128, (size_t)8 * 65536 * 64 * 4, (size_t)65536 * 64 * 4, big);
  return 0;
}
