// XCD-team micro-benchmark (round 2): can the workgroups that sit on ONE XCD (shared 4 MiB L2) hand
// data to each other inside a launch with plain stores + a vmcnt drain + an L2-served flag, and what
// does such a team barrier cost against a kernel boundary?
//
//   census      which XCC / CU every workgroup of a 256-block grid lands on
//   barrier     time per team barrier (atomic-counter form, flag-line form), teams formed by XCC_ID
//   pingpong    P dependent phases of a 7-tap column stencil over 8 "clips" of 512 rows x 128 words,
//               one clip per team, 16 rows per workgroup, halo rows come from the neighbours:
//               checks every word against the host, L1-warm (the same lines are re-read every second
//               phase), optional uneven load; loads plain or sc1; teams by XCC_ID or by blockIdx / 32
//   launches    the same P phases as P graph-captured launches (the boundary this would replace)
//
// build: hipcc --offload-arch=gfx950 -O3 -o scratch/xcd_team_bench scratch/xcd_team_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int MAXT = 8, LINE = 32;   // 32 words = 128 B
constexpr unsigned SPIN_LIMIT = 1u << 22;

struct TeamState {
  unsigned registered[LINE];
  unsigned members[MAXT][LINE];
  unsigned bar[MAXT][LINE];
  unsigned flags[MAXT][2 * LINE];   // one word per member (<= 64 members)
  unsigned timeout[LINE];
};

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 15;
}
__device__ __forceinline__ unsigned hw_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(x));
  return x;
}
__device__ __forceinline__ unsigned long long rt_clock() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz

struct Team { int id, rank, size; bool ok; };

// every workgroup: team = its XCC (BYXCC) or blockIdx / 32; rank by arrival; one grid-wide rendezvous so that
// every member knows the team's size.  Bounded spin.
template <bool BYXCC>
__device__ Team team_form(TeamState* st) {
  __shared__ int s_id, s_rank, s_size, s_ok;
  if (threadIdx.x == 0) {
    const int id = BYXCC ? (int)xcc_id() % MAXT : (int)(blockIdx.x / 32) % MAXT;
    const int rank = (int)__hip_atomic_fetch_add(&st->members[id][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    drain();   // the membership add has returned before the arrival is counted
    __hip_atomic_fetch_add(&st->registered[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    bool ok = true;
    while (ld_sc1(&st->registered[0]) < gridDim.x) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > SPIN_LIMIT || ld_sc1(&st->timeout[0])) { st->timeout[0] = 1; ok = false; break; }
    }
    s_id = id; s_rank = rank; s_size = (int)ld_sc1(&st->members[id][0]); s_ok = ok;
  }
  __syncthreads();
  Team t{s_id, s_rank, s_size, s_ok != 0};
  return t;
}

// FORM 0: one counter per team (agent-scope atomic add, sc1 poll).  FORM 1: one flag word per member in a line of the
// team (plain store of the epoch, drained; lanes of wave 0 poll the line with sc1 loads).
template <int FORM>
__device__ bool team_barrier(TeamState* st, const Team& t, unsigned epoch) {
  drain();            // every storing wave: its stores have reached L2
  __syncthreads();
  __shared__ int s_ok;
  if (threadIdx.x < 64) {
    bool ok = true;
    if (FORM == 0) {
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&st->bar[t.id][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (ld_sc1(&st->bar[t.id][0]) < epoch * (unsigned)t.size) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > SPIN_LIMIT) { st->timeout[0] = 2; ok = false; break; }
        }
        s_ok = ok;
      }
    } else {
      if (threadIdx.x == 0) {
        __hip_atomic_store(&st->flags[t.id][t.rank], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // plain store
        drain();
      }
      unsigned spins = 0;
      for (;;) {
        const unsigned v = (int)threadIdx.x < t.size ? ld_sc1(&st->flags[t.id][threadIdx.x]) : epoch;
        if (__all((int)(v - epoch) >= 0)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) { if (threadIdx.x == 0) st->timeout[0] = 3; ok = false; break; }
      }
      if (threadIdx.x == 0) s_ok = ok;
    }
  }
  __syncthreads();
  return s_ok != 0;
}

__global__ __launch_bounds__(256) void census_kernel(unsigned* out) {
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc_id(); out[2 * blockIdx.x + 1] = hw_id(); }
}

template <int FORM, bool BYXCC>
__global__ __launch_bounds__(256) void barrier_kernel(TeamState* st, int nbar, unsigned long long* ticks) {
  const Team t = team_form<BYXCC>(st);
  if (!t.ok) return;
  team_barrier<FORM>(st, t, 1);
  const unsigned long long t0 = rt_clock();
  for (int i = 0; i < nbar; ++i)
    if (!team_barrier<FORM>(st, t, 2 + i)) return;
  const unsigned long long t1 = rt_clock();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------ ping-pong stencil
constexpr int ROWS = 512, COLS = 128, RPW = 16;   // rows per clip, words per row, rows per workgroup

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// one phase of one workgroup: rows [r0, r0 + RPW) of clip `clip`:  dst[l][c] = sum_{k=-3..3} src[l + k][c] + phase
template <int LOADMODE>
__device__ __forceinline__ void stencil_rows(const unsigned* src, unsigned* dst, int clip, int r0, unsigned phase) {
  const unsigned* base = src + (size_t)clip * ROWS * COLS;
  unsigned* out = dst + (size_t)clip * ROWS * COLS;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, ROWS * COLS * 4, 0x00020000);
  // 256 threads: 16 rows x 32 chunks of 4 words -> 2 passes of 8 rows
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int row = r0 + pass * 8 + (threadIdx.x >> 5);
    const int c4 = (threadIdx.x & 31) * 4;
    u32x4 acc = {phase, phase, phase, phase};
#pragma unroll
    for (int k = -3; k <= 3; ++k) {
      const int l = row + k;
      const int lc = l < 0 ? 0 : (l >= ROWS ? ROWS - 1 : l);
      u32x4 v;
      if (LOADMODE == 0) v = *reinterpret_cast<const u32x4*>(base + (size_t)lc * COLS + c4);
      else v = __builtin_amdgcn_raw_buffer_load_b128(rs, (lc * COLS + c4) * 4, 0, 16);
      if (l < 0 || l >= ROWS) v = u32x4{0, 0, 0, 0};
      acc += v;
    }
    *reinterpret_cast<u32x4*>(out + (size_t)row * COLS + c4) = acc;
  }
}

template <int FORM, bool BYXCC, int LOADMODE, bool UNEVEN>
__global__ __launch_bounds__(256) void pingpong_kernel(TeamState* st, unsigned* buf0, unsigned* buf1, int phases,
                                                       unsigned long long* ticks) {
  const Team t = team_form<BYXCC>(st);
  if (!t.ok) return;
  const unsigned long long t0 = rt_clock();
  for (int p = 0; p < phases; ++p) {
    const unsigned* src = (p & 1) ? buf1 : buf0;
    unsigned* dst = (p & 1) ? buf0 : buf1;
    if (UNEVEN) {
      const unsigned n = hash32(blockIdx.x * 977u + p) & 63;
      for (unsigned i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
    }
    for (int w = t.rank; w < ROWS / RPW; w += t.size) stencil_rows<LOADMODE>(src, dst, t.id, w * RPW, (unsigned)p);
    if (!team_barrier<FORM>(st, t, 1 + p)) return;
  }
  const unsigned long long t1 = rt_clock();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(256) void phase_kernel(const unsigned* src, unsigned* dst, unsigned phase) {
  const int clip = blockIdx.x % 8, w = blockIdx.x / 8;
  stencil_rows<0>(src, dst, clip, w * RPW, phase);
}

static void host_ref(std::vector<unsigned>& a, int phases) {
  std::vector<unsigned> b(a.size());
  for (int p = 0; p < phases; ++p) {
    for (int clip = 0; clip < 8; ++clip)
      for (int l = 0; l < ROWS; ++l)
        for (int c = 0; c < COLS; ++c) {
          unsigned acc = (unsigned)p;
          for (int k = -3; k <= 3; ++k) {
            const int ll = l + k;
            if (ll >= 0 && ll < ROWS) acc += a[((size_t)clip * ROWS + ll) * COLS + c];
          }
          b[((size_t)clip * ROWS + l) * COLS + c] = acc;
        }
    a.swap(b);
  }
}

template <typename K, typename... A>
static float timed(K kernel, dim3 grid, TeamState* st, A... args) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipMemset(st, 0, sizeof(TeamState)));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kernel, grid, dim3(256), 0, 0, st, args...);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

static unsigned read_timeout(TeamState* st) {
  unsigned v = 0;
  CHECK(hipMemcpy(&v, reinterpret_cast<char*>(st) + offsetof(TeamState, timeout), 4, hipMemcpyDeviceToHost));
  return v;
}

static double median_ticks(unsigned long long* d_ticks, int n) {
  std::vector<unsigned long long> h(n);
  CHECK(hipMemcpy(h.data(), d_ticks, n * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  return (double)h[n / 2] / 100.0;   // us
}

template <int FORM, bool BYXCC, int LOADMODE, bool UNEVEN>
static void run_pingpong(const char* name, TeamState* st, unsigned* b0, unsigned* b1, unsigned long long* d_ticks,
                         const std::vector<unsigned>& init, const std::vector<unsigned>& want, int phases) {
  const size_t n = init.size();
  CHECK(hipMemcpy(b0, init.data(), n * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(b1, 0xEE, n * 4));
  CHECK(hipMemset(d_ticks, 0, 256 * 8));
  const float ms = timed(pingpong_kernel<FORM, BYXCC, LOADMODE, UNEVEN>, dim3(256), st, b0, b1, phases, d_ticks);
  std::vector<unsigned> got(n);
  CHECK(hipMemcpy(got.data(), (phases & 1) ? b1 : b0, n * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < n; ++i) bad += got[i] != want[i];
  printf("pingpong %-44s phases %d: %8.1f us total, %6.2f us/phase (in-kernel median %6.2f), wrong words %zu / %zu, timeout %u\n",
         name, phases, ms * 1e3, ms * 1e3 / phases, median_ticks(d_ticks, 256) / phases, bad, n, read_timeout(st));
}

int main() {
  TeamState* st;
  CHECK(hipMalloc(&st, sizeof(TeamState)));
  unsigned long long* d_ticks;
  CHECK(hipMalloc(&d_ticks, 256 * 8));

  // ---- census
  {
    unsigned* d;
    CHECK(hipMalloc(&d, 256 * 8));
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(census_kernel, dim3(256), dim3(256), 0, 0, d);
      unsigned h[512];
      CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
      int cnt[16] = {0}, rr = 0;
      for (int b = 0; b < 256; ++b) { cnt[h[2 * b] & 15]++; rr += (h[2 * b] == h[2 * (b % 8)]); }
      printf("census rep %d: per-XCC counts", rep);
      for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
      printf("; blocks with xcc(b) == xcc(b %% 8): %d / 256; xcc of blocks 0..7:", rr);
      for (int b = 0; b < 8; ++b) printf(" %u", h[2 * b]);
      // distinct (xcc, se, cu) triples: HW_ID bits: [3:0] wave, [5:4] simd, [7:6]?, [11:8] cu, [12] sh, [15:13] se
      std::vector<unsigned> keys;
      for (int b = 0; b < 256; ++b) keys.push_back((h[2 * b] << 16) | ((h[2 * b + 1] >> 8) & 0xFF));
      std::sort(keys.begin(), keys.end());
      const size_t distinct = std::unique(keys.begin(), keys.end()) - keys.begin();
      printf("; distinct (xcc, cu/sh/se) = %zu\n", distinct);
    }
    CHECK(hipFree(d));
  }

  // ---- barrier latency
  const int NB = 2000;
  {
    float ms = timed(barrier_kernel<0, true>, dim3(256), st, NB, d_ticks);
    printf("barrier atomic-counter, teams by XCC : %.3f us per barrier (kernel %.1f us, in-kernel median %.3f), timeout %u\n", ms * 1e3 / NB, ms * 1e3,
           median_ticks(d_ticks, 256) / NB, read_timeout(st));
    ms = timed(barrier_kernel<1, true>, dim3(256), st, NB, d_ticks);
    printf("barrier flag-line,      teams by XCC : %.3f us per barrier (kernel %.1f us, in-kernel median %.3f), timeout %u\n", ms * 1e3 / NB, ms * 1e3,
           median_ticks(d_ticks, 256) / NB, read_timeout(st));
    ms = timed(barrier_kernel<0, false>, dim3(256), st, NB, d_ticks);
    printf("barrier atomic-counter, teams by b/32: %.3f us per barrier (kernel %.1f us, in-kernel median %.3f), timeout %u\n", ms * 1e3 / NB, ms * 1e3,
           median_ticks(d_ticks, 256) / NB, read_timeout(st));
  }

  // ---- ping-pong
  const int P = 64;
  const size_t n = (size_t)8 * ROWS * COLS;
  std::vector<unsigned> init(n);
  for (size_t i = 0; i < n; ++i) init[i] = (unsigned)(i * 2654435761u) >> 20;
  std::vector<unsigned> want = init;
  host_ref(want, P);
  unsigned *b0, *b1;
  CHECK(hipMalloc(&b0, n * 4)); CHECK(hipMalloc(&b1, n * 4));
  for (int rep = 0; rep < 2; ++rep) {
    run_pingpong<0, true, 1, false>("atomic bar, XCC teams, sc1 loads", st, b0, b1, d_ticks, init, want, P);
    run_pingpong<1, true, 1, false>("flag bar,   XCC teams, sc1 loads", st, b0, b1, d_ticks, init, want, P);
    run_pingpong<0, true, 1, true>("atomic bar, XCC teams, sc1 loads, uneven", st, b0, b1, d_ticks, init, want, P);
    run_pingpong<1, true, 1, true>("flag bar,   XCC teams, sc1 loads, uneven", st, b0, b1, d_ticks, init, want, P);
    run_pingpong<0, true, 0, false>("atomic bar, XCC teams, PLAIN loads", st, b0, b1, d_ticks, init, want, P);
    run_pingpong<0, true, 0, true>("atomic bar, XCC teams, PLAIN loads, uneven", st, b0, b1, d_ticks, init, want, P);
    run_pingpong<0, false, 1, false>("atomic bar, b/32 teams, sc1 loads", st, b0, b1, d_ticks, init, want, P);
    run_pingpong<0, false, 1, true>("atomic bar, b/32 teams, sc1 loads, uneven", st, b0, b1, d_ticks, init, want, P);
  }

  // ---- the same phases as graph-captured launches
  {
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    CHECK(hipMemcpy(b0, init.data(), n * 4, hipMemcpyHostToDevice));
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int p = 0; p < P; ++p)
      hipLaunchKernelGGL(phase_kernel, dim3(256), dim3(256), 0, s, (p & 1) ? b1 : b0, (p & 1) ? b0 : b1, (unsigned)p);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    std::vector<unsigned> got(n);
    CHECK(hipMemcpy(got.data(), (P & 1) ? b1 : b0, n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) bad += got[i] != want[i];
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int reps = 20;
    CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("launches: %d graph-captured phase kernels: %.2f us/phase (first replay wrong words %zu)\n", P, ms * 1e3 / reps / P, bad);
  }
  printf("done\n");
  return 0;
}
