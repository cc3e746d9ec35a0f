// XCD teams: in-launch hand-offs between the workgroups that share ONE XCD's L2.
//
// MI355X = 8 XCDs x 32 CUs, each XCD with a private 4 MiB L2 that is coherent for all of ITS compute
// units and not with the other seven.  The clips of a batch are independent (the reference vmaps the
// model, train.py:56-58), so a dependent run of per-clip operators (the Blocks of a CNN stage,
// model.py:120-167) can stay inside one launch if every clip is owned by the workgroups of one XCD:
// a producer's plain stores are in that L2 once its `s_waitcnt vmcnt(0)` has returned, and a consumer
// on the same XCD that loads with `sc1` (bypasses its CU's vector L1, served by L2) reads them.  Nothing
// has to cross the fabric, no `buffer_wbl2` / `buffer_inv`, and the barrier is one flag line in that L2.
//
// Measured (scratch/xcd_team_bench.hip, profiles/r02_notes.md): team barrier 0.48 us (flag line) /
// 0.88 us (atomic counter); 64 dependent stencil phases 2.0 us/phase in one launch, every word correct
// under uneven load with an L1-warm consumer; the same protocol with PLAIN loads or with teams that
// span XCDs reads stale data in every word — hence team membership is the hardware's XCC_ID, never
// blockIdx arithmetic (HIP promises no placement; round-robin over XCDs is only what is observed).
//
// Rules for code that uses a team:
//   * every load of bytes another workgroup of the team wrote in this launch is an sc1 load (team_load*),
//   * every storing wave drains (team_barrier does it) before the workgroup signals,
//   * spins are bounded; a timeout sets TeamState::timeout (sticky) and every later barrier falls through,
//   * the state is zero when a launch starts: the last workgroup to leave a launch clears it.
#pragma once
#include "common.h"

namespace amt {

constexpr int TEAM_MAX = 8;            // XCDs
constexpr int TEAM_MAX_MEMBERS = 64;   // workgroups per team (flag words per team)
constexpr int TEAM_LINE = 32;          // words per 128-byte line
constexpr unsigned TEAM_SPIN_LIMIT = 1u << 21;

struct TeamState {
  unsigned members[TEAM_MAX][TEAM_LINE];          // [xcc][0]: workgroups registered on that XCD
  unsigned flags[TEAM_MAX][TEAM_MAX_MEMBERS];     // [xcc][rank]: last barrier epoch that member reached
  unsigned counter[TEAM_MAX][TEAM_LINE];          // [xcc][0]: barrier arrivals (experiment form AMT_TEAM_BAR == 2 only)
  unsigned exited_team[TEAM_MAX][TEAM_LINE];      // [xcc][0]: members of that team that have left the launch
  unsigned exited[TEAM_LINE];                     // [0]: workgroups that have left the launch
  unsigned timeout[TEAM_LINE];                    // [0]: sticky, != 0 after a bounded spin gave up (never cleared)
};

struct Team {
  int xcc;        // hardware XCC id of this workgroup
  int rank;       // arrival rank inside the team
  int size;       // workgroups of the team
  int index;      // index of this team among the non-empty teams
  int n_teams;    // non-empty teams
  unsigned epoch; // barriers passed
  bool ok;
};

AMT_DEV unsigned team_ld_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
AMT_DEV void team_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
AMT_DEV unsigned team_xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 7;
}

// a pointer every lane read from LDS with the same value -> scalar registers
template <typename P> AMT_DEV P* team_uniform(P* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<P*>(((unsigned long long)hi << 32) | lo);
}
// "this generic pointer points to global memory": a round trip through address space 1, which the compiler's
// address-space inference propagates to the accesses (global_* instead of flat_* instructions)
template <typename P> AMT_DEV P* team_global(P* p) {
  typedef __attribute__((address_space(1))) P* G;
  return (P*)(G)p;
}

// ---- sc1 loads (buffer form: the compiler tracks their vmcnt).  `rs` covers the tensor; offsets in bytes.
typedef __attribute__((ext_vector_type(4))) unsigned team_u32x4;
AMT_DEV __amdgpu_buffer_rsrc_t team_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
template <bool SC1> AMT_DEV void team_load8f(__amdgpu_buffer_rsrc_t rs, const bf16* base, long elem, float (&o)[8]) {
  if constexpr (SC1) {
    const team_u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(elem * 2), 0, 16);
    const bf16x8 v = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  } else {
    load8f(base + elem, o);
  }
}
template <bool SC1> AMT_DEV void team_load8f(__amdgpu_buffer_rsrc_t rs, const f16* base, long elem, float (&o)[8]) {
  if constexpr (SC1) {
    const team_u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(elem * 2), 0, 16);
    const f16x8 v = __builtin_bit_cast(f16x8, raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  } else {
    load8f(base + elem, o);
  }
}
template <bool SC1> AMT_DEV void team_load8f(__amdgpu_buffer_rsrc_t rs, const float* base, long elem, float (&o)[8]) {
  if constexpr (SC1) {
    const team_u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(elem * 4), 0, 16);
    const team_u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(elem * 4 + 16), 0, 16);
    // whole-vector casts: with a per-element __builtin_bit_cast(float, a[i]) hipcc (ROCm 7.2) narrowed the load to
    // buffer_load_dword and used element 0 for all four (profiles/r02_notes.md)
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = fa[i]; o[4 + i] = fb[i]; }
  } else {
    load8f(base + elem, o);
  }
}

// raw 8-element chunk (no conversion) at element offset `elem` of `base`; SC1 = false: a plain load
template <bool SC1> AMT_DEV Chunk8<bf16> team_chunk_load(const bf16* base, size_t elem) {
  if constexpr (SC1) {
    Chunk8<bf16> c;
    c.v = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(team_rsrc(base), (int)(elem * 2), 0, 16));
    return c;
  } else {
    return chunk_load(base + elem);
  }
}
template <bool SC1> AMT_DEV Chunk8<f16> team_chunk_load(const f16* base, size_t elem) {
  if constexpr (SC1) {
    Chunk8<f16> c;
    c.v = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(team_rsrc(base), (int)(elem * 2), 0, 16));
    return c;
  } else {
    return chunk_load(base + elem);
  }
}
template <bool SC1> AMT_DEV Chunk8<float> team_chunk_load(const float* base, size_t elem) {
  if constexpr (SC1) {
    Chunk8<float> c;
    const __amdgpu_buffer_rsrc_t rs = team_rsrc(base);
    c.a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(elem * 4), 0, 16));
    c.b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(elem * 4 + 16), 0, 16));
    return c;
  } else {
    return chunk_load(base + elem);
  }
}

// Every workgroup of the launch calls team_register once, first, and team_assemble after it (work that needs only the
// XCC id and the arrival rank - e.g. warming the first operands into the XCD's L2 - fits in between).
// Team = XCC id; rank by arrival; team_assemble waits (bounded) until all gridDim.x workgroups have registered, so every
// member knows every team's size.
struct TeamTicket { int xcc, rank; };
AMT_DEV TeamTicket team_register(TeamState* st) {
  __shared__ int s_ticket[2];
  if (threadIdx.x == 0) {
    const int xcc = (int)team_xcc_id();
    s_ticket[0] = xcc;
    s_ticket[1] = (int)__hip_atomic_fetch_add(&st->members[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  TeamTicket k;
  k.xcc = __builtin_amdgcn_readfirstlane(s_ticket[0]);
  k.rank = __builtin_amdgcn_readfirstlane(s_ticket[1]);
  return k;
}
AMT_DEV Team team_assemble(TeamState* st, const TeamTicket& k) {
  __shared__ int s_team[4];
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    unsigned spins = 0;
    // the timeout word is STICKY: once any hand-off of this context has given up, team sizes / flag lines may be in a
    // partial state, so every later team launch falls through at once (the host then stops launching team
    // forms as well: chain.hip: block_run_fwd)
    bool ok = team_ld_u32(&st->timeout[0]) == 0u;
    unsigned cnt = 0;
    for (; ok;) {
      cnt = lane < TEAM_MAX ? team_ld_u32(&st->members[lane][0]) : 0u;
      unsigned total = cnt;
#pragma unroll
      for (int o = 1; o < TEAM_MAX; o <<= 1) total += __shfl_xor(total, o);
      if (__shfl(total, 0) >= gridDim.x) break;
      __builtin_amdgcn_s_sleep(2);
      if (++spins > TEAM_SPIN_LIMIT || team_ld_u32(&st->timeout[0]) != 0u) { ok = false; break; }
    }
    const unsigned long long nonempty = __ballot(cnt != 0u);
    if (lane == 0) {
      if (!ok) __hip_atomic_store(&st->timeout[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_team[1] = __popcll(nonempty & ((1ull << k.xcc) - 1ull));
      s_team[2] = __popcll(nonempty);
      s_team[3] = ok ? 1 : 0;
    }
    if (lane == k.xcc) s_team[0] = (int)cnt;
  }
  __syncthreads();
  Team t;
  // wave-uniform by construction: keep them in scalar registers
  t.xcc = k.xcc; t.rank = k.rank;
  t.size = __builtin_amdgcn_readfirstlane(s_team[0]); t.index = __builtin_amdgcn_readfirstlane(s_team[1]);
  t.n_teams = __builtin_amdgcn_readfirstlane(s_team[2]);
  t.epoch = 0; t.ok = __builtin_amdgcn_readfirstlane(s_team[3]) != 0 && t.size > 0 && t.size <= TEAM_MAX_MEMBERS;
  __syncthreads();
  return t;
}
AMT_DEV Team team_form(TeamState* st) { return team_assemble(st, team_register(st)); }

// Warm `bytes` at `base` into this XCD's L2: member `rank` of `size` touches every size-th piece (16 B per
// thread), by LDS-DMA into a dummy LDS region (`dummy`: 1 KiB per wave, never read), so no register waits for the data.
// Why: the members of a team stream the SAME weights in lockstep.  Cold, every line is one L2 miss with 31 requests
// queued behind it, and whatever else goes through that L2 queues too - instruction fetches included: the other
// members stood still for the ~6 us the first member's stream took (profiles/r02_notes.md).  Operands that depend on
// nothing (weights) are therefore requested one step ahead, each line by ONE member.
AMT_DEV void team_warm(const void* base, size_t bytes, int rank, int size, void* dummy) {
  typedef const __attribute__((address_space(1))) void* GlobalPtr;
  typedef __attribute__((address_space(3))) void* LdsPtr;
  const char* p = static_cast<const char*>(base);
  char* d = static_cast<char*>(dummy) + (threadIdx.x >> 6) * 1024;
  const size_t piece = (size_t)blockDim.x * 16;   // bytes one workgroup touches per step
  for (size_t off = (size_t)rank * piece + (size_t)threadIdx.x * 16; off + 16 <= bytes; off += (size_t)size * piece)
    __builtin_amdgcn_global_load_lds((GlobalPtr)(p + off), (LdsPtr)d, 16, 0, 0);
}

// All workgroups of the team: stores issued before the call are visible to sc1 loads issued after it by every member.
// Also a workgroup barrier.  Returns false (for the whole workgroup) after a timeout.
#ifndef AMT_TEAM_BAR
#define AMT_TEAM_BAR 0   // 0: flag line, plain stores (production); experiments: 1 sc1 flag stores, 2 atomic counter, 3 flag line without s_sleep
#endif
AMT_DEV bool team_barrier(TeamState* st, Team& t) {
  __shared__ int s_ok;
  team_drain();                 // every storing wave: its stores have reached the XCD's L2
  __syncthreads();
  t.epoch += 1;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    bool ok = t.ok;
    unsigned spins = 0;
#if AMT_TEAM_BAR == 2
    if (lane == 0) {
      __hip_atomic_fetch_add(&st->counter[t.xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (ok && team_ld_u32(&st->counter[t.xcc][0]) < t.epoch * (unsigned)t.size) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > TEAM_SPIN_LIMIT) { ok = false; __hip_atomic_store(&st->timeout[0], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      }
      s_ok = ok ? 1 : 0;
    }
#else
    if (lane == 0) {
#if AMT_TEAM_BAR == 1
      __hip_atomic_store(&st->flags[t.xcc][t.rank], t.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
      // plain store into the team's flag line (kept in this XCD's L2), then drained
      __hip_atomic_store(&st->flags[t.xcc][t.rank], t.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
      team_drain();
    }
    while (ok) {
      const unsigned v = lane < t.size ? team_ld_u32(&st->flags[t.xcc][lane]) : t.epoch;
      if (__all((int)(v - t.epoch) >= 0)) break;
#if AMT_TEAM_BAR != 3
      __builtin_amdgcn_s_sleep(1);
#endif
      if (++spins > TEAM_SPIN_LIMIT) {
        ok = false;
        if (lane == 0) __hip_atomic_store(&st->timeout[0], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (lane == 0) s_ok = ok ? 1 : 0;
#endif
  }
  __syncthreads();
  t.ok = __builtin_amdgcn_readfirstlane(s_ok) != 0;
  return t.ok;
}

// Every workgroup calls this last (after its final barrier).  The state must be zero when the next launch starts:
//   * the flag line of a team is dirty in THAT XCD's L2 (plain stores), so only a member of the team may clear it (with
//     plain stores again: a write-through store from another XCD would be overwritten by this L2's write-back at the
//     end of the launch) — the last member of each team to leave does;
//   * the registration counters are only ever touched by memory-side atomics and may still be polled by a workgroup
//     that has not seen the full count yet, so they are cleared by the last workgroup of the whole launch.
AMT_DEV void team_leave(TeamState* st, const Team& t) {
  __shared__ int s_last[2];
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned pt = __hip_atomic_fetch_add(&st->exited_team[t.xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned pa = __hip_atomic_fetch_add(&st->exited[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last[0] = (int)(pt + 1) == t.size;
    s_last[1] = pa + 1 == gridDim.x;
  }
  __syncthreads();
  if (s_last[0]) {
    if (threadIdx.x < TEAM_MAX_MEMBERS)
      __hip_atomic_store(&st->flags[t.xcc][threadIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (threadIdx.x == 0) {
      __hip_atomic_store(&st->exited_team[t.xcc][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&st->counter[t.xcc][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (s_last[1]) {
    if (threadIdx.x < TEAM_MAX) __hip_atomic_store(&st->members[threadIdx.x][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) __hip_atomic_store(&st->exited[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace amt
