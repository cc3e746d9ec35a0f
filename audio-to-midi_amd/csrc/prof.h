// Optional per-kernel-class timing with HIP events on the launch stream (bench.py's roofline leg).
// Disabled by default: a ProfScope then costs one branch.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

namespace amt {

struct ProfRec { int cls; hipEvent_t e0, e1; double flops, bytes; };

struct Profiler {
  bool on = false;
  std::vector<std::string> names;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  int class_id(const char* name);
  hipEvent_t event();
};
Profiler& profiler();

struct ProfScope {
  hipStream_t s;
  int idx;
  ProfScope(hipStream_t stream, const char* name, double flops, double bytes) : s(stream), idx(-1) {
    Profiler& p = profiler();
    if (!p.on) return;
    ProfRec r;
    r.cls = p.class_id(name);
    r.e0 = p.event();
    r.e1 = p.event();
    r.flops = flops;
    r.bytes = bytes;
    (void)hipEventRecord(r.e0, s);
    idx = (int)p.recs.size();
    p.recs.push_back(r);
  }
  ~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(profiler().recs[idx].e1, s);
  }
};

}  // namespace amt
