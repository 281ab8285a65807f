// ktx_prof.hip — per-launch timing of the library's kernels (measurement aid for bench.py; never on in the product path).
//
// While enabled, every instrumented launch site brackets its kernel with two HIP events on the launch stream and appends
// (label, algorithmic bytes, events) to an ordered log; ktx_timing_collect synchronises the device and returns the log as
// text, one line per launch, in launch order.  Mode 2 logs the labels only (no events): rocprofv3 --pmc child passes use
// it to map the dispatches they see back to kernel classes.  Launches made while timing is on are not graph-capturable.
// The reference's analogue is its FORWARD_TIME_PROFILE per-stage timers (kt-kernel/operators/amx/moe_base.hpp:200-206).
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "ktx_common.h"

namespace {
struct Rec {
  std::string label;
  double bytes;
  hipEvent_t e0, e1;
};
int g_mode = 0;
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_free;

hipEvent_t get_event() {
  if (!g_free.empty()) {
    hipEvent_t e = g_free.back();
    g_free.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

int ktx_timing_mode() { return g_mode; }

std::string ktx_fmt(const char* fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return std::string(buf);
}

KtxTimeScope::KtxTimeScope(hipStream_t st, double bytes, std::string label) : st_(st), idx_(-1) {
  if (!g_mode || label.empty()) return;
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r;
  r.label = std::move(label);
  r.bytes = bytes;
  r.e0 = r.e1 = nullptr;
  if (g_mode == 1) {
    r.e0 = get_event();
    r.e1 = get_event();
    (void)hipEventRecord(r.e0, st);
  }
  idx_ = (long)g_recs.size();
  g_recs.push_back(std::move(r));
}

KtxTimeScope::~KtxTimeScope() {
  if (idx_ < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if ((size_t)idx_ < g_recs.size() && g_recs[idx_].e1) (void)hipEventRecord(g_recs[idx_].e1, st_);
}

extern "C" int ktx_timing_enable(int mode) {
  KTX_REQUIRE(mode >= 0 && mode <= 2, "ktx_timing_enable: mode must be 0 (off), 1 (events) or 2 (labels only)");
  g_mode = mode;
  return 0;
}

// Text log of every launch since the previous collect: "<label>\t<algorithmic bytes>\t<microseconds or -1>\n".
// Returns the number of bytes the full log needs (including the terminating NUL) through *needed; writes at most `cap`.
extern "C" int ktx_timing_collect(char* buf, size_t cap, size_t* needed) {
  KTX_HIP(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lk(g_mu);
  std::string out;
  for (auto& r : g_recs) {
    float ms = -1e-3f;
    if (r.e0 && r.e1) {
      if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = -1e-3f;
    }
    char line[64];
    snprintf(line, sizeof(line), "\t%.0f\t%.3f\n", r.bytes, (double)ms * 1e3);
    out += r.label;
    out += line;
  }
  if (needed) *needed = out.size() + 1;
  if (buf && cap >= out.size() + 1) {
    memcpy(buf, out.c_str(), out.size() + 1);
    for (auto& r : g_recs) {
      if (r.e0) g_free.push_back(r.e0);
      if (r.e1) g_free.push_back(r.e1);
    }
    g_recs.clear();
  }
  return 0;
}

// ---- dev probe: evict the instruction caches (scripts/lin_stamps.py --thrash) --------------------------------------------
// Inside a model every kernel of a decode step finds the instruction cache (64 KB per CU pair) filled by the ~10 kernels
// that ran since its previous launch; a chain of identical launches does not.  Two kernels of 48 KB of straight-line code on
// every CU put an isolated chain into the model's state, so cold-code costs can be measured per kernel.
namespace {
template <int ID>
__global__ __launch_bounds__(256) void icache_thrash_kernel(float* out, float seed) {
  float a0 = seed + ID, a1 = seed * 2.f, a2 = seed * 3.f, a3 = seed * 4.f;
  const float b = seed + 1.5f, c = (float)ID;
#pragma unroll
  for (int i = 0; i < 48 * 32; i++) {   // 4 x v_fma_f32 (8 bytes each) = 32 bytes per iteration
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
  }
  if (a0 + a1 + a2 + a3 == 12345.678f) out[blockIdx.x] = a0;
}
}  // namespace
extern "C" int ktx_debug_icache_thrash(void* d_scratch, void* stream) {
  hipLaunchKernelGGL(icache_thrash_kernel<0>, dim3(512), dim3(256), 0, (hipStream_t)stream, (float*)d_scratch, 1.0f);
  hipLaunchKernelGGL(icache_thrash_kernel<1>, dim3(512), dim3(256), 0, (hipStream_t)stream, (float*)d_scratch, 1.0f);
  return 0;
}
