// tlb_probe.hip — VERDICT r3 item 2: why does a dependent first touch cost 3-5 us inside the model's decode graph on some boxes
// and 1-1.5 us on others, while isolated chains of the same kernels are fast everywhere?
//
// Hypothesis under test: address translation.  A decode step streams 11 GB of weights out of a 150 GB working set between two
// touches of the small hot buffers (activation rows, norm weights, tickets), so their translations are gone from the TLBs when
// the next kernel's first load needs them; an isolated chain never leaves a few MB.  The probe reproduces exactly that
// difference and nothing else:
//     [ stream(region r of the arena) ; touch(hot) ] x N      captured in ONE graph, r cycling over the arena
// with the streamed region either fixed (cache thrash only: its pages stay translated) or walking over `arena_gb` of distinct
// memory (cache + TLB thrash).  touch = thread 0 of every workgroup times, with s_memtime, a first load from the hot buffer
// (translation + cache miss), a second dependent load from another line of the SAME page (cache miss only) and a third from
// a line of a different, never-touched-this-round 2 MB page of the hot allocation.
//   hipcc --offload-arch=gfx950 -O3 scripts/tlb_probe.hip -o scripts/tlb_probe && scripts/tlb_probe [arena_gb=48]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef unsigned int u4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void stream_kernel(const u4v* __restrict__ src, size_t n16, unsigned* sink) {
  unsigned acc = 0;
  const size_t stride = (size_t)gridDim.x * 512;
  size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const u4v a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
    const u4v c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
    acc += a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// hot: >= 6 MB.  out[wg][4] = {first load, second load same page, third load other 2 MB page, total} in 100 MHz ticks
__global__ __launch_bounds__(512) void touch_kernel(const unsigned* hot, unsigned long long* out, int round, unsigned* sink) {
  if (threadIdx.x != 0) return;
  const unsigned* p = hot + (size_t)blockIdx.x * 64;   // one 256-byte line per workgroup inside the first 64 KB
  const unsigned long long t0 = wall_clock64();
  unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = wall_clock64();
  const unsigned* p2 = hot + (128 * 1024 / 4) + (size_t)blockIdx.x * 64 + (v & 1);   // other line, same 2 MB page
  v += __hip_atomic_load(p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = wall_clock64();
  const unsigned* p3 = hot + ((size_t)(2 + (round & 1) * 2) << 20) / 4 + (size_t)blockIdx.x * 64 + (v & 1);   // another 2 MB page
  v += __hip_atomic_load(p3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t3 = wall_clock64();
  // the model's actual pattern: a line that ANOTHER workgroup (another XCD: block b runs on XCD b % 8) wrote with a plain store in the
  // previous kernel of the chain; then this workgroup writes its own line for the next round
  unsigned* h2 = const_cast<unsigned*>(hot) + ((size_t)6 << 20) / 4;
  const unsigned* p4 = h2 + (size_t)((blockIdx.x + 1) % gridDim.x) * 64 + (v & 1);
  v += *(volatile const unsigned*)p4;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t4 = wall_clock64();
  h2[(size_t)blockIdx.x * 64] = (unsigned)round;
  unsigned long long* o = out + ((size_t)round * gridDim.x + blockIdx.x) * 4;
  o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = t4 - t3;
  if (v == 0x9abcdef0u) sink[1] = v;
}

// Part 2: latency of ONE load per workgroup from memory nobody has touched for a long time, by what is cold about it:
//   mode 0  the same line every round                      (everything hot)
//   mode 1  a new 256-byte line of an often-used 2 MB page (cache miss, translation hot)
//   mode 2  a new 2 MB page every round                    (cache miss + translation miss)
//   mode 3  a new 2 MB page, then a second line of it      (second = cache miss with the translation just fetched)
__global__ __launch_bounds__(64) void cold_kernel(const unsigned* arena, size_t arena_words, unsigned long long* out, int round, int mode,
                                                  unsigned* sink) {
  if (threadIdx.x != 0) return;
  const size_t wg = blockIdx.x, nwg = gridDim.x;
  size_t off = 0;
  if (mode == 0) off = wg * 64;
  else if (mode == 1) off = wg * (4096 / 4) + (size_t)(round % 16) * 64;
  else off = (((size_t)round * nwg + wg) * ((size_t)2 << 20) / 4 + (wg % 7) * 1024) % (arena_words - (1 << 20));
  const unsigned* p = arena + off;
  const unsigned long long t0 = wall_clock64();
  unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = wall_clock64();
  v += __hip_atomic_load(p + 4096 + (v & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // +16 KB: same 2 MB page, other line
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = wall_clock64();
  unsigned long long* o = out + ((size_t)round * nwg + wg) * 4;
  o[0] = t1 - t0; o[1] = t2 - t1;
  if (v == 0x9abcdef0u) sink[2] = v;
}

// Part 3: the in-model situation the two parts above do not reproduce — a small DEPENDENT load issued while the same kernel's bulk
// weight requests are in flight chip-wide.  Wavefronts 0..6 of every workgroup request `bulk_kb` KiB of never-touched memory (nt),
// wavefront 7 (no bulk of its own: a wavefront's replies come back in order) times one load of
//   kind 0  a line another workgroup (another XCD) wrote with a plain store in the PREVIOUS kernel   (an activation row)
//   kind 1  a read-only line every workgroup reads in every kernel                                   (norm weights)
// and then a returning device-scope atomic on a word of its own (a ticket).
__global__ __launch_bounds__(512) void mix_kernel(const u4v* __restrict__ bulk, size_t bulk_stride16, int bulk_loads, unsigned* hot,
                                                  unsigned long long* out, int round, unsigned* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned acc = 0;
  if (wave < 7) {
    const u4v* p = bulk + (size_t)blockIdx.x * bulk_stride16 + (size_t)wave * 64 * bulk_loads + lane;
    for (int i = 0; i < bulk_loads; i++) { const u4v v = __builtin_nontemporal_load(p + (size_t)i * 64); acc += v.x ^ v.w; }
  } else if (lane == 0) {
    unsigned* h2 = hot + ((size_t)6 << 20) / 4;                       // peer-written lines
    unsigned* tick = hot + ((size_t)7 << 20) / 4 + blockIdx.x * 64;   // this workgroup's ticket word
    const unsigned long long t0 = wall_clock64();
    unsigned v = *(volatile const unsigned*)(h2 + (size_t)((blockIdx.x + 1) % gridDim.x) * 64);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = wall_clock64();
    v += *(volatile const unsigned*)(hot + (size_t)(blockIdx.x & 15) * 64 + (v & 1));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = wall_clock64();
    v += __hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = wall_clock64();
    h2[(size_t)blockIdx.x * 64] = (unsigned)round + v * 0;
    unsigned long long* o = out + ((size_t)round * gridDim.x + blockIdx.x) * 4;
    o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2;
    acc = v;
  }
  if (acc == 0x12345678u) sink[3] = acc;
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
  const int arena_gb = argc > 1 ? atoi(argv[1]) : 48;
  const int N = 48, NWG = 256;
  std::vector<char*> chunks;
  for (int i = 0; i < arena_gb; i++) {
    char* p = nullptr;
    if (hipMalloc((void**)&p, (size_t)1 << 30) != hipSuccess) break;
    chunks.push_back(p);
  }
  printf("arena: %zu GiB in 1 GiB allocations\n", chunks.size());
  if (chunks.size() < 4) return 1;
  for (auto c : chunks) CK(hipMemsetAsync(c, 1, (size_t)1 << 30, 0));
  unsigned *hot, *sink;
  unsigned long long* out;
  CK(hipMalloc((void**)&hot, 8 << 20));
  CK(hipMemset(hot, 0, 8 << 20));
  CK(hipMalloc((void**)&sink, 64));
  CK(hipMalloc((void**)&out, (size_t)N * NWG * 4 * 8));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  CK(hipDeviceSynchronize());
  struct Cfg { const char* name; size_t bytes; int walk; };   // walk: 0 = the same region every round, 1 = a new region every round
  const Cfg cfgs[] = {
      {"no stream between touches            ", 0, 0},
      {"stream 64 MB, same region            ", (size_t)64 << 20, 0},
      {"stream 64 MB, walking the arena      ", (size_t)64 << 20, 1},
      {"stream 512 MB, same region           ", (size_t)512 << 20, 0},
      {"stream 512 MB, walking the arena     ", (size_t)512 << 20, 1},
      {"stream 1 GB, walking the arena       ", (size_t)1 << 30, 1},
  };
  printf("%-40s %10s %10s %10s %10s | %s\n", "between two touches", "1st load", "2nd same", "3rd other", "peer-wrote", "us, median over workgroups and rounds (p90 of the 1st and of the peer-written line in brackets)");
  for (const Cfg& c : cfgs) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int r = 0; r < N; r++) {
      if (c.bytes) {
        const size_t per = (size_t)1 << 30;
        const size_t off = c.walk ? ((size_t)r * c.bytes) % (chunks.size() * per) : 0;
        const char* base = chunks[off / per] + off % per;
        hipLaunchKernelGGL(stream_kernel, dim3(NWG), dim3(512), 0, st, (const u4v*)base, c.bytes / 16, sink);
      }
      hipLaunchKernelGGL(touch_kernel, dim3(NWG), dim3(512), 0, st, hot, out, r, sink);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; w++) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h((size_t)N * NWG * 4);
    CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> a, b, d, e;
    for (int r = 8; r < N; r++)
      for (int w = 0; w < NWG; w++) {
        const unsigned long long* o = &h[((size_t)r * NWG + w) * 4];
        a.push_back(o[0] * 0.01); b.push_back(o[1] * 0.01); d.push_back(o[2] * 0.01); e.push_back(o[3] * 0.01);
      }
    std::vector<double> as = a;
    std::sort(as.begin(), as.end());
    std::vector<double> es = e;
    std::sort(es.begin(), es.end());
    printf("%-40s %10.2f %10.2f %10.2f %10.2f | (%.2f, %.2f)\n", c.name, med(a), med(b), med(d), med(e), as[as.size() * 9 / 10], es[es.size() * 9 / 10]);
    (void)hipGraphExecDestroy(ge);
    (void)hipGraphDestroy(g);
  }
  // ---- part 2: cold lines / cold pages (one arena allocation of 1 GiB chunks is contiguous in VA only per chunk: stay inside chunks)
  printf("\nfirst load of a workgroup from ...                  1st load   2nd line of the same page | us, median (p90)\n");
  const char* names[3] = {"the same line every round (hot)          ", "a new line of an often-used page         ", "a new 2 MB page every round              "};
  for (int mode = 0; mode < 3; mode++) {
    for (int nwg : {256, 16}) {
      hipGraph_t g;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int r = 0; r < N; r++) {
        const char* base = chunks[(r * 7) % chunks.size()];
        hipLaunchKernelGGL(cold_kernel, dim3(nwg), dim3(64), 0, st, (const unsigned*)base, ((size_t)1 << 30) / 4, out, r, mode, sink);
      }
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      std::vector<unsigned long long> h((size_t)N * nwg * 4);
      CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> a, b;
      for (int r = 4; r < N; r++)
        for (int w = 0; w < nwg; w++) { a.push_back(h[((size_t)r * nwg + w) * 4] * 0.01); b.push_back(h[((size_t)r * nwg + w) * 4 + 1] * 0.01); }
      std::vector<double> as = a, bs = b;
      std::sort(as.begin(), as.end());
      std::sort(bs.begin(), bs.end());
      printf("%s %3d WGs %8.2f (%5.2f) %8.2f (%5.2f)\n", names[mode], nwg, med(a), as[as.size() * 9 / 10], med(b), bs[bs.size() * 9 / 10]);
      (void)hipGraphExecDestroy(ge);
      (void)hipGraphDestroy(g);
    }
  }
  // ---- part 3: dependent loads under the kernel's own bulk requests
  printf("\nwavefront 7's loads while wavefronts 0..6 of all 256 workgroups request bulk:   peer-written line   read-only line   returning atomic | us, median (p90)\n");
  for (int kb : {0, 32, 128, 448}) {
    const int loads = kb * 1024 / (7 * 64 * 16);   // 16-byte loads per thread of the seven streaming wavefronts
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int r = 0; r < N; r++) {
      const char* base = chunks[(r * 5 + 1) % chunks.size()];
      hipLaunchKernelGGL(mix_kernel, dim3(NWG), dim3(512), 0, st, (const u4v*)base, (size_t)(1 << 20) / 16 * 2, loads, hot, out, r, sink);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h((size_t)N * NWG * 4);
    CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> a, b, c;
    for (int r = 4; r < N; r++)
      for (int w = 0; w < NWG; w++) {
        const unsigned long long* o = &h[((size_t)r * NWG + w) * 4];
        a.push_back(o[0] * 0.01); b.push_back(o[1] * 0.01); c.push_back(o[2] * 0.01);
      }
    auto p90 = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() * 9 / 10]; };
    printf("bulk %3d KiB per workgroup (%5.1f MB per kernel) %8.2f (%5.2f) %8.2f (%5.2f) %8.2f (%5.2f)\n", kb, kb * NWG / 1024.0, med(a), p90(a),
           med(b), p90(b), med(c), p90(c));
    (void)hipGraphExecDestroy(ge);
    (void)hipGraphDestroy(g);
  }
  return 0;
}
