// Dev calibration of the box a gpurun call landed on: effective shader clock under a latency-bound load, dependent-load
// latency (L2 / MALL / HBM), the cost of a dependent kernel boundary inside a HIP graph, and a streaming read rate.
// The numbers explain why latency-bound decode kernels differ 1.5-2x between boxes while streaming kernels do not.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/box_probe scripts/box_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty() {}
// one lane follows a pointer chain; out[0] = shader cycles (s_memtime), out[1] = 100 MHz wall ticks
__global__ void k_chase(const unsigned* __restrict__ next, int hops, unsigned start, unsigned long long* out) {
  unsigned i = start;
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int h = 0; h < hops; h++) i = __builtin_nontemporal_load(next + (size_t)i * 32);   // one 128-B line per hop
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  out[0] = c1 - c0; out[1] = w1 - w0; out[2] = i;
}
__global__ void k_chase_ptr(const unsigned long long* start, int hops, unsigned long long* out) {
  const unsigned long long* q = start;
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int h = 0; h < hops; h++) q = (const unsigned long long*)__builtin_nontemporal_load(q);
  out[0] = __builtin_readcyclecounter() - c0; out[1] = wall_clock64() - w0; out[2] = (unsigned long long)q;
}
// ALU spin: shader cycles vs wall ticks with every CU busy
__global__ void k_spin(long long cycles, unsigned long long* out) {
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  float a = threadIdx.x;
  while ((long long)(__builtin_readcyclecounter() - c0) < cycles) a = a * 1.0001f + 0.5f;
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = __builtin_readcyclecounter() - c0; out[1] = wall_clock64() - w0; out[2] = (unsigned long long)a; }
}
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream(const u4v* __restrict__ p, size_t n, uint4* sink) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const u4v v = __builtin_nontemporal_load(p + i);
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if (acc.x == 0x12345678u) sink[0] = acc;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("device %s  CUs %d  clockRate %d kHz  memClock %d kHz  L2 %d\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate, pr.memoryClockRate, pr.l2CacheSize);
  unsigned long long* out; CK(hipMalloc(&out, 64)); unsigned long long h[3];
  // --- shader clock: busy chip
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, st, 20000000LL, out); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
    printf("spin all CUs: %.0f MHz shader clock (%.2f ms)\n", (double)h[0] / ((double)h[1] / 100.0), (double)h[1] / 1e5);
  }
  // --- dependent-load latency at several footprints (lines of 128 B, random cycle)
  for (size_t mb : {1, 16, 128, 2048}) {
    const size_t lines = mb * (1 << 20) / 128;
    std::vector<unsigned> perm(lines), nx(lines * 32, 0);
    for (size_t i = 0; i < lines; i++) perm[i] = (unsigned)i;
    srand(1);
    for (size_t i = lines - 1; i > 0; i--) { size_t j = ((size_t)rand() * 32768 + rand()) % (i + 1); std::swap(perm[i], perm[j]); }
    for (size_t i = 0; i < lines; i++) nx[(size_t)perm[i] * 32] = perm[(i + 1) % lines];
    unsigned* d; CK(hipMalloc(&d, lines * 128)); CK(hipMemcpy(d, nx.data(), lines * 128, hipMemcpyHostToDevice));
    const int hops = 4000;
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, st, d, hops, perm[rep * 7], out); CK(hipStreamSynchronize(st));
      CK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
      printf("chase %5zu MB: %.0f ns/hop  %.0f cyc/hop  (clock %.0f MHz)\n", mb, (double)h[1] * 10.0 / hops, (double)h[0] / hops,
             (double)h[0] / ((double)h[1] / 100.0));
    }
    CK(hipFree(d));
  }
  // --- dependent-load latency across MANY allocations (the model's weights are thousands of 8-60 MB hipMallocs): every hop
  // lands in another allocation, so it needs another translation; 64 GB in 32 MB pieces vs in 2 GB pieces
  for (size_t piece_mb : {32, 2048}) {
    const size_t total_mb = 64 * 1024, npiece = total_mb / piece_mb, piece = piece_mb << 20;
    std::vector<char*> ptrs(npiece);
    bool ok = true;
    for (size_t i = 0; i < npiece && ok; i++) ok = hipMalloc(&ptrs[i], piece) == hipSuccess;
    if (!ok) { printf("chase across pieces: allocation failed\n"); break; }
    // chain: hop h sits at a pseudo-random line of piece (h * 7919) % npiece and holds the ADDRESS of the next hop
    const int hops = 4000;
    std::vector<unsigned long long> addr(hops);
    srand(3);
    for (int h = 0; h < hops; h++) addr[h] = (unsigned long long)(ptrs[((size_t)h * 7919) % npiece] + ((size_t)rand() % (piece / 128)) * 128);
    for (int h = 0; h < hops; h++) { unsigned long long nxt = addr[(h + 1) % hops]; CK(hipMemcpy((void*)addr[h], &nxt, 8, hipMemcpyHostToDevice)); }
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k_chase_ptr, dim3(1), dim3(1), 0, st, (const unsigned long long*)addr[0], hops, out); CK(hipStreamSynchronize(st));
      CK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
      printf("chase across %4zu x %4zu MB allocations (64 GB): %.0f ns/hop (pass %d)\n", npiece, piece_mb, (double)h[1] * 10.0 / hops, rep);
    }
    for (auto q : ptrs) CK(hipFree(q));
  }
  // --- boundary cost inside a graph
  for (int grid : {1, 256, 1024}) {
    hipGraph_t g; hipGraphExec_t ge; const int n = 200;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st);
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 5; i++) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < 20; i++) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 20;
    printf("graph of %d empty kernels, grid %4d: %.2f us per kernel\n", n, grid, us / n);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  // --- streaming read
  {
    const size_t bytes = (size_t)4 << 30; uint4* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes));
    uint4* sink; CK(hipMalloc(&sink, 64));
    for (int rep = 0; rep < 3; rep++) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a, st);
      hipLaunchKernelGGL(k_stream, dim3(256 * 8), dim3(256), 0, st, (const u4v*)d, bytes / 16, sink);
      hipEventRecord(b, st); CK(hipStreamSynchronize(st));
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("stream 4 GiB: %.2f TB/s\n", (double)bytes / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
