// Dev probe: where do the ~9 us of the single-workgroup router go?  Variants of "one workgroup reads a cold 256 KiB
// matrix": loads only, + dot2, + per-wave reduce/LDS, with 1..16 workgroups, and a dependent chain of 26 launches each on
// its own buffer (so every launch is cold), timed per launch from a graph.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
  union { uint32_t u; v2bf v; } ua, ub; ua.u = a; ub.u = b;
  return __builtin_amdgcn_fdot2_f32_bf16(ua.v, ub.v, c, false);
}
// MODE 0: loads + trivial use; 1: + dot2 + reduce + LDS + one-wave epilogue.  NWG workgroups split the 64 rows.
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k_gate(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, float* out, int rows_per_wg) {
  __shared__ float s[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = THREADS / 64;
  const int row0 = blockIdx.x * rows_per_wg;
  const int epw = rows_per_wg / nw;   // rows per wave
  uint4 a[4];
  for (int jj = 0; jj < 4; jj++) a[jj] = *reinterpret_cast<const uint4*>(x + lane * 8 + jj * 512);
  float tot = 0.f;
  for (int i = 0; i < epw; i++) {
    const int e = row0 + wave + nw * i;
    uint4 b[4];
#pragma unroll
    for (int jj = 0; jj < 4; jj++) b[jj] = *reinterpret_cast<const uint4*>(w + (size_t)e * 2048 + lane * 8 + jj * 512);
    float acc = 0.f;
    if (MODE == 0) {
#pragma unroll
      for (int jj = 0; jj < 4; jj++) acc += __uint_as_float(b[jj].x ^ a[jj].x);
    } else {
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        acc = dot2(a[jj].x, b[jj].x, acc); acc = dot2(a[jj].y, b[jj].y, acc);
        acc = dot2(a[jj].z, b[jj].z, acc); acc = dot2(a[jj].w, b[jj].w, acc);
      }
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
      if (lane == 0) s[(wave + nw * i) & 63] = acc;
    }
    tot += acc;
  }
  if (MODE == 0) { if (tot == 1234.5f) out[0] = tot; return; }
  __syncthreads();
  if (wave == 0) {
    float v = s[lane & (rows_per_wg - 1)];
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if (lane == 0) out[blockIdx.x] = v;
  }
}
template <class F> double run(hipStream_t st, int n, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < n; i++) launch(i);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 3; i++) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  auto t0 = std::chrono::high_resolution_clock::now();
  const int reps = 30;
  for (int i = 0; i < reps; i++) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return us / reps / n;
}
__global__ void k_flush(float* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int L = 26;
  std::vector<uint16_t*> w(L);
  for (auto& p : w) { CK(hipMalloc(&p, 64 * 2048 * 2)); CK(hipMemset(p, 0x3c, 64 * 2048 * 2)); }
  uint16_t* x; CK(hipMalloc(&x, 4096)); CK(hipMemset(x, 0x3c, 4096));
  float* out; CK(hipMalloc(&out, 4096));
  float* big; const size_t NB = 96u << 20; CK(hipMalloc(&big, NB * 4));   // 384 MB: evicts L2 + MALL between launches
  auto flush = [&]() { hipLaunchKernelGGL(k_flush, dim3((unsigned)(NB / 256)), dim3(256), 0, st, big, NB); };
  double fl = run(st, L, [&](int) { flush(); });
  printf("flush alone: %.2f us per launch\n", fl);
#define T(MODE, THREADS, NWG, label)                                                                                       \
  { double hot = run(st, L, [&](int i) { hipLaunchKernelGGL((k_gate<MODE, THREADS>), dim3(NWG), dim3(THREADS), 0, st, x, w[i], out, 64 / NWG); });   \
    double cold = run(st, L, [&](int i) { flush(); hipLaunchKernelGGL((k_gate<MODE, THREADS>), dim3(NWG), dim3(THREADS), 0, st, x, w[i], out, 64 / NWG); }) - fl; \
    printf("%-34s warm(L2/MALL) %.2f us   cold(after flush) %.2f us\n", label, hot, cold); }
  T(0, 1024, 1, "1 WG x1024, loads only");
  T(1, 1024, 1, "1 WG x1024, dot2+reduce+epilogue");
  T(1, 256, 1, "1 WG x256");
  T(1, 256, 4, "4 WG x256");
  T(1, 256, 16, "16 WG x256 (4 rows each)");
  T(1, 64, 64, "64 WG x64 (1 row each)");
  T(0, 256, 16, "16 WG x256 loads only");
  return 0;
}
