// Dev calibration: per-kernel floor inside a HIP graph on this box (empty / tiny kernels, various grids).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_empty() {}
__global__ void k_touch(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void k_read(const int* p, int* q) { int v = p[blockIdx.x & 63]; if (v == 12345) q[0] = v; }
__global__ void k_spin(long long cycles) { long long t0 = clock64(); while (clock64() - t0 < cycles) {} }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <class F> double run(hipStream_t st, int n, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < n; i++) launch(i);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 5; i++) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  auto t0 = std::chrono::high_resolution_clock::now();
  const int reps = 50;
  for (int i = 0; i < reps; i++) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return us / reps;
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  int *p, *q; CK(hipMalloc(&p, 4096)); CK(hipMalloc(&q, 4096)); CK(hipMemset(p, 0, 4096));
  for (int n : {1, 52, 208}) {
    double a = run(st, n, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); });
    double b = run(st, n, [&](int) { hipLaunchKernelGGL(k_empty, dim3(132), dim3(256), 0, st); });
    double c = run(st, n, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st); });
    double d = run(st, n, [&](int) { hipLaunchKernelGGL(k_empty, dim3(448), dim3(512), 16384, st); });
    double e = run(st, n, [&](int) { hipLaunchKernelGGL(k_read, dim3(132), dim3(256), 0, st, p, q); });
    double f = run(st, n, [&](int) { hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, st, p); });
    printf("n=%3d per-replay us: empty1x64 %.1f | empty132x256 %.1f | empty1024x256 %.1f | empty448x512+lds %.1f | read132 %.1f | touch %.1f\n", n, a, b, c, d, e, f);
    if (n > 1) printf("        per-kernel us: %.2f | %.2f | %.2f | %.2f | %.2f | %.2f\n", a / n, b / n, c / n, d / n, e / n, f / n);
  }
  for (long long cyc : {1000LL, 5000LL, 20000LL}) {
    double a = run(st, 52, [&](int) { hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, st, cyc); });
    printf("spin %lld cycles x52: per-kernel %.2f us\n", cyc, a / 52);
  }
  return 0;
}
