// Dev probe: what does COLD CODE cost?  Kernels whose body is KB kilobytes of straight-line VALU code (no loop), timed
// inside a HIP graph chain  (a) warm: the same kernel back to back,  (b) L1I-cold: 8 distinct instances round-robin
// (8 x KB > the 64 KB instruction cache),  (c) L2/MALL-cold: a 1 GiB streaming read between every two kernels (its time
// is measured on its own and subtracted).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/icache_probe scripts/icache_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int ID, int KB, int PF = 0>   // PF = 1: the kernel first reads its own code as DATA (one burst) so the instruction fetches hit the L2
__global__ __launch_bounds__(256) void k_code(float* out, float seed) {
  unsigned pfv = 0;
  if (PF) {
    const char* pc = (const char*)__builtin_amdgcn_s_getpc();
#pragma unroll
    for (int i = 0; i < (KB + 7) / 8; i++) pfv ^= *reinterpret_cast<const volatile unsigned*>(pc + (size_t)(i * 64 + (threadIdx.x & 63)) * 128);
  }
  float a0 = seed + ID, a1 = seed * 2.f, a2 = seed * 3.f, a3 = seed * 4.f;
  const float b = seed + 1.5f, c = (float)ID;
#pragma unroll
  for (int i = 0; i < KB * 32; i++) {   // 4 x v_fma_f32 (8 bytes each) = 32 bytes per iteration
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
  }
  if (threadIdx.x == 0) out[blockIdx.x] = a0 + a1 + a2 + a3 + (pfv == 0x1234567u ? 1.f : 0.f);
}
__global__ void k_pc(unsigned long long* out) { out[0] = (unsigned long long)__builtin_amdgcn_s_getpc(); }
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_flush(const u4v* __restrict__ p, size_t n, unsigned* sink) {
  u4v acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= p[i];
  if (acc.x == 0x12345u) sink[0] = acc.y;
}

typedef void (*kfn)(float*, float);
template <int KB, int PF = 0> struct Set {
  static kfn get(int id) {
    switch (id & 7) {
      case 0: return k_code<0, KB, PF>; case 1: return k_code<1, KB, PF>; case 2: return k_code<2, KB, PF>; case 3: return k_code<3, KB, PF>;
      case 4: return k_code<4, KB, PF>; case 5: return k_code<5, KB, PF>; case 6: return k_code<6, KB, PF>; default: return k_code<7, KB, PF>;
    }
  }
};

template <class F> double chain(hipStream_t st, int n, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < n; i++) launch(i);
  hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 3; i++) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  auto t0 = std::chrono::high_resolution_clock::now();
  const int reps = 5;
  for (int i = 0; i < reps; i++) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return us / n;
}

template <int KB> int run(hipStream_t st, float* out, const u4v* big, size_t nbig, unsigned* sink, double flush_us) {
  for (int grid : {1, 256}) {
    const int n = 64;
    double warm = chain(st, n, [&](int) { hipLaunchKernelGGL(Set<KB>::get(0), dim3(grid), dim3(256), 0, st, out, 1.0f); });
    double rr = chain(st, n, [&](int i) { hipLaunchKernelGGL(Set<KB>::get(i), dim3(grid), dim3(256), 0, st, out, 1.0f); });
    double cold = chain(st, n, [&](int i) {
      hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, st, big, nbig, sink);
      hipLaunchKernelGGL(Set<KB>::get(i), dim3(grid), dim3(256), 0, st, out, 1.0f);
    }) - flush_us;
    double coldpf = chain(st, n, [&](int i) {
      hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, st, big, nbig, sink);
      hipLaunchKernelGGL((Set<KB, 1>::get(i)), dim3(grid), dim3(256), 0, st, out, 1.0f);
    }) - flush_us;
    printf("code %3d KB, grid %3d: warm %6.2f us | 8 instances round-robin %6.2f us | after a 1 GiB flush %6.2f us | flush + self-prefetch %6.2f us\n", KB, grid, warm, rr, cold, coldpf);
  }
  return 0;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  float* out; CK(hipMalloc(&out, 4096));
  const size_t bytes = (size_t)1 << 30; u4v* big; CK(hipMalloc(&big, bytes)); CK(hipMemset(big, 1, bytes));
  unsigned* sink; CK(hipMalloc(&sink, 64));
  {   // where does kernel code live?
    unsigned long long* d; CK(hipMalloc(&d, 8)); unsigned long long pc = 0;
    hipLaunchKernelGGL(k_pc, dim3(1), dim3(1), 0, st, d); CK(hipStreamSynchronize(st)); CK(hipMemcpy(&pc, d, 8, hipMemcpyDeviceToHost));
    hipPointerAttribute_t at; hipError_t e = hipPointerGetAttributes(&at, (void*)pc);
    printf("kernel code at %#llx (a hipMalloc'ed buffer: %p); hipPointerGetAttributes: %s", pc, (void*)big, hipGetErrorString(e));
    if (e == hipSuccess) printf(" type %d device %d", (int)at.type, at.device);
    printf("\n"); (void)hipGetLastError();
  }
  double flush_us = chain(st, 32, [&](int) { hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, st, big, bytes / 16, sink); });
  printf("flush kernel alone: %.1f us (%.2f TB/s)\n", flush_us, bytes / flush_us / 1e6);
  run<1>(st, out, big, bytes / 16, sink, flush_us);
  run<4>(st, out, big, bytes / 16, sink, flush_us);
  run<12>(st, out, big, bytes / 16, sink, flush_us);
  run<48>(st, out, big, bytes / 16, sink, flush_us);
  return 0;
}
