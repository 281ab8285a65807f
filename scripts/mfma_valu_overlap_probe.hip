// Dev probe (round 6): do vector-ALU instructions issue in the shadow of an int8 MFMA on gfx950, and for which MFMA shape?
// One workgroup per CU, W wavefronts per SIMD; every wavefront runs ITER iterations of {NM independent-chain MFMAs, NV independent
// integer VALU instructions} written so that the compiler cannot fuse or drop them.  Reports shader cycles per iteration (s_memtime
// of wavefront 0) for: MFMA only, VALU only, both — for v_mfma_i32_16x16x64_i8 (4 passes) and v_mfma_i32_32x32x32_i8 (8 passes).
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_valu_overlap_probe.hip -o scripts/mfma_valu_overlap_probe && scripts/mfma_valu_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE, int NM, int NV>
__global__ __launch_bounds__(512) void probe(const int* in, int* out, long long* cycles, int iters) {
  const int lane = threadIdx.x;
  v4i a = {in[lane], in[lane + 1], in[lane + 2], in[lane + 3]};
  v4i b = {in[lane + 4], in[lane + 5], in[lane + 6], in[lane + 7]};
  v4i c4[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  v16i c16[2] = {};
  uint32_t v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = in[lane + 8 + i];
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if constexpr (NM > 0) {
        if constexpr (SHAPE == 16) {
#pragma unroll
          for (int m = 0; m < NM; m++) c4[m & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c4[m & 3], 0, 0, 0);
        } else {
#pragma unroll
          for (int m = 0; m < NM; m++) c16[m & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c16[m & 1], 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < NV; i++) {
        const int j = i & 7;
        asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(v[(j + 1) & 7]), "v"(it));   // one full-rate VALU op, opaque to the compiler
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  int s = 0;
#pragma unroll
  for (int m = 0; m < 4; m++) s += c4[m][0] + c4[m][3];
  s += c16[0][0] + c16[1][5];
#pragma unroll
  for (int i = 0; i < 8; i++) s += (int)v[i];
  out[blockIdx.x * blockDim.x + lane] = s;
  if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int NM, int NV>
static void run(const char* name, int waves_per_simd, const int* din, int* dout, long long* dcyc) {
  const int iters = 2000, threads = 256 * waves_per_simd;
  hipLaunchKernelGGL((probe<SHAPE, NM, NV>), dim3(256), dim3(threads), 0, 0, din, dout, dcyc, iters);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<SHAPE, NM, NV>), dim3(256), dim3(threads), 0, 0, din, dout, dcyc, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long long cyc[256];
  hipMemcpy(cyc, dcyc, sizeof(cyc), hipMemcpyDeviceToHost);
  const double per_round = (double)cyc[0] / iters / 4;   // one round = NM MFMAs + NV VALU per wavefront
  printf("%-34s W=%d  NM=%d NV=%2d  %7.1f cycles/round/wave  (%6.1f per SIMD-round of W waves)  wall %.3f ms\n", name, waves_per_simd, NM, NV,
         per_round, per_round, ms);
}

int main() {
  int *din, *dout;
  long long* dcyc;
  hipMalloc(&din, 4096); hipMalloc(&dout, 256 * 512 * 4); hipMalloc(&dcyc, 256 * 8);
  int h[1024];
  for (int i = 0; i < 1024; i++) h[i] = i * 2654435761u;
  hipMemcpy(din, h, 4096, hipMemcpyHostToDevice);
  for (int w = 1; w <= 2; w++) {
    run<16, 4, 0>("16x16x64 mfma only", w, din, dout, dcyc);
    run<16, 0, 8>("valu only", w, din, dout, dcyc);
    run<16, 0, 16>("valu only", w, din, dout, dcyc);
    run<16, 4, 4>("16x16x64 + valu", w, din, dout, dcyc);
    run<16, 4, 8>("16x16x64 + valu", w, din, dout, dcyc);
    run<16, 4, 12>("16x16x64 + valu", w, din, dout, dcyc);
    run<16, 4, 16>("16x16x64 + valu", w, din, dout, dcyc);
    run<32, 2, 0>("32x32x32 mfma only", w, din, dout, dcyc);
    run<32, 2, 4>("32x32x32 + valu", w, din, dout, dcyc);
    run<32, 2, 8>("32x32x32 + valu", w, din, dout, dcyc);
    run<32, 2, 12>("32x32x32 + valu", w, din, dout, dcyc);
    run<32, 2, 16>("32x32x32 + valu", w, din, dout, dcyc);
  }
  return 0;
}
