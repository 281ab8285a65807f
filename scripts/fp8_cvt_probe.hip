// Stand-alone probe: v_cvt_scalef32_pk_bf16_fp8 (scale 1.0) against v_cvt_pk_f32_fp8 + truncation for every 16-bit pair of e4m3 codes.
//   hipcc --offload-arch=gfx950 -O2 scripts/fp8_cvt_probe.hip -o /tmp/fp8_cvt_probe && /tmp/fp8_cvt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* bad, unsigned* first) {
  const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;      // low 16 bits = two codes
  union { bf2 v; unsigned u; } a;
  a.v = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, 1.0f, false);
  const v2f f = __builtin_amdgcn_cvt_pk_f32_fp8(v, false);
  const unsigned b = (__float_as_uint(f[0]) >> 16) | (__float_as_uint(f[1]) & 0xffff0000u);
  if (a.u != b) { if (atomicAdd(bad, 1u) == 0) { first[0] = v; first[1] = a.u; first[2] = b; } }
}
int main() {
  unsigned *d, h[4] = {0, 0, 0, 0};
  hipMalloc(&d, 16); hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d, d + 1);
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("pairs that differ: %u of 65536", h[0]);
  if (h[0]) printf("  (first: codes %#06x -> %#010x vs %#010x)", h[1], h[2], h[3]);
  printf("\n");
  return 0;
}
