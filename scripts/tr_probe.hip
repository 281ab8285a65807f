// Dev probe: semantics of ds_read_b64_tr_b16 on gfx950 (which element lands in which lane).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(v4s* out, int stride) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = i;
  __syncthreads();
  // lane i of each 16-lane group reads a 4-element chunk at row (i/4) [stride elements apart], cols (i%4)*4, group g -> +g*4 rows
  int i = threadIdx.x & 15, g = threadIdx.x >> 4;
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + (g * 4 + (i >> 2)) * stride + (i & 3) * 4));
  out[threadIdx.x] = r;
}
int main() {
  v4s* d; hipMalloc(&d, 64 * sizeof(v4s));
  for (int stride : {16, 64}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
    v4s h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d\n", stride);
    for (int l = 0; l < 64; l += 1) if (l < 20 || l % 16 == 0) printf("lane %2d: %d %d %d %d\n", l, h[l][0], h[l][1], h[l][2], h[l][3]);
  }
  return 0;
}
