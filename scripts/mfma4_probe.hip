// Dev probe: operand/result layout of v_mfma_i32_4x4x4i8 (16 blocks) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const int* a, const int* b, v4i* out) {
  v4i c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_4x4x4i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
  out[threadIdx.x] = c;
}
int main() {
  int ha[64], hb[64];
  // A: lane l -> bytes (l, 0, 0, 0) pattern: only k=0 nonzero, value = 1 + (l & 3)      (row id within block)
  // B: lane l -> bytes (1,0,0,0) * (10^(l&3))? keep small: value = 1 + 4*(l & 3)          (col id within block)
  for (int l = 0; l < 64; l++) { ha[l] = (1 + (l & 3)) & 0xff; hb[l] = (1 + 4 * (l & 3)) & 0xff; }
  int *da, *db; v4i* dout; hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 64 * 16);
  hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
  v4i h[64]; hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  printf("test1: A[row i]=1+i (k0), B[col j]=1+4j (k0): expect D[i][j]=(1+i)*(1+4j)\n");
  for (int l = 0; l < 8; l++) printf("lane %d: %d %d %d %d\n", l, h[l][0], h[l][1], h[l][2], h[l][3]);
  // test2: block dependence: A = block id + 1 in k0 for all rows; B = 1
  for (int l = 0; l < 64; l++) { ha[l] = (1 + (l >> 2)); hb[l] = 1; }
  hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
  hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  printf("test2: A=1+block, B=1: lanes 0,4,8,..: ");
  for (int l = 0; l < 64; l += 4) printf("%d ", h[l][0]);
  printf("\n");
  // test3: k order: A bytes = (1,2,3,4) , B bytes = (1,10,100,0)... use small: B=(1,0,0,0) then (0,1,0,0)
  for (int kk = 0; kk < 4; kk++) {
    for (int l = 0; l < 64; l++) { ha[l] = 0x04030201; hb[l] = 1 << (8 * kk); }
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
    hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    printf("test3 B byte %d set: D=%d\n", kk, h[0][0]);
  }
  return 0;
}
