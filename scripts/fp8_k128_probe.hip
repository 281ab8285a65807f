// Stand-alone probe: one v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, both block scales 2^0) against the four chained
// v_mfma_f32_16x16x32_fp8_fp8 it would replace in lin_fp8_gemm_kernel, on the same per-lane bytes (lane = row r, k quarter kq; 32
// consecutive k per lane), and both against a double-precision dot product on the host.  Also times 64 of each per wavefront.
//   hipcc --offload-arch=gfx950 -O2 scripts/fp8_k128_probe.hip -o /tmp/fp8_k128_probe && /tmp/fp8_k128_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const uint4* A, const uint4* B, float* o_new, float* o_old, long long* cyc) {
  const int lane = threadIdx.x;
  const uint4 a0 = A[lane * 2], a1 = A[lane * 2 + 1], b0 = B[lane * 2], b1 = B[lane * 2 + 1];
  const v8i a = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
  const v8i b = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  auto l64 = [](unsigned x, unsigned y) { return (long)(((unsigned long long)y << 32) | x); };
  v4f d = {0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(l64(a0.x, a0.y), l64(b0.x, b0.y), d, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(l64(a0.z, a0.w), l64(b0.z, b0.w), d, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(l64(a1.x, a1.y), l64(b1.x, b1.y), d, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(l64(a1.z, a1.w), l64(b1.z, b1.w), d, 0, 0, 0);
  for (int q = 0; q < 4; q++) { o_new[lane * 4 + q] = c[q]; o_old[lane * 4 + q] = d[q]; }
  // timing: 8 independent accumulators x 64 rounds
  v4f t[8];
  for (int i = 0; i < 8; i++) t[i] = c;
  long long s0 = __builtin_readcyclecounter();
  for (int r = 0; r < 64; r++)
    for (int i = 0; i < 8; i++) t[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, t[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  long long s1 = __builtin_readcyclecounter();
  for (int i = 0; i < 8; i++) asm volatile("" :: "v"(t[i]));
  for (int i = 0; i < 8; i++) t[i] = d;
  long long s2 = __builtin_readcyclecounter();
  for (int r = 0; r < 64; r++)
    for (int i = 0; i < 8; i++) {
      t[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(l64(a0.x, a0.y), l64(b0.x, b0.y), t[i], 0, 0, 0);
      t[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(l64(a0.z, a0.w), l64(b0.z, b0.w), t[i], 0, 0, 0);
      t[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(l64(a1.x, a1.y), l64(b1.x, b1.y), t[i], 0, 0, 0);
      t[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(l64(a1.z, a1.w), l64(b1.z, b1.w), t[i], 0, 0, 0);
    }
  long long s3 = __builtin_readcyclecounter();
  for (int i = 0; i < 8; i++) asm volatile("" :: "v"(t[i]));
  if (lane == 0) { cyc[0] = s1 - s0; cyc[1] = s3 - s2; }
}
static float e4m3(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  if (e == 15 && m == 7) return NAN;
  const float x = e ? ldexpf(1.0f + m / 8.0f, e - 7) : ldexpf(m / 8.0f, -6);
  return s ? -x : x;
}
int main() {
  std::vector<unsigned char> A(64 * 32), B(64 * 32);
  srand(5);
  for (auto& x : A) { x = rand() & 0xff; if ((x & 0x7f) == 0x7f) x ^= 1; }
  for (auto& x : B) { x = rand() & 0xff; if ((x & 0x7f) == 0x7f) x ^= 1; }
  unsigned char *dA, *dB; float *dn, *dold; long long* dc;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dn, 1024); hipMalloc(&dold, 1024); hipMalloc(&dc, 16);
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const uint4*)dA, (const uint4*)dB, dn, dold, dc);
  float hn[256], ho[256]; long long hc[2];
  hipMemcpy(hn, dn, 1024, hipMemcpyDeviceToHost); hipMemcpy(ho, dold, 1024, hipMemcpyDeviceToHost); hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
  // lane (r = lane & 15, kq = lane >> 4) holds row r's k = kq*32 + [0, 32) for A (rows of D's... token side) and for B (column side)
  int differ = 0; double worst_new = 0, worst_old = 0;
  for (int lane = 0; lane < 64; lane++)
    for (int q = 0; q < 4; q++) {
      const int col = lane & 15, row = (lane >> 4) * 4 + q;         // D[row][col] = sum_k A[row][k] * B[col][k]
      double ref = 0, mag = 0;
      for (int kq = 0; kq < 4; kq++)
        for (int j = 0; j < 32; j++) {
          const double p = (double)e4m3(A[(kq * 16 + row) * 32 + j]) * e4m3(B[(kq * 16 + col) * 32 + j]);
          ref += p; mag += fabs(p);
        }
      worst_new = fmax(worst_new, fabs(hn[lane * 4 + q] - ref) / mag);
      worst_old = fmax(worst_old, fabs(ho[lane * 4 + q] - ref) / mag);
      differ += hn[lane * 4 + q] != ho[lane * 4 + q];
    }
  printf("K=128 vs 4 x K=32: %d of 256 outputs differ in bits; max |err| / sum|products|: K=128 %.3g, 4 x K=32 %.3g\n", differ, worst_new, worst_old);
  printf("cycles per 128-k fragment (8 independent accumulators): K=128 %.1f, 4 x K=32 %.1f\n", hc[0] / 512.0, hc[1] / 512.0);
  return 0;
}
