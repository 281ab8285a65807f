// Dev probe (round 6): cycles per LDS read instruction for the fragment-read patterns of the folded GGUF prompt kernel
// (csrc/ktx_moe_gguf.inc) — which layouts are bank-conflict-free for ds_read_b64 / ds_read_b128 on gfx950.
//   hipcc --offload-arch=gfx950 -O3 scripts/lds_pattern_probe.hip -o scripts/lds_pattern_probe && scripts/lds_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int W>   // W = 8: ds_read_b64, 16: ds_read_b128
__global__ __launch_bounds__(256) void probe(const uint32_t* offs, long long* cycles, uint32_t* sink, int iters, int ustride) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<uint32_t*>(smem)[i] = i;
  __syncthreads();
  const uint32_t a0 = offs[threadIdx.x & 63];
  uint32_t acc = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    asm volatile("" ::: "memory");          // (the reads are loop-invariant otherwise: hoisted)
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const uint32_t a = a0 + u * ustride;   // 16 independent reads per iteration, same bank pattern (ustride 4096) or the real loop's step
      if constexpr (W == 8) { const uint2 v = *reinterpret_cast<const uint2*>(smem + a); acc += v.x ^ v.y; }
      else { const uint4 v = *reinterpret_cast<const uint4*>(smem + a); acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int W>
static void run(const char* name, const uint32_t* h_offs, uint32_t* d_offs, long long* d_cyc, uint32_t* d_sink, int ustride = 4096) {
  hipMemcpy(d_offs, h_offs, 256, hipMemcpyHostToDevice);
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<W>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4096 * 2);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(probe<W>, dim3(256), dim3(256), 65536 + 8192, 0, d_offs, d_cyc, d_sink, iters, ustride);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
  printf("%-58s %6.2f cycles per wave-instruction (4 waves per CU issuing)\n", name, (double)c / iters / 16);
}

int main() {
  uint32_t *d_offs, *d_sink;
  long long* d_cyc;
  hipMalloc(&d_offs, 256); hipMalloc(&d_cyc, 256 * 8); hipMalloc(&d_sink, 256 * 256 * 4);
  uint32_t o[64];
  // (1) the folded kernel's b64 fragment read: lane = (t = lane & 15, kc = lane >> 4): row t * 256 + ((p ^ t) * 16) + (kc & 1) * 8, p = kc >> 1
  for (int l = 0; l < 64; l++) { const int t = l & 15, kc = l >> 4; o[l] = t * 256 + (((kc >> 1) ^ t) * 16) + (kc & 1) * 8; }
  run<8>("b64  swizzled rows of 256 B (folded kernel, round 6)", o, d_offs, d_cyc, d_sink);
  // (2) the same without the swizzle (every row's piece p in slot p): 16 rows on one bank group
  for (int l = 0; l < 64; l++) { const int t = l & 15, kc = l >> 4; o[l] = t * 256 + ((kc >> 1) * 16) + (kc & 1) * 8; }
  run<8>("b64  unswizzled rows of 256 B", o, d_offs, d_cyc, d_sink);
  // (3) b64 fully linear (lane * 8)
  for (int l = 0; l < 64; l++) o[l] = l * 8;
  run<8>("b64  linear", o, d_offs, d_cyc, d_sink);
  // (4) the unit layout of the unfolded kernel: b128 at kc * US + t * 16, US = 1040
  for (int l = 0; l < 64; l++) { const int t = l & 15, kc = l >> 4; o[l] = kc * 1040 + t * 16; }
  run<16>("b128 unit rows, stride 1040 (unfolded kernel)", o, d_offs, d_cyc, d_sink);
  for (int l = 0; l < 64; l++) { const int t = l & 15, kc = l >> 4; o[l] = kc * 1024 + t * 16; }
  run<16>("b128 unit rows, stride 1024", o, d_offs, d_cyc, d_sink);
  for (int l = 0; l < 64; l++) o[l] = l * 16;
  run<16>("b128 linear", o, d_offs, d_cyc, d_sink);
  // (5) b128 on swizzled 256-B rows: lane (t, kc) reads the whole 16-byte piece kc ^ t'... slot = (kc ^ t) & 15
  for (int l = 0; l < 64; l++) { const int t = l & 15, kc = l >> 4; o[l] = t * 256 + ((kc ^ t) & 15) * 16; }
  run<16>("b128 swizzled rows of 256 B (piece kc)", o, d_offs, d_cyc, d_sink);
  // (6) b64 with the row index in the LOW slot bits only: slot = p ^ (t & 7) ... variants for the record
  for (int l = 0; l < 64; l++) { const int t = l & 15, kc = l >> 4; o[l] = t * 256 + ((((kc >> 1) * 2) ^ t) * 16) + (kc & 1) * 8; }
  run<8>("b64  swizzled, pieces p = 2 (kc >> 1)", o, d_offs, d_cyc, d_sink);
  // (7) random 8-byte reads from a 16 KiB table (IQ1_S codebook look-ups)
  uint32_t s = 12345;
  for (int l = 0; l < 64; l++) { s = s * 1664525u + 1013904223u; o[l] = ((s >> 8) & 2047) * 8; }
  run<8>("b64  random 8-byte entries of a 16 KiB table", o, d_offs, d_cyc, d_sink);
  // (8)-(12) the MLA split-KV tile loop (csrc/ktx_mla.hip, rows of 1168 B): V fragments by ds_read_b64_tr_b16 — lane (g = lane >> 4,
  // i = lane & 15) addresses row 8 g + (i >> 2) (+ 4 for the second read), 8 bytes at (i & 3) * 8, 32 bytes further per output tile
  for (int l = 0; l < 64; l++) { const int g = l >> 4, i = l & 15; o[l] = (g * 8 + (i >> 2)) * 1168 + (i & 3) * 8; }
  run<8>("b64  MLA V fragment, rows 8g + i/4 (product)", o, d_offs, d_cyc, d_sink, 32);
  for (int l = 0; l < 64; l++) { const int g = l >> 4, i = l & 15; o[l] = (g * 8 + 2 * (i >> 2)) * 1168 + (i & 3) * 8; }
  run<8>("b64  MLA V fragment, rows 8g + 2 (i/4) (even rows)", o, d_offs, d_cyc, d_sink, 32);
  for (int l = 0; l < 64; l++) { const int g = l >> 4, i = l & 15; o[l] = (g * 8 + 2 * (i >> 2) + 1) * 1168 + (i & 3) * 8; }
  run<8>("b64  MLA V fragment, rows 8g + 2 (i/4) + 1 (odd rows)", o, d_offs, d_cyc, d_sink, 32);
  for (int l = 0; l < 64; l++) o[l] = (l & 15) * 1168 + (l >> 4) * 16;
  run<16>("b128 MLA K fragment, row lane & 15, piece lane >> 4", o, d_offs, d_cyc, d_sink, 64);
  for (int l = 0; l < 64; l++) o[l] = (l & 15) * 64 + (l >> 4) * 16;
  run<16>("b128 MLA P fragment, 64-byte rows", o, d_offs, d_cyc, d_sink, 0);
  return 0;
}
