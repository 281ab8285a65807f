// Dev probe: where does the fixed cost of a dependent decode kernel go on this box?  Chains of N dependent kernels in a
// HIP graph, each stamping wall_clock64 (100 MHz) at its first instruction, after its kernel-argument read, after its
// first global load, and at its end:
//     gap    = start[i+1] - end[i]     (boundary: end-of-kernel release, dispatch, wave launch)
//     args   = kernel-argument read    (scalar loads from the kernarg segment: host or device memory?)
//     load1  = first dependent global load
//     body   = the rest
//   hipcc --offload-arch=gfx950 -O3 -o scripts/dep_chain scripts/dep_chain.hip
//   HIP_FORCE_DEV_KERNARG=0/1 ./scripts/dep_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Big { const unsigned* x; unsigned* y; unsigned long long* stamps; int idx; int pad[40]; };

// 1 WG (or many): every thread reads 16 B of x (written by the predecessor), block-reduces, writes y
__global__ __launch_bounds__(256) void k_small(Big p) {
  const unsigned long long t0 = wall_clock64();
  const int idx = p.idx + p.pad[3];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = wall_clock64();
  const uint4 v = reinterpret_cast<const uint4*>(p.x)[blockIdx.x * 256 + threadIdx.x];
  unsigned s = v.x + v.y + v.z + v.w;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = wall_clock64();
  __shared__ unsigned red[4];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3];
  reinterpret_cast<uint4*>(p.y)[blockIdx.x * 256 + threadIdx.x] = make_uint4(s, v.y, v.z, v.w + 1);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* st = p.stamps + (size_t)idx * 4;
    st[0] = t0; st[1] = t1; st[2] = t2; st[3] = wall_clock64();
  }
}

typedef unsigned int u4v __attribute__((ext_vector_type(4)));
// GEMV-like: each WG reads the whole x row (xbytes) into LDS, each wave streams `tiles` 1-KiB weight tiles, writes 64 B
template <int ORDER>   // 0: weights then x (the first probe), 1: x first then weights, 2: x only (no weight stream)
__global__ __launch_bounds__(512) void k_gemv(Big p, const u4v* __restrict__ w, int xvec, int tiles) {
  extern __shared__ uint4 xs[];
  const unsigned long long t0 = wall_clock64();
  const int idx = p.idx + p.pad[3];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = wall_clock64();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u4v* wp = w + ((size_t)(blockIdx.x * 8 + wave) * tiles) * 64 + lane;
  u4v acc = {0, 0, 0, 0};
  u4v t[8];
  uint4 xv[4];
  if (ORDER >= 1) {
#pragma unroll
    for (int i = 0; i < 4; i++) xv[i] = reinterpret_cast<const uint4*>(p.x)[min((int)threadIdx.x + i * 512, xvec - 1)];
  }
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = (ORDER == 2 || i >= tiles) ? u4v{1, 2, 3, 4} : __builtin_nontemporal_load(wp + i * 64);
  if (ORDER >= 1) {
#pragma unroll
    for (int i = 0; i < 4; i++) if ((int)threadIdx.x + i * 512 < xvec) xs[threadIdx.x + i * 512] = xv[i];
  } else {
    for (int i = threadIdx.x; i < xvec; i += 512) xs[i] = reinterpret_cast<const uint4*>(p.x)[i];
  }
  __syncthreads();
  const unsigned long long t2 = wall_clock64();
#pragma unroll
  for (int i = 0; i < 8; i++) if (i < tiles) { acc += t[i] * xs[(i * 64 + lane) % xvec].x; }
  if (ORDER != 2) for (int i = 8; i < tiles; i++) acc += __builtin_nontemporal_load(wp + i * 64) * xs[(i * 64 + lane) % xvec].x;
  unsigned s = acc.x + acc.y + acc.z + acc.w;
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane < 4) p.y[(blockIdx.x * 8 + wave) * 4 + lane] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* st = p.stamps + (size_t)idx * 4;
    st[0] = t0; st[1] = t1; st[2] = t2; st[3] = wall_clock64();
  }
}

template <class F> int chain(const char* name, hipStream_t st, int n, unsigned long long* d_st, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < n; i++) launch(i);
  CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 10; i++) hipGraphLaunch(ge, st);
  CK(hipStreamSynchronize(st));
  auto t0 = std::chrono::high_resolution_clock::now();
  const int reps = 20;
  for (int i = 0; i < reps; i++) hipGraphLaunch(ge, st);
  CK(hipStreamSynchronize(st));
  double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / reps;
  std::vector<unsigned long long> h(n * 4);
  CK(hipMemcpy(h.data(), d_st, n * 32, hipMemcpyDeviceToHost));
  double gap = 0, args = 0, load1 = 0, body = 0;
  for (int i = 1; i < n; i++) {
    gap += (double)(h[i * 4] - h[(i - 1) * 4 + 3]) / 100.0;
    args += (double)(h[i * 4 + 1] - h[i * 4]) / 100.0;
    load1 += (double)(h[i * 4 + 2] - h[i * 4 + 1]) / 100.0;
    body += (double)(h[i * 4 + 3] - h[i * 4 + 2]) / 100.0;
  }
  const int m = n - 1;
  printf("%-44s %6.2f us/kernel | WG0: gap %5.2f  args %5.2f  load1 %5.2f  rest %5.2f\n", name, us / n, gap / m, args / m, load1 / m, body / m);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return 0;
}

int main() {
  printf("HIP_FORCE_DEV_KERNARG=%s\n", getenv("HIP_FORCE_DEV_KERNARG") ? getenv("HIP_FORCE_DEV_KERNARG") : "(unset)");
  hipStream_t st; CK(hipStreamCreate(&st));
  const int n = 100;
  unsigned *a, *b; CK(hipMalloc(&a, 4 << 20)); CK(hipMalloc(&b, 4 << 20)); CK(hipMemset(a, 1, 4 << 20)); CK(hipMemset(b, 1, 4 << 20));
  unsigned long long* d_st; CK(hipMalloc(&d_st, n * 32));
  u4v* w; const size_t wbytes = (size_t)1 << 30; CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 1, wbytes));
  auto big = [&](int i) { Big p{}; p.x = (i & 1) ? b : a; p.y = (i & 1) ? a : b; p.stamps = d_st; p.idx = i; return p; };
  for (int grid : {1, 256, 1024})
    { char nm[64]; snprintf(nm, 64, "small: %d WG x 256 (16 B/thread)", grid);
      if (chain(nm, st, n, d_st, [&](int i) { hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, st, big(i)); })) return 1; }
  // GEMV-like: 8 MB over 132 WGs (q_a|kv_a), 8 MB over 264, 37 MB over 256, 62 MB over 448
  struct Cfg { int grid, tiles, xvec; const char* nm; } cfgs[] = {
      {132, 7, 896, "gemv 132 WG x 7 tiles/wave (7.4 MB), x 14 KB"}, {264, 4, 896, "gemv 264 WG x 4 tiles (8.4 MB), x 14 KB"},
      {256, 18, 192, "gemv 256 WG x 18 tiles (37 MB), x 3 KB"}, {448, 16, 2048, "gemv 448 WG x 16 tiles (58 MB), x 32 KB"},
      {256, 56, 896, "gemv 256 WG x 56 tiles (117 MB), x 14 KB"}};
  for (int order = 0; order < 3; order++) {
    printf("-- %s\n", order == 0 ? "weights requested first, then x" : order == 1 ? "x requested first, then weights" : "x only");
    for (auto& c : cfgs) {
      // a different weight region per chain position so nothing is served from the L2 / MALL
      auto launch = [&](int i) {
        const size_t per = (size_t)c.grid * 8 * c.tiles * 1024;
        const size_t off = ((size_t)i * per) % (wbytes - per);
        const u4v* wq = (const u4v*)((const char*)w + (off & ~(size_t)1023));
        if (order == 0) hipLaunchKernelGGL(k_gemv<0>, dim3(c.grid), dim3(512), c.xvec * 16, st, big(i), wq, c.xvec, c.tiles);
        else if (order == 1) hipLaunchKernelGGL(k_gemv<1>, dim3(c.grid), dim3(512), c.xvec * 16, st, big(i), wq, c.xvec, c.tiles);
        else hipLaunchKernelGGL(k_gemv<2>, dim3(c.grid), dim3(512), c.xvec * 16, st, big(i), wq, c.xvec, c.tiles);
      };
      if (chain(c.nm, st, n, d_st, launch)) return 1;
    }
  }
  return 0;
}
