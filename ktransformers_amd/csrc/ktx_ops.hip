// ktx_ops.hip — RMSNorm / fused add+RMSNorm / SiLU·mul / MLA prep (RMSNorm + YaRN RoPE) for gfx950.  C ABI: include/ktx_ops.h.
// All of these are latency-bound row kernels: one 256-thread workgroup per token, 16-byte loads, the row stays in
// registers between the reduction and the scaling pass.
#include "ktx_common.h"

#include "../../include/ktx_ops.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    f[2 * i] = __uint_as_float(d[i] << 16);
    f[2 * i + 1] = __uint_as_float(d[i] & 0xffff0000u);
  }
}
// torch's fp32 -> bf16 cast: plain round-to-nearest-even (no denormal flush)
__device__ __forceinline__ uint32_t rne_bf16(float f) {
  const uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x0040u;
  return (u + (0x7fffu + ((u >> 16) & 1u))) >> 16;
}
__device__ __forceinline__ float rbf(float f) { return __uint_as_float(rne_bf16(f) << 16); }
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(rne_bf16(f[0]) | (rne_bf16(f[1]) << 16), rne_bf16(f[2]) | (rne_bf16(f[3]) << 16),
                    rne_bf16(f[4]) | (rne_bf16(f[5]) << 16), rne_bf16(f[6]) | (rne_bf16(f[7]) << 16));
}

__device__ __forceinline__ float block_sum(float v, float* s_red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) s_red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < nw; w++) t += s_red[w];
  __syncthreads();
  return t;
}

constexpr int NORM_VPT = 8;   // 16-byte vectors per thread: dim <= 256 * 8 * 8 = 16384

// MODE 0: single rounding; 1: forward_native's double rounding; 2: fused add (x += residual first)
template <int MODE>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* x, long ldx, bf16_t* res, const bf16_t* __restrict__ w,
                                                      bf16_t* y, long ldy, int T, int dim, float eps, const int32_t* d_bsz) {
  __shared__ float s_red[4];
  const int t = blockIdx.x;
  int bsz = T;
  if (d_bsz) bsz = min(max(*d_bsz, 0), T);
  if (t >= bsz) return;
  const int nvec = dim >> 3;
  float v[NORM_VPT][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_VPT; i++) {
    const int c = threadIdx.x + i * 256;
    if (c < nvec) {
      unpack8(*reinterpret_cast<const uint4*>(x + (size_t)t * ldx + c * 8), v[i]);
      if constexpr (MODE == 2) {
        float r[8];
        unpack8(*reinterpret_cast<const uint4*>(res + (size_t)t * dim + c * 8), r);
#pragma unroll
        for (int e = 0; e < 8; e++) v[i][e] += r[e];
        *reinterpret_cast<uint4*>(res + (size_t)t * dim + c * 8) = pack8(v[i]);
      }
#pragma unroll
      for (int e = 0; e < 8; e++) ss += v[i][e] * v[i][e];
    }
  }
  ss = block_sum(ss, s_red);
  const float r = 1.0f / sqrtf(ss / (float)dim + eps);
#pragma unroll
  for (int i = 0; i < NORM_VPT; i++) {
    const int c = threadIdx.x + i * 256;
    if (c < nvec) {
      float wf[8], o[8];
      unpack8(*reinterpret_cast<const uint4*>(w + c * 8), wf);
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = MODE == 1 ? wf[e] * rbf(v[i][e] * r) : v[i][e] * r * wf[e];
      *reinterpret_cast<uint4*>(y + (size_t)t * ldy + c * 8) = pack8(o);
    }
  }
}

// act_fn(gate) * up in bf16 arithmetic: silu evaluated in fp32 and rounded to bf16 (torch's SiLU on a bf16 tensor), then
// the bf16 product.
__global__ __launch_bounds__(256) void silu_mul_kernel(const bf16_t* __restrict__ gu, long ldg, bf16_t* __restrict__ y, int T,
                                                       int inter, const int32_t* d_bsz) {
  int bsz = T;
  if (d_bsz) bsz = min(max(*d_bsz, 0), T);
  const int t = blockIdx.y;
  if (t >= bsz) return;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c * 8 >= inter) return;
  float g[8], u[8], o[8];
  unpack8(*reinterpret_cast<const uint4*>(gu + (size_t)t * ldg + c * 8), g);
  unpack8(*reinterpret_cast<const uint4*>(gu + (size_t)t * ldg + inter + c * 8), u);
#pragma unroll
  for (int e = 0; e < 8; e++) o[e] = rbf(g[e] / (1.0f + expf(-g[e]))) * u[e];
  *reinterpret_cast<uint4*>(y + (size_t)t * inter + c * 8) = pack8(o);
}

struct PrepParams {
  int T, H, nope, rope, kvl;
  const bf16_t* q; long q_rs;
  bf16_t* q_pe;
  const bf16_t* kv; long kv_rs;
  const bf16_t* nw; float eps;
  bf16_t *ckv, *kpe;
  const int64_t* pos; const float* inv_freq; float mscale;
};

// one (token, head) vector of `rope` elements: thread i < rope/2 owns the interleaved pair (2i, 2i+1)
__device__ __forceinline__ void rope_pair(const bf16_t* src, bf16_t* dst, int i, int half, float pos, const float* inv_freq,
                                          float mscale) {
  const uint32_t pr = *reinterpret_cast<const uint32_t*>(src + 2 * i);
  const float u1 = __uint_as_float(pr << 16), u2 = __uint_as_float(pr & 0xffff0000u);   // de-interleaved halves
  const float fr = pos * inv_freq[i];
  const float c = rbf(cosf(fr) * mscale), s = rbf(sinf(fr) * mscale);
  // q*cos + rotate_half(q)*sin, every product and the sum rounded to bf16
  dst[i] = (bf16_t)rne_bf16(rbf(u1 * c) + rbf(-u2 * s));
  dst[half + i] = (bf16_t)rne_bf16(rbf(u2 * c) + rbf(u1 * s));
}

// grid (T, 1 + ceil(H / 8)): y == 0 -> latent RMSNorm + k_pe RoPE; y >= 1 -> q_pe RoPE of 8 heads
__global__ __launch_bounds__(256) void mla_prep_kernel(PrepParams p) {
  __shared__ float s_red[4];
  const int t = blockIdx.x, tid = threadIdx.x;
  const float pos = (float)p.pos[t];
  const int half = p.rope >> 1;
  if (blockIdx.y == 0) {
    if (p.kv) {
      const bf16_t* row = p.kv + (size_t)t * p.kv_rs;
      if (tid < half) rope_pair(row + p.kvl, p.kpe + (size_t)t * p.rope, tid, half, pos, p.inv_freq, p.mscale);
      const int nvec = p.kvl >> 3;   // kv_lora <= 2048
      float v[8];
      float ss = 0.f;
      if (tid < nvec) {
        unpack8(*reinterpret_cast<const uint4*>(row + tid * 8), v);
#pragma unroll
        for (int e = 0; e < 8; e++) ss += v[e] * v[e];
      }
      ss = block_sum(ss, s_red);
      const float r = 1.0f / sqrtf(ss / (float)p.kvl + p.eps);
      if (tid < nvec) {
        float wf[8], o[8];
        unpack8(*reinterpret_cast<const uint4*>(p.nw + tid * 8), wf);
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = wf[e] * rbf(v[e] * r);
        *reinterpret_cast<uint4*>(p.ckv + (size_t)t * p.kvl + tid * 8) = pack8(o);
      }
    }
    return;
  }
  if (!p.q) return;
  const int per = 256 / half;                     // heads per workgroup pass (rope 64 -> 8)
  const int h = (blockIdx.y - 1) * per + tid / half, i = tid % half;
  if (tid < per * half && h < p.H)
    rope_pair(p.q + (size_t)t * p.q_rs + (size_t)h * (p.nope + p.rope) + p.nope, p.q_pe + ((size_t)t * p.H + h) * p.rope, i,
              half, pos, p.inv_freq, p.mscale);
}


// ---- greedy sampling: argmax over a row of bf16 logits (first maximum wins, like torch.argmax on the device) -----------------
// One launch: every workgroup scans a contiguous chunk and hands its (value, index) to the last arriver (arrival ticket,
// payload through write-through sc1 stores / sc1 loads — the fence-free hand-off of the router, ktx_gate_dev.inc), which
// reduces the partials and writes the int64 index.  NaNs never win a comparison.
constexpr int ARGMAX_WGS = 64;
__device__ __forceinline__ void argmax_pick(float& bv, int& bi, float v, int i) {
  if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}
__global__ __launch_bounds__(256) void argmax_bf16_kernel(const bf16_t* __restrict__ x, long ld, int n, float* __restrict__ pval,
                                                          int* __restrict__ pidx, int* __restrict__ counters,
                                                          int64_t* __restrict__ out) {
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  __shared__ int s_last;
  const int row = blockIdx.y, wg = blockIdx.x, nwg = gridDim.x, tid = threadIdx.x, lane = tid & 63;
  const bf16_t* xr = x + (size_t)row * ld;
  const int npiece = n >> 3, per = (npiece + nwg - 1) / nwg;   // whole 16-byte pieces; the last n % 8 elements one by one
  const int p0 = wg * per, p1 = min(npiece, p0 + per);
  float bv = -__builtin_inff();
  int bi = 0x7fffffff;
  for (int p = p0 + tid; p < p1; p += 256) {
    const uint4 v = *reinterpret_cast<const uint4*>(xr + (size_t)p * 8);
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      argmax_pick(bv, bi, __uint_as_float(d[q] << 16), p * 8 + 2 * q);
      argmax_pick(bv, bi, __uint_as_float(d[q] & 0xffff0000u), p * 8 + 2 * q + 1);
    }
  }
  if (wg == nwg - 1 && tid < (n & 7)) argmax_pick(bv, bi, bf16_to_f32(xr[npiece * 8 + tid]), npiece * 8 + tid);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) argmax_pick(bv, bi, __shfl_xor(bv, o, 64), __shfl_xor(bi, o, 64));
  if (lane == 0) { s_v[tid >> 6] = bv; s_i[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; w++) argmax_pick(bv, bi, s_v[w], s_i[w]);
    __hip_atomic_store(&pval[row * ARGMAX_WGS + wg], bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&pidx[row * ARGMAX_WGS + wg], bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int ticket = __hip_atomic_fetch_add(&counters[row], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == nwg - 1;
    if (last) __hip_atomic_store(&counters[row], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    s_last = last;
  }
  __syncthreads();
  if (!s_last || tid >= 64) return;
  bv = -__builtin_inff();
  bi = 0x7fffffff;
  if (lane < nwg) {
    bv = __hip_atomic_load(&pval[row * ARGMAX_WGS + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bi = __hip_atomic_load(&pidx[row * ARGMAX_WGS + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) argmax_pick(bv, bi, __shfl_xor(bv, o, 64), __shfl_xor(bi, o, 64));
  if (lane == 0) out[row] = bi == 0x7fffffff ? 0 : bi;
}

}  // namespace

extern "C" int ktx_rmsnorm(const void* d_x, int64_t ldx, const void* d_w, void* d_y, int64_t ldy, int T, int dim, float eps,
                           int native_rounding, const int32_t* d_bsz, ktx_stream_t stream) {
  KTX_REQUIRE(d_x && d_w && d_y, "ktx_rmsnorm: null argument");
  KTX_REQUIRE(dim > 0 && dim % 8 == 0 && dim <= 256 * 8 * NORM_VPT, "ktx_rmsnorm: dim must be a multiple of 8 and <= 16384");
  KTX_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0, "ktx_rmsnorm: row strides must be multiples of 8 elements");
  if (T <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  KTX_TIMED(st, (double)T * dim * 4.0 + dim * 2.0, "rmsnorm_kernel T=%d dim=%d", T, dim);
  if (native_rounding)
    hipLaunchKernelGGL(rmsnorm_kernel<1>, dim3(T), dim3(256), 0, st, (const bf16_t*)d_x, (long)ldx, (bf16_t*)nullptr,
                       (const bf16_t*)d_w, (bf16_t*)d_y, (long)ldy, T, dim, eps, d_bsz);
  else
    hipLaunchKernelGGL(rmsnorm_kernel<0>, dim3(T), dim3(256), 0, st, (const bf16_t*)d_x, (long)ldx, (bf16_t*)nullptr,
                       (const bf16_t*)d_w, (bf16_t*)d_y, (long)ldy, T, dim, eps, d_bsz);
  KTX_HIP(hipGetLastError());
  return 0;
}

extern "C" int ktx_fused_add_rmsnorm(void* d_x, void* d_residual, const void* d_w, int T, int dim, float eps,
                                     const int32_t* d_bsz, ktx_stream_t stream) {
  KTX_REQUIRE(d_x && d_residual && d_w, "ktx_fused_add_rmsnorm: null argument");
  KTX_REQUIRE(dim > 0 && dim % 8 == 0 && dim <= 256 * 8 * NORM_VPT, "ktx_fused_add_rmsnorm: dim must be a multiple of 8 and <= 16384");
  if (T <= 0) return 0;
  hipLaunchKernelGGL(rmsnorm_kernel<2>, dim3(T), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)d_x, (long)dim,
                     (bf16_t*)d_residual, (const bf16_t*)d_w, (bf16_t*)d_x, (long)dim, T, dim, eps, d_bsz);
  KTX_HIP(hipGetLastError());
  return 0;
}

extern "C" int ktx_silu_mul(const void* d_gu, int64_t ldg, void* d_y, int T, int inter, const int32_t* d_bsz,
                            ktx_stream_t stream) {
  KTX_REQUIRE(d_gu && d_y, "ktx_silu_mul: null argument");
  KTX_REQUIRE(inter > 0 && inter % 8 == 0 && ldg % 8 == 0, "ktx_silu_mul: sizes must be multiples of 8");
  if (T <= 0) return 0;
  KTX_TIMED((hipStream_t)stream, (double)T * inter * 6.0, "silu_mul_kernel T=%d I=%d", T, inter);
  hipLaunchKernelGGL(silu_mul_kernel, dim3((inter / 8 + 255) / 256, T), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)d_gu, (long)ldg, (bf16_t*)d_y, T, inter, d_bsz);
  KTX_HIP(hipGetLastError());
  return 0;
}

extern "C" size_t ktx_argmax_workspace_bytes(int rows) { return (size_t)(rows > 0 ? rows : 0) * (2 * ARGMAX_WGS + 1) * 4; }

extern "C" int ktx_argmax_bf16(const void* d_x, int64_t ldx, int rows, int n, int64_t* d_out, void* d_workspace,
                               ktx_stream_t stream) {
  KTX_REQUIRE(d_x && d_out && d_workspace, "ktx_argmax_bf16: null argument");
  KTX_REQUIRE(n > 0 && ldx >= n && (rows == 1 || ldx % 8 == 0) && ((uintptr_t)d_x & 15) == 0,
              "ktx_argmax_bf16: rows must start on 16-byte boundaries (ldx % 8 == 0, ldx >= n)");
  if (rows <= 0) return 0;
  float* pval = (float*)d_workspace;
  int* pidx = (int*)(pval + (size_t)rows * ARGMAX_WGS);
  int* counters = pidx + (size_t)rows * ARGMAX_WGS;
  const int nwg = std::min(ARGMAX_WGS, std::max(1, (n / 8 + 255) / 256));
  KTX_TIMED((hipStream_t)stream, (double)rows * n * 2.0, "argmax_bf16_kernel T=%d n=%d", rows, n);
  hipLaunchKernelGGL(argmax_bf16_kernel, dim3(nwg, rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)d_x, (long)ldx, n,
                     pval, pidx, counters, d_out);
  KTX_HIP(hipGetLastError());
  return 0;
}

extern "C" int ktx_mla_prep(int T, int num_heads, int nope_dim, int rope_dim, int kv_lora, const void* d_q,
                            int64_t q_row_stride, void* d_q_pe_out, const void* d_kv, int64_t kv_row_stride,
                            const void* d_kv_norm_w, float eps, void* d_ckv_out, void* d_kpe_out, const int64_t* d_pos,
                            const float* d_inv_freq, float mscale, ktx_stream_t stream) {
  KTX_REQUIRE(d_pos && d_inv_freq, "ktx_mla_prep: null positions / inv_freq");
  KTX_REQUIRE(rope_dim > 0 && rope_dim % 2 == 0 && rope_dim <= 512 && 256 % (rope_dim / 2) == 0, "ktx_mla_prep: unsupported rope_dim");
  KTX_REQUIRE(!d_kv || (d_kv_norm_w && d_ckv_out && d_kpe_out), "ktx_mla_prep: kv part needs norm weight and both outputs");
  KTX_REQUIRE(!d_kv || (kv_lora % 8 == 0 && kv_lora <= 2048 && kv_row_stride % 8 == 0), "ktx_mla_prep: kv_lora must be a multiple of 8, <= 2048");
  KTX_REQUIRE(!d_q || (d_q_pe_out && (nope_dim + rope_dim) % 2 == 0 && nope_dim % 2 == 0 && q_row_stride % 2 == 0), "ktx_mla_prep: bad q layout");
  if (T <= 0) return 0;
  PrepParams p{};
  p.T = T; p.H = num_heads; p.nope = nope_dim; p.rope = rope_dim; p.kvl = kv_lora;
  p.q = (const bf16_t*)d_q; p.q_rs = q_row_stride; p.q_pe = (bf16_t*)d_q_pe_out;
  p.kv = (const bf16_t*)d_kv; p.kv_rs = kv_row_stride; p.nw = (const bf16_t*)d_kv_norm_w; p.eps = eps;
  p.ckv = (bf16_t*)d_ckv_out; p.kpe = (bf16_t*)d_kpe_out;
  p.pos = d_pos; p.inv_freq = d_inv_freq; p.mscale = mscale;
  const int per = 256 / (rope_dim / 2);
  KTX_TIMED((hipStream_t)stream, (double)T * (num_heads * rope_dim * 4.0 + (kv_lora + rope_dim) * 4.0), "mla_prep_kernel T=%d Hq=%d", T, num_heads);
  hipLaunchKernelGGL(mla_prep_kernel, dim3(T, 1 + (d_q ? (num_heads + per - 1) / per : 0)), dim3(256), 0, (hipStream_t)stream, p);
  KTX_HIP(hipGetLastError());
  return 0;
}
