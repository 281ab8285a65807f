// ktx_gate.hip — MoE router for gfx950.  C ABI in include/ktx_gate.h.
//
// Restates MoEGate.forward (archive/ktransformers/models/modeling_deepseek_v3.py:430-481 and the V2 variant
// archive/ktransformers/models/modeling_deepseek.py:413-455) as two launches: an HBM/L2-bound fp32 GEMV per (token,
// expert) and a one-wavefront-per-token selection kernel that keeps all E scores in registers.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/ktx_gate.h"
#include "ktx_common.h"

#define KTX_GATE_MAX_E 1024  // 16 scores per lane

// ---- logits: one wavefront per (token, expert); 16-byte loads of both bf16 rows, fp32 FMA, butterfly reduce --------
__global__ __launch_bounds__(256) void gate_logits_kernel(const int32_t* d_bsz, int qlen, int E, int H,
                                                          const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          float* __restrict__ logits) {
  int T = qlen;
  if (d_bsz) T = min(max(*d_bsz, 0), qlen);
  const int t = blockIdx.y;
  if (t >= T) return;
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= E) return;
  const bf16_t* xr = x + (size_t)t * H;
  const bf16_t* wr = w + (size_t)e * H;
  float acc = 0.0f;
  for (int j = lane * 8; j < H; j += 512) {
    const uint4 a = *reinterpret_cast<const uint4*>(xr + j);
    const uint4 b = *reinterpret_cast<const uint4*>(wr + j);
    const uint32_t av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int q = 0; q < 4; q++) acc = ktx_dot2_bf16(av[q], bv[q], acc);   // v_dot2c_f32_bf16: two products per instruction
  }
  acc = wave_sum(acc);   // same order and tree as gate_fused_kernel: the two paths give identical logits
  if (lane == 0) logits[(size_t)t * E + e] = acc;
}

// ---- selection ---------------------------------------------------------------------------------------------------------
// Wave-wide reductions of the selection run on DPP + v_readlane (ktx_common.h), not on ds_bpermute shuffles.
#define dpp_i ktx_dpp_i
#define dpp_f ktx_dpp_f
#define gate_wave_max wave_max
#define gate_wave_sum wave_sum
// (value, index) argmax over the wave; ties -> lower index.  Every lane returns the winner.
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#define KTX_ARGMAX_STEP(CTRL)                                               \
  {                                                                         \
    const float ov = dpp_f<CTRL>(v);                                        \
    const int oi = dpp_i<CTRL>(i);                                          \
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }                  \
  }
  KTX_ARGMAX_STEP(KTX_DPP_QUAD_1032)
  KTX_ARGMAX_STEP(KTX_DPP_QUAD_2301)
  KTX_ARGMAX_STEP(KTX_DPP_ROW_HALF_MIRROR)
  KTX_ARGMAX_STEP(KTX_DPP_ROW_MIRROR)
#undef KTX_ARGMAX_STEP
  const int vb = __float_as_int(v);
  float bv = __int_as_float(__builtin_amdgcn_readlane(vb, 0));
  int bi = __builtin_amdgcn_readlane(i, 0);
#pragma unroll
  for (int r = 1; r < 4; r++) {
    const float ov = __int_as_float(__builtin_amdgcn_readlane(vb, 16 * r));
    const int oi = __builtin_amdgcn_readlane(i, 16 * r);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  v = bv;
  i = bi;
}

template <int EPL>  // scores per lane: expert e lives on lane e % 64, slot e / 64 (E <= 64*EPL)
__device__ __forceinline__ void gate_select_token(const ktx_gate_config& c, int t, int lane, const float* __restrict__ logits,
                                                  const float* __restrict__ bias, int64_t* __restrict__ topk_idx,
                                                  float* __restrict__ topk_w) {
  const int E = c.n_routed_experts;
  const float NEG = -__builtin_inff();
  float score[EPL], choice[EPL];
  // scores (modeling_deepseek_v3.py:438-444 sigmoid; modeling_deepseek.py:421-424 softmax in fp32)
  float mx = NEG;
#pragma unroll
  for (int s = 0; s < EPL; s++) {
    const int e = s * 64 + lane;
    score[s] = e < E ? logits[e] : NEG;   // `logits` = this token's row
    mx = fmaxf(mx, score[s]);
  }
  if (c.scoring == KTX_GATE_SOFTMAX) {
    mx = gate_wave_max(mx);
    float sum = 0.0f;
#pragma unroll
    for (int s = 0; s < EPL; s++) {
      score[s] = (s * 64 + lane < E) ? expf(score[s] - mx) : 0.0f;
      sum += score[s];
    }
    sum = gate_wave_sum(sum);
#pragma unroll
    for (int s = 0; s < EPL; s++) score[s] = score[s] / sum;
  } else {
#pragma unroll
    for (int s = 0; s < EPL; s++) score[s] = 1.0f / (1.0f + expf(-score[s]));
  }
#pragma unroll
  for (int s = 0; s < EPL; s++) {
    const int e = s * 64 + lane;
    choice[s] = e < E ? score[s] + ((bias && c.topk_method == KTX_GATE_NOAUX_TC) ? bias[e] : 0.0f) : NEG;
  }

  // group limitation (modeling_deepseek_v3.py:449-468 / modeling_deepseek.py:431-448)
  if (c.topk_method != KTX_GATE_GREEDY && c.n_group > 1) {
    const int gsz = E / c.n_group;
    const bool top2 = c.topk_method == KTX_GATE_NOAUX_TC;   // group score: sum of the group's top-2 (V3) | group max (V2)
    unsigned long long keep = 0ull;
    if (gsz == 32 && 2 * EPL <= 64) {
      // DeepSeek-V3 / R1 (256 experts, 8 groups): the group of expert (slot s, lane) is 2s + (lane >> 5), so every slot
      // yields two group scores from 16-lane DPP reductions — no loop over groups, no per-group argmax passes.
      float gsc[2 * EPL];
#pragma unroll
      for (int s = 0; s < EPL; s++) {
        const float v = choice[s];
        const int r1 = __float_as_int(row16_max(v));
        const float h0 = fmaxf(__int_as_float(__builtin_amdgcn_readlane(r1, 0)), __int_as_float(__builtin_amdgcn_readlane(r1, 16)));
        const float h1 = fmaxf(__int_as_float(__builtin_amdgcn_readlane(r1, 32)), __int_as_float(__builtin_amdgcn_readlane(r1, 48)));
        float q0 = 0.0f, q1 = 0.0f;
        if (top2) {   // second largest = max after retiring ONE instance of the largest (a tied pair counts twice, like topk(2))
          const unsigned long long hit = __ballot(v == (lane < 32 ? h0 : h1));
          const int f0 = __ffs((int)(unsigned)(hit & 0xffffffffull)) - 1, f1 = 32 + __ffs((int)(unsigned)(hit >> 32)) - 1;
          const float v2 = (lane == (lane < 32 ? f0 : f1)) ? NEG : v;
          const int r2 = __float_as_int(row16_max(v2));
          q0 = fmaxf(__int_as_float(__builtin_amdgcn_readlane(r2, 0)), __int_as_float(__builtin_amdgcn_readlane(r2, 16)));
          q1 = fmaxf(__int_as_float(__builtin_amdgcn_readlane(r2, 32)), __int_as_float(__builtin_amdgcn_readlane(r2, 48)));
        }
        gsc[2 * s] = top2 ? h0 + q0 : h0;
        gsc[2 * s + 1] = top2 ? h1 + q1 : h1;
      }
      // the topk_group best groups by rank (ties -> lower index); wave-uniform arithmetic
#pragma unroll
      for (int g = 0; g < 2 * EPL; g++) {
        int rank = 0;
#pragma unroll
        for (int o = 0; o < 2 * EPL; o++)
          rank += (o < c.n_group && o != g && (gsc[o] > gsc[g] || (gsc[o] == gsc[g] && o < g))) ? 1 : 0;
        if (g < c.n_group && rank < c.topk_group) keep |= 1ull << g;
      }
    } else {
      int grp[EPL];
#pragma unroll
      for (int s = 0; s < EPL; s++) grp[s] = (s * 64 + lane) / gsz;
      float gscore = NEG;  // lane g < n_group holds group g's score
      for (int g = 0; g < c.n_group; g++) {
        float v1 = NEG, v2 = NEG;  // per-lane top-2 inside group g
#pragma unroll
        for (int s = 0; s < EPL; s++) {
          if (s * 64 + lane < E && grp[s] == g) {
            const float v = choice[s];
            if (v > v1) { v2 = v1; v1 = v; } else if (v > v2) { v2 = v; }
          }
        }
        // wave top-2 via two argmax passes
        float m1 = v1; int i1 = lane;
        wave_argmax(m1, i1);
        float cand = (lane == i1) ? v2 : v1;
        int i2 = lane;
        wave_argmax(cand, i2);
        const float gs = top2 ? (m1 + cand) : m1;
        if (lane == g) gscore = gs;
      }
      // pick topk_group groups; everything outside them is masked out
      float gs = (lane < c.n_group) ? gscore : NEG;
      for (int r = 0; r < c.topk_group; r++) {
        float v = gs; int i = lane;
        wave_argmax(v, i);
        keep |= 1ull << i;
        if (lane == i) gs = NEG;
      }
    }
    const float masked = top2 ? NEG : 0.0f;  // V3 fills -inf, V2 fills 0.0
#pragma unroll
    for (int s = 0; s < EPL; s++) {
      const int e = s * 64 + lane;
      if (e < E && !((keep >> (e / gsz)) & 1ull)) choice[s] = masked;
    }
  }

  // top-k experts, descending choice score, ties -> lower index
  float wsum = 0.0f;
  float myw = 0.0f;   // lane r keeps the weight of the r-th pick
  int myi = 0;
  for (int r = 0; r < c.top_k; r++) {
    float bv = NEG; int bi = 0x7fffffff;
#pragma unroll
    for (int s = 0; s < EPL; s++) {
      const int e = s * 64 + lane;
      if (e < E && (choice[s] > bv || (choice[s] == bv && e < bi))) { bv = choice[s]; bi = e; }
    }
    wave_argmax(bv, bi);
    // gather the UNBIASED score of the winner (modeling_deepseek_v3.py:472) and retire it
    float sc = 0.0f;
#pragma unroll
    for (int s = 0; s < EPL; s++)
      if (s * 64 + lane == bi) { sc = score[s]; choice[s] = NEG; }
    sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc), bi & 63));
    // V2 (modeling_deepseek.py:426-448) takes the weight straight from the (masked) score it ranked on; V3 gathers
    // the unbiased score of the winner (modeling_deepseek_v3.py:472)
    if (c.topk_method != KTX_GATE_NOAUX_TC) sc = bv;
    wsum += sc;
    if (lane == r) { myw = sc; myi = bi; }
  }
  if (lane < c.top_k) {
    float wv = myw;
    if (c.top_k > 1 && c.norm_topk_prob) {
      wv = wv / (wsum + 1e-20f);
      if (c.topk_method == KTX_GATE_NOAUX_TC) wv = wv * c.routed_scaling_factor;  // V3 always scales (:481)
    } else {
      wv = wv * c.routed_scaling_factor;  // V3 :481; V2 scales only when it does not normalise (:453-455)
    }
    topk_idx[(size_t)t * c.top_k + lane] = myi;
    topk_w[(size_t)t * c.top_k + lane] = wv;
  }
}

template <int EPL>
__global__ __launch_bounds__(64) void gate_select_kernel(ktx_gate_config c, const int32_t* d_bsz, int qlen,
                                                         const float* __restrict__ logits, const float* __restrict__ bias,
                                                         int64_t* __restrict__ topk_idx, float* __restrict__ topk_w) {
  int T = qlen;
  if (d_bsz) T = min(max(*d_bsz, 0), qlen);
  const int t = blockIdx.x;
  if (t >= T) return;
  gate_select_token<EPL>(c, t, threadIdx.x, logits + (size_t)t * c.n_routed_experts, bias, topk_idx, topk_w);
}

// ---- fused router for decode-sized batches: logits GEMV + selection in ONE launch -----------------------------------------
// Every workgroup computes 4 experts' logits for one token; the last workgroup of a token to finish (arrival ticket on a
// per-token counter) performs the selection.  Hand-off WITHOUT fences (MI355X_MICROARCH.md, "handoff-flag": sc1 payload ->
// s_waitcnt vmcnt(0) -> flag): the logits are written with write-through `sc1` stores (relaxed agent-scope atomic stores),
// drained, then the ticket is taken; the last arriver reads them back with `sc1` loads, which bypass its L1 — no
// buffer_wbl2 / buffer_inv (~3.5 us the pair) is needed because neither side keeps the payload in a non-coherent cache.
// No workgroup ever waits, so residency does not matter.  The counter is reset by the last arriver (zeroed at allocation),
// keeping the launch graph-replayable without a memset node.  (A single-workgroup router was tried for the 64 x 2048
// DeepSeek-V2-Lite gate: one CU pulls only ~75 GB/s, 4.6-6.7 us for the 256 KiB, vs 2.4-2.9 us on 16 CUs —
// scripts/gate_probe.hip.)
template <int EPL, int NJ>   // NJ = 512-column blocks held in registers by the fused-RMSNorm variant (0: no norm)
__global__ __launch_bounds__(256) void gate_fused_kernel(ktx_gate_config c, const int32_t* d_bsz, int qlen,
                                                         const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ logits,
                                                         int32_t* __restrict__ counters, int64_t* __restrict__ topk_idx,
                                                         float* __restrict__ topk_w, const bf16_t* __restrict__ norm_w,
                                                         float norm_eps, bf16_t* __restrict__ xn_out) {
  __shared__ int s_last;
  __shared__ float s_red[4];
  __shared__ float s_logits[KTX_GATE_MAX_E];
  extern __shared__ __attribute__((aligned(16))) uint8_t gate_smem[];   // NJ > 0: the normalised row, bf16 [H]
  uint4* xs = reinterpret_cast<uint4*>(gate_smem);
  int T = qlen;
  if (d_bsz) T = min(max(*d_bsz, 0), qlen);
  const int t = blockIdx.y;
  if (t >= T) return;
  const int E = c.n_routed_experts, H = c.hidden_size;
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  // (NJ > 0: the workgroup-wide barriers below need every wavefront, so an out-of-range expert clamps to the last row and
  //  simply does not store its logit)
  const bool e_ok = e < E;
  if (e_ok || NJ > 0) {
    const bf16_t* xr = x + (size_t)t * H;
    const bf16_t* wr = w + (size_t)(e_ok ? e : E - 1) * H;
    float acc = 0.0f;
    if constexpr (NJ > 0) {
      // fused post_attention_layernorm (H <= 512*NJ).  The workgroup normalises the row ONCE into LDS — 256 threads x <= 4
      // pieces of 8 (DeepseekV3RMSNorm: w * bf16(x * r), both roundings) — and each wavefront then takes its expert's dot
      // product against the LDS copy.  (The first version had every wavefront normalise the whole row in registers: 112
      // elements per lane at H = 7168 unrolled into ~12k instructions — 100 KB of code, more than the instruction cache,
      // fetched cold by every launch: 78 us per call inside a DeepSeek-V3 decode step.)  Workgroup 0 writes the normalised
      // row out for the experts that run after the router.  The router-row loads are issued first: they depend on nothing.
      uint4 b[NJ];
#pragma unroll
      for (int u = 0; u < NJ; u++) {
        const int j = lane * 8 + u * 512;
        b[u] = j < H ? *reinterpret_cast<const uint4*>(wr + j) : make_uint4(0, 0, 0, 0);
      }
      const int npiece = H >> 3;
      uint4 xa[4], wn[4];
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int p = threadIdx.x + i * 256;
        const bool ok = p < npiece;
        xa[i] = ok ? *reinterpret_cast<const uint4*>(xr + p * 8) : make_uint4(0, 0, 0, 0);
        wn[i] = ok ? *reinterpret_cast<const uint4*>(norm_w + p * 8) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint32_t av[4] = {xa[i].x, xa[i].y, xa[i].z, xa[i].w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float lo = bf16_to_f32((bf16_t)(av[q] & 0xffffu)), hi = bf16_to_f32((bf16_t)(av[q] >> 16));
          ss += lo * lo + hi * hi;
        }
      }
      ss = wave_sum(ss);
      if (lane == 0) s_red[threadIdx.x >> 6] = ss;
      __syncthreads();
      const float r = 1.0f / sqrtf((((s_red[0] + s_red[1]) + s_red[2]) + s_red[3]) / (float)H + norm_eps);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int p = threadIdx.x + i * 256;
        if (p < npiece) {
          const uint4 ov = make_uint4(ktx_norm_pk(xa[i].x, r, wn[i].x), ktx_norm_pk(xa[i].y, r, wn[i].y),
                                      ktx_norm_pk(xa[i].z, r, wn[i].z), ktx_norm_pk(xa[i].w, r, wn[i].w));
          xs[p] = ov;
          if (xn_out && blockIdx.x == 0) *reinterpret_cast<uint4*>(xn_out + (size_t)t * H + p * 8) = ov;
        }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < NJ; u++) {
        const int j = lane * 8 + u * 512;
        if (j < H) {
          const uint4 a = xs[lane + u * 64];
          const uint32_t av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b[u].x, b[u].y, b[u].z, b[u].w};
#pragma unroll
          for (int q = 0; q < 4; q++) acc = ktx_dot2_bf16(av[q], bv[q], acc);
        }
      }
    } else
    for (int j0 = lane * 8; j0 < H; j0 += 512 * 8) {   // 8 column blocks' loads in flight before the first FMA
      uint4 a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int j = j0 + u * 512;
        a[u] = j < H ? *reinterpret_cast<const uint4*>(xr + j) : make_uint4(0, 0, 0, 0);
        b[u] = j < H ? *reinterpret_cast<const uint4*>(wr + j) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t av[4] = {a[u].x, a[u].y, a[u].z, a[u].w}, bv[4] = {b[u].x, b[u].y, b[u].z, b[u].w};
#pragma unroll
        for (int q = 0; q < 4; q++) acc = ktx_dot2_bf16(av[q], bv[q], acc);
      }
    }
    acc = wave_sum(acc);
    if (lane == 0 && e_ok) __hip_atomic_store(&logits[(size_t)t * E + e], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1 store
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int ticket = __hip_atomic_fetch_add(&counters[t], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == (int)gridDim.x - 1;
    if (last) {
      __hip_atomic_store(&counters[t], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  for (int i = threadIdx.x; i < E; i += 256)   // sc1 loads: served past this CU's L1
    s_logits[i] = __hip_atomic_load(&logits[(size_t)t * E + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (threadIdx.x < 64) gate_select_token<EPL>(c, t, threadIdx.x, s_logits, bias, topk_idx, topk_w);
}

extern "C" int ktx_gate_logits(const ktx_gate_config* cfg, const int32_t* d_bsz, int qlen, const void* d_x,
                               const void* d_w, float* d_logits, void* stream) {
  KTX_REQUIRE(cfg && d_x && d_w && d_logits && qlen > 0, "ktx_gate_logits: bad argument");
  KTX_REQUIRE(cfg->hidden_size % 8 == 0, "ktx_gate_logits: hidden_size must be a multiple of 8");
  const dim3 grid((cfg->n_routed_experts + 3) / 4, qlen);
  hipLaunchKernelGGL(gate_logits_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_bsz, qlen, cfg->n_routed_experts,
                     cfg->hidden_size, (const bf16_t*)d_x, (const bf16_t*)d_w, d_logits);
  KTX_HIP(hipGetLastError());
  return 0;
}

static int gate_forward_impl(const ktx_gate_config* cfg, const int32_t* d_bsz, int qlen, const void* d_x, const void* d_w,
                             const float* d_bias, float* d_logits, int32_t* d_counters, int64_t* d_topk_idx,
                             float* d_topk_weight, const void* d_norm_w, float norm_eps, void* d_xn_out, void* stream) {
  KTX_REQUIRE(cfg && d_x && d_w && d_logits && d_counters && d_topk_idx && d_topk_weight && qlen > 0, "ktx_gate_forward: bad argument");
  const int E = cfg->n_routed_experts;
  KTX_REQUIRE(cfg->hidden_size % 8 == 0, "ktx_gate_forward: hidden_size must be a multiple of 8");
  KTX_REQUIRE(E > 0 && E <= KTX_GATE_MAX_E, "ktx_gate_forward: n_routed_experts out of range (1..1024)");
  KTX_REQUIRE(cfg->top_k > 0 && cfg->top_k <= 64 && cfg->top_k <= E, "ktx_gate_forward: top_k out of range");
  KTX_REQUIRE(cfg->n_group >= 1 && cfg->n_group <= 64 && E % cfg->n_group == 0, "ktx_gate_forward: bad n_group");
  KTX_REQUIRE(cfg->topk_group >= 1 && cfg->topk_group <= cfg->n_group, "ktx_gate_forward: bad topk_group");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((E + 3) / 4, qlen);
  const int epl = (E + 63) / 64;
#define KTX_FUSED2(N, J) hipLaunchKernelGGL((gate_fused_kernel<N, J>), grid, dim3(256), (J) > 0 ? (size_t)cfg->hidden_size * 2 : 0, st, *cfg, d_bsz, qlen, (const bf16_t*)d_x, (const bf16_t*)d_w, d_bias, d_logits, d_counters, d_topk_idx, d_topk_weight, (const bf16_t*)d_norm_w, norm_eps, (bf16_t*)d_xn_out)
#define KTX_FUSED(N)                                                          \
  do {                                                                        \
    if (!d_norm_w) KTX_FUSED2(N, 0);                                          \
    else if (cfg->hidden_size <= 2048) KTX_FUSED2(N, 4);                      \
    else if (cfg->hidden_size <= 4096) KTX_FUSED2(N, 8);                      \
    else KTX_FUSED2(N, 16);                                                   \
  } while (0)
  KTX_REQUIRE(!d_norm_w || cfg->hidden_size <= 8192, "ktx_gate_forward_norm: the fused RMSNorm needs hidden_size <= 8192");
  KTX_TIMED(st, (double)E * cfg->hidden_size * 2.0 + (double)qlen * cfg->hidden_size * (d_norm_w ? 6.0 : 2.0) + E * 8.0,
            "gate_fused_kernel%s T=%d E=%d H=%d", d_norm_w ? "+rmsnorm" : "", qlen, E, cfg->hidden_size);
  if (epl <= 1) KTX_FUSED(1);
  else if (epl <= 2) KTX_FUSED(2);
  else if (epl <= 4) KTX_FUSED(4);
  else if (epl <= 6) KTX_FUSED(6);
  else if (epl <= 8) KTX_FUSED(8);
  else KTX_FUSED(16);
#undef KTX_FUSED2
#undef KTX_FUSED
  KTX_HIP(hipGetLastError());
  return 0;
}

extern "C" int ktx_gate_forward(const ktx_gate_config* cfg, const int32_t* d_bsz, int qlen, const void* d_x, const void* d_w,
                                const float* d_bias, float* d_logits, int32_t* d_counters, int64_t* d_topk_idx,
                                float* d_topk_weight, void* stream) {
  return gate_forward_impl(cfg, d_bsz, qlen, d_x, d_w, d_bias, d_logits, d_counters, d_topk_idx, d_topk_weight, nullptr, 0.f,
                           nullptr, stream);
}

extern "C" int ktx_gate_forward_norm(const ktx_gate_config* cfg, const int32_t* d_bsz, int qlen, const void* d_x,
                                     const void* d_norm_weight, float norm_eps, void* d_xn_out, const void* d_w,
                                     const float* d_bias, float* d_logits, int32_t* d_counters, int64_t* d_topk_idx,
                                     float* d_topk_weight, void* stream) {
  KTX_REQUIRE(d_norm_weight && d_xn_out, "ktx_gate_forward_norm: null norm weight / output");
  return gate_forward_impl(cfg, d_bsz, qlen, d_x, d_w, d_bias, d_logits, d_counters, d_topk_idx, d_topk_weight, d_norm_weight,
                           norm_eps, d_xn_out, stream);
}

extern "C" int ktx_gate_select(const ktx_gate_config* cfg, const int32_t* d_bsz, int qlen, const float* d_logits,
                               const float* d_bias, int64_t* d_topk_idx, float* d_topk_weight, void* stream) {
  KTX_REQUIRE(cfg && d_logits && d_topk_idx && d_topk_weight && qlen > 0, "ktx_gate_select: bad argument");
  const int E = cfg->n_routed_experts;
  KTX_REQUIRE(E > 0 && E <= KTX_GATE_MAX_E, "ktx_gate_select: n_routed_experts out of range (1..1024)");
  KTX_REQUIRE(cfg->top_k > 0 && cfg->top_k <= 64 && cfg->top_k <= E, "ktx_gate_select: top_k out of range");
  KTX_REQUIRE(cfg->n_group >= 1 && cfg->n_group <= 64 && E % cfg->n_group == 0, "ktx_gate_select: bad n_group");
  KTX_REQUIRE(cfg->topk_group >= 1 && cfg->topk_group <= cfg->n_group, "ktx_gate_select: bad topk_group");
  hipStream_t st = (hipStream_t)stream;
  const int epl = (E + 63) / 64;
#define KTX_SEL(N) hipLaunchKernelGGL(gate_select_kernel<N>, dim3(qlen), dim3(64), 0, st, *cfg, d_bsz, qlen, d_logits, d_bias, d_topk_idx, d_topk_weight)
  if (epl <= 1) KTX_SEL(1);
  else if (epl <= 2) KTX_SEL(2);
  else if (epl <= 4) KTX_SEL(4);
  else if (epl <= 6) KTX_SEL(6);
  else if (epl <= 8) KTX_SEL(8);
  else KTX_SEL(16);
#undef KTX_SEL
  KTX_HIP(hipGetLastError());
  return 0;
}
