// ktx_mla.hip — MLA compressed-KV paged attention (absorbed form) for gfx950.  C ABI in include/ktx_mla.h.
//
// All Hq query heads share one latent K = [ckv(512) | k_pe(64)] and V = ckv, so decode attention is two skinny GEMMs per
// KV tile:  S[heads, tokens] = Q[heads, 576] K^T  and  O[heads, 512] += P[heads, tokens] V[tokens, 512], both on
// v_mfma_f32_16x16x32_bf16 with fp32 softmax statistics (flash-decoding: the KV range is split across workgroups and
// the partial (m, l, O) triples are merged by a second launch).
//
// Workgroup = NWV waves; each wave owns 16 heads (Q fragments and the 16x512 fp32 output tile live in registers), all
// waves share one staged 32-token KV tile in LDS:
//   Kt [32][576+8]  bf16 row-major (pad 16 B)  -> B operand of S = Q K^T  (16-byte reads, token = lane&15)
//   Vt [512][32+8]  bf16 transposed            -> B operand of O += P V   (16-byte reads, dim = lane&15)
//   Pt [NWV][16][32] bf16                      -> P re-laid from the MFMA C layout to the A layout (wave-private)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "../../include/ktx_mla.h"
#include "ktx_common.h"

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

#define MLA_TILE 32
#define MLA_DC 512
#define MLA_DR 64
#define MLA_KROW (MLA_DC + MLA_DR + 8)   // 584 elements = 1168 B
#define MLA_VROW (MLA_TILE + 8)          // 40 elements = 80 B

struct MlaParams {
  const bf16_t *q_nope, *q_pe, *ckv, *k_pe;
  long long ckv_ts, kpe_ts;
  const int32_t *qo_indptr, *kv_indptr, *kv_indices, *kv_len, *d_bsz;
  int batch, total_q, Hq, page_size, nsplit;
  float sm_scale;
  float* part_o;   // [total_q][Hq][nsplit][512]
  float* part_ml;  // [total_q][Hq][nsplit][2]
};

__device__ __forceinline__ v8bf as_v8bf(const uint4& u) {
  union { uint4 u; v8bf v; } c;
  c.u = u;
  return c.v;
}

template <int NWV>
__global__ __launch_bounds__(NWV * 64) void mla_decode_kernel(MlaParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  bf16_t* Kt = reinterpret_cast<bf16_t*>(smem);                        // [32][584]
  bf16_t* Vt = Kt + MLA_TILE * MLA_KROW;                               // [512][40]
  bf16_t* Pt = Vt + MLA_DC * MLA_VROW;                                 // [NWV][16][32]
  __shared__ int s_req[4];  // request id, kv_end, page base, first tile / tile count packed below
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x, hb = blockIdx.y, qt = blockIdx.z;
  int B = p.batch;
  if (p.d_bsz) B = min(max(*p.d_bsz, 0), p.batch);

  // which request owns query token qt?  (qo_indptr is tiny: linear scan by one lane)
  if (tid == 0) {
    int b = -1;
    for (int i = 0; i < B; i++)
      if (qt >= p.qo_indptr[i] && qt < p.qo_indptr[i + 1]) { b = i; break; }
    int kv_end = 0;
    if (b >= 0) {
      const int qo_len = p.qo_indptr[b + 1] - p.qo_indptr[b];
      kv_end = p.kv_len[b] - qo_len + (qt - p.qo_indptr[b]) + 1;   // causal: positions < kv_end are visible
      kv_end = max(kv_end, 0);
    }
    s_req[0] = b;
    s_req[1] = kv_end;
    s_req[2] = b >= 0 ? p.kv_indptr[b] : 0;
  }
  __syncthreads();
  const int req = s_req[0], kv_end = s_req[1], page_base = s_req[2];
  const int head0 = hb * NWV * 16 + wave * 16;
  const size_t pidx = ((size_t)qt * p.Hq + head0) * p.nsplit + split;  // + head*nsplit per head
  if (req < 0) return;

  const int ntiles = (kv_end + MLA_TILE - 1) / MLA_TILE;
  const int per = (ntiles + p.nsplit - 1) / p.nsplit;
  const int t_begin = split * per, t_end = min(ntiles, t_begin + per);

  // ---- Q fragments: A[m = head (lane&15)][k = (lane>>4)*8 .. +7] for 18 k-steps of 32 (16 nope + 2 rope) ------------
  v8bf qf[18];
  {
    const int h = head0 + (lane & 15);
    const bf16_t* qn = p.q_nope + ((size_t)qt * p.Hq + h) * MLA_DC + (lane >> 4) * 8;
    const bf16_t* qr = p.q_pe + ((size_t)qt * p.Hq + h) * MLA_DR + (lane >> 4) * 8;
#pragma unroll
    for (int s = 0; s < 16; s++) qf[s] = as_v8bf(*reinterpret_cast<const uint4*>(qn + s * 32));
#pragma unroll
    for (int s = 0; s < 2; s++) qf[16 + s] = as_v8bf(*reinterpret_cast<const uint4*>(qr + s * 32));
  }

  v4f o[32];
#pragma unroll
  for (int i = 0; i < 32; i++) o[i] = v4f{0.f, 0.f, 0.f, 0.f};
  float m_run[4], l_run[4];
#pragma unroll
  for (int r = 0; r < 4; r++) { m_run[r] = -__builtin_inff(); l_run[r] = 0.f; }
  bf16_t* Pw = Pt + wave * 16 * MLA_TILE;

  for (int tile = t_begin; tile < t_end; tile++) {
    const int tok0 = tile * MLA_TILE;
    __syncthreads();  // previous tile fully consumed
    // ---- stage the tile: 32 tokens x (512 + 64) bf16; each thread moves 16-byte pieces ----------------------------
    for (int u = tid; u < MLA_TILE * 72; u += NWV * 64) {  // 72 = 576/8 pieces per token
      const int tk = u / 72, piece = u % 72;
      const int pos = tok0 + tk;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (pos < kv_end) {
        const int page = p.kv_indices[page_base + pos / p.page_size];
        const size_t trow = (size_t)page * p.page_size + pos % p.page_size;
        if (piece < 64) v = *reinterpret_cast<const uint4*>(p.ckv + trow * p.ckv_ts + piece * 8);
        else v = *reinterpret_cast<const uint4*>(p.k_pe + trow * p.kpe_ts + (piece - 64) * 8);
      }
      *reinterpret_cast<uint4*>(Kt + tk * MLA_KROW + piece * 8) = v;
      if (piece < 64) {  // transposed copy for the PV GEMM
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
          Vt[(piece * 8 + 2 * q) * MLA_VROW + tk] = (bf16_t)(w[q] & 0xffffu);
          Vt[(piece * 8 + 2 * q + 1) * MLA_VROW + tk] = (bf16_t)(w[q] >> 16);
        }
      }
    }
    __syncthreads();

    // ---- S = Q K^T for 2 x 16 tokens ---------------------------------------------------------------------------------
    v4f s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    const bf16_t* kb0 = Kt + (lane & 15) * MLA_KROW + (lane >> 4) * 8;
    const bf16_t* kb1 = kb0 + 16 * MLA_KROW;
#pragma unroll
    for (int s = 0; s < 18; s++) {
      const v8bf b0 = as_v8bf(*reinterpret_cast<const uint4*>(kb0 + s * 32));
      const v8bf b1 = as_v8bf(*reinterpret_cast<const uint4*>(kb1 + s * 32));
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[s], b0, s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[s], b1, s1, 0, 0, 0);
    }
    // lane holds S[head = (lane>>4)*4 + r][token = tok0 + (lane&15) (+16)]
    const bool v0 = tok0 + (lane & 15) < kv_end, v1 = tok0 + 16 + (lane & 15) < kv_end;
    float alpha[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float a = v0 ? s0[r] * p.sm_scale : -__builtin_inff();
      const float b = v1 ? s1[r] * p.sm_scale : -__builtin_inff();
      float mx = fmaxf(a, b);
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      const float m_new = fmaxf(m_run[r], mx);        // finite: every processed tile has >= 1 visible token
      const float pa = __expf(a - m_new), pb = __expf(b - m_new);
      float sum = pa + pb;
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
      alpha[r] = __expf(m_run[r] - m_new);
      l_run[r] = l_run[r] * alpha[r] + sum;
      m_run[r] = m_new;
      // P to LDS in [head][token] order for the A-operand re-read
      const int hrow = (lane >> 4) * 4 + r;
      Pw[hrow * MLA_TILE + (lane & 15)] = f32_to_bf16(pa);
      Pw[hrow * MLA_TILE + 16 + (lane & 15)] = f32_to_bf16(pb);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const v8bf pf = as_v8bf(*reinterpret_cast<const uint4*>(Pw + (lane & 15) * MLA_TILE + (lane >> 4) * 8));

    // ---- O = O*alpha + P V --------------------------------------------------------------------------------------------
    const bf16_t* vb = Vt + (lane & 15) * MLA_VROW + (lane >> 4) * 8;
#pragma unroll
    for (int i = 0; i < 32; i++) {
#pragma unroll
      for (int r = 0; r < 4; r++) o[i][r] *= alpha[r];
      const v8bf b = as_v8bf(*reinterpret_cast<const uint4*>(vb + i * 16 * MLA_VROW));
      o[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, b, o[i], 0, 0, 0);
    }
  }

  // ---- partial results: un-normalised O plus (m, l) per head ------------------------------------------------------------
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int hrow = (lane >> 4) * 4 + r;
    float* po = p.part_o + (pidx + (size_t)hrow * p.nsplit) * MLA_DC;
#pragma unroll
    for (int i = 0; i < 32; i++) po[i * 16 + (lane & 15)] = o[i][r];
    if ((lane & 15) == 0) {
      float* ml = p.part_ml + (pidx + (size_t)hrow * p.nsplit) * 2;
      ml[0] = m_run[r];
      ml[1] = l_run[r];
    }
  }
}

// merge the KV splits: one wave per (query token, head); lane handles 8 of the 512 output dims
__global__ __launch_bounds__(64) void mla_merge_kernel(MlaParams p, bf16_t* out, float* lse) {
  const int qt = blockIdx.y, h = blockIdx.x, lane = threadIdx.x;
  int B = p.batch;
  if (p.d_bsz) B = min(max(*p.d_bsz, 0), p.batch);
  if (qt >= p.qo_indptr[B]) return;
  const size_t base = ((size_t)qt * p.Hq + h) * p.nsplit;
  float mstar = -__builtin_inff();
  for (int s = 0; s < p.nsplit; s++) {
    const float l = p.part_ml[(base + s) * 2 + 1];
    if (l > 0.f) mstar = fmaxf(mstar, p.part_ml[(base + s) * 2]);
  }
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float lsum = 0.f;
  for (int s = 0; s < p.nsplit; s++) {
    const float l = p.part_ml[(base + s) * 2 + 1];
    if (!(l > 0.f)) continue;
    const float w = __expf(p.part_ml[(base + s) * 2] - mstar);
    lsum += l * w;
    const float* po = p.part_o + (base + s) * MLA_DC + lane * 8;
    const float4 a = *reinterpret_cast<const float4*>(po), b = *reinterpret_cast<const float4*>(po + 4);
    acc[0] += a.x * w; acc[1] += a.y * w; acc[2] += a.z * w; acc[3] += a.w * w;
    acc[4] += b.x * w; acc[5] += b.y * w; acc[6] += b.z * w; acc[7] += b.w * w;
  }
  const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
  uint32_t w[4];
#pragma unroll
  for (int q = 0; q < 4; q++)
    w[q] = (uint32_t)f32_to_bf16(acc[2 * q] * inv) | ((uint32_t)f32_to_bf16(acc[2 * q + 1] * inv) << 16);
  *reinterpret_cast<uint4*>(out + ((size_t)qt * p.Hq + h) * MLA_DC + lane * 8) = make_uint4(w[0], w[1], w[2], w[3]);
  if (lse && lane == 0) lse[(size_t)qt * p.Hq + h] = lsum > 0.f ? (mstar + __logf(lsum)) * 1.44269504089f : -__builtin_inff();
}

__global__ void mla_cache_append_kernel(bf16_t* cache, long long ts, int page_size, const bf16_t* ckv, const bf16_t* kpe,
                                        const int32_t* page_idx, const int32_t* page_off, const int32_t* ntok,
                                        int max_tokens) {
  int T = max_tokens;
  if (ntok) T = min(max(*ntok, 0), max_tokens);
  const int t = blockIdx.x;
  if (t >= T) return;
  bf16_t* dst = cache + ((size_t)page_idx[t] * page_size + page_off[t]) * ts;
  const int i = threadIdx.x;  // 72 threads x 16 B = 576 bf16
  if (i < 64) *reinterpret_cast<uint4*>(dst + i * 8) = *reinterpret_cast<const uint4*>(ckv + (size_t)t * MLA_DC + i * 8);
  else if (i < 72) *reinterpret_cast<uint4*>(dst + i * 8) = *reinterpret_cast<const uint4*>(kpe + (size_t)t * MLA_DR + (i - 64) * 8);
}

extern "C" size_t ktx_mla_workspace_bytes(const ktx_mla_config* cfg, int max_q_tokens) {
  if (!cfg || max_q_tokens <= 0) return 0;
  const size_t n = (size_t)max_q_tokens * cfg->num_heads * std::max(1, cfg->max_splits);
  return n * (MLA_DC + 2) * sizeof(float);
}

extern "C" int ktx_mla_decode(const ktx_mla_config* cfg, const void* d_q_nope, const void* d_q_pe, const void* d_ckv,
                              const void* d_k_pe, int64_t ckv_token_stride, int64_t kpe_token_stride,
                              const int32_t* d_qo_indptr, const int32_t* d_kv_indptr, const int32_t* d_kv_indices,
                              const int32_t* d_kv_len_arr, const int32_t* d_bsz, int batch, int total_q_tokens,
                              void* d_out, float* d_lse, void* d_workspace, size_t workspace_bytes, void* stream) {
  KTX_REQUIRE(cfg && d_q_nope && d_q_pe && d_ckv && d_k_pe && d_out && d_workspace, "ktx_mla_decode: null pointer");
  KTX_REQUIRE(d_qo_indptr && d_kv_indptr && d_kv_indices && d_kv_len_arr, "ktx_mla_decode: null index array");
  KTX_REQUIRE(cfg->head_dim_ckv == MLA_DC && cfg->head_dim_kpe == MLA_DR, "ktx_mla_decode: only kv_lora_rank 512 + rope 64");
  KTX_REQUIRE(cfg->num_heads > 0 && cfg->num_heads % 16 == 0, "ktx_mla_decode: num_heads must be a multiple of 16");
  KTX_REQUIRE(batch > 0 && total_q_tokens > 0 && cfg->page_size > 0, "ktx_mla_decode: bad sizes");
  KTX_REQUIRE(ckv_token_stride % 8 == 0 && kpe_token_stride % 8 == 0, "ktx_mla_decode: token strides must be multiples of 8 elements");
  hipStream_t st = (hipStream_t)stream;
  const int Hq = cfg->num_heads;
  const int nwv = (Hq % 64 == 0) ? 4 : 1;
  const int hblocks = Hq / (16 * nwv);
  // enough workgroups to cover the chip a few times over, bounded by the workspace
  int nsplit = std::max(1, 1024 / std::max(1, hblocks * total_q_tokens));
  nsplit = std::min(nsplit, std::max(1, cfg->max_splits));
  const size_t need = (size_t)total_q_tokens * Hq * nsplit * (MLA_DC + 2) * sizeof(float);
  KTX_REQUIRE(workspace_bytes >= need, "ktx_mla_decode: workspace too small");
  MlaParams p;
  p.q_nope = (const bf16_t*)d_q_nope; p.q_pe = (const bf16_t*)d_q_pe; p.ckv = (const bf16_t*)d_ckv; p.k_pe = (const bf16_t*)d_k_pe;
  p.ckv_ts = ckv_token_stride; p.kpe_ts = kpe_token_stride;
  p.qo_indptr = d_qo_indptr; p.kv_indptr = d_kv_indptr; p.kv_indices = d_kv_indices; p.kv_len = d_kv_len_arr; p.d_bsz = d_bsz;
  p.batch = batch; p.total_q = total_q_tokens; p.Hq = Hq; p.page_size = cfg->page_size; p.nsplit = nsplit;
  p.sm_scale = cfg->sm_scale;
  p.part_o = (float*)d_workspace;
  p.part_ml = p.part_o + (size_t)total_q_tokens * Hq * nsplit * MLA_DC;
  const size_t lds = (size_t)(MLA_TILE * MLA_KROW + MLA_DC * MLA_VROW + nwv * 16 * MLA_TILE) * sizeof(bf16_t);
  const dim3 grid(nsplit, hblocks, total_q_tokens);
  if (nwv == 4) {
    static hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void*>(mla_decode_kernel<4>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    KTX_HIP(e4);
    hipLaunchKernelGGL(mla_decode_kernel<4>, grid, dim3(256), lds, st, p);
  } else {
    static hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(mla_decode_kernel<1>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    KTX_HIP(e1);
    hipLaunchKernelGGL(mla_decode_kernel<1>, grid, dim3(64), lds, st, p);
  }
  KTX_HIP(hipGetLastError());
  hipLaunchKernelGGL(mla_merge_kernel, dim3(Hq, total_q_tokens), dim3(64), 0, st, p, (bf16_t*)d_out, d_lse);
  KTX_HIP(hipGetLastError());
  return 0;
}

extern "C" int ktx_mla_cache_append(const ktx_mla_config* cfg, void* d_kv_cache, int64_t token_stride,
                                    const void* d_ckv_new, const void* d_kpe_new, const int32_t* d_page_idx,
                                    const int32_t* d_page_offset, const int32_t* d_ntokens, int max_tokens, void* stream) {
  KTX_REQUIRE(cfg && d_kv_cache && d_ckv_new && d_kpe_new && d_page_idx && d_page_offset && max_tokens > 0,
              "ktx_mla_cache_append: bad argument");
  KTX_REQUIRE(token_stride >= MLA_DC + MLA_DR && token_stride % 8 == 0, "ktx_mla_cache_append: bad token stride");
  hipLaunchKernelGGL(mla_cache_append_kernel, dim3(max_tokens), dim3(128), 0, (hipStream_t)stream, (bf16_t*)d_kv_cache,
                     (long long)token_stride, cfg->page_size, (const bf16_t*)d_ckv_new, (const bf16_t*)d_kpe_new,
                     d_page_idx, d_page_offset, d_ntokens, max_tokens);
  KTX_HIP(hipGetLastError());
  return 0;
}
