// ktx_moe.hip — routed-expert MoE forward for gfx950 (MI355X), hand-written HIP + MFMA.  C ABI in include/ktx_moe.h.
//
// Restates on the GPU the arithmetic of the reference's CPU expert path (file:line relative to
// /root/reference/kt-kernel/):
//   token->expert bucketing        operators/amx/moe_base.hpp:208-319        -> moe_prep_kernel (block 0)
//   activation quant (per row i8)  operators/amx/la/amx_buffers.hpp:47-98    -> moe_prep_kernel / moe_actquant_kernel
//   int4/int8 GEMM, exact int32    operators/amx/la/amx_kernels.hpp:1735-1846,2482-2506 -> moe_gemm_kernel (MFMA i8)
//   fp32 -> bf16 after each GEMM   operators/amx/la/amx_buffers.hpp:1716-1732 -> epilogue of moe_gemm_kernel
//   silu(gate)*up, poly exp        operators/amx/moe_base.hpp:693-726, la/amx.hpp:22-76 -> epilogue (GATE_UP)
//   weighted combine, slot order   operators/amx/moe_base.hpp:413-436,620-638 -> moe_combine_kernel
//   merge (+incremental) -> bf16   operators/amx/moe_base.hpp:749-791        -> moe_combine_kernel
//   load-time int4/int8 quantiser  operators/amx/la/amx_buffers.hpp:527-627, la/amx_kernels.hpp:1103-1150
//                                                                             -> quant_rows_kernel + pack_w*_kernel
//
// HBM layout of one weight matrix [N][K] ("W tile layout"; N%16==0, K%128==0):
//   strip s = n/16, k-step ks = k/128.  Tile (s,ks) is contiguous; lane l = kc*16 + i of a wavefront (i = row within
//   the strip, kc = 0..3) owns the MFMA A-fragments of both 64-deep halves h of the k-step, where fragment element
//   e (0..15) multiplies k = ks*128 + kc*32 + h*16 + e.
//     W4: tile = 1024 B, lane l -> 16 B = dwords P[0..3]; half h uses P[2h], P[2h+1]:
//           P[2h  ].byte[b] = (q[e=4+b] & 0xF0)  | ((q[e=b]   >> 4) & 0x0F)
//           P[2h+1].byte[b] = (q[e=12+b] & 0xF0) | ((q[e=8+b] >> 4) & 0x0F)        q = nibble*16 (int8)
//         so the four int8 fragment registers are ((P<<4)&0xF0F0F0F0, P&0xF0F0F0F0) per dword — 3 VALU ops / 8 weights.
//     W8: tile = 2048 B, half h at +h*1024, lane l -> 16 int8 = the fragment itself.
//   One wavefront streams a whole strip (all k-steps) with 1 KiB fully-coalesced dwordx4 loads.
// Activations are int8 [rows][K]; a chunk of KC = SPC*128 k is staged in LDS as [mt][col=KC/16][tok=16][16 B] so a
// B-fragment ds_read_b128 (lane = kc*16+tok) is bank-conflict free, and staging stores are 1 KiB contiguous per wave.
#include <hip/hip_runtime.h>
#include <type_traits>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ktx_moe.h"
#include "ktx_common.h"
#include "ktx_internal.h"

// =====================================================================================================
// error slot
// =====================================================================================================
std::string& ktx_err_slot() {
  static thread_local std::string s;
  return s;
}
int ktx_fail(const std::string& msg) {
  ktx_err_slot() = msg;
  return -1;
}
extern "C" const char* ktx_last_error(void) { return ktx_err_slot().c_str(); }

// =====================================================================================================
// device helpers
// =====================================================================================================

#include "ktx_moe_dec.inc"
namespace ktxw4 {
#include "ktx_w4_step.inc"
}

struct Tile {
  int expert;  // local expert index
  int row0;    // first sorted row
  int nrows;   // <= 16*MT
  int pad;
};

// =====================================================================================================
// K0: prep — block 0 buckets (t,j) pairs by expert; blocks 1.. quantise one token's activations (a5, a6)
// =====================================================================================================
#define KTX_EMAX 1024

struct PrepParams {
  const int32_t* d_bsz;
  int qlen, k, E, expert_begin, H;
  int rows_per_tile;  // 16*MT
  const int64_t* ids;
  const uint8_t* mask;
  const bf16_t* x;
  int8_t* x_q;
  float* x_d;
  int32_t* row_of_pair;  // [qlen*k]  -> sorted row or -1
  int32_t* src_of_row;   // [qlen*k]  sorted row -> token index
  Tile* tiles;
  int32_t* counters;  // [0] = num_tiles, [1] = num_rows
  int bm_words;       // set by launch_moe_prep: 32-token words per expert in one window of the scatter's bitmaps
};

__device__ __forceinline__ void quant_row_block(const bf16_t* __restrict__ src, int K, int8_t* __restrict__ dst,
                                                float* __restrict__ d_out, float* red /* LDS [nwaves] */) {
  // per row: d = amax/127, id = d ? 1/d : 0, q = sat8(rne(x*id))   (amx_buffers.hpp:47-98)
  const int tid = threadIdx.x, nthr = blockDim.x;
  float amax = 0.0f;
  for (int j = tid * 8; j < K; j += nthr * 8) {
    uint4 v = *reinterpret_cast<const uint4*>(src + j);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      amax = fmaxf(amax, fabsf(bf16_to_f32((bf16_t)(w[q] & 0xffffu))));
      amax = fmaxf(amax, fabsf(bf16_to_f32((bf16_t)(w[q] >> 16))));
    }
  }
  amax = wave_max(amax);
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  float m = 0.0f;
  for (int w = 0; w < (nthr >> 6); w++) m = fmaxf(m, red[w]);
  const float d = m / 127.0f;
  const float id = d ? 1.0f / d : 0.0f;
  for (int j = tid * 8; j < K; j += nthr * 8) {
    uint4 v = *reinterpret_cast<const uint4*>(src + j);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      int a = quant_rne_sat8(bf16_to_f32((bf16_t)(w[q] & 0xffffu)) * id);
      int b = quant_rne_sat8(bf16_to_f32((bf16_t)(w[q] >> 16)) * id);
      o[q >> 1] |= ((uint32_t)(a & 0xff) | ((uint32_t)(b & 0xff) << 8)) << ((q & 1) * 16);
    }
    *reinterpret_cast<uint2*>(dst + j) = make_uint2(o[0], o[1]);
  }
  if (tid == 0) *d_out = d;
}

__global__ __launch_bounds__(1024) void moe_prep_kernel(PrepParams p) {
  __shared__ int s_cnt[KTX_EMAX];
  __shared__ int s_off[KTX_EMAX];
  __shared__ int s_toff[KTX_EMAX];
  __shared__ int s_cur[KTX_EMAX];
  __shared__ float s_red[16];
  const int tid = threadIdx.x;
  int T = p.qlen;
  if (p.d_bsz) T = min(max(*p.d_bsz, 0), p.qlen);

  if (blockIdx.x > 0) {
    const int t = blockIdx.x - 1;
    if (t >= T) return;
    quant_row_block(p.x + (size_t)t * p.H, p.H, p.x_q + (size_t)t * p.H, p.x_d + t, s_red);
    return;
  }

  // ---- block 0: histogram -> offsets -> STABLE scatter -> tile list -------------------------------------
  // Rows of an expert are laid out in (token, slot) order — the reference's m_local_pos_ (moe_base.hpp:208-227): the row of a
  // pair depends on the routing table only, never on arrival order.  Two implementations of the rank "how many earlier pairs
  // chose the same expert":
  //  * per-expert token bitmaps (the common case: a token names an expert at most once).  A window of 32 * bm_words tokens at a
  //    time: every pair sets its token's bit in its expert's bitmap (an OR: order-free), one thread per expert turns the words
  //    into exclusive popcount prefixes, and a pair's rank is prefix[word] + popcount(bits below its token) — O(1) per pair;
  //  * a wavefront-ballot counting sort for routing tables in which a token names one expert twice (detected by the OR): a wave
  //    finds the lanes holding the same expert with one ballot per distinct expert (match-any loop), the rank inside the wave
  //    is a popcount of the lower lanes, the 16 waves of a 1024-pair chunk are chained through a per-(wave, expert) count
  //    window in LDS, chunks through a running count.
  extern __shared__ int s_dyn[];
  __shared__ int s_dup;
  const int E = p.E, npairs = T * p.k;
  const int lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < E; e += blockDim.x) { s_cnt[e] = 0; s_cur[e] = 0; }
  if (tid == 0) s_dup = 0;
  __syncthreads();
  auto expert_of = [&](int i) -> int {
    if (i >= npairs) return -1;
    const long long id = p.ids[i] - p.expert_begin;
    return (id >= 0 && id < E && !(p.mask && p.mask[id])) ? (int)id : -1;
  };
  for (int i = tid; i < npairs; i += blockDim.x) {   // per-expert totals (an integer sum: the order of the adds does not matter)
    const int e = expert_of(i);
    if (e >= 0) atomicAdd(&s_cnt[e], 1);
  }
  __syncthreads();
  if (tid < 64) {  // wave 0: exclusive scans of counts and of tile counts
    const int per = (E + 63) / 64;
    const int e0 = tid * per;
    int sum = 0, tsum = 0;
    for (int e = e0; e < min(e0 + per, E); e++) {
      sum += s_cnt[e];
      tsum += (s_cnt[e] + p.rows_per_tile - 1) / p.rows_per_tile;
    }
    int inc = sum, tinc = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int a = __shfl_up(inc, o, 64), b = __shfl_up(tinc, o, 64);
      if (tid >= o) { inc += a; tinc += b; }
    }
    int run = inc - sum, trun = tinc - tsum;
    for (int e = e0; e < min(e0 + per, E); e++) {
      s_off[e] = run;
      s_toff[e] = trun;
      run += s_cnt[e];
      trun += (s_cnt[e] + p.rows_per_tile - 1) / p.rows_per_tile;
    }
    if (tid == 63) { p.counters[0] = tinc; p.counters[1] = inc; }
  }
  __syncthreads();
  {   // ---- bitmap scatter, window by window
    const int nw = p.bm_words, WT = nw * 32;
    unsigned* bm = reinterpret_cast<unsigned*>(s_dyn);   // [E][nw]
    int* pc = s_dyn + E * nw;                             // [E][nw + 1]: exclusive popcount prefixes, window total last
    for (int t0 = 0; t0 < T && !s_dup; t0 += WT) {
      const int i0 = t0 * p.k, i1 = min(T, t0 + WT) * p.k;
      for (int i = tid; i < E * nw; i += blockDim.x) bm[i] = 0u;
      __syncthreads();
      for (int i = i0 + tid; i < i1; i += blockDim.x) {
        const int e = expert_of(i);
        if (e >= 0) {
          const int t = i / p.k - t0;
          const unsigned bit = 1u << (t & 31);
          if (atomicOr(&bm[e * nw + (t >> 5)], bit) & bit) s_dup = 1;   // the token names this expert twice
        }
      }
      __syncthreads();
      if (s_dup) break;
      for (int e = wave; e < E; e += 16) {   // one wave per expert: lane = bitmap word (nw <= 64), shuffle prefix sum
        const int c = lane < nw ? __popc(bm[e * nw + lane]) : 0;
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int a = __shfl_up(inc, o, 64);
          if (lane >= o) inc += a;
        }
        if (lane < nw) pc[e * (nw + 1) + lane] = inc - c;
        if (lane == 63) pc[e * (nw + 1) + nw] = inc;
      }
      __syncthreads();
      for (int i = i0 + tid; i < i1; i += blockDim.x) {
        const int e = expert_of(i);
        int r = -1;
        if (e >= 0) {
          const int t = i / p.k - t0, w = t >> 5;
          r = s_off[e] + s_cur[e] + pc[e * (nw + 1) + w] + __popc(bm[e * nw + w] & ((1u << (t & 31)) - 1u));
          p.src_of_row[r] = i / p.k;
        }
        p.row_of_pair[i] = r;
      }
      __syncthreads();
      for (int e = tid; e < E; e += blockDim.x) s_cur[e] += pc[e * (nw + 1) + nw];
      __syncthreads();
    }
  }
  if (s_dup) {   // ---- ballot counting sort over ALL pairs (rows already written by the bitmap windows are rewritten identically)
    int* s_wcnt = s_dyn;   // [16 waves][E]: zero outside the current chunk's window
    for (int e = tid; e < E; e += blockDim.x) s_cur[e] = 0;
    for (int i = tid; i < 16 * E; i += blockDim.x) s_wcnt[i] = 0;
    __syncthreads();
    for (int base = 0; base < npairs; base += 1024) {
      const int i = base + tid;
      const int e = expert_of(i);
      int rank = 0;
      unsigned long long rem = __ballot(e >= 0);
      while (rem) {   // wave-uniform loop: one round per distinct expert among the wave's 64 pairs
        const int lead = __builtin_amdgcn_readfirstlane(__ffsll((long long)rem) - 1);
        const int e0 = __builtin_amdgcn_readlane(e, lead);
        const unsigned long long m = __ballot(e == e0);
        if (e == e0) rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == lead) s_wcnt[wave * E + e0] = __popcll(m);
        rem &= ~m;
      }
      __syncthreads();
      int r = -1;
      if (e >= 0) {
        int pre = 0;
        for (int w = 0; w < wave; w++) pre += s_wcnt[w * E + e];
        r = s_off[e] + s_cur[e] + pre + rank;
        p.src_of_row[r] = i / p.k;
      }
      if (i < npairs) p.row_of_pair[i] = r;
      __syncthreads();
      if (e >= 0 && rank == 0) {   // one lane per (wave, expert): advance the running count, close the window
        atomicAdd(&s_cur[e], s_wcnt[wave * E + e]);
        s_wcnt[wave * E + e] = 0;
      }
      __syncthreads();
    }
  }
  for (int e = tid; e < E; e += blockDim.x) {
    const int c = s_cnt[e];
    for (int i = 0, r = 0; r < c; i++, r += p.rows_per_tile) {
      Tile t;
      t.expert = e; t.row0 = s_off[e] + r; t.nrows = min(p.rows_per_tile, c - r); t.pad = 0;
      p.tiles[s_toff[e] + i] = t;
    }
  }
}

// launch: the scatter's bitmaps / count window live in dynamic LDS (beyond 48 KB the attribute is needed)
static hipError_t launch_moe_prep(const PrepParams& pp, int nblocks, hipStream_t st) {
  const hipError_t attr_err = ktx_set_max_lds(reinterpret_cast<const void*>(moe_prep_kernel), 96 * 1024);
  if (attr_err != hipSuccess) return attr_err;
  PrepParams p2 = pp;
  const int E = std::max(pp.E, 1);
  p2.bm_words = std::max(1, std::min(64, (64 * 1024) / (8 * E)));   // ~64 KB of bitmaps + prefixes per window
  const size_t smem = std::max((size_t)16 * E * sizeof(int), (size_t)E * (2 * p2.bm_words + 1) * sizeof(int));
  hipLaunchKernelGGL(moe_prep_kernel, dim3(nblocks), dim3(1024), smem, st, p2);
  return hipGetLastError();
}

// =====================================================================================================
// Kq: per-row int8 quantisation of the activated intermediate (a11: down_ba_->from_mat, moe_base.hpp:378-384)
// =====================================================================================================
__device__ __forceinline__ uint2 quant8(const uint4& v, float id);
__device__ __forceinline__ float amax8(const uint4& v, float m);
// One workgroup per sorted row.  gu = [row][g | u] bf16 from the gate/up GEMM: a = bf16(act_fn(g, u)) (a10,
// moe_base.hpp:693-726) is formed in registers (K <= 8192: four 8-element pieces per thread), then quantised per row exactly
// like quant_row_block (d = amax/127, q = sat8(rne(a * (1/d)))).
__global__ __launch_bounds__(256) void moe_actquant_kernel(const bf16_t* __restrict__ gu, int K, int8_t* __restrict__ a_q,
                                                           float* __restrict__ a_d, const int32_t* counters) {
  __shared__ float s_red[4];
  const int row = blockIdx.x;
  if (row >= counters[1]) return;
  const int tid = threadIdx.x;
  const bf16_t* gr = gu + (size_t)row * 2 * K;
  uint4 av[4];
  float amax = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int j = (tid + i * 256) * 8;
    av[i] = make_uint4(0, 0, 0, 0);
    if (j < K) {
      const uint4 g = *reinterpret_cast<const uint4*>(gr + j), u = *reinterpret_cast<const uint4*>(gr + K + j);
      const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, uw[4] = {u.x, u.y, u.z, u.w};
      uint32_t o[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const bf16_t lo = f32_to_bf16(act_fn(bf16_to_f32((bf16_t)(gw[q] & 0xffffu)), bf16_to_f32((bf16_t)(uw[q] & 0xffffu))));
        const bf16_t hi = f32_to_bf16(act_fn(bf16_to_f32((bf16_t)(gw[q] >> 16)), bf16_to_f32((bf16_t)(uw[q] >> 16))));
        o[q] = (uint32_t)lo | ((uint32_t)hi << 16);
      }
      av[i] = make_uint4(o[0], o[1], o[2], o[3]);
      amax = amax8(av[i], amax);
    }
  }
  amax = wave_max(amax);
  if ((tid & 63) == 0) s_red[tid >> 6] = amax;
  __syncthreads();
  const float m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  const float d = m / 127.0f;
  const float id = d ? 1.0f / d : 0.0f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int j = (tid + i * 256) * 8;
    if (j < K) *reinterpret_cast<uint2*>(a_q + (size_t)row * K + j) = quant8(av[i], id);
  }
  if (tid == 0) a_d[row] = d;
}

// =====================================================================================================
// K1 / K2: grouped W4A8 / W8A8 GEMM on MFMA i8 (a8, a9, a10)
// =====================================================================================================
struct GemmParams {
  const uint8_t* w0;   // gate (GATE_UP) or down
  const uint8_t* w1;   // up   (GATE_UP) or nullptr
  const float* s0;     // scales [E][N]
  const float* s1;
  size_t expert_stride;  // bytes of one expert's matrix
  int N, K;
  const int8_t* act_q;      // [src rows][K]
  const float* act_d;       // [src rows]
  const int32_t* row_src;   // sorted row -> source row (nullptr: identity)
  const Tile* tiles;
  const int32_t* counters;
  bf16_t* out;  // [sorted rows][N]
};

template <int WBITS, int MT, int SPC, bool GATE_UP>
__global__ __launch_bounds__(256) void moe_gemm_kernel(GemmParams p) {
  constexpr int NMAT = GATE_UP ? 2 : 1;
  constexpr int KC = SPC * 128;
  constexpr int COLS = KC / 16;                       // 16-byte columns per chunk
  constexpr int BUF_BYTES = MT * COLS * 256;          // one LDS activation buffer
  constexpr int GROUPS = MT * SPC * 2;                // staging groups of (16 tok x 4 cols)
  constexpr int UPT = (GROUPS + 3) / 4;               // staging units per thread
  constexpr int TILE_BYTES = (WBITS == 4) ? 1024 : 2048;

  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* Bs = smem;                                               // [2][BUF_BYTES]
  float* s_ad = reinterpret_cast<float*>(smem + 2 * BUF_BYTES);     // [MT*16]
  int* s_src = reinterpret_cast<int*>(s_ad + MT * 16);              // [MT*16]

  // (an XCD-aware remap that runs the 3-4 token tiles sharing an expert's weight strips back to back on one XCD was
  // measured: no change at T = 2048 — the re-reads are absorbed by the 256 MB Infinity Cache; the kernel is bound by the
  // one-chunk prefetch distance)
  const int tile_idx = blockIdx.y;
  if (tile_idx >= p.counters[0]) return;
  const Tile tile = p.tiles[tile_idx];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int strip = blockIdx.x * 4 + wave;
  const int NKS = p.K / 128;
  const int NC = (NKS + SPC - 1) / SPC;
  const bool strip_ok = strip * 16 < p.N;  // wave-uniform

  // rows of this tile: source row + activation scale
  if (tid < MT * 16) {
    int src = -1;
    float ad = 0.0f;
    if (tid < tile.nrows) {
      src = p.row_src ? p.row_src[tile.row0 + tid] : (tile.row0 + tid);
      ad = p.act_d[src];
    }
    s_src[tid] = src;
    s_ad[tid] = ad;
  }
  __syncthreads();

  const uint8_t* wbase[NMAT];
  wbase[0] = p.w0 + (size_t)tile.expert * p.expert_stride + (size_t)strip * NKS * TILE_BYTES;
  if constexpr (GATE_UP) wbase[1] = p.w1 + (size_t)tile.expert * p.expert_stride + (size_t)strip * NKS * TILE_BYTES;

  v4i acc[NMAT][MT];
#pragma unroll
  for (int m = 0; m < NMAT; m++)
#pragma unroll
    for (int t = 0; t < MT; t++) acc[m][t] = v4i{0, 0, 0, 0};

  WFrag<WBITS> wA[SPC][NMAT], wB[SPC][NMAT];
  uint4 breg[UPT];

  auto load_w = [&](WFrag<WBITS>(&dst)[SPC][NMAT], int c) {
#pragma unroll
    for (int s = 0; s < SPC; s++) {
      const int ks = c * SPC + s;
      if (strip_ok && ks < NKS) {
#pragma unroll
        for (int m = 0; m < NMAT; m++) dst[s][m] = load_wfrag<WBITS>(wbase[m] + (size_t)ks * TILE_BYTES, lane);
      }
    }
  };
  // staging unit `it` of this thread: a fixed (token row, 16-byte column) — its global row pointer and LDS slot are
  // computed once, the chunk loop only adds c*KC (the per-chunk address arithmetic was ~1/4 of the kernel's instructions)
  const int8_t* gsrc[UPT];
  int gcol[UPT], lds_off[UPT];
#pragma unroll
  for (int it = 0; it < UPT; it++) {
    const int g = it * 4 + wave;
    gsrc[it] = nullptr;
    gcol[it] = 0;
    lds_off[it] = 0;
    if (g < GROUPS) {
      const int mt = g / (SPC * 2), col = (g % (SPC * 2)) * 4 + (lane >> 4);
      const int src = s_src[mt * 16 + (lane & 15)];
      gcol[it] = col * 16;
      lds_off[it] = ((mt * COLS + col) * 16 + (lane & 15)) * 16;
      if (src >= 0) gsrc[it] = p.act_q + (size_t)src * p.K + col * 16;
    }
  }
  auto load_b = [&](int c) {
#pragma unroll
    for (int it = 0; it < UPT; it++) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (gsrc[it] && c * KC + gcol[it] < p.K) v = *reinterpret_cast<const uint4*>(gsrc[it] + c * KC);
      breg[it] = v;
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int it = 0; it < UPT; it++)
      if (it * 4 + wave < GROUPS) *reinterpret_cast<uint4*>(Bs + buf * BUF_BYTES + lds_off[it]) = breg[it];
  };
  auto compute = [&](WFrag<WBITS>(&w)[SPC][NMAT], int c, int buf) {
    if (!strip_ok) return;
    const uint8_t* bb = Bs + buf * BUF_BYTES + ((lane >> 4) * 2 * 16 + (lane & 15)) * 16;
#pragma unroll
    for (int s = 0; s < SPC; s++) {
      if (c * SPC + s < NKS) {
        v4i a[NMAT][2];
#pragma unroll
        for (int m = 0; m < NMAT; m++) unpack_wfrag<WBITS>(w[s][m], a[m][0], a[m][1]);
        // both halves of every (matrix, token-tile) accumulator are issued a full round apart: back-to-back MFMAs on the
        // same accumulator would each wait out the previous one's latency
        v4i b[MT][2];
#pragma unroll
        for (int t = 0; t < MT; t++) {
          b[t][0] = *reinterpret_cast<const v4i*>(bb + ((t * COLS + s * 8) * 16) * 16);
          b[t][1] = *reinterpret_cast<const v4i*>(bb + ((t * COLS + s * 8 + 1) * 16) * 16);
        }
#pragma unroll
        for (int hh = 0; hh < 2; hh++)
#pragma unroll
          for (int t = 0; t < MT; t++)
#pragma unroll
            for (int m = 0; m < NMAT; m++)
              acc[m][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m][hh], b[t][hh], acc[m][t], 0, 0, 0);
      }
    }
  };

  // ---- software pipeline: weights for chunk c+1 and activations for chunk c+1 are in flight while chunk c computes
  load_w(wA, 0);
  load_b(0);
  store_b(0);
  __syncthreads();
  for (int c = 0; c < NC; c += 2) {
    if (c + 1 < NC) { load_b(c + 1); load_w(wB, c + 1); }
    compute(wA, c, 0);
    if (c + 1 < NC) store_b(1);
    __syncthreads();
    if (c + 1 < NC) {
      if (c + 2 < NC) { load_b(c + 2); load_w(wA, c + 2); }
      compute(wB, c + 1, 1);
      if (c + 2 < NC) store_b(0);
      __syncthreads();
    }
  }

  // ---- epilogue: scale, round to bf16 (a9), activation (a10), store 4 consecutive n per lane --------------------
  if (!strip_ok) return;
  const int tok = lane & 15;
  const int n0 = strip * 16 + (lane >> 4) * 4;
  const float4 sc0 = *reinterpret_cast<const float4*>(p.s0 + (size_t)tile.expert * p.N + n0);
  float4 sc1 = sc0;
  if constexpr (GATE_UP) sc1 = *reinterpret_cast<const float4*>(p.s1 + (size_t)tile.expert * p.N + n0);
  const float s0v[4] = {sc0.x, sc0.y, sc0.z, sc0.w};
  const float s1v[4] = {sc1.x, sc1.y, sc1.z, sc1.w};
#pragma unroll
  for (int t = 0; t < MT; t++) {
    const int row = t * 16 + tok;
    if (row < tile.nrows) {
      const float ad = s_ad[row];
      bf16_t o[4], o2[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        // GemmKernel224Int4::apply_scale: (a_d * b_d) * float(acc)   (la/amx_kernels.hpp:1808-1846), then bf16 (a9)
        o[r] = f32_to_bf16((ad * s0v[r]) * (float)acc[0][t][r]);
        if constexpr (GATE_UP) o2[r] = f32_to_bf16((ad * s1v[r]) * (float)acc[1][t][r]);
      }
      // gate/up: both bf16 results are stored ([row][g | u]); SiLU(g)*u (a10) is applied by moe_actquant_kernel, which has
      // to read the row anyway — the ~40-instruction polynomial per element was a third of this kernel's issue slots
      const int ld = GATE_UP ? 2 * p.N : p.N;
      *reinterpret_cast<uint2*>(p.out + (size_t)(tile.row0 + row) * ld + n0) =
          make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
      if constexpr (GATE_UP)
        *reinterpret_cast<uint2*>(p.out + (size_t)(tile.row0 + row) * ld + p.N + n0) =
            make_uint2((uint32_t)o2[0] | ((uint32_t)o2[1] << 16), (uint32_t)o2[2] | ((uint32_t)o2[3] << 16));
    }
  }
}

// =====================================================================================================
// K1s / K2s: "streaming" grouped GEMM for prompts (token tiles of 64 rows).
// The chunk-pipelined kernel above re-reads one 1 KB activation fragment from LDS for every two MFMAs of a wave — exactly
// the LDS port's 128 B/clk at full MFMA rate — and synchronises all waves every chunk (PMC at T=2048: MFMA busy 21 %, 55 %
// of wave cycles waiting).  Here:
//   * the tile's int8 activation rows are staged in LDS once per <= 2048-column chunk of K (one chunk for DeepSeek-V2-Lite
//     and for every down projection) — no barrier inside a chunk;
//   * each of the 8 waves owns FOUR 16-row weight strips (gate/up: 2 strips x {gate, up}; down: 4 strips), so one
//     activation fragment read feeds 4 MFMAs (LDS port at <= 50 %) and 16 independent accumulators hide MFMA latency;
//   * weights stream through a D-deep register ring per wave, D k-steps (>= 0.8 us of MFMA work) ahead of use.
// Integer arithmetic and epilogue are those of moe_gemm_kernel, so results are bit-identical.
// =====================================================================================================
template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}

template <int WBITS, int MT, int D, bool GATE_UP>
__global__ __launch_bounds__(512) void moe_gemm_stream_kernel(GemmParams p, int kch) {
  constexpr int NJ = 4;                              // (strip, matrix) units per wave
  constexpr int TOK = MT * 16;
  constexpr int CS = TOK * 16;                       // LDS stride between 16-byte columns; token slot XOR-swizzled per column
  constexpr int TILE_BYTES = (WBITS == 4) ? 1024 : 2048;
  constexpr int SPW = GATE_UP ? 2 : 4;               // strips per wave
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int NKS = p.K / 128, KSC = kch / 128, CCOL = kch / 16;
  uint8_t* xs = smem;                                                // [CCOL][TOK][16 B] (+pad)
  float* s_ad = reinterpret_cast<float*>(smem + (size_t)CCOL * CS);  // [TOK]
  int* s_src = reinterpret_cast<int*>(s_ad + TOK);                   // [TOK]

  // (grid mappings that give all tiles x strip groups of one expert consecutive slots of ONE XCD, so that its L2 serves
  // the re-reads, were measured: -5 % at best with uniform routing, +25 % with skewed routing because of the padding
  // workgroups a fixed slots-per-expert layout needs — plain tile-major it is)
  const int bx = blockIdx.x, tile_idx = blockIdx.y;
  if (tile_idx >= p.counters[0]) return;
  const Tile tile = p.tiles[tile_idx];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip0 = (bx * 8 + wave) * SPW;
  const int nstrips = p.N / 16;

  // ---- weight ring first (addresses depend on the tile record only); strips past N alias strip 0 and are not stored
  const uint8_t* wb[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int st = strip0 + (GATE_UP ? (j >> 1) : j);
    const uint8_t* base = (GATE_UP && (j & 1)) ? p.w1 : p.w0;
    wb[j] = base + (size_t)tile.expert * p.expert_stride + (size_t)(st < nstrips ? st : 0) * NKS * TILE_BYTES;
  }
  WFrag<WBITS> ring[D][NJ];
#pragma unroll
  for (int d = 0; d < D; d++) {
    const int ks = d < NKS ? d : NKS - 1;
#pragma unroll
    for (int j = 0; j < NJ; j++) ring[d][j] = load_wfrag<WBITS>(wb[j] + (size_t)ks * TILE_BYTES, lane);
  }

  if (tid < TOK) {
    int src = -1;
    float ad = 0.0f;
    if (tid < tile.nrows) {
      src = p.row_src ? p.row_src[tile.row0 + tid] : (tile.row0 + tid);
      ad = p.act_d[src];
    }
    s_src[tid] = src;
    s_ad[tid] = ad;
  }

  // B fragment of k-step s (within the chunk), half hh, token tile t: column s*8 + kc*2 + hh, token t*16 + (lane&15).
  // LDS address of (col, tok) = col*CS + ((tok ^ f(col)) << 4), f(col) = (col & 3) | (col & 4 ? 12 : 0): ds_read_b128 is
  // served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... over a 256-byte bank row — f keeps the token sets
  // {0-3,12-15} / {4-11} of the two kc in a group disjoint (conflict-free reads; the unswizzled layout was 2-way), and gives
  // the 8 consecutive columns one staging ds_write_b128 group covers 8 different 16-byte slots (conflict-free writes).
  const int kc = lane >> 4;
  const int fsw = ((kc & 1) << 1) | ((kc & 2) ? 12 : 0);
  const uint8_t* bb0 = xs + (size_t)(kc * 2) * CS + (((lane & 15) ^ fsw) << 4);
  const uint8_t* bb1 = xs + (size_t)(kc * 2 + 1) * CS + (((lane & 15) ^ (fsw | 1)) << 4);
  const bool wave_ok = strip0 < nstrips;  // wave-uniform

  v4i acc[NJ][MT];
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int t = 0; t < MT; t++) acc[j][t] = v4i{0, 0, 0, 0};
  for (int c0 = 0; c0 < NKS; c0 += KSC) {
    const int c1 = c0 + KSC < NKS ? c0 + KSC : NKS;
    const int ncol = (c1 - c0) * 8;
    __syncthreads();  // previous chunk fully consumed (and s_src visible on the first pass)
    // ---- stage the chunk: wavefront -> rows wave, wave + 8, ...; lane -> the row's 16-byte columns lane, lane + 64 (2 KB of one
    // row per pass: coalesced, and NO integer division — the round-1..5 mapping idx -> (idx / ncol, idx % ncol) spent as many
    // VALU instructions as the whole k loop: 4.5 VALU per MFMA in profiles/r06_w_pmc_stream8192.txt; the launch time did not
    // move).  4 rows = 8 loads in flight per thread before the LDS writes.
    // (unconditional loads from clamped rows / columns + an explicit vmcnt(0): every path into the k loop then has no load
    // pending, which lets the compiler's waitcnt pass keep the ring's vmcnt(12..15) waits exact)
    auto stage = [&](auto uc, int r0) {
      constexpr int U = decltype(uc)::value;
      uint4 v[U][2];
      int off[U][2];
      bool live[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int row = r0 + u * 8;
        const int src = s_src[row];
        live[u] = src >= 0;
        const int8_t* rp = p.act_q + (size_t)(src >= 0 ? src : 0) * p.K + (size_t)c0 * 128;
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int col = lane + h * 64 < ncol ? lane + h * 64 : ncol - 1;
          off[u][h] = col * CS + (((row & ~15) | ((row & 15) ^ ((col & 3) | ((col & 4) ? 12 : 0)))) << 4);
          v[u][h] = *reinterpret_cast<const uint4*>(rp + col * 16);
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int h = 0; h < 2; h++)
          if (lane + h * 64 < ncol) *reinterpret_cast<uint4*>(xs + off[u][h]) = live[u] ? v[u][h] : make_uint4(0, 0, 0, 0);
    };
    static_assert(TOK % 16 == 0, "rows per wavefront");
    {                                  // (kch <= 2048: at most two 64-column passes per row)
      int r0 = wave;
      for (; r0 + 24 < TOK; r0 += 32) stage(std::integral_constant<int, 4>{}, r0);
      for (; r0 < TOK; r0 += 16) stage(std::integral_constant<int, 2>{}, r0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();
    if (wave_ok) {
      // one k-step: consume ring slot d, refill it D steps ahead (clamped: branch-free, so the compiler keeps exact
      // vmcnt(N) waits instead of draining the ring at every control-flow join), 2 x MT x NJ MFMAs
      auto step = [&](auto dc, auto refill, int ks) {
        constexpr int d = decltype(dc)::value;
        v4i a[NJ][2];
#pragma unroll
        for (int j = 0; j < NJ; j++) unpack_wfrag<WBITS>(ring[d][j], a[j][0], a[j][1]);
        if constexpr (decltype(refill)::value) {
          const int kn = ks + D < NKS ? ks + D : NKS - 1;
#pragma unroll
          for (int j = 0; j < NJ; j++) ring[d][j] = load_wfrag<WBITS>(wb[j] + (size_t)kn * TILE_BYTES, lane);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the refill HERE (the scheduler otherwise sinks all loads to the loop end)
        const size_t koff = (size_t)((ks - c0) * 8) * CS;
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
          const uint8_t* bk = (hh ? bb1 : bb0) + koff;
          v4i b[MT];
#pragma unroll
          for (int t = 0; t < MT; t++) b[t] = *reinterpret_cast<const v4i*>(bk + t * 256);
#pragma unroll
          for (int t = 0; t < MT; t++)
#pragma unroll
            for (int j = 0; j < NJ; j++)
              acc[j][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[j][hh], b[t], acc[j][t], 0, 0, 0);
        }
      };
      int s0 = c0;
      for (; s0 + D <= c1; s0 += D)
        static_for<D>([&](auto dc) { step(dc, std::true_type{}, s0 + decltype(dc)::value); });
      // tail (< D steps, only ever at the end of K): nothing left to prefetch
      static_for<D>([&](auto dc) {
        if (s0 + decltype(dc)::value < c1) step(dc, std::false_type{}, s0 + decltype(dc)::value);
      });
    }
  }
  if (!wave_ok) return;

  // ---- epilogue: the arithmetic of moe_gemm_kernel's (GemmKernel224Int4::apply_scale, la/amx_kernels.hpp:1808-1846)
  const int tok = lane & 15;
  const int ld = GATE_UP ? 2 * p.N : p.N;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int st = strip0 + (GATE_UP ? (j >> 1) : j);
    if (st < nstrips) {
      const int n0 = st * 16 + (lane >> 4) * 4;
      const float* sp = (GATE_UP && (j & 1)) ? p.s1 : p.s0;
      const float4 sc = *reinterpret_cast<const float4*>(sp + (size_t)tile.expert * p.N + n0);
      const float sv[4] = {sc.x, sc.y, sc.z, sc.w};
      const int coff = (GATE_UP && (j & 1)) ? p.N : 0;
#pragma unroll
      for (int t = 0; t < MT; t++) {
        const int row = t * 16 + tok;
        if (row < tile.nrows) {
          const float ad = s_ad[row];
          bf16_t o[4];
#pragma unroll
          for (int r = 0; r < 4; r++) o[r] = f32_to_bf16((ad * sv[r]) * (float)acc[j][t][r]);
          *reinterpret_cast<uint2*>(p.out + (size_t)(tile.row0 + row) * ld + coff + n0) =
              make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
        }
      }
    }
  }
}

// =====================================================================================================
// K1r / K2r: register-tile grouped GEMM for prompts (dev knob 4 = 3).  Written at the end of round 1 from the PMC evidence in
// bit-exact on hardware (tests/test_moe_gpu.py::test_register_tile_prompt_kernels) but NOT yet timed —
// the GPU budget was spent — so the product path does not select it yet.
// The 64-row kernels move ~4x the unique operand bytes across the fabric; here one workgroup owns a 256-row x 16-unit
// (256-feature) int32 accumulator tile in registers (8 waves = 4 row groups x 2 unit groups, 128 accumulator VGPRs each)
// and walks K one 128-wide step at a time with BOTH operands double-buffered in LDS (activations gathered + swizzled as in
// the streaming kernel, weights stored un-packed to int8 so the loop has no VALU): 350 op per fabric byte instead of 210.
// Tiles come from moe_prep_kernel with rows_per_tile = 256.  Arithmetic and epilogue as in moe_gemm_kernel.
// =====================================================================================================
template <int WBITS, bool GATE_UP>
__global__ __launch_bounds__(512) void moe_gemm_rt_kernel(GemmParams p) {
  constexpr int ROWS = 256, UNITS = 16, WU = 8, WT = 4;      // workgroup tile; per-wave units / 16-row token tiles
  constexpr int TILE_BYTES = (WBITS == 4) ? 1024 : 2048;
  constexpr int ACT_BYTES = 8 * ROWS * 16;                   // one k-step of activations: 8 x 16-byte columns x 256 rows
  constexpr int W_BYTES = UNITS * 2048;                      // one k-step of weights, int8: [unit][half][lane][16 B]
  constexpr int BUF = ACT_BYTES + W_BYTES;                   // 64 KB per buffer, two buffers
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  float* s_ad = reinterpret_cast<float*>(smem + 2 * BUF);    // [ROWS]
  int* s_src = reinterpret_cast<int*>(s_ad + ROWS);          // [ROWS]

  const int tile_idx = blockIdx.y;
  if (tile_idx >= p.counters[0]) return;
  const Tile tile = p.tiles[tile_idx];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;                   // row group (64 rows), unit group (8 units)
  const int NKS = p.K / 128, nstrips = p.N / 16;

  if (tid < ROWS) {
    int src = -1;
    float ad = 0.0f;
    if (tid < tile.nrows) {
      src = p.row_src ? p.row_src[tile.row0 + tid] : (tile.row0 + tid);
      ad = p.act_d[src];
    }
    s_src[tid] = src;
    s_ad[tid] = ad;
  }
  __syncthreads();

  // ---- staging roles.  activations: 4 x (row, column) per thread; weights: units tid>>6 and 8 + (tid>>6), lane tid&63
  const int8_t* a_src[4];
  int a_off[4];
  bool a_live[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int idx = tid + i * 512, row = idx >> 3, col = idx & 7;
    const int src = s_src[row];
    a_live[i] = src >= 0;
    a_src[i] = p.act_q + (size_t)(src >= 0 ? src : 0) * p.K + col * 16;
    a_off[i] = col * (ROWS * 16) + (((row & ~15) | ((row & 15) ^ ((col & 3) | ((col & 4) ? 12 : 0)))) << 4);
  }
  const uint8_t* w_src[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int u = (tid >> 6) + i * 8;
    const int strip = GATE_UP ? blockIdx.x * 8 + (u >> 1) : blockIdx.x * 16 + u;
    const uint8_t* base = (GATE_UP && (u & 1)) ? p.w1 : p.w0;
    w_src[i] = base + (size_t)tile.expert * p.expert_stride + (size_t)(strip < nstrips ? strip : 0) * NKS * TILE_BYTES + lane * 16;
  }
  uint4 ra[4];
  WFrag<WBITS> rw[2];
  auto load_regs = [&](int ks) {
#pragma unroll
    for (int i = 0; i < 4; i++) ra[i] = *reinterpret_cast<const uint4*>(a_src[i] + (size_t)ks * 128);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const uint8_t* t = w_src[i] + (size_t)ks * TILE_BYTES;
      rw[i].v[0] = *reinterpret_cast<const uint4*>(t);
      if constexpr (WBITS == 8) rw[i].v[1] = *reinterpret_cast<const uint4*>(t + 1024);
    }
  };
  auto store_lds = [&](int buf) {
    uint8_t* ab = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < 4; i++) *reinterpret_cast<uint4*>(ab + a_off[i]) = a_live[i] ? ra[i] : make_uint4(0, 0, 0, 0);
    uint8_t* wb = ab + ACT_BYTES;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      v4i h0, h1;
      unpack_wfrag<WBITS>(rw[i], h0, h1);
      const int u = (tid >> 6) + i * 8;
      *reinterpret_cast<v4i*>(wb + u * 2048 + lane * 16) = h0;
      *reinterpret_cast<v4i*>(wb + u * 2048 + 1024 + lane * 16) = h1;
    }
  };

  v4i acc[WU][WT];
#pragma unroll
  for (int u = 0; u < WU; u++)
#pragma unroll
    for (int t = 0; t < WT; t++) acc[u][t] = v4i{0, 0, 0, 0};
  // this wave's fragment addresses inside a buffer
  const int kc = lane >> 4;
  const int fsw = ((kc & 1) << 1) | ((kc & 2) ? 12 : 0);
  const int b_off0 = (kc * 2) * (ROWS * 16) + (wm * 64) * 16 + (((lane & 15) ^ fsw) << 4);
  const int b_off1 = (kc * 2 + 1) * (ROWS * 16) + (wm * 64) * 16 + (((lane & 15) ^ (fsw | 1)) << 4);
  const int w_off = ACT_BYTES + (wn * WU) * 2048 + lane * 16;
  const bool rows_live = wm * 64 < tile.nrows;  // wave-uniform: a row group past the tile only keeps the barriers company

  load_regs(0);
  store_lds(0);
  __syncthreads();
  for (int c = 0; c < NKS; c++) {
    if (c + 1 < NKS) load_regs(c + 1);
    if (rows_live) {
      const uint8_t* buf = smem + (c & 1) * BUF;
#pragma unroll
      for (int hh = 0; hh < 2; hh++) {
        v4i b[WT];
#pragma unroll
        for (int t = 0; t < WT; t++) b[t] = *reinterpret_cast<const v4i*>(buf + (hh ? b_off1 : b_off0) + t * 256);
#pragma unroll
        for (int u = 0; u < WU; u++) {
          const v4i a = *reinterpret_cast<const v4i*>(buf + w_off + u * 2048 + hh * 1024);
#pragma unroll
          for (int t = 0; t < WT; t++) acc[u][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b[t], acc[u][t], 0, 0, 0);
        }
      }
    }
    if (c + 1 < NKS) store_lds((c + 1) & 1);
    __syncthreads();
  }
  if (!rows_live) return;

  // ---- epilogue (GemmKernel224Int4::apply_scale, la/amx_kernels.hpp:1808-1846; bf16 rounding a9)
  const int tok = lane & 15;
  const int ld = GATE_UP ? 2 * p.N : p.N;
#pragma unroll
  for (int u = 0; u < WU; u++) {
    const int gu = wn * WU + u;
    const int strip = GATE_UP ? blockIdx.x * 8 + (gu >> 1) : blockIdx.x * 16 + gu;
    if (strip < nstrips) {
      const int n0 = strip * 16 + (lane >> 4) * 4;
      const float* sp = (GATE_UP && (gu & 1)) ? p.s1 : p.s0;
      const float4 sc = *reinterpret_cast<const float4*>(sp + (size_t)tile.expert * p.N + n0);
      const float sv[4] = {sc.x, sc.y, sc.z, sc.w};
      const int coff = (GATE_UP && (gu & 1)) ? p.N : 0;
#pragma unroll
      for (int t = 0; t < WT; t++) {
        const int row = wm * 64 + t * 16 + tok;
        if (row < tile.nrows) {
          const float ad = s_ad[row];
          bf16_t o[4];
#pragma unroll
          for (int r = 0; r < 4; r++) o[r] = f32_to_bf16((ad * sv[r]) * (float)acc[u][t][r]);
          *reinterpret_cast<uint2*>(p.out + (size_t)(tile.row0 + row) * ld + coff + n0) =
              make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
        }
      }
    }
  }
}

// =====================================================================================================
// Decode fast path (qlen*k <= KTX_DEC_MAX_PAIRS): two launches per layer instead of five.
//   moe_dec_gateup_kernel : one workgroup per ((t,j) pair, 4 strips); quantises x[t] itself (a6), streams the gate/up
//                           strips of expert ids[t][j] through a D-deep register ring (every wave keeps D KiB-sized
//                           loads in flight), MFMA, SiLU*up epilogue -> a_buf[pair]
//   moe_dec_down_kernel   : one workgroup per (token, 16-row strip of H), wave j = slot j: quantises a_buf[t,j] itself
//                           (a11), streams the down strip of its expert, then the weighted combine in slot order (a12)
//                           and the merge/incremental/bf16 step (a4) happen in-workgroup through LDS.
// Every column of the 16-wide MFMA B operand carries the same token, so no lane masking is needed.
// =====================================================================================================
template <int WBITS, int D, int NW, bool EXACT, int KS = 1>
__global__ __launch_bounds__(NW * KS * 64) void moe_dec_gateup_kernel(DecParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  moe_dec_gateup_body<WBITS, D, NW, EXACT, KS>(p, blockIdx.x, blockIdx.y, smem);
}

template <int WBITS>
__device__ __forceinline__ void dec_step1(const WFrag<WBITS>& f, const v4i& b0, const v4i& b1, v4i& acc0, v4i& acc1) {
  v4i a0, a1;
  unpack_wfrag<WBITS>(f, a0, a1);
  acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, acc0, 0, 0, 0);  // two independent accumulators: the halves
  acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, acc1, 0, 0, 0);  // are summed (exactly, int32) at the end
}

// SIDE_G > 0: the workgroup also computes its 16 outputs of a W4 (group SIDE_G) dense linear on side_x — the shared experts'
// down_proj, which the reference runs beside the routed experts (operators/experts.py:974-1012) — and the block's closing
// adds ride in the epilogue.  The side strip's k-steps are dealt out to the k wavefronts (ceil(side_nks / k) each, at most
// SIDE_MAX), their loads issued up front with the routed weight ring; the fp32 partial sums meet in LDS in wavefront order,
// i.e. the k-slice structure of lin_dec_kernel (ktx_linear.hip) with the same k-step arithmetic (ktx_w4_step.inc).
// (with the side strip the kernel is held to 128 registers — two workgroups per CU like the plain kernel, so that the whole
// grid of H/16 workgroups is resident at once.  The variants with four side k-steps per wavefront on the 16-deep W4 ring or
// a W8 ring then keep 5-9 values in scratch (20-36 B per lane, -Rpass-analysis=kernel-resource-usage); giving them 168
// registers removes the scratch but leaves ONE workgroup per CU — two rounds of workgroups for H >= 4112 — which is the
// worse trade for a latency-bound launch, so the cap stays.  W8 with the 11-deep ring is refused at dispatch.)
template <int WBITS, int D, bool EXACT, int SIDE_G = 0, int SIDE_MAX = 2>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(SIDE_G > 0 ? 4 : 1)))
void moe_dec_down_kernel(DecParams p) {
  constexpr int TILE_BYTES = (WBITS == 4) ? 1024 : 2048;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // [k][I + 128] int8 activations | [k][16] fp32 down outputs | [k] valid flags | [k] routing weights
  // | SIDE: [k][16] fp32 side partials | side activations [side_nks * 256 B] | group sums [side_nks * GPK][4] fp32
  const int IP = p.I + 128;
  uint8_t* aq_all = smem;
  float* s_dn = reinterpret_cast<float*>(smem + (size_t)p.k * IP);
  int* s_valid = reinterpret_cast<int*>(s_dn + p.k * 16);
  float* s_wt = reinterpret_cast<float*>(s_valid + p.k);
  int T = p.qlen;
  if (p.d_bsz) T = min(max(*p.d_bsz, 0), p.qlen);
  const int t = blockIdx.y;
  if (t >= T) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave j = routing slot j
  const int pair = t * p.k + j, strip = blockIdx.x;
  // ---- side linear, part 1: this wave's weight tiles and scales in flight first, then its slice of the activation row ------
  constexpr int SGPK = SIDE_G > 0 ? 128 / SIDE_G : 1;
  float* s_side = s_wt + p.k;                                                      // [k][16]
  uint8_t* side_xs = smem + (((size_t)(reinterpret_cast<uint8_t*>(s_side + p.k * 16) - smem) + 15) & ~(size_t)15);
  float* side_aux = reinterpret_cast<float*>(side_xs + (size_t)p.side_nks * 256);
  uint4 sw[SIDE_G > 0 ? SIDE_MAX : 1];
  uint2 ssc[SIDE_G > 0 ? SIDE_MAX : 1];
  const int spw = SIDE_G > 0 ? (p.side_nks + p.k - 1) / p.k : 0;
  if constexpr (SIDE_G > 0) {
    const uint8_t* wp = p.side_w + (size_t)strip * p.side_nks * 1024 + lane * 16;
    const bf16_t* sp = p.side_sc + ((size_t)strip * p.side_nks * 16 + (lane & 15)) * SGPK;
#pragma unroll
    for (int i = 0; i < SIDE_MAX; i++) {
      const int ks = j * spw + i;
      if (i < spw && ks < p.side_nks) {
        typedef unsigned int u4v __attribute__((ext_vector_type(4)));
        const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(wp + (size_t)ks * 1024));
        sw[i] = make_uint4(v.x, v.y, v.z, v.w);
        ssc[i] = ktxw4::load_w4_scales<SGPK>(sp + (size_t)ks * 16 * SGPK);
      }
    }
    // the wave stages exactly the k-steps it will multiply (16 pieces of 8 each: lane = (k-step i, piece)), so the staging
    // needs no workgroup barrier — only the wave reads these LDS bytes back
    {
      const int i = lane >> 4, ks = j * spw + i;
      const bool mine = i < spw && ks < p.side_nks;
      const int idx = ks * 16 + (lane & 15);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (mine) v = *reinterpret_cast<const uint4*>(p.side_x + (size_t)t * p.side_nks * 128 + idx * 8);
      float sm = ktxw4::sum8_bf16(v);
      constexpr int PPG = SIDE_G > 0 ? SIDE_G / 8 : 1;   // 8-element pieces per scale group (<= 16: inside the 16-lane run)
#pragma unroll
      for (int o = 1; o < PPG; o <<= 1) sm += __shfl_xor(sm, o, 64);
      if (mine) {
        *reinterpret_cast<uint4*>(side_xs + idx * 16) = v;
        if ((idx & (PPG - 1)) == 0) {
          float* ax = side_aux + (idx / PPG) * 4;
          ax[0] = sm; ax[1] = sm; ax[2] = sm; ax[3] = sm;
        }
      }
    }
  }
  const long long idl = p.ids[pair] - p.expert_begin;
  const bool valid = !(idl < 0 || idl >= p.E || (p.mask && p.mask[idl]));
  const int e = valid ? (int)idl : 0;
  const int NKS = p.I / 128;
  const uint8_t* wd = p.down_w + (size_t)e * p.dn_stride + (size_t)strip * NKS * TILE_BYTES;
  uint8_t* aq = aq_all + (size_t)j * IP;

  float ad = 0.0f;
  v4i acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  if (valid) {
    // ---- a11 part 1: the activated row of this (t,j) pair (issued before the weight ring: needed first) ----------
    const bf16_t* ar = p.a_buf + (size_t)pair * p.I;
    uint4 av[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = lane * 8 + i * 512;
      av[i] = make_uint4(0, 0, 0, 0);
      if (c < p.I) av[i] = *reinterpret_cast<const uint4*>(ar + c);
    }
    WFrag<WBITS> ring[D];
#pragma unroll
    for (int d = 0; d < D; d++)
      if (EXACT || d < NKS) ring[d] = load_wfrag_nt<WBITS>(wd + (size_t)d * TILE_BYTES, lane);
    const float4 sd = *reinterpret_cast<const float4*>(p.down_s + (size_t)e * p.H + strip * 16 + (lane >> 4) * 4);
    const float wt = p.weights[pair];
    // ---- a11 part 2: wave-local per-row int8 quantisation (moe_base.hpp:378-384 -> amx_buffers.hpp:47-98) ---------
    float amax = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) amax = amax8(av[i], amax);
    amax = wave_max(amax);
    ad = amax / 127.0f;
    const float aid = ad ? 1.0f / ad : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = lane * 8 + i * 512;
      if (c < p.I) *reinterpret_cast<uint2*>(aq + c) = quant8(av[i], aid);
    }
    if (lane == 0) { s_valid[j] = 1; s_wt[j] = wt; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the wave reads back its own LDS row: order only
    __builtin_amdgcn_wave_barrier();

    const uint8_t* bb = aq + (lane >> 4) * 32;
    if constexpr (EXACT) {
      const int G = NKS / D;
      for (int g = 0; g < G - 1; g++) {
        const uint8_t* bg = bb + g * D * 128;
        const uint8_t* wn = wd + (size_t)(g + 1) * D * TILE_BYTES;
#pragma unroll
        for (int d = 0; d < D; d++) {
          const v4i b0 = *reinterpret_cast<const v4i*>(bg + d * 128);
          const v4i b1 = *reinterpret_cast<const v4i*>(bg + d * 128 + 16);
          dec_step1<WBITS>(ring[d], b0, b1, acc0, acc1);
          ring[d] = load_wfrag_nt<WBITS>(wn + (size_t)d * TILE_BYTES, lane);
        }
      }
      const uint8_t* bg = bb + (G - 1) * D * 128;
#pragma unroll
      for (int d = 0; d < D; d++) {
        const v4i b0 = *reinterpret_cast<const v4i*>(bg + d * 128);
        const v4i b1 = *reinterpret_cast<const v4i*>(bg + d * 128 + 16);
        dec_step1<WBITS>(ring[d], b0, b1, acc0, acc1);
      }
    } else {
      for (int s0 = 0; s0 < NKS; s0 += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
          const int ks = s0 + d;
          if (ks < NKS) {
            const v4i b0 = *reinterpret_cast<const v4i*>(bb + ks * 128);
            const v4i b1 = *reinterpret_cast<const v4i*>(bb + ks * 128 + 16);
            dec_step1<WBITS>(ring[d], b0, b1, acc0, acc1);
            if (ks + D < NKS) ring[d] = load_wfrag_nt<WBITS>(wd + (size_t)(ks + D) * TILE_BYTES, lane);
          }
        }
      }
    }
    if ((lane & 15) == 0) {
      const int r0 = (lane >> 4) * 4;
      const float sdv[4] = {sd.x, sd.y, sd.z, sd.w};
#pragma unroll
      for (int r = 0; r < 4; r++)
        s_dn[j * 16 + r0 + r] = bf16_to_f32(f32_to_bf16((ad * sdv[r]) * (float)(acc0[r] + acc1[r])));
    }
  } else if (lane == 0) {
    s_valid[j] = 0;
    s_wt[j] = 0.0f;
  }
  if constexpr (SIDE_G > 0) {   // ---- side linear, part 2: this wave's k-steps (token slot 0 of the MFMA tile = this token)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the wave reads back its own staged pieces: order only
    __builtin_amdgcn_wave_barrier();
    v4f sacc = {0.f, 0.f, 0.f, 0.f};
    const uint8_t* xb0 = side_xs + (lane >> 4) * 16;
#pragma unroll
    for (int i = 0; i < SIDE_MAX; i++) {
      const int ks = j * spw + i;
      if (i < spw && ks < p.side_nks)
        ktxw4::w4_kstep<SIDE_G>(sw[i], ssc[i], xb0 + (size_t)ks * 256, 16, side_aux + ks * SGPK * 4, 4, sacc);
    }
    if (lane < 16) s_side[j * 16 + lane] = sacc[0];
  }
  __syncthreads();
  if (tid < 16) {  // a12: weighted combine in slot order, then a4
    float acc = 0.0f;
    for (int jj = 0; jj < p.k; jj++)
      if (s_valid[jj]) acc = fmaf(s_dn[jj * 16 + tid], s_wt[jj], acc);
    const size_t o = (size_t)t * p.H + strip * 16 + tid;
    if (p.partial_f32) {
      reinterpret_cast<float*>(p.y)[o] = acc;
    } else {
      bf16_t* yp = reinterpret_cast<bf16_t*>(p.y) + o;
      if (p.incremental) acc = acc + bf16_to_f32(*yp);
      bf16_t ob = f32_to_bf16(acc);
      if constexpr (SIDE_G > 0) {
        // the k-slices of the side strip in wavefront order; then the adds of the block with torch's bf16 tensor arithmetic as
        // lin_addends does them (ktx_linear.hip): shared + routed (experts.py:1004-1006), then residual + mlp (modeling_deepseek_v3.py:1225)
        float sv = 0.f;
        for (int jj = 0; jj < p.k; jj++) sv += s_side[jj * 16 + tid];
        const bf16_t sb = f32_to_bf16(sv);
        ob = f32_to_bf16(bf16_to_f32(ob) + bf16_to_f32(sb));
        if (p.side_add2) ob = f32_to_bf16(bf16_to_f32(p.side_add2[o]) + bf16_to_f32(ob));
      }
      *yp = ob;
    }
  }
}

// =====================================================================================================
// K3: weighted combine in slot order (a12) + merge/incremental + bf16 (a4)
// =====================================================================================================
struct CombineParams {
  const int32_t* d_bsz;
  int qlen, k, H;
  const bf16_t* dn;            // [sorted rows][H]
  const int32_t* row_of_pair;  // [qlen*k]
  const float* weights;        // [qlen][k]
  bf16_t* y;                   // [qlen][H]  (float* when partial_f32)
  int incremental;
  int partial_f32;
  int dn_f32;                  // GGUF path: dn is fp32 and the sum is `out += dn * w` (llamafile/moe.hpp:447-449)
};

__global__ __launch_bounds__(256) void moe_combine_kernel(CombineParams p) {
  int T = p.qlen;
  if (p.d_bsz) T = min(max(*p.d_bsz, 0), p.qlen);
  const int t = blockIdx.y;
  if (t >= T) return;
  const int h = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (h >= p.H) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < p.k; j++) {
    const int r = p.row_of_pair[t * p.k + j];
    if (r < 0) continue;
    const float w = p.weights[t * p.k + j];
    if (p.dn_f32) {
      const float4 f = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.dn) + (size_t)r * p.H + h);
      acc[0] = acc[0] + f.x * w; acc[1] = acc[1] + f.y * w; acc[2] = acc[2] + f.z * w; acc[3] = acc[3] + f.w * w;
      continue;
    }
    const uint2 v = *reinterpret_cast<const uint2*>(p.dn + (size_t)r * p.H + h);
    acc[0] = fmaf(bf16_to_f32((bf16_t)(v.x & 0xffffu)), w, acc[0]);
    acc[1] = fmaf(bf16_to_f32((bf16_t)(v.x >> 16)), w, acc[1]);
    acc[2] = fmaf(bf16_to_f32((bf16_t)(v.y & 0xffffu)), w, acc[2]);
    acc[3] = fmaf(bf16_to_f32((bf16_t)(v.y >> 16)), w, acc[3]);
  }
  if (p.partial_f32) {  // expert-parallel partial: un-rounded fp32 sums, reduced across ranks by the caller
    float* fp = reinterpret_cast<float*>(p.y) + (size_t)t * p.H + h;
    *reinterpret_cast<float4*>(fp) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    return;
  }
  bf16_t* yp = p.y + (size_t)t * p.H + h;
  if (p.incremental) {
    const uint2 o = *reinterpret_cast<const uint2*>(yp);
    acc[0] = acc[0] + bf16_to_f32((bf16_t)(o.x & 0xffffu));
    acc[1] = acc[1] + bf16_to_f32((bf16_t)(o.x >> 16));
    acc[2] = acc[2] + bf16_to_f32((bf16_t)(o.y & 0xffffu));
    acc[3] = acc[3] + bf16_to_f32((bf16_t)(o.y >> 16));
  }
  const uint32_t lo = (uint32_t)f32_to_bf16(acc[0]) | ((uint32_t)f32_to_bf16(acc[1]) << 16);
  const uint32_t hi = (uint32_t)f32_to_bf16(acc[2]) | ((uint32_t)f32_to_bf16(acc[3]) << 16);
  *reinterpret_cast<uint2*>(yp) = make_uint2(lo, hi);
}

// =====================================================================================================
// load-time: quantise bf16 rows (a7) and pack into the W tile layout
// =====================================================================================================
// One block per row.  fmt 0 (AMXINT4): d = float(double(amax)/112.0), q = round_4bit_s8(sat8(rne(w*(1/d)))).
//                     fmt 1 (AMXINT8): d = amax/127,                 q = sat8(rne(w*(1/d))).
__global__ __launch_bounds__(256) void quant_rows_kernel(const bf16_t* __restrict__ w, int K, int fmt,
                                                         int8_t* __restrict__ q, float* __restrict__ d_out) {
  __shared__ float s_red[4];
  const int tid = threadIdx.x;
  const size_t row = blockIdx.x;
  const bf16_t* src = w + row * K;
  float amax = 0.0f;
  for (int j = tid; j < K; j += 256) amax = fmaxf(amax, fabsf(bf16_to_f32(src[j])));
  amax = wave_max(amax);
  if ((tid & 63) == 0) s_red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  float d;
  if (fmt == 0) d = (float)((double)amax / 112.0);
  else d = amax / 127.0f;
  const float id = d ? 1.0f / d : 0.0f;
  for (int j = tid; j < K; j += 256) {
    int v = quant_rne_sat8(bf16_to_f32(src[j]) * id);
    if (fmt == 0) {  // round_4bit_s8 (amx_buffers.hpp:527-539): sign(i) * ((|i| + 8) & 0xF0)
      int a = v < 0 ? -v : v;
      a = (a + 8) & 0xF0;
      v = v < 0 ? -a : a;
    }
    q[row * K + j] = (int8_t)v;
  }
  if (tid == 0) d_out[row] = d;
}

// q: int8 [N][K] row-major -> W4 tiles.  One thread per packed dword.
__global__ void pack_w4_kernel(const int8_t* __restrict__ q, int N, int K, uint32_t* __restrict__ out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // dword index
  const size_t total = (size_t)N * K / 8;
  if (idx >= total) return;
  const int NKS = K / 128;
  const size_t tile = idx / 256;
  const int within = (int)(idx % 256);
  const int lane = within / 4, P = within % 4;
  const int strip = (int)(tile / NKS), ks = (int)(tile % NKS);
  const int i = lane & 15, kc = lane >> 4, h = P >> 1, hi8 = (P & 1) * 8;
  const int8_t* row = q + (size_t)(strip * 16 + i) * K + ks * 128 + kc * 32 + h * 16 + hi8;
  uint32_t v = 0;
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const uint32_t lo = ((uint32_t)(uint8_t)row[b] >> 4) & 0x0Fu;
    const uint32_t hi = (uint32_t)(uint8_t)row[4 + b] & 0xF0u;
    v |= (hi | lo) << (8 * b);
  }
  out[idx] = v;
}

// q: int8 [N][K] row-major -> W8 tiles.  One thread per dword.
__global__ void pack_w8_kernel(const int8_t* __restrict__ q, int N, int K, uint32_t* __restrict__ out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)N * K / 4;
  if (idx >= total) return;
  const int NKS = K / 128;
  const size_t tile = idx / 512;
  const int within = (int)(idx % 512);
  const int h = within / 256, lane = (within % 256) / 4, dw = within % 4;
  const int strip = (int)(tile / NKS), ks = (int)(tile % NKS);
  const int i = lane & 15, kc = lane >> 4;
  const int8_t* row = q + (size_t)(strip * 16 + i) * K + ks * 128 + kc * 32 + h * 16 + dw * 4;
  out[idx] = (uint32_t)(uint8_t)row[0] | ((uint32_t)(uint8_t)row[1] << 8) | ((uint32_t)(uint8_t)row[2] << 16) |
             ((uint32_t)(uint8_t)row[3] << 24);
}

// =====================================================================================================
// FP8 (DeepSeek e4m3, 128x128 block scale_inv) and BF16 experts: bf16 activations, bf16 MFMA, fp32 accumulation.
// Reference: AMX_FP8_MOE_TP / AMX_BF16_MOE_TP (operators/amx/fp8-moe.hpp:93-108, bf16-moe.hpp) over
// GemmKernel224FP8 / GemmKernel224BF16 (operators/amx/la/amx_raw_kernels.hpp:17-566): weights are widened to bf16
// exactly, activations are NOT quantised, products accumulate in fp32 (per 128-K group for FP8, then
// c = fma(group_sum, scale_inv[n/128][k/128], c)), every stage output is rounded to bf16.
// The reference's fp32 summation order inside a group is a sequential VDPBF16PS chain (and differs again on its AMX tile
// path); the MFMA sums the same exact products in a different order, so outputs agree to fp32 rounding of the group
// sums (tests: <= 1 bf16 ulp on a small fraction of elements), not bit-for-bit.
//
// W tile layout for these formats (k-step = 128, four 16x16x32 MFMAs j = 0..3; fragment element e <-> k = ks*128 +
// kc*32 + j*8 + e):  FP8: tile 2048 B, lane l owns bytes [l*16, +16) of each of two 1 KiB halves (elements 0..15, 16..31
// of its 32-k run);  BF16: tile 4096 B, four 1 KiB quarters, quarter j = the lane's fragment for MFMA j.
// Activations are staged per 256-k chunk as [mt][col = 32][tok = 16][16 B] (same conflict-free scheme as the int path).
// =====================================================================================================
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

__device__ __forceinline__ v8bf u4_as_v8bf(const uint4& u) {
  union { uint4 u; v8bf v; } c;
  c.u = u;
  return c.v;
}

// 4 e4m3 bytes -> 4 bf16 (two dwords): v_cvt_pk_f32_fp8 is exact, and every e4m3 value is exact in bf16, so taking the
// upper halves of the fp32 results is exact (OCP e4m3fn on gfx950; 0x7F/0xFF are NaN here, +-480 in the reference's LUT).
__device__ __forceinline__ uint2 fp8x4_to_bf16x4(uint32_t v) {
  // gfx950: v_cvt_scalef32_pk_bf16_fp8 converts two e4m3 bytes straight to two bf16 (scale 1.0): 2 VALU instructions per four weights
  // instead of two converts + two packs (scripts/fp8_cvt_probe.hip: the same bits as the two-step path for all 256 codes)
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  union { bf2 v; uint32_t u; } lo, hi;
  lo.v = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, 1.0f, false);
  hi.v = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, 1.0f, true);
  return make_uint2(lo.u, hi.u);
}

struct FpGemmParams {
  const uint8_t *w0, *w1;     // tiled weights (gate | down, up)
  const float *s0, *s1;       // FP8: scale_inv [E][N/128][K/128]; BF16: nullptr
  const float *r0, *r1;       // FP8_PERCHANNEL: scale per output row [E][N] (the block scales are then all 1), else nullptr
  size_t expert_stride;       // bytes per expert matrix
  size_t scale_stride;        // floats per expert
  int N, K;
  const bf16_t* act;          // [src rows][K] bf16
  const int32_t* row_src;     // sorted row -> source row (nullptr: identity)
  const Tile* tiles;
  const int32_t* counters;
  bf16_t* out;                // [sorted rows][N]
};

// WIDE (round 4, the prompt configuration MT = 4): 8 wavefronts per workgroup and two B operands per wavefront for BOTH GEMMs (gate and up
// of a strip; two strips of down).  At 64 rows per expert every weight byte is read once, but every workgroup re-reads the tile's
// bf16 activations — with 4 strips per workgroup that was as many bytes as the fp8 weights themselves (7.5 GB against 6.4 GB per
// DeepSeek-V3 gate|up layer; 3.5x the weights for down) and the kernel ran at the CUs' ingest rate, not HBM's (profiles/r04_bench2).
// Per output the arithmetic is unchanged (same MFMA chain per 128-k group, same scale fma): bit-identical results.
template <bool FP8, int MT, bool GATE_UP, bool WIDE = false>
__global__ __launch_bounds__(WIDE ? 512 : 256) void moe_gemm_fp_kernel(FpGemmParams p) {
  constexpr int NMAT = (GATE_UP || WIDE) ? 2 : 1;      // B operands (units) per wavefront
  constexpr int NWV = WIDE ? 8 : 4;
  constexpr int SPC = 2;                      // k-steps per chunk
  constexpr int COLS = SPC * 16;              // 16-byte columns per chunk (128 bf16 = 256 B per step)
  constexpr int BUF_BYTES = MT * COLS * 256;
  constexpr int GROUPS = MT * COLS / 4;       // staging groups of (16 tok x 4 cols)
  constexpr int UPT = (GROUPS + NWV - 1) / NWV;
  constexpr int TILE_BYTES = FP8 ? 2048 : 4096;
  constexpr int NQ = FP8 ? 2 : 4;             // uint4 per lane per k-step per matrix

  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t* Bs = smem;                                            // [2][BUF_BYTES]
  int* s_src = reinterpret_cast<int*>(smem + 2 * BUF_BYTES);     // [MT*16]

  const int tile_idx = blockIdx.y;
  if (tile_idx >= p.counters[0]) return;
  const Tile tile = p.tiles[tile_idx];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NKS = p.K / 128;
  const int NC = (NKS + SPC - 1) / SPC;
  int ustrip[NMAT];
  bool uok[NMAT];
#pragma unroll
  for (int m = 0; m < NMAT; m++) {
    ustrip[m] = GATE_UP ? blockIdx.x * NWV + wave : (WIDE ? (blockIdx.x * NWV + wave) * 2 + m : blockIdx.x * NWV + wave);
    uok[m] = ustrip[m] * 16 < p.N;
  }
  const bool strip_ok = uok[0];      // (a wavefront's second strip of down can only be missing when the first is present)

  if (tid < MT * 16) s_src[tid] = tid < tile.nrows ? (p.row_src ? p.row_src[tile.row0 + tid] : tile.row0 + tid) : -1;
  __syncthreads();

  const uint8_t* wbase[NMAT];
  const float* sbase[NMAT];
#pragma unroll
  for (int m = 0; m < NMAT; m++) {
    const int st = uok[m] ? ustrip[m] : 0;                  // a strip past N aliases strip 0 and is not stored
    const uint8_t* w = (GATE_UP && m) ? p.w1 : p.w0;
    const float* sc = (GATE_UP && m) ? p.s1 : p.s0;
    wbase[m] = w + (size_t)tile.expert * p.expert_stride + (size_t)st * NKS * TILE_BYTES;
    sbase[m] = FP8 ? sc + (size_t)tile.expert * p.scale_stride + (size_t)(st * 16 / 128) * NKS : nullptr;
  }

  v4f acc[NMAT][MT];
#pragma unroll
  for (int m = 0; m < NMAT; m++)
#pragma unroll
    for (int t = 0; t < MT; t++) acc[m][t] = v4f{0.f, 0.f, 0.f, 0.f};

  uint4 wA[SPC][NMAT][NQ], wB[SPC][NMAT][NQ];
  uint4 breg[UPT];

  auto load_w = [&](uint4(&dst)[SPC][NMAT][NQ], int c) {
#pragma unroll
    for (int s = 0; s < SPC; s++) {
      const int ks = c * SPC + s;
      if (strip_ok && ks < NKS) {
#pragma unroll
        for (int m = 0; m < NMAT; m++)
#pragma unroll
          for (int q = 0; q < NQ; q++)
            dst[s][m][q] = *reinterpret_cast<const uint4*>(wbase[m] + (size_t)ks * TILE_BYTES + q * 1024 + lane * 16);
      }
    }
  };
  auto load_b = [&](int c) {
#pragma unroll
    for (int it = 0; it < UPT; it++) {
      const int g = it * NWV + wave;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (g < GROUPS) {
        const int mt = g / (COLS / 4), col = (g % (COLS / 4)) * 4 + (lane >> 4);
        const int src = s_src[mt * 16 + (lane & 15)];
        const int kb = c * SPC * 128 + col * 8;   // bf16 element index of this 16-byte column
        if (src >= 0 && kb < p.K) v = *reinterpret_cast<const uint4*>(p.act + (size_t)src * p.K + kb);
      }
      breg[it] = v;
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int it = 0; it < UPT; it++) {
      const int g = it * NWV + wave;
      if (g < GROUPS) {
        const int mt = g / (COLS / 4), col = (g % (COLS / 4)) * 4 + (lane >> 4);
        *reinterpret_cast<uint4*>(Bs + buf * BUF_BYTES + ((mt * COLS + col) * 16 + (lane & 15)) * 16) = breg[it];
      }
    }
  };
  auto compute = [&](uint4(&w)[SPC][NMAT][NQ], int c, int buf) {
    if (!strip_ok) return;
    // B fragment of MFMA j in step s: column s*16 + kc*4 + j, token lane&15
    const uint8_t* bb = Bs + buf * BUF_BYTES + ((lane >> 4) * 4 * 16 + (lane & 15)) * 16;
#pragma unroll
    for (int s = 0; s < SPC; s++) {
      const int ks = c * SPC + s;
      if (ks < NKS) {
        v8bf a[NMAT][4];
#pragma unroll
        for (int m = 0; m < NMAT; m++) {
          if constexpr (FP8) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
              const uint2 e0 = fp8x4_to_bf16x4(w[s][m][q].x), e1 = fp8x4_to_bf16x4(w[s][m][q].y);
              const uint2 e2 = fp8x4_to_bf16x4(w[s][m][q].z), e3 = fp8x4_to_bf16x4(w[s][m][q].w);
              a[m][q * 2] = u4_as_v8bf(make_uint4(e0.x, e0.y, e1.x, e1.y));
              a[m][q * 2 + 1] = u4_as_v8bf(make_uint4(e2.x, e2.y, e3.x, e3.y));
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++) a[m][j] = u4_as_v8bf(w[s][m][j]);
          }
        }
        float sc[NMAT];
#pragma unroll
        for (int m = 0; m < NMAT; m++) sc[m] = FP8 ? sbase[m][ks] : 1.0f;
#pragma unroll
        for (int t = 0; t < MT; t++) {
          v4f tmp[NMAT];
#pragma unroll
          for (int m = 0; m < NMAT; m++) tmp[m] = FP8 ? v4f{0.f, 0.f, 0.f, 0.f} : acc[m][t];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const v8bf b = u4_as_v8bf(*reinterpret_cast<const uint4*>(bb + ((t * COLS + s * 16 + j) * 16) * 16));
#pragma unroll
            for (int m = 0; m < NMAT; m++) tmp[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][j], b, tmp[m], 0, 0, 0);
          }
#pragma unroll
          for (int m = 0; m < NMAT; m++) {
            if constexpr (FP8) {
#pragma unroll
              for (int r = 0; r < 4; r++) acc[m][t][r] = fmaf(tmp[m][r], sc[m], acc[m][t][r]);  // apply_scale_kgroup
            } else {
              acc[m][t] = tmp[m];
            }
          }
        }
      }
    }
  };

  load_w(wA, 0);
  load_b(0);
  store_b(0);
  __syncthreads();
  for (int c = 0; c < NC; c += 2) {
    if (c + 1 < NC) { load_b(c + 1); load_w(wB, c + 1); }
    compute(wA, c, 0);
    if (c + 1 < NC) store_b(1);
    __syncthreads();
    if (c + 1 < NC) {
      if (c + 2 < NC) { load_b(c + 2); load_w(wA, c + 2); }
      compute(wB, c + 1, 1);
      if (c + 2 < NC) store_b(0);
      __syncthreads();
    }
  }

  if (!strip_ok) return;
  const int tok = lane & 15;
#pragma unroll
  for (int m = 0; m < (GATE_UP ? 1 : NMAT); m++) {       // gate|up: one output per wavefront (both units); down: one per unit
    if (!uok[m]) continue;
    const int n0 = ustrip[m] * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int t = 0; t < MT; t++) {
      const int row = t * 16 + tok;
      if (row < tile.nrows) {
        bf16_t o[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          // FP8_PERCHANNEL: the whole-K sum times the row's scale (apply_scale_perchannel, amx_raw_kernels.hpp:686-700); x * 1.0f
          // is exact, so the other formats are untouched
          const float rs0 = p.r0 ? p.r0[(size_t)tile.expert * p.N + n0 + r] : 1.0f;
          const bf16_t g = f32_to_bf16(acc[m][t][r] * rs0);
          if constexpr (GATE_UP) {
            const float rs1 = p.r1 ? p.r1[(size_t)tile.expert * p.N + n0 + r] : 1.0f;
            const bf16_t u = f32_to_bf16(acc[1][t][r] * rs1);
            o[r] = f32_to_bf16(act_fn(bf16_to_f32(g), bf16_to_f32(u)));
          } else {
            o[r] = g;
          }
        }
        *reinterpret_cast<uint2*>(p.out + (size_t)(tile.row0 + row) * p.N + n0) =
            make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
      }
    }
  }
}

// =====================================================================================================
// Decode fast path of the FP8 / BF16 experts (qlen*k <= KTX_DEC_MAX_PAIRS): the two-launch structure of the int formats
// (moe_dec_gateup_kernel / moe_dec_down_kernel) with the arithmetic of moe_gemm_fp_kernel — bf16 activations as they are,
// weights widened to bf16 exactly, bf16 MFMA, fp32 accumulation (per 128-K group for FP8, then c = fma(group, scale, c)),
// every stage rounded to bf16.  Reference analogue: forward_decode of operators/amx/moe_base.hpp:464-654 over
// GemmKernel224FP8 / GemmKernel224BF16 (la/amx_raw_kernels.hpp:334-566, :17-256).  Before this path existed a decode token
// of these formats went bucket -> grouped GEMM -> grouped GEMM -> combine: four launches of M-tiled kernels at M = 1.
// All 16 B-operand columns of the MFMA carry the same token, so no lane masking is needed.
// =====================================================================================================
struct DecFpParams {
  const int32_t* d_bsz;
  int qlen, k, E, expert_begin, H, I;
  const int64_t* ids;
  const uint8_t* mask;
  const bf16_t* x;
  const float* weights;
  const uint8_t *gate_w, *up_w, *down_w;
  const float *gate_s, *up_s, *down_s;   // FP8: scale_inv [E][N/128][K/128]
  const float *gate_r, *up_r, *down_r;   // FP8_PERCHANNEL: scale per output row [E][N] (block scales all 1), else nullptr
  size_t gu_stride, dn_stride;           // bytes per expert matrix
  bf16_t* a_buf;                         // [qlen*k][I]
  void* y;
  int incremental, partial_f32;
};

template <bool FP8>
struct FpSlot {   // one k-step of one matrix in registers: the lane's 32 weights (+ the group's scale)
  uint4 q[FP8 ? 2 : 4];
  float sc;
};
template <bool FP8>
__device__ __forceinline__ FpSlot<FP8> fp_load_slot(const uint8_t* tile, const float* sc, int lane) {
  typedef unsigned int u4v __attribute__((ext_vector_type(4)));
  FpSlot<FP8> s;
#pragma unroll
  for (int q = 0; q < (FP8 ? 2 : 4); q++) {
    const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(tile + q * 1024 + lane * 16));
    s.q[q] = make_uint4(v.x, v.y, v.z, v.w);
  }
  s.sc = FP8 ? *sc : 1.0f;
  return s;
}
// acc += W_slot (16 features x 128 k) . x (128 k, one token); xs = LDS address of this lane's part of the k-step
template <bool FP8>
__device__ __forceinline__ void fp_dec_step(const FpSlot<FP8>& w, const uint8_t* xs, v4f& acc) {
  v8bf a[4];
  if constexpr (FP8) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const uint2 e0 = fp8x4_to_bf16x4(w.q[q].x), e1 = fp8x4_to_bf16x4(w.q[q].y);
      const uint2 e2 = fp8x4_to_bf16x4(w.q[q].z), e3 = fp8x4_to_bf16x4(w.q[q].w);
      a[q * 2] = u4_as_v8bf(make_uint4(e0.x, e0.y, e1.x, e1.y));
      a[q * 2 + 1] = u4_as_v8bf(make_uint4(e2.x, e2.y, e3.x, e3.y));
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) a[j] = u4_as_v8bf(w.q[j]);
  }
  v4f tmp = FP8 ? v4f{0.f, 0.f, 0.f, 0.f} : acc;
#pragma unroll
  for (int j = 0; j < 4; j++)
    tmp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], u4_as_v8bf(*reinterpret_cast<const uint4*>(xs + j * 16)), tmp, 0, 0, 0);
  if constexpr (FP8) {
#pragma unroll
    for (int r = 0; r < 4; r++) acc[r] = fmaf(tmp[r], w.sc, acc[r]);   // apply_scale_kgroup
  } else {
    acc = tmp;
  }
}

// one workgroup per ((t,j) pair, NW strips): x[t] staged in LDS as bf16, gate and up strips of expert ids[t][j] through a
// D-deep register ring (NKS % D == 0: branch-free), SiLU*up epilogue -> a_buf[pair]
template <bool FP8, int D, int NW>
__global__ __launch_bounds__(NW * 64) void moe_dec_fp_gateup_kernel(DecFpParams p) {
  constexpr int TILE_BYTES = FP8 ? 2048 : 4096;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // [H] bf16
  int T = p.qlen;
  if (p.d_bsz) T = min(max(*p.d_bsz, 0), p.qlen);
  const int pair = blockIdx.y, t = pair / p.k;
  if (t >= T) return;
  const long long idl = p.ids[pair] - p.expert_begin;
  if (idl < 0 || idl >= p.E || (p.mask && p.mask[idl])) return;
  const int e = (int)idl;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = blockIdx.x * NW + wave;
  const bool strip_ok = strip * 16 < p.I;
  const int strip_c = strip_ok ? strip : 0;
  const int NKS = p.H / 128;
  // activations first (vmcnt retires in order), then the ring
  const bf16_t* xr = p.x + (size_t)t * p.H;
  uint4 xv[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int j = tid * 8 + i * NW * 512;
    xv[i] = make_uint4(0, 0, 0, 0);
    if (j < p.H) xv[i] = *reinterpret_cast<const uint4*>(xr + j);
  }
  const uint8_t* wg = p.gate_w + (size_t)e * p.gu_stride + (size_t)strip_c * NKS * TILE_BYTES;
  const uint8_t* wu = p.up_w + (size_t)e * p.gu_stride + (size_t)strip_c * NKS * TILE_BYTES;
  const size_t sstride = (size_t)(p.I / 128) * NKS;
  const float* sg = FP8 ? p.gate_s + (size_t)e * sstride + (size_t)(strip_c * 16 / 128) * NKS : nullptr;
  const float* su = FP8 ? p.up_s + (size_t)e * sstride + (size_t)(strip_c * 16 / 128) * NKS : nullptr;
  FpSlot<FP8> ring[D][2];
#pragma unroll
  for (int d = 0; d < D; d++) {
    ring[d][0] = fp_load_slot<FP8>(wg + (size_t)d * TILE_BYTES, sg + d, lane);
    ring[d][1] = fp_load_slot<FP8>(wu + (size_t)d * TILE_BYTES, su + d, lane);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int j = tid * 8 + i * NW * 512;
    if (j < p.H) *reinterpret_cast<uint4*>(smem + j * 2) = xv[i];
  }
  __syncthreads();
  if (!strip_ok) return;
  v4f accg = {0.f, 0.f, 0.f, 0.f}, accu = {0.f, 0.f, 0.f, 0.f};
  const uint8_t* xb = smem + (lane >> 4) * 64;
  const int G = NKS / D;
  for (int g = 0; g < G - 1; g++) {
#pragma unroll
    for (int d = 0; d < D; d++) {
      const int ks = g * D + d;
      fp_dec_step<FP8>(ring[d][0], xb + ks * 256, accg);
      fp_dec_step<FP8>(ring[d][1], xb + ks * 256, accu);
      ring[d][0] = fp_load_slot<FP8>(wg + (size_t)(ks + D) * TILE_BYTES, sg + ks + D, lane);
      ring[d][1] = fp_load_slot<FP8>(wu + (size_t)(ks + D) * TILE_BYTES, su + ks + D, lane);
    }
  }
#pragma unroll
  for (int d = 0; d < D; d++) {
    const int ks = (G - 1) * D + d;
    fp_dec_step<FP8>(ring[d][0], xb + ks * 256, accg);
    fp_dec_step<FP8>(ring[d][1], xb + ks * 256, accu);
  }
  if ((lane & 15) == 0) {
    const int n0 = strip * 16 + (lane >> 4) * 4;
    bf16_t o[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      // FP8_PERCHANNEL: whole-K sum times the row's scale (x * 1.0f is exact: the other formats are untouched)
      const float rg = p.gate_r ? p.gate_r[(size_t)e * p.I + n0 + r] : 1.0f, ru = p.up_r ? p.up_r[(size_t)e * p.I + n0 + r] : 1.0f;
      const bf16_t gq = f32_to_bf16(accg[r] * rg), uq = f32_to_bf16(accu[r] * ru);
      o[r] = f32_to_bf16(act_fn(bf16_to_f32(gq), bf16_to_f32(uq)));
    }
    *reinterpret_cast<uint2*>(p.a_buf + (size_t)pair * p.I + n0) =
        make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
  }
}

// one workgroup per (token, 16-row strip of H), wave j = slot j: the activated row of pair (t,j) staged (wave-private) in
// LDS, the down strip of its expert streamed, then the slot-ordered weighted combine (a12) and the merge step (a4)
template <bool FP8, int D>
__global__ __launch_bounds__(512) void moe_dec_fp_down_kernel(DecFpParams p) {
  constexpr int TILE_BYTES = FP8 ? 2048 : 4096;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // [k][I] bf16 activations | [k][16] fp32 down outputs | [k] valid flags | [k] routing weights
  uint8_t* a_all = smem;
  float* s_dn = reinterpret_cast<float*>(smem + (size_t)p.k * p.I * 2);
  int* s_valid = reinterpret_cast<int*>(s_dn + p.k * 16);
  float* s_wt = reinterpret_cast<float*>(s_valid + p.k);
  int T = p.qlen;
  if (p.d_bsz) T = min(max(*p.d_bsz, 0), p.qlen);
  const int t = blockIdx.y;
  if (t >= T) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = t * p.k + j, strip = blockIdx.x;
  const long long idl = p.ids[pair] - p.expert_begin;
  const bool valid = !(idl < 0 || idl >= p.E || (p.mask && p.mask[idl]));
  const int e = valid ? (int)idl : 0;
  const int NKS = p.I / 128;
  if (valid) {
    const bf16_t* ar = p.a_buf + (size_t)pair * p.I;
    uint8_t* aw = a_all + (size_t)j * p.I * 2;
    uint4 av[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int c = lane * 8 + i * 512;
      av[i] = make_uint4(0, 0, 0, 0);
      if (c < p.I) av[i] = *reinterpret_cast<const uint4*>(ar + c);
    }
    const uint8_t* wd = p.down_w + (size_t)e * p.dn_stride + (size_t)strip * NKS * TILE_BYTES;
    const float* sd = FP8 ? p.down_s + (size_t)e * (size_t)(p.H / 128) * NKS + (size_t)(strip * 16 / 128) * NKS : nullptr;
    FpSlot<FP8> ring[D];
#pragma unroll
    for (int d = 0; d < D; d++) ring[d] = fp_load_slot<FP8>(wd + (size_t)d * TILE_BYTES, sd + d, lane);
    const float wt = p.weights[pair];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int c = lane * 8 + i * 512;
      if (c < p.I) *reinterpret_cast<uint4*>(aw + c * 2) = av[i];
    }
    if (lane == 0) { s_valid[j] = 1; s_wt[j] = wt; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the wave reads back its own LDS row: order only
    __builtin_amdgcn_wave_barrier();
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    const uint8_t* xb = aw + (lane >> 4) * 64;
    const int G = NKS / D;
    for (int g = 0; g < G - 1; g++) {
#pragma unroll
      for (int d = 0; d < D; d++) {
        const int ks = g * D + d;
        fp_dec_step<FP8>(ring[d], xb + ks * 256, acc);
        ring[d] = fp_load_slot<FP8>(wd + (size_t)(ks + D) * TILE_BYTES, sd + ks + D, lane);
      }
    }
#pragma unroll
    for (int d = 0; d < D; d++) fp_dec_step<FP8>(ring[d], xb + ((G - 1) * D + d) * 256, acc);
    if ((lane & 15) == 0) {
      const int r0 = (lane >> 4) * 4;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float rd = p.down_r ? p.down_r[(size_t)e * p.H + strip * 16 + r0 + r] : 1.0f;   // FP8_PERCHANNEL row scale
        s_dn[j * 16 + r0 + r] = bf16_to_f32(f32_to_bf16(acc[r] * rd));
      }
    }
  } else if (lane == 0) {
    s_valid[j] = 0;
    s_wt[j] = 0.0f;
  }
  __syncthreads();
  if (tid < 16) {  // a12: weighted combine in slot order, then a4
    float acc = 0.0f;
    for (int jj = 0; jj < p.k; jj++)
      if (s_valid[jj]) acc = fmaf(s_dn[jj * 16 + tid], s_wt[jj], acc);
    const size_t o = (size_t)t * p.H + strip * 16 + tid;
    if (p.partial_f32) {
      reinterpret_cast<float*>(p.y)[o] = acc;
    } else {
      bf16_t* yp = reinterpret_cast<bf16_t*>(p.y) + o;
      if (p.incremental) acc = acc + bf16_to_f32(*yp);
      *yp = f32_to_bf16(acc);
    }
  }
}

// row-major [N][K] (fp8 bytes or bf16) -> W tiles of the fp formats; one thread per 16-byte piece
__global__ void pack_wfp_kernel(const uint8_t* __restrict__ src, int N, int K, int fp8, uint4* __restrict__ out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int esz = fp8 ? 1 : 2, nq = fp8 ? 2 : 4, per16 = 16 / esz;   // elements per 16-byte piece
  const size_t total = (size_t)N * K * esz / 16;
  if (idx >= total) return;
  const int NKS = K / 128;
  const size_t tile = idx / (64 * nq);
  const int within = (int)(idx % (64 * nq));
  const int q = within / 64, lane = within % 64;
  const int strip = (int)(tile / NKS), ks = (int)(tile % NKS);
  const int i = lane & 15, kc = lane >> 4;
  const size_t k0 = (size_t)ks * 128 + kc * 32 + q * per16;
  out[idx] = *reinterpret_cast<const uint4*>(src + ((size_t)(strip * 16 + i) * K + k0) * esz);
}

// =====================================================================================================
// RAWINT4 (Kimi-K2 compressed-tensors int4, group 32, bf16 scales): bit-exact restatement of
// GemmKernel224Int4SmallKGroup (operators/amx/la/amx_kernels.hpp:3344-3597) on v_mfma_i32_4x4x4i8.
//
// The reference keeps, per output, SIXTEEN fp32 lane accumulators: AVX lane L owns k = 64*kb + 4L..4L+3 of every 64-K
// block, adds fma(as[g]*bs[g], float(dot4), s_L) block after block (g = 2kb + (L>=8)), and only at the end reduces the
// 16 lanes with _mm512_reduce_add_ps' fixed tree and divides by 16.  A 64-deep MFMA cannot reproduce that association,
// but the 16-block 4x4x4 int8 MFMA computes exactly those dot4 partial sums: block b = AVX lane L, A rows = 4 tokens,
// B columns = 4 weight rows, K = the lane's 4 k (layout probed: scripts/mfma4_probe.hip).  GPU lane (L, j) therefore
// holds s_L for weight row j and tokens r = 0..3, and the final butterfly over lane bits 5,4,3,2 is the reference's tree.
//
// RAW tile layout: weight rows in groups of 4, k in steps of 512 (8 blocks of 64): tile (rg, step) = 1 KiB, lane
// l = L*4 + j owns 16 B = dwords P[0..3]; byte b of P[p]: low nibble = q[row j][64*(8*step+2p) + 4L + b],
// high nibble = q[row j][64*(8*step+2p+1) + 4L + b] (two's complement nibbles), so ((P<<4)&0xF0F0F0F0, P&0xF0F0F0F0)
// are the two B operands (multiplicand 16q, as in the reference).  Scales stay bf16: [rg][step][row j][h][8 blocks].
// =====================================================================================================
struct RawGemmParams {
  const uint8_t *w0, *w1;      // RAW tiles (gate | down, up)
  const bf16_t *s0, *s1;       // bf16 scales in the tile order above
  size_t expert_stride;        // bytes of weights per expert matrix
  size_t scale_stride;         // bf16 elements of scales per expert matrix
  int N, K;
  const int8_t* act_q;         // [src rows][K]
  const float* act_d;          // [src rows][K/32]
  const int32_t* row_src;
  const Tile* tiles;           // <= 4 rows per tile
  const int32_t* counters;
  bf16_t* out;                 // [sorted rows][N]
};

template <int NT, bool GATE_UP>
__global__ __launch_bounds__(256) void moe_rawint4_gemm_kernel(RawGemmParams p) {
  constexpr int NMAT = GATE_UP ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int KP = p.K + 32;                                        // padded activation row (bank spread of the 4 tokens)
  int8_t* xq = reinterpret_cast<int8_t*>(smem);                   // [4][K+32]
  float* as_l = reinterpret_cast<float*>(smem + 4 * KP);          // [K/32][4]
  __shared__ int s_src[4];

  const int tile_idx = blockIdx.y;
  if (tile_idx >= p.counters[0]) return;
  const Tile tile = p.tiles[tile_idx];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = blockIdx.x * 4 + wave;
  const bool strip_ok = strip * 16 < p.N;
  const int NS = p.K / 512, G = p.K / 32;
  const int L = lane >> 2, j = lane & 3, hsel = L >> 3;

  if (tid < 4) s_src[tid] = tid < tile.nrows ? (p.row_src ? p.row_src[tile.row0 + tid] : tile.row0 + tid) : -1;
  __syncthreads();
  for (int u = tid; u < 4 * (p.K / 16); u += 256) {
    const int r = u / (p.K / 16), piece = u % (p.K / 16);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (s_src[r] >= 0) v = *reinterpret_cast<const uint4*>(p.act_q + (size_t)s_src[r] * p.K + piece * 16);
    *reinterpret_cast<uint4*>(xq + r * KP + piece * 16) = v;
  }
  for (int u = tid; u < 4 * G; u += 256) {
    const int g = u >> 2, r = u & 3;
    as_l[u] = s_src[r] >= 0 ? p.act_d[(size_t)s_src[r] * G + g] : 0.0f;
  }
  __syncthreads();
  if (!strip_ok) return;

  float acc[NMAT][4][NT];
#pragma unroll
  for (int m = 0; m < NMAT; m++)
#pragma unroll
    for (int rg = 0; rg < 4; rg++)
#pragma unroll
      for (int r = 0; r < NT; r++) acc[m][rg][r] = 0.0f;

  const uint8_t* wb[NMAT];
  const bf16_t* sb[NMAT];
  wb[0] = p.w0 + (size_t)tile.expert * p.expert_stride + (size_t)(strip * 4) * NS * 1024 + lane * 16;
  sb[0] = p.s0 + (size_t)tile.expert * p.scale_stride + (size_t)(strip * 4) * NS * 64 + (j * 2 + hsel) * 8;
  if constexpr (GATE_UP) {
    wb[1] = p.w1 + (size_t)tile.expert * p.expert_stride + (size_t)(strip * 4) * NS * 1024 + lane * 16;
    sb[1] = p.s1 + (size_t)tile.expert * p.scale_stride + (size_t)(strip * 4) * NS * 64 + (j * 2 + hsel) * 8;
  }
  const int8_t* xa = xq + j * KP + 4 * L;   // A operand: lane (L, i = j) supplies token i's 4 int8 of AVX lane L

  // Double buffer over (512-k step, matrix) pairs, ONE matrix per buffer: gate and up of a step take turns (the activation
  // fragments are re-read from LDS for the second).  Round 2 kept both matrices of the current AND the next step in
  // registers (128 VGPRs of operands beside 32 accumulators): the 4-token gate|up variant spilled 100 B per lane to scratch.
  // Every accumulator still receives its blocks in the same order: results unchanged.
  uint4 wcur[4], scur[4], wnxt[4], snxt[4];
  auto load_step = [&](auto mc, uint4(&w)[4], uint4(&sc)[4], int st) {
    constexpr int m = decltype(mc)::value;
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      w[rg] = *reinterpret_cast<const uint4*>(wb[m] + ((size_t)rg * NS + st) * 1024);
      sc[rg] = *reinterpret_cast<const uint4*>(sb[m] + ((size_t)rg * NS + st) * 64);
    }
  };
  auto compute_step = [&](auto mc, uint4(&w)[4], uint4(&sc)[4], int st) {
    constexpr int m = decltype(mc)::value;
#pragma unroll
    for (int kb8 = 0; kb8 < 8; kb8++) {
      const int kb = st * 8 + kb8;
      const int a_op = *reinterpret_cast<const int*>(xa + 64 * kb);
      const float4 as4 = *reinterpret_cast<const float4*>(as_l + (2 * kb + hsel) * 4);
      const float asv[4] = {as4.x, as4.y, as4.z, as4.w};
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const uint32_t P = kb8 < 2 ? w[rg].x : kb8 < 4 ? w[rg].y : kb8 < 6 ? w[rg].z : w[rg].w;
        const int b_op = (kb8 & 1) ? (int)(P & 0xF0F0F0F0u) : (int)((P << 4) & 0xF0F0F0F0u);
        const uint32_t S = kb8 < 2 ? sc[rg].x : kb8 < 4 ? sc[rg].y : kb8 < 6 ? sc[rg].z : sc[rg].w;
        const float bs = __uint_as_float((kb8 & 1) ? (S & 0xffff0000u) : (S << 16));
        const v4i d = __builtin_amdgcn_mfma_i32_4x4x4i8(a_op, b_op, v4i{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NT; r++) acc[m][rg][r] = fmaf(asv[r] * bs, (float)d[r], acc[m][rg][r]);
      }
    }
  };
  using M0 = std::integral_constant<int, 0>;
  using M1 = std::integral_constant<int, NMAT - 1>;
  load_step(M0{}, wcur, scur, 0);
  if constexpr (GATE_UP) {
    for (int st = 0; st < NS; st++) {
      load_step(M1{}, wnxt, snxt, st);
      compute_step(M0{}, wcur, scur, st);
      if (st + 1 < NS) load_step(M0{}, wcur, scur, st + 1);
      compute_step(M1{}, wnxt, snxt, st);
    }
  } else {
    for (int st = 0; st < NS; st += 2) {
      if (st + 1 < NS) load_step(M0{}, wnxt, snxt, st + 1);
      compute_step(M0{}, wcur, scur, st);
      if (st + 1 < NS) {
        if (st + 2 < NS) load_step(M0{}, wcur, scur, st + 2);
        compute_step(M0{}, wnxt, snxt, st + 1);
      }
    }
  }

  // _mm512_reduce_add_ps tree over the 16 AVX lanes (lane bits 5,4,3,2), then /16 (amx_kernels.hpp:3385-3450)
#pragma unroll
  for (int m = 0; m < NMAT; m++)
#pragma unroll
    for (int rg = 0; rg < 4; rg++)
#pragma unroll
      for (int r = 0; r < NT; r++) {
        float v = acc[m][rg][r];
        v = v + __shfl_xor(v, 32, 64);
        v = v + __shfl_xor(v, 16, 64);
        v = v + __shfl_xor(v, 8, 64);
        v = v + __shfl_xor(v, 4, 64);
        acc[m][rg][r] = v / 16.0f;
      }
  if (L == 0) {
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const int n = strip * 16 + rg * 4 + j;
#pragma unroll
      for (int r = 0; r < NT; r++) {
        if (r < tile.nrows) {
          const bf16_t g = f32_to_bf16(acc[0][rg][r]);
          bf16_t o = g;
          if constexpr (GATE_UP) {
            const bf16_t u = f32_to_bf16(acc[1][rg][r]);
            o = f32_to_bf16(act_fn(bf16_to_f32(g), bf16_to_f32(u)));
          }
          p.out[(size_t)(tile.row0 + r) * p.N + n] = o;
        }
      }
    }
  }
}

// =====================================================================================================
// RAWINT4 prompt chunks: the same products on the 16x16x32 int8 MFMA, one MFMA per (32-k group) x (16 rows) x (16 tokens).
//
// The exact kernel above keeps the reference's SIXTEEN fp32 lane accumulators per output, i.e. one convert + multiply + fma per
// FOUR int8 products (4x4x4 MFMA blocks, four tokens per tile, every weight byte re-read and re-unpacked per four tokens): 38 ms
// per Kimi-K2 layer at a 2048-token chunk against 1.4 ms for the AMXINT4 format (profiles/r04_final).  A prompt chunk does not
// need that association: the terms of an output are the per-group sums  (as[t][g] * bs[n][g]) * float(sum_{k in g} a[t][k] * 16 q[n][k])
// — the reference adds them as 16 interleaved chains + a tree, this kernel as ONE chain over the groups in k order (the
// integer group sum is exact either way, so only the association of <= K/32 fp32 additions differs: <= a few fp32 ulps before the
// one bf16 rounding; tests/test_moe_gpu.py::test_rawint4_prompt_chunks holds it to the bound of the FP8 / BF16 formats, whose
// prompt kernels re-associate in the same way).  Decode and short batches (qlen < 64) keep the exact kernels; dev knob 29 = 1
// forces them for every size.
//
// No second copy of the weights: the RAW tiles are read as they lie in HBM.  MFMA lane (i = lane & 15, kc = lane >> 4) needs
// row i's eight nibbles k = 64 kb + 32 h + 8 kc + [0, 8) of group (kb, h): those are the pieces (L, j) = (8h + 2kc + o, i & 3),
// o = 0 / 1, of RAW tile rg = i >> 2 — so the lane loads exactly its four pieces q = (h, o) of the strip's four tiles (16 bytes
// each, every 128-byte line of the 4 KiB used by one wave instruction pair) and nothing moves between lanes.
//   workgroup = 8 wavefronts = 8 strips of 16 weight rows x (gate, up) x 64 tokens; K walks in the RAW step of 512 (16 groups):
//   activations + their group scales of the step are double-buffered in LDS, the step's weights + scales in registers.
// =====================================================================================================
template <bool GATE_UP>
__global__ __launch_bounds__(512) void moe_rawint4_chunk_kernel(RawGemmParams p) {
  constexpr int NU = 2;                             // B operands per wavefront that share one activation fragment: (gate, up) of a
                                                    // strip, or two strips of down (one fragment + scale read per two MFMAs: LDS port)
  constexpr int TOK = 64, MT = 4;
  constexpr int CS = TOK * 16 + 16;                 // bytes between 16-byte activation columns (+16: staging writes spread over banks)
  constexpr int XB = 32 * CS;                       // one step of activations: 32 columns x 64 tokens x 16 B
  constexpr int AB = 16 * TOK * 4;                  // one step of activation scales: [16 groups][64 tokens] fp32
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // [2][XB] | [2][AB]
  __shared__ int s_src[TOK];
  uint8_t* xs = smem;
  float* as_l = reinterpret_cast<float*>(smem + 2 * XB);

  const int tile_idx = blockIdx.y;
  if (tile_idx >= p.counters[0]) return;
  const Tile tile = p.tiles[tile_idx];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nstrips = p.N / 16;
  const int NS = p.K / 512, G = p.K / 32;
  const int i = lane & 15, kc = lane >> 4, rg = i >> 2, j = i & 3;
  int ustrip[NU];
#pragma unroll
  for (int u = 0; u < NU; u++) ustrip[u] = GATE_UP ? blockIdx.x * 8 + wave : (blockIdx.x * 8 + wave) * 2 + u;
  const bool wave_ok = ustrip[0] < nstrips;

  if (tid < TOK) s_src[tid] = tid < tile.nrows ? (p.row_src ? p.row_src[tile.row0 + tid] : tile.row0 + tid) : -1;
  __syncthreads();

  // ---- this lane's four pieces of each unit's four RAW tiles, and its row's scales (a strip past N aliases strip 0, not stored)
  const uint8_t* wb[NU];
  const bf16_t* sb[NU];
#pragma unroll
  for (int u = 0; u < NU; u++) {
    const int st = ustrip[u] < nstrips ? ustrip[u] : 0;
    const uint8_t* w = (GATE_UP && u) ? p.w1 : p.w0;
    const bf16_t* sc = (GATE_UP && u) ? p.s1 : p.s0;
    wb[u] = w + (size_t)tile.expert * p.expert_stride + ((size_t)(st * 4 + rg) * NS) * 1024 + (size_t)((2 * kc) * 4 + j) * 16;
    sb[u] = sc + (size_t)tile.expert * p.scale_stride + ((size_t)(st * 4 + rg) * NS) * 64 + (size_t)(j * 2) * 8;
  }
  struct WStep { uint4 w[NU][4]; uint4 s[NU][2]; };   // w[u][h * 2 + o]: piece L = 8h + 2kc + o; s[u][h]: 8 bf16 scales (kb = 0..7)
  auto load_w = [&](WStep& d, int st) {
#pragma unroll
    for (int u = 0; u < NU; u++) {
#pragma unroll
      for (int q = 0; q < 4; q++)
        d.w[u][q] = *reinterpret_cast<const uint4*>(wb[u] + (size_t)st * 1024 + ((q >> 1) * 8 + (q & 1)) * 64);
#pragma unroll
      for (int h = 0; h < 2; h++) d.s[u][h] = *reinterpret_cast<const uint4*>(sb[u] + (size_t)st * 64 + h * 8);
    }
  };

  // ---- staging: thread -> 4 x (token, 16-byte piece) of the step's 64 x 512 activation bytes + (256 threads) one float4 of group
  // scales.  The requests go out BEFORE the step's MFMAs, the LDS writes after them (20 registers across the step buy the overlap:
  // with one workgroup per CU nobody else keeps the memory pipe busy while this one computes).
  struct XStage { uint4 x0, x1, x2, x3; float4 a; };
  auto load_x = [&](XStage& xg, int st) {
    uint4 xr[4];
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int idx = it * 512 + tid, tok = idx >> 5, piece = idx & 31;
      const int src = s_src[tok];      // unconditional loads from a clamped row (a branch per load serialises the requests)
      xr[it] = *reinterpret_cast<const uint4*>(p.act_q + (size_t)(src >= 0 ? src : 0) * p.K + (size_t)st * 512 + piece * 16);
    }
    const int tok4 = (tid & 255) >> 2, src4 = s_src[tok4];
    const float4 av = *reinterpret_cast<const float4*>(p.act_d + (size_t)(src4 >= 0 ? src4 : 0) * G + (size_t)st * 16 + (tid & 3) * 4);
    const float keep = src4 >= 0 ? 1.0f : 0.0f;              // rows past the tile: scale 0 (their int8 values are then irrelevant)
    xg.x0 = xr[0]; xg.x1 = xr[1]; xg.x2 = xr[2]; xg.x3 = xr[3];
    xg.a = make_float4(av.x * keep, av.y * keep, av.z * keep, av.w * keep);
  };
  auto store_x = [&](const XStage& xg, int buf) {
    const uint4 xr[4] = {xg.x0, xg.x1, xg.x2, xg.x3};
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int idx = it * 512 + tid, tok = idx >> 5, piece = idx & 31;
      *reinterpret_cast<uint4*>(xs + buf * XB + piece * CS + tok * 16) = xr[it];
    }
    if (tid < 256) {
      float* ab = as_l + buf * (AB / 4) + ((tid & 3) * 4) * TOK + (tid >> 2);
      ab[0] = xg.a.x; ab[TOK] = xg.a.y; ab[2 * TOK] = xg.a.z; ab[3 * TOK] = xg.a.w;
    }
  };

  float acc[NU][MT][4];
#pragma unroll
  for (int u = 0; u < NU; u++)
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[u][t][r] = 0.0f;

  // One step = 4 dword positions p4 of the lane's pieces x 2 nibble halves (kb8 = 2 p4 + par) x 2 group halves h = 16 groups.
  // Two things the compiler does to this loop when left alone, both measured (profiles/r04_j_*): (1) fully unrolled, it unpacks a
  // whole step's operands up front and spills 460 registers — every group's operand words therefore pass an empty volatile asm
  // right before their use, which pins the unpack behind the previous group; (2) it SLP-packs the four scale products and fmas of
  // a fragment into v_pk_mul_f32 / v_pk_fma_f32 plus ~6 register-pair moves per MFMA — slower than the scalar forms beside MFMAs
  // (MI355X_MICROARCH.md, per-instruction constants) — so the two operations are spelled as single instructions.
  auto mul1 = [](float a, float b) { float d; asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; };
  auto fma1 = [](float a, float b, float c) { float d; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; };
  // MTL = 16-token groups of the tile that hold rows (a Kimi-K2 chunk of 2048 tokens leaves ~43 rows per expert: three groups,
  // not four, for most tiles); one instantiation per count, chosen once per workgroup
  auto compute_n = [&](auto mtl, const WStep& w, int buf) __attribute__((always_inline)) {
    constexpr int MTL = decltype(mtl)::value;
    if (!wave_ok) return;
    const uint8_t* xb = xs + buf * XB + (kc >> 1) * CS + i * 16 + (kc & 1) * 8;
    const float* ab = as_l + buf * (AB / 4) + kc * 4;
    // PMC of the first version (profiles/r04_j_rawint4_chunk_pmc.txt): 48 % of the wave cycles parked on s_waitcnt (every group
    // read its activation fragments and scales from LDS and waited for them), 24 % in issue stalls (a fragment converted right
    // behind its MFMA).  So: the fragments of group g + 1 are requested before the MFMAs of group g, and a group's eight MFMAs are
    // issued back to back before the first result is touched.
    struct Frag { long a[MTL]; float4 s[MTL]; };
    auto load_frag = [&](Frag& f, int g) __attribute__((always_inline)) {      // g = kb8 * 2 + h: column 2 g, scale row g
#pragma unroll
      for (int t = 0; t < MTL; t++) {
        f.a[t] = *reinterpret_cast<const long*>(xb + (size_t)(2 * g) * CS + t * 256);
        f.s[t] = *reinterpret_cast<const float4*>(ab + g * TOK + t * 16);
      }
    };
    Frag fr[2];
    load_frag(fr[0], 0);
#pragma unroll
    for (int g = 0; g < 16; g++) {
      const int kb8 = g >> 1, h = g & 1;
      const Frag& cur = fr[g & 1];
      long bop[NU];
      float bs[NU];
#pragma unroll
      for (int u = 0; u < NU; u++) {
        const uint4 &w0 = w.w[u][h * 2], &w1 = w.w[u][h * 2 + 1], &sv = w.s[u][h];
        uint32_t P0 = kb8 < 2 ? w0.x : kb8 < 4 ? w0.y : kb8 < 6 ? w0.z : w0.w;
        uint32_t P1 = kb8 < 2 ? w1.x : kb8 < 4 ? w1.y : kb8 < 6 ? w1.z : w1.w;
        const uint32_t S = kb8 < 2 ? sv.x : kb8 < 4 ? sv.y : kb8 < 6 ? sv.z : sv.w;
        asm volatile("" : "+v"(P0), "+v"(P1));
        const uint32_t b0 = (kb8 & 1) ? (P0 & 0xF0F0F0F0u) : ((P0 << 4) & 0xF0F0F0F0u);
        const uint32_t b1 = (kb8 & 1) ? (P1 & 0xF0F0F0F0u) : ((P1 << 4) & 0xF0F0F0F0u);
        bop[u] = (long)(((unsigned long long)b1 << 32) | b0);
        bs[u] = __uint_as_float((kb8 & 1) ? (S & 0xffff0000u) : (S << 16));
      }
      if (g + 1 < 16) load_frag(fr[(g + 1) & 1], g + 1);
#pragma unroll
      for (int t0 = 0; t0 < MTL; t0 += 2) {          // four MFMAs in flight (two token groups x two operands), then their 16 results
        v4i d[2][NU];
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
#pragma unroll
          for (int u = 0; u < NU; u++)
            if (t0 + tt < MTL) d[tt][u] = __builtin_amdgcn_mfma_i32_16x16x32_i8(cur.a[t0 + tt], bop[u], v4i{0, 0, 0, 0}, 0, 0, 0);
        // products, conversions, then the fmas: a dependent pair of VALU instructions back to back stalls the issue (the first
        // version's `mul -> fma` / `cvt -> fma` chains showed as 30 % SQ_WAIT_INST_ANY)
#pragma unroll
        for (int tt = 0; tt < 2; tt++) {
          if (t0 + tt >= MTL) continue;
          const int t = t0 + tt;
          const float asv[4] = {cur.s[t].x, cur.s[t].y, cur.s[t].z, cur.s[t].w};
          float pm[NU][4], cf[NU][4];
#pragma unroll
          for (int u = 0; u < NU; u++)
#pragma unroll
            for (int r = 0; r < 4; r++) pm[u][r] = mul1(asv[r], bs[u]);
#pragma unroll
          for (int u = 0; u < NU; u++)
#pragma unroll
            for (int r = 0; r < 4; r++) cf[u][r] = (float)d[tt][u][r];
#pragma unroll
          for (int u = 0; u < NU; u++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[u][t][r] = fma1(pm[u][r], cf[u][r], acc[u][t][r]);
        }
      }
    }
  };
  const int mt_live = (tile.nrows + 15) >> 4;
  auto compute = [&](const WStep& w, int buf) __attribute__((always_inline)) {
    if (mt_live >= 4) compute_n(std::integral_constant<int, 4>{}, w, buf);
    else if (mt_live == 3) compute_n(std::integral_constant<int, 3>{}, w, buf);
    else if (mt_live == 2) compute_n(std::integral_constant<int, 2>{}, w, buf);
    else compute_n(std::integral_constant<int, 1>{}, w, buf);
  };

  WStep wa, wb2;
  XStage xg;
  load_w(wa, 0);
  load_x(xg, 0);
  store_x(xg, 0);
  __syncthreads();
  for (int st = 0; st < NS; st += 2) {
    if (st + 1 < NS) { load_w(wb2, st + 1); load_x(xg, st + 1); }
    compute(wa, 0);
    if (st + 1 < NS) store_x(xg, 1);     // buffer 1 was last read in step st - 1: every wavefront is past that step's barrier
    __syncthreads();
    if (st + 1 < NS) {
      if (st + 2 < NS) { load_w(wa, st + 2); load_x(xg, st + 2); }
      compute(wb2, 1);
      if (st + 2 < NS) store_x(xg, 0);
      __syncthreads();
    }
  }

  if (!wave_ok) return;
  if constexpr (GATE_UP) {
    const int n = ustrip[0] * 16 + i;
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = t * 16 + kc * 4 + r;
        if (row < tile.nrows) {
          const bf16_t gv = f32_to_bf16(acc[0][t][r] / 16.0f), uv = f32_to_bf16(acc[1][t][r] / 16.0f);
          p.out[(size_t)(tile.row0 + row) * p.N + n] = f32_to_bf16(act_fn(bf16_to_f32(gv), bf16_to_f32(uv)));
        }
      }
  } else {
#pragma unroll
    for (int u = 0; u < NU; u++) {
      if (ustrip[u] >= nstrips) continue;
      const int n = ustrip[u] * 16 + i;
#pragma unroll
      for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = t * 16 + kc * 4 + r;
          if (row < tile.nrows) p.out[(size_t)(tile.row0 + row) * p.N + n] = f32_to_bf16(acc[u][t][r] / 16.0f);
        }
    }
  }
}

// per-(row, 32-group) int8 quantisation (BufferASmallKGroupImpl::from_mat, amx_buffers.hpp:431-495): one block per row,
// 8 elements per thread, a group = 4 adjacent lanes
__device__ __forceinline__ void quant_row_kgroup_block(const bf16_t* __restrict__ src, int K, int8_t* __restrict__ dst,
                                                       float* __restrict__ d_out) {
  for (int c = threadIdx.x * 8; c < K; c += blockDim.x * 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(src + c);
    float amax = amax8(v, 0.0f);
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    const float d = amax / 127.0f;
    const float id = d ? 1.0f / d : 0.0f;
    *reinterpret_cast<uint2*>(dst + c) = quant8(v, id);
    if ((threadIdx.x & 3) == 0) d_out[c >> 5] = d;
  }
}

__global__ __launch_bounds__(256) void moe_actquant_kgroup_kernel(const bf16_t* __restrict__ a, int K,
                                                                  int8_t* __restrict__ a_q, float* __restrict__ a_d,
                                                                  const int32_t* counters, const int32_t* d_bsz, int qlen,
                                                                  int rows_are_tokens) {
  const int row = blockIdx.x;
  if (rows_are_tokens) {
    int T = qlen;
    if (d_bsz) T = min(max(*d_bsz, 0), qlen);
    if (row >= T) return;
  } else if (row >= counters[1]) {
    return;
  }
  quant_row_kgroup_block(a + (size_t)row * K, K, a_q + (size_t)row * K, a_d + (size_t)row * (K / 32));
}

// =====================================================================================================
// Decode fast path of the RAWINT4 experts (qlen*k <= KTX_DEC_MAX_PAIRS): the two-launch structure of the other formats with
// the arithmetic of moe_rawint4_gemm_kernel above for ONE token: lane (L, j) needs only token 0's entry of each 4x4x4 block
// product, which is a plain 4-element int8 dot (v_dot4c_i32_i8) — same integers, a quarter of the registers.  Before this path a decode token went bucket -> x-quant -> gate/up -> requant -> down -> combine: six
// launches with a one-step prefetch; here the per-32-group activation quantisation (BufferASmallKGroupImpl::from_mat,
// amx_buffers.hpp:431-495) happens in the consuming workgroup and the RAW tiles stream through a D-step register ring.
// =====================================================================================================
struct RawDecParams {
  const int32_t* d_bsz;
  int qlen, k, E, expert_begin, H, I;
  const int64_t* ids;
  const uint8_t* mask;
  const bf16_t* x;
  const float* weights;
  const uint8_t *gate_w, *up_w, *down_w;
  const bf16_t *gate_s, *up_s, *down_s;
  size_t gu_stride, dn_stride;     // bytes of RAW tiles per expert matrix
  size_t gu_sstride, dn_sstride;   // bf16 scales per expert matrix
  bf16_t* a_buf;                   // [qlen*k][I]
  void* y;
  int incremental, partial_f32;
};

struct RawSlot {   // one 512-K step of a 16-row strip: 4 row groups of nibbles + their bf16 group scales
  uint4 w[4], s[4];
};
__device__ __forceinline__ RawSlot raw_load_slot(const uint8_t* wb, const bf16_t* sb, int NS, int st) {
  typedef unsigned int u4v __attribute__((ext_vector_type(4)));
  RawSlot r;
#pragma unroll
  for (int rg = 0; rg < 4; rg++) {
    const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(wb + ((size_t)rg * NS + st) * 1024));
    r.w[rg] = make_uint4(v.x, v.y, v.z, v.w);
    r.s[rg] = *reinterpret_cast<const uint4*>(sb + ((size_t)rg * NS + st) * 64);
  }
  return r;
}
// acc[rg] += the step's 8 blocks of 64 k for the lane's (AVX lane L, weight row j); xa = LDS int8 row + 4L, as1 = LDS group
// scales + hsel.  Same expression order as compute_step of moe_rawint4_gemm_kernel (token 0 of its 4).
__device__ __forceinline__ void raw_dec_step(const RawSlot& sl, const int8_t* xa, const float* as1, int st, float (&acc)[4]) {
#pragma unroll
  for (int kb8 = 0; kb8 < 8; kb8++) {
    const int kb = st * 8 + kb8;
    const int a_op = *reinterpret_cast<const int*>(xa + 64 * kb);
    const float as = as1[2 * kb];
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const uint32_t P = kb8 < 2 ? sl.w[rg].x : kb8 < 4 ? sl.w[rg].y : kb8 < 6 ? sl.w[rg].z : sl.w[rg].w;
      const int b_op = (kb8 & 1) ? (int)(P & 0xF0F0F0F0u) : (int)((P << 4) & 0xF0F0F0F0u);
      const uint32_t S = kb8 < 2 ? sl.s[rg].x : kb8 < 4 ? sl.s[rg].y : kb8 < 6 ? sl.s[rg].z : sl.s[rg].w;
      const float bs = __uint_as_float((kb8 & 1) ? (S & 0xffff0000u) : (S << 16));
      const int d0 = __builtin_amdgcn_sdot4(a_op, b_op, 0, false);   // token 0's entry of the 4x4x4 block product, as one v_dot4
      acc[rg] = fmaf(as * bs, (float)d0, acc[rg]);
    }
  }
}
// _mm512_reduce_add_ps' tree over the 16 AVX lanes (lane bits 5,4,3,2), then /16 (amx_kernels.hpp:3385-3450)
__device__ __forceinline__ float raw_reduce16(float v) {
  v = v + __shfl_xor(v, 32, 64);
  v = v + __shfl_xor(v, 16, 64);
  v = v + __shfl_xor(v, 8, 64);
  v = v + __shfl_xor(v, 4, 64);
  return v / 16.0f;
}
// 8 bf16 of a row held by this lane (4 adjacent lanes = one 32-group) -> int8 + the group's scale, as quant_row_kgroup_block
__device__ __forceinline__ uint2 raw_quant_piece(const uint4& v, float& d) {
  float amax = amax8(v, 0.0f);
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
  d = amax / 127.0f;
  const float id = d ? 1.0f / d : 0.0f;
  return quant8(v, id);
}

// one workgroup per ((t,j) pair, NW strips of I): x[t] quantised per 32-group into LDS, gate and up strips of expert
// ids[t][j] through the ring (NS % D == 0: branch-free), SiLU*up epilogue -> a_buf[pair]
template <int D, int NW>
__global__ __launch_bounds__(NW * 64) void moe_dec_raw_gateup_kernel(RawDecParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // [H] int8 | [H/32] fp32
  int8_t* xq = reinterpret_cast<int8_t*>(smem);
  float* xd = reinterpret_cast<float*>(smem + p.H);
  int T = p.qlen;
  if (p.d_bsz) T = min(max(*p.d_bsz, 0), p.qlen);
  const int pair = blockIdx.y, t = pair / p.k;
  if (t >= T) return;
  const long long idl = p.ids[pair] - p.expert_begin;
  if (idl < 0 || idl >= p.E || (p.mask && p.mask[idl])) return;
  const int e = (int)idl;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = blockIdx.x * NW + wave;
  const bool strip_ok = strip * 16 < p.I;
  const int strip_c = strip_ok ? strip : 0;
  const int NS = p.H / 512;
  const int L = lane >> 2, j = lane & 3, hsel = L >> 3;
  // activations first (vmcnt retires in order), then the ring
  const bf16_t* xr = p.x + (size_t)t * p.H;
  uint4 xv[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = tid * 8 + i * NW * 512;
    xv[i] = make_uint4(0, 0, 0, 0);
    if (c < p.H) xv[i] = *reinterpret_cast<const uint4*>(xr + c);
  }
  const uint8_t* wg = p.gate_w + (size_t)e * p.gu_stride + (size_t)(strip_c * 4) * NS * 1024 + lane * 16;
  const uint8_t* wu = p.up_w + (size_t)e * p.gu_stride + (size_t)(strip_c * 4) * NS * 1024 + lane * 16;
  const bf16_t* sg = p.gate_s + (size_t)e * p.gu_sstride + (size_t)(strip_c * 4) * NS * 64 + (j * 2 + hsel) * 8;
  const bf16_t* su = p.up_s + (size_t)e * p.gu_sstride + (size_t)(strip_c * 4) * NS * 64 + (j * 2 + hsel) * 8;
  RawSlot ring[D][2];
#pragma unroll
  for (int d = 0; d < D; d++) {
    ring[d][0] = raw_load_slot(wg, sg, NS, d);
    ring[d][1] = raw_load_slot(wu, su, NS, d);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = tid * 8 + i * NW * 512;
    float d;
    const uint2 q = raw_quant_piece(xv[i], d);   // every lane takes part in the shuffles; rows past H hold zeros
    if (c < p.H) {
      *reinterpret_cast<uint2*>(xq + c) = q;
      if ((tid & 3) == 0) xd[c >> 5] = d;
    }
  }
  __syncthreads();
  if (!strip_ok) return;
  float accg[4] = {0.f, 0.f, 0.f, 0.f}, accu[4] = {0.f, 0.f, 0.f, 0.f};
  const int8_t* xa = xq + 4 * L;
  const float* as1 = xd + hsel;
  const int G = NS / D;
  for (int g = 0; g < G - 1; g++) {
#pragma unroll
    for (int d = 0; d < D; d++) {
      const int st = g * D + d;
      raw_dec_step(ring[d][0], xa, as1, st, accg);
      raw_dec_step(ring[d][1], xa, as1, st, accu);
      ring[d][0] = raw_load_slot(wg, sg, NS, st + D);
      ring[d][1] = raw_load_slot(wu, su, NS, st + D);
    }
  }
#pragma unroll
  for (int d = 0; d < D; d++) {
    const int st = (G - 1) * D + d;
    raw_dec_step(ring[d][0], xa, as1, st, accg);
    raw_dec_step(ring[d][1], xa, as1, st, accu);
  }
#pragma unroll
  for (int rg = 0; rg < 4; rg++) {
    const float gv = raw_reduce16(accg[rg]), uv = raw_reduce16(accu[rg]);
    if (L == 0) {
      const bf16_t gq = f32_to_bf16(gv), uq = f32_to_bf16(uv);
      p.a_buf[(size_t)pair * p.I + strip * 16 + rg * 4 + j] = f32_to_bf16(act_fn(bf16_to_f32(gq), bf16_to_f32(uq)));
    }
  }
}

// one workgroup per (token, 16-row strip of H), wave j = slot j: the activated row of pair (t,j) re-quantised per 32-group
// into the wave's own LDS row, the down strip of its expert streamed, then the slot-ordered weighted combine (a12) and the
// merge step (a4) exactly as moe_combine_kernel does them
template <int D>
__global__ __launch_bounds__(512) void moe_dec_raw_down_kernel(RawDecParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // [k][I] int8 | [k][I/32] fp32 | [k][16] fp32 down outputs | [k] valid flags | [k] routing weights
  int8_t* aq_all = reinterpret_cast<int8_t*>(smem);
  float* ad_all = reinterpret_cast<float*>(smem + (size_t)p.k * p.I);
  float* s_dn = ad_all + (size_t)p.k * (p.I / 32);
  int* s_valid = reinterpret_cast<int*>(s_dn + p.k * 16);
  float* s_wt = reinterpret_cast<float*>(s_valid + p.k);
  int T = p.qlen;
  if (p.d_bsz) T = min(max(*p.d_bsz, 0), p.qlen);
  const int t = blockIdx.y;
  if (t >= T) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int slot = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = t * p.k + slot, strip = blockIdx.x;
  const long long idl = p.ids[pair] - p.expert_begin;
  const bool valid = !(idl < 0 || idl >= p.E || (p.mask && p.mask[idl]));
  const int e = valid ? (int)idl : 0;
  const int NS = p.I / 512;
  const int L = lane >> 2, j = lane & 3, hsel = L >> 3;
  if (valid) {
    const bf16_t* ar = p.a_buf + (size_t)pair * p.I;
    int8_t* aq = aq_all + (size_t)slot * p.I;
    float* ad = ad_all + (size_t)slot * (p.I / 32);
    uint4 av[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int c = lane * 8 + i * 512;
      av[i] = make_uint4(0, 0, 0, 0);
      if (c < p.I) av[i] = *reinterpret_cast<const uint4*>(ar + c);
    }
    const uint8_t* wd = p.down_w + (size_t)e * p.dn_stride + (size_t)(strip * 4) * NS * 1024 + lane * 16;
    const bf16_t* sd = p.down_s + (size_t)e * p.dn_sstride + (size_t)(strip * 4) * NS * 64 + (j * 2 + hsel) * 8;
    RawSlot ring[D];
#pragma unroll
    for (int d = 0; d < D; d++) ring[d] = raw_load_slot(wd, sd, NS, d);
    const float wt = p.weights[pair];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int c = lane * 8 + i * 512;
      float d;
      const uint2 q = raw_quant_piece(av[i], d);
      if (c < p.I) {
        *reinterpret_cast<uint2*>(aq + c) = q;
        if ((lane & 3) == 0) ad[c >> 5] = d;
      }
    }
    if (lane == 0) { s_valid[slot] = 1; s_wt[slot] = wt; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the wave reads back its own LDS row: order only
    __builtin_amdgcn_wave_barrier();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int8_t* xa = aq + 4 * L;
    const float* as1 = ad + hsel;
    const int G = NS / D;
    for (int g = 0; g < G - 1; g++) {
#pragma unroll
      for (int d = 0; d < D; d++) {
        const int st = g * D + d;
        raw_dec_step(ring[d], xa, as1, st, acc);
        ring[d] = raw_load_slot(wd, sd, NS, st + D);
      }
    }
#pragma unroll
    for (int d = 0; d < D; d++) raw_dec_step(ring[d], xa, as1, (G - 1) * D + d, acc);
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const float v = raw_reduce16(acc[rg]);
      if (L == 0) s_dn[slot * 16 + rg * 4 + j] = bf16_to_f32(f32_to_bf16(v));
    }
  } else if (lane == 0) {
    s_valid[slot] = 0;
    s_wt[slot] = 0.0f;
  }
  __syncthreads();
  if (tid < 16) {  // a12: weighted combine in slot order, then a4
    float acc = 0.0f;
    for (int jj = 0; jj < p.k; jj++)
      if (s_valid[jj]) acc = fmaf(s_dn[jj * 16 + tid], s_wt[jj], acc);
    const size_t o = (size_t)t * p.H + strip * 16 + tid;
    if (p.partial_f32) {
      reinterpret_cast<float*>(p.y)[o] = acc;
    } else {
      bf16_t* yp = reinterpret_cast<bf16_t*>(p.y) + o;
      if (p.incremental) acc = acc + bf16_to_f32(*yp);
      *yp = f32_to_bf16(acc);
    }
  }
}

// raw row-major nibbles [N][K/2] (byte = ((q1+8)<<4)|(q0+8), even k low) -> RAW tiles; one thread per packed dword
__global__ void pack_rawint4_kernel(const uint8_t* __restrict__ src, int N, int K, uint32_t* __restrict__ out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)N * K / 8;
  if (idx >= total) return;
  const int NS = K / 512;
  const size_t tile = idx / 256;
  const int within = (int)(idx % 256), lane = within / 4, pp = within % 4;
  const int rg = (int)(tile / NS), st = (int)(tile % NS);
  const int L = lane >> 2, jrow = lane & 3;
  const uint8_t* row = src + (size_t)(rg * 4 + jrow) * (K / 2);
  uint32_t v = 0;
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const int k_lo = 64 * (8 * st + 2 * pp) + 4 * L + b, k_hi = k_lo + 64;
    const uint32_t n_lo = ((k_lo & 1) ? (row[k_lo >> 1] >> 4) : (row[k_lo >> 1] & 15)) ^ 8u;
    const uint32_t n_hi = ((k_hi & 1) ? (row[k_hi >> 1] >> 4) : (row[k_hi >> 1] & 15)) ^ 8u;
    v |= ((n_hi << 4) | n_lo) << (8 * b);
  }
  out[idx] = v;
}

// bf16 scales [N][K/32] -> tile order [rg][step][row j][h][8 blocks]; one thread per element
__global__ void pack_rawint4_scales_kernel(const bf16_t* __restrict__ src, int N, int K, bf16_t* __restrict__ out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int G = K / 32, NS = K / 512;
  if (idx >= (size_t)N * G) return;
  const int kb8 = (int)(idx % 8), h = (int)((idx / 8) % 2), jrow = (int)((idx / 16) % 4);
  const size_t t = idx / 64;
  const int st = (int)(t % NS), rg = (int)(t / NS);
  out[idx] = src[(size_t)(rg * 4 + jrow) * G + 2 * (8 * st + kb8) + h];
}

#include "ktx_moe_gguf.inc"
#include "ktx_moe_legacy.inc"

// =====================================================================================================
// host side
// =====================================================================================================
// Scratch for one forward.  Like the reference's shared_mem_buffer arena (cpu_backend/shared_mem_buffer.h:37-55) it is
// shared by the MoE layers of a device: forwards of different layers are ordered on ONE stream (the caller's decode /
// prefill stream — handles that run concurrently on different streams of one device must not share an arena and are not
// supported), so one arena sized for the largest request serves them all.  An arena is NEVER grown in place or freed
// while a handle refers to it: captured HIP graphs bake its addresses in.  A handle that needs more than the device's
// current arena holds gets a NEW arena (which becomes the device's current one); the old one lives on for the handles
// (and graphs) that already use it and is released with the last of them.
struct Workspace {
  int8_t *x_q = nullptr, *a_q = nullptr;
  float *x_d = nullptr, *a_d = nullptr;
  bf16_t *a_buf = nullptr, *dn_buf = nullptr;
  int32_t *row_of_pair = nullptr, *src_of_row = nullptr, *counters = nullptr;
  Tile* tiles = nullptr;
  int16_t *x_bs = nullptr, *a_bs = nullptr;   // Q8_K 16-sums (GGUF path)
  size_t cap[12] = {0};
  int device = 0;
  ~Workspace() {
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    void* ptrs[] = {x_q, a_q, x_d, a_d, a_buf, dn_buf, row_of_pair, src_of_row, counters, tiles, x_bs, a_bs};
    for (void* q : ptrs)
      if (q) (void)hipFree(q);
    if (have) (void)hipSetDevice(cur);
  }
};
static std::mutex g_ws_mu;
static std::shared_ptr<Workspace> g_ws[64];

// current device saved on entry, restored on exit (callers with several GPUs in one process keep their own current device)
struct DeviceGuard {
  int prev = -1;
  hipError_t err;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) err = hipSetDevice(dev);
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};
#define KTX_ON_DEVICE(dev) DeviceGuard _dg(dev); KTX_HIP(_dg.err)

struct ktx_moe_s {
  ktx_moe_config cfg;
  int wbits;
  size_t gu_stride, dn_stride;  // bytes per expert matrix
  uint8_t *gate_w = nullptr, *up_w = nullptr, *down_w = nullptr;
  float *gate_s = nullptr, *up_s = nullptr, *down_s = nullptr;
  float *gate_r = nullptr, *up_r = nullptr, *down_r = nullptr;   // FP8_PERCHANNEL: fp32 scale per output row [E][N]
  uint8_t* mask = nullptr;
  std::shared_ptr<Workspace> ws_own;   // keeps the arena alive as long as this handle (and graphs captured through it)
  Workspace* ws = nullptr;
  int max_pairs = 0, max_tiles = 0;
  int gg_type[3] = {0, 0, 0};     // GGUF: ggml type of gate / up / down
  size_t gg_stride[3] = {0, 0, 0};
  bool loaded_gguf = false;
  bool exact = false;             // ktx_moe_set_exact: no path of this handle may re-associate the reference's fp32 sums
};

static int pick_mt(int qlen, int k, int E) {
  // tokens per expert on average; the MFMA N dimension is 16 tokens per M-tile
  const double per = (double)qlen * k / std::max(1, E);
  if (per <= 16.0) return 1;
  if (per <= 48.0) return 2;
  return 4;   // 128-token tiles (MT = 8) were measured slower at 192 tokens/expert: 2 waves/SIMD less, half-empty last tile
}

extern "C" int ktx_moe_create(const ktx_moe_config* cfg, ktx_moe_t* out) {
  KTX_REQUIRE(cfg && out, "ktx_moe_create: null argument");
  KTX_REQUIRE(cfg->format >= KTX_FMT_AMXINT4 && cfg->format <= KTX_FMT_FP8_PERCHANNEL, "ktx_moe_create: unknown format");
  KTX_REQUIRE(cfg->format != KTX_FMT_GGUF || (cfg->hidden_size % 256 == 0 && cfg->intermediate_size % 256 == 0),
              "ktx_moe_create: GGUF k-quants need hidden_size and intermediate_size to be multiples of 256 (QK_K)");
  KTX_REQUIRE(cfg->format != KTX_FMT_RAWINT4 || cfg->group_size == 32,
              "ktx_moe_create: RAWINT4 supports group_size 32 (Kimi-K2 native int4) only");
  KTX_REQUIRE(cfg->format != KTX_FMT_RAWINT4 || (cfg->hidden_size % 512 == 0 && cfg->intermediate_size % 512 == 0),
              "ktx_moe_create: RAWINT4 needs hidden_size and intermediate_size to be multiples of 512");
  KTX_REQUIRE(cfg->format != KTX_FMT_FP8 || cfg->group_size == 0 || cfg->group_size == 128,
              "ktx_moe_create: FP8 supports 128x128 block scales only");
  KTX_REQUIRE(cfg->expert_num > 0 && cfg->expert_num <= KTX_EMAX, "ktx_moe_create: expert_num out of range (1..1024)");
  KTX_REQUIRE(cfg->num_experts_per_tok > 0, "ktx_moe_create: num_experts_per_tok must be positive");
  KTX_REQUIRE(cfg->hidden_size % 128 == 0 && cfg->intermediate_size % 128 == 0,
              "ktx_moe_create: hidden_size and intermediate_size must be multiples of 128 (reference: K % 128 == 0)");
  KTX_REQUIRE(cfg->max_len > 0, "ktx_moe_create: max_len must be positive");
  KTX_REQUIRE(cfg->intermediate_size <= 8192 || cfg->format > KTX_FMT_AMXINT8, "ktx_moe_create: intermediate_size > 8192 is not supported for the int formats");
  KTX_ON_DEVICE(cfg->device);
  ktx_moe_s* h = new ktx_moe_s();
  h->cfg = *cfg;
  if (h->cfg.global_expert_num <= 0) h->cfg.global_expert_num = cfg->expert_num;
  h->wbits = (cfg->format == KTX_FMT_AMXINT4 || cfg->format == KTX_FMT_RAWINT4) ? 4 : (cfg->format == KTX_FMT_BF16 ? 16 : 8);
  const size_t E = cfg->expert_num, H = cfg->hidden_size, I = cfg->intermediate_size;
  h->gu_stride = I * H * h->wbits / 8;
  h->dn_stride = H * I * h->wbits / 8;
  h->max_pairs = cfg->max_len * cfg->num_experts_per_tok;
  h->max_tiles = std::min<int>(h->max_pairs, (int)E) + h->max_pairs / (cfg->format == KTX_FMT_RAWINT4 ? 4 : 16) + 1;
  const bool gguf = cfg->format == KTX_FMT_GGUF;   // tiles are allocated by ktx_moe_load_gguf (size depends on the types)
  if (!gguf) {
  KTX_HIP(hipMalloc(&h->gate_w, E * h->gu_stride));
  KTX_HIP(hipMalloc(&h->up_w, E * h->gu_stride));
  KTX_HIP(hipMalloc(&h->down_w, E * h->dn_stride));
  }
  // scales: fp32 per row (int formats) | fp32 per 128x128 block (FP8) | bf16 per (row, 32-group) (RAWINT4)
  size_t gu_sbytes = E * I * sizeof(float), dn_sbytes = E * H * sizeof(float);
  if (cfg->format == KTX_FMT_RAWINT4) gu_sbytes = dn_sbytes = E * I * (H / 32) * sizeof(bf16_t);
  if (!gguf) {
  KTX_HIP(hipMalloc(&h->gate_s, gu_sbytes));
  KTX_HIP(hipMalloc(&h->up_s, gu_sbytes));
  KTX_HIP(hipMalloc(&h->down_s, dn_sbytes));
  }
  if (cfg->format == KTX_FMT_FP8_PERCHANNEL) {   // the block-scale arrays above are filled with 1, these carry the row scales
    KTX_HIP(hipMalloc(&h->gate_r, E * I * sizeof(float)));
    KTX_HIP(hipMalloc(&h->up_r, E * I * sizeof(float)));
    KTX_HIP(hipMalloc(&h->down_r, E * H * sizeof(float)));
  }
  {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    KTX_REQUIRE(cfg->device >= 0 && cfg->device < 64, "ktx_moe_create: device ordinal out of range");
    const bool raw = cfg->format == KTX_FMT_RAWINT4;
    size_t need[12] = {0};
    need[0] = (size_t)cfg->max_len * H;
    need[1] = (size_t)cfg->max_len * sizeof(float) * (raw || gguf ? H / 32 : 1);   // (GGUF: Q8_K has a scale per 256, Q8_0 per 32)
    // int formats: the grouped gate/up GEMM stores g | u (2*I bf16 per row); GGUF: fp32 intermediates
    need[2] = (size_t)h->max_pairs * I * (gguf ? sizeof(float) : 2 * sizeof(bf16_t));
    need[3] = (size_t)h->max_pairs * I;
    need[4] = (size_t)h->max_pairs * sizeof(float) * (raw || gguf ? I / 32 : 1);
    need[5] = (size_t)h->max_pairs * H * (gguf ? sizeof(float) : sizeof(bf16_t));
    need[6] = need[7] = (size_t)h->max_pairs * sizeof(int32_t);
    need[8] = (size_t)h->max_tiles * sizeof(Tile);
    need[9] = (4 + KTX_DEC_MAX_PAIRS * 64) * sizeof(int32_t);   // [4] bucket counters | arrival tickets of the GGUF decode launches
    if (gguf) {
      // Q8_K 16-sums, and behind them the 32-sums split into two int8 planes (16 B per 256-block: the folded Q4_K kernel's min operand)
      need[10] = (size_t)cfg->max_len * (H / 16) * sizeof(int16_t) + (size_t)cfg->max_len * (H / 256) * (16 + 4);   // (+ the block sums)
      need[11] = (size_t)h->max_pairs * (I / 16) * sizeof(int16_t) + (size_t)h->max_pairs * (I / 256) * (16 + 4);
    }
    std::shared_ptr<Workspace> cur = g_ws[cfg->device];
    bool fits = cur != nullptr;
    for (int i = 0; fits && i < 12; i++) fits = need[i] <= cur->cap[i];
    if (!fits) {
      auto w = std::make_shared<Workspace>();
      w->device = cfg->device;
      for (int i = 0; i < 12; i++) w->cap[i] = std::max(need[i], cur ? cur->cap[i] : (size_t)0);
      hipError_t e = hipSuccess;
      auto alloc = [&](auto*& ptr, int i) {
        if (e == hipSuccess && w->cap[i]) e = hipMalloc(reinterpret_cast<void**>(&ptr), w->cap[i]);
      };
      alloc(w->x_q, 0); alloc(w->x_d, 1); alloc(w->a_buf, 2); alloc(w->a_q, 3); alloc(w->a_d, 4); alloc(w->dn_buf, 5);
      alloc(w->row_of_pair, 6); alloc(w->src_of_row, 7); alloc(w->tiles, 8); alloc(w->counters, 9);
      alloc(w->x_bs, 10); alloc(w->a_bs, 11);
      if (e == hipSuccess) e = hipMemset(w->counters, 0, w->cap[9]);
      if (e != hipSuccess) {
        ktx_moe_destroy(h);
        return ktx_fail(std::string("ktx_moe_create: workspace: ") + hipGetErrorString(e));
      }
      g_ws[cfg->device] = cur = w;
    }
    h->ws_own = cur;
    h->ws = cur.get();
  }
  *out = h;
  return 0;
}

extern "C" int ktx_moe_destroy(ktx_moe_t h) {
  if (!h) return 0;
  DeviceGuard _dg(h->cfg.device);
  void* ptrs[] = {h->gate_w, h->up_w, h->down_w, h->gate_s, h->up_s, h->down_s, h->gate_r, h->up_r, h->down_r, h->mask};
  for (void* p : ptrs)
    if (p) hipFree(p);
  delete h;
  return 0;
}

extern "C" int ktx_moe_set_exact(ktx_moe_t h, int exact) {
  KTX_REQUIRE(h, "ktx_moe_set_exact: null handle");
  h->exact = exact != 0;
  return 0;
}

extern "C" size_t ktx_moe_weight_bytes(ktx_moe_t h) {
  if (!h) return 0;
  const size_t E = h->cfg.expert_num, H = h->cfg.hidden_size, I = h->cfg.intermediate_size;
  if (h->cfg.format == KTX_FMT_GGUF) return E * (h->gg_stride[0] + h->gg_stride[1] + h->gg_stride[2]);
  return E * (2 * h->gu_stride + h->dn_stride) + E * (2 * I + H) * sizeof(float);
}

// GGUF k-quant experts: raw ggml blocks [E][N][K/256] per matrix (DEVICE pointers) -> W tiles (ktx_moe_gguf.inc)
extern "C" int ktx_moe_load_gguf(ktx_moe_t h, const void* d_gate, const void* d_up, const void* d_down, int gate_type,
                                 int up_type, int down_type) {
  KTX_REQUIRE(h && d_gate && d_up && d_down, "ktx_moe_load_gguf: null argument");
  KTX_REQUIRE(h->cfg.format == KTX_FMT_GGUF, "ktx_moe_load_gguf: handle was not created with KTX_FMT_GGUF");
  const int types[3] = {gate_type, up_type, down_type};
  // two families, by the activation format ggml pairs them with (vec_dot_type): Q8_K for the k- / i-quants, Q8_0 for the legacy types.
  // One expert set stays inside one family (the intermediate is quantised once, to the down matrix's partner: moe.hpp:388).
  const bool legacy = gl_known(types[0]);
  for (int t : types)
    KTX_REQUIRE(legacy ? gl_known(t) : gg_known(t),
                "ktx_moe_load_gguf: supported ggml types are Q2_K (10), Q3_K (11), Q4_K (12), Q5_K (13), Q6_K (14), IQ1_S (19), IQ4_XS (23) "
                "— or Q4_0 (2), Q5_0 (6), Q8_0 (8) for all three matrices");
  KTX_ON_DEVICE(h->cfg.device);
  const int E = h->cfg.expert_num, H = h->cfg.hidden_size, I = h->cfg.intermediate_size;
  const int Ns[3] = {I, I, H}, Ks[3] = {H, H, I};
  const uint8_t* src[3] = {(const uint8_t*)d_gate, (const uint8_t*)d_up, (const uint8_t*)d_down};
  uint8_t** dst[3] = {&h->gate_w, &h->up_w, &h->down_w};
  for (int m = 0; m < 3; m++) {
    if (*dst[m]) { KTX_HIP(hipFree(*dst[m])); *dst[m] = nullptr; }
    h->gg_type[m] = types[m];
    h->gg_stride[m] = legacy ? gl_matrix_bytes(types[m], Ns[m], Ks[m]) : gg_matrix_bytes(types[m], Ns[m], Ks[m]);
    KTX_HIP(hipMalloc(dst[m], (size_t)E * h->gg_stride[m]));
    const size_t src_stride = (size_t)Ns[m] * (legacy ? gl_src_row_bytes(types[m], Ks[m]) : gg_src_row_bytes(types[m], Ks[m]));
    const int ntiles = (Ns[m] / 16) * (Ks[m] / (legacy ? 32 : 256));
    for (int e = 0; e < E; e++) {
      const uint8_t* sp = src[m] + e * src_stride;
      uint8_t* dp = *dst[m] + e * h->gg_stride[m];
      if (types[m] == GG_Q4_0) hipLaunchKernelGGL(gl_pack_kernel<GG_Q4_0>, dim3(ntiles), dim3(64), 0, 0, sp, Ns[m], Ks[m], dp);
      else if (types[m] == GG_Q5_0) hipLaunchKernelGGL(gl_pack_kernel<GG_Q5_0>, dim3(ntiles), dim3(64), 0, 0, sp, Ns[m], Ks[m], dp);
      else if (types[m] == GG_Q8_0) hipLaunchKernelGGL(gl_pack_kernel<GG_Q8_0>, dim3(ntiles), dim3(64), 0, 0, sp, Ns[m], Ks[m], dp);
      else if (types[m] == GG_Q4K) hipLaunchKernelGGL(gg_pack_q4k_kernel, dim3(ntiles), dim3(64), 0, 0, sp, Ns[m], Ks[m], dp);
      else if (types[m] == GG_IQ1S) hipLaunchKernelGGL(gg_pack_iq1s_kernel, dim3(ntiles), dim3(64), 0, 0, sp, Ns[m], Ks[m], dp);
      else if (types[m] == GG_Q5K) hipLaunchKernelGGL(gg_pack_q5k_kernel, dim3(ntiles), dim3(64), 0, 0, sp, Ns[m], Ks[m], dp);
      else if (types[m] == GG_Q2K) hipLaunchKernelGGL(gg_pack_q23k_kernel<GG_Q2K>, dim3(ntiles), dim3(64), 0, 0, sp, Ns[m], Ks[m], dp);
      else if (types[m] == GG_Q3K) hipLaunchKernelGGL(gg_pack_q23k_kernel<GG_Q3K>, dim3(ntiles), dim3(64), 0, 0, sp, Ns[m], Ks[m], dp);
      else if (types[m] == GG_IQ4XS) hipLaunchKernelGGL(gg_pack_iq4xs_kernel, dim3(ntiles), dim3(64), 0, 0, sp, Ns[m], Ks[m], dp);
      else
        hipLaunchKernelGGL(gg_pack_q6k_kernel, dim3(ntiles), dim3(64), 0, 0, src[m] + e * src_stride, Ns[m], Ks[m], *dst[m] + e * h->gg_stride[m]);
    }
  }
  KTX_HIP(hipGetLastError());
  KTX_HIP(hipDeviceSynchronize());
  h->loaded_gguf = true;
  return 0;
}

static int pack_matrix(ktx_moe_s* h, const int8_t* d_q, int N, int K, uint8_t* d_dst, hipStream_t st) {
  if (h->wbits == 4) {
    const size_t total = (size_t)N * K / 8;
    hipLaunchKernelGGL(pack_w4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_q, N, K,
                       reinterpret_cast<uint32_t*>(d_dst));
  } else {
    const size_t total = (size_t)N * K / 4;
    hipLaunchKernelGGL(pack_w8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_q, N, K,
                       reinterpret_cast<uint32_t*>(d_dst));
  }
  KTX_HIP(hipGetLastError());
  return 0;
}

extern "C" int ktx_moe_load_bf16(ktx_moe_t h, const void* d_gate, const void* d_up, const void* d_down) {
  KTX_REQUIRE(h && d_gate && d_up && d_down, "ktx_moe_load_bf16: null argument");
  KTX_ON_DEVICE(h->cfg.device);
  const int E = h->cfg.expert_num, H = h->cfg.hidden_size, I = h->cfg.intermediate_size;
  KTX_REQUIRE(h->cfg.format != KTX_FMT_FP8 && h->cfg.format != KTX_FMT_FP8_PERCHANNEL && h->cfg.format != KTX_FMT_RAWINT4,
              "ktx_moe_load_bf16: FP8 / RAWINT4 handles take pre-quantised weights (ktx_moe_load_fp8 / ktx_moe_load_rawint4)");
  if (h->cfg.format == KTX_FMT_BF16) {  // no quantisation: re-tile only (BufferBBF16Impl::from_mat is a re-layout too)
    const size_t pieces = (size_t)I * H * 2 / 16;
    for (int e = 0; e < E; e++) {
      const uint8_t* src[3] = {(const uint8_t*)d_gate + (size_t)e * I * H * 2, (const uint8_t*)d_up + (size_t)e * I * H * 2,
                               (const uint8_t*)d_down + (size_t)e * H * I * 2};
      uint8_t* dst[3] = {h->gate_w + e * h->gu_stride, h->up_w + e * h->gu_stride, h->down_w + e * h->dn_stride};
      const int Ns[3] = {I, I, H}, Ks[3] = {H, H, I};
      for (int m = 0; m < 3; m++)
        hipLaunchKernelGGL(pack_wfp_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, 0, src[m], Ns[m], Ks[m], 0,
                           reinterpret_cast<uint4*>(dst[m]));
    }
    KTX_HIP(hipGetLastError());
    KTX_HIP(hipDeviceSynchronize());
    return 0;
  }
  int8_t* tmp = nullptr;
  KTX_HIP(hipMalloc(&tmp, (size_t)I * H));
  const int fmt = h->cfg.format == KTX_FMT_AMXINT4 ? 0 : 1;
  for (int e = 0; e < E; e++) {
    const bf16_t* src[3] = {(const bf16_t*)d_gate + (size_t)e * I * H, (const bf16_t*)d_up + (size_t)e * I * H,
                            (const bf16_t*)d_down + (size_t)e * H * I};
    uint8_t* dst[3] = {h->gate_w + e * h->gu_stride, h->up_w + e * h->gu_stride, h->down_w + e * h->dn_stride};
    float* sc[3] = {h->gate_s + (size_t)e * I, h->up_s + (size_t)e * I, h->down_s + (size_t)e * H};
    const int Ns[3] = {I, I, H}, Ks[3] = {H, H, I};
    for (int m = 0; m < 3; m++) {
      hipLaunchKernelGGL(quant_rows_kernel, dim3(Ns[m]), dim3(256), 0, 0, src[m], Ks[m], fmt, tmp, sc[m]);
      if (pack_matrix(h, tmp, Ns[m], Ks[m], dst[m], 0)) { hipFree(tmp); return -1; }
    }
  }
  KTX_HIP(hipDeviceSynchronize());
  KTX_HIP(hipFree(tmp));
  return 0;
}

extern "C" int ktx_moe_load_quantized(ktx_moe_t h, int expert, int which, const int8_t* q, const float* scale) {
  KTX_REQUIRE(h && q && scale, "ktx_moe_load_quantized: null argument");
  KTX_REQUIRE(expert >= 0 && expert < h->cfg.expert_num, "ktx_moe_load_quantized: expert out of range");
  KTX_REQUIRE(h->cfg.format == KTX_FMT_AMXINT4 || h->cfg.format == KTX_FMT_AMXINT8,
              "ktx_moe_load_quantized: AMXINT4 / AMXINT8 handles only");
  KTX_REQUIRE(which >= 0 && which <= 2, "ktx_moe_load_quantized: bad matrix selector");
  KTX_ON_DEVICE(h->cfg.device);
  const int H = h->cfg.hidden_size, I = h->cfg.intermediate_size;
  const int N = which == KTX_MAT_DOWN ? H : I, K = which == KTX_MAT_DOWN ? I : H;
  uint8_t* dst = which == KTX_MAT_GATE ? h->gate_w + expert * h->gu_stride
                 : which == KTX_MAT_UP ? h->up_w + expert * h->gu_stride
                                       : h->down_w + expert * h->dn_stride;
  float* sc = which == KTX_MAT_GATE ? h->gate_s + (size_t)expert * I
              : which == KTX_MAT_UP ? h->up_s + (size_t)expert * I
                                    : h->down_s + (size_t)expert * H;
  int8_t* tmp = nullptr;
  KTX_HIP(hipMalloc(&tmp, (size_t)N * K));
  KTX_HIP(hipMemcpy(tmp, q, (size_t)N * K, hipMemcpyHostToDevice));
  KTX_HIP(hipMemcpy(sc, scale, (size_t)N * sizeof(float), hipMemcpyHostToDevice));
  if (pack_matrix(h, tmp, N, K, dst, 0)) { hipFree(tmp); return -1; }
  KTX_HIP(hipDeviceSynchronize());
  KTX_HIP(hipFree(tmp));
  return 0;
}

extern "C" int ktx_moe_load_fp8(ktx_moe_t h, const void* d_gate, const void* d_up, const void* d_down,
                                const float* d_gate_scale, const float* d_up_scale, const float* d_down_scale) {
  KTX_REQUIRE(h && d_gate && d_up && d_down && d_gate_scale && d_up_scale && d_down_scale, "ktx_moe_load_fp8: null argument");
  KTX_REQUIRE(h->cfg.format == KTX_FMT_FP8, "ktx_moe_load_fp8: handle was not created with KTX_FMT_FP8");
  KTX_ON_DEVICE(h->cfg.device);
  const int E = h->cfg.expert_num, H = h->cfg.hidden_size, I = h->cfg.intermediate_size;
  const size_t pieces = (size_t)I * H / 16, nsc = (size_t)(I / 128) * (H / 128);
  for (int e = 0; e < E; e++) {
    const uint8_t* src[3] = {(const uint8_t*)d_gate + (size_t)e * I * H, (const uint8_t*)d_up + (size_t)e * I * H,
                             (const uint8_t*)d_down + (size_t)e * H * I};
    uint8_t* dst[3] = {h->gate_w + e * h->gu_stride, h->up_w + e * h->gu_stride, h->down_w + e * h->dn_stride};
    const int Ns[3] = {I, I, H}, Ks[3] = {H, H, I};
    for (int m = 0; m < 3; m++)
      hipLaunchKernelGGL(pack_wfp_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, 0, src[m], Ns[m], Ks[m], 1,
                         reinterpret_cast<uint4*>(dst[m]));
  }
  KTX_HIP(hipGetLastError());
  KTX_HIP(hipMemcpy(h->gate_s, d_gate_scale, E * nsc * sizeof(float), hipMemcpyDeviceToDevice));
  KTX_HIP(hipMemcpy(h->up_s, d_up_scale, E * nsc * sizeof(float), hipMemcpyDeviceToDevice));
  KTX_HIP(hipMemcpy(h->down_s, d_down_scale, E * nsc * sizeof(float), hipMemcpyDeviceToDevice));
  KTX_HIP(hipDeviceSynchronize());
  return 0;
}

extern "C" int ktx_moe_load_fp8_perchannel(ktx_moe_t h, const void* d_gate, const void* d_up, const void* d_down,
                                           const float* d_gate_scale, const float* d_up_scale, const float* d_down_scale) {
  KTX_REQUIRE(h && d_gate && d_up && d_down && d_gate_scale && d_up_scale && d_down_scale, "ktx_moe_load_fp8_perchannel: null argument");
  KTX_REQUIRE(h->cfg.format == KTX_FMT_FP8_PERCHANNEL, "ktx_moe_load_fp8_perchannel: handle was not created with KTX_FMT_FP8_PERCHANNEL");
  KTX_ON_DEVICE(h->cfg.device);
  const int E = h->cfg.expert_num, H = h->cfg.hidden_size, I = h->cfg.intermediate_size;
  const size_t pieces = (size_t)I * H / 16, nsc = (size_t)(I / 128) * (H / 128);
  for (int e = 0; e < E; e++) {
    const uint8_t* src[3] = {(const uint8_t*)d_gate + (size_t)e * I * H, (const uint8_t*)d_up + (size_t)e * I * H,
                             (const uint8_t*)d_down + (size_t)e * H * I};
    uint8_t* dst[3] = {h->gate_w + e * h->gu_stride, h->up_w + e * h->gu_stride, h->down_w + e * h->dn_stride};
    const int Ns[3] = {I, I, H}, Ks[3] = {H, H, I};
    for (int m = 0; m < 3; m++)
      hipLaunchKernelGGL(pack_wfp_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, 0, src[m], Ns[m], Ks[m], 1,
                         reinterpret_cast<uint4*>(dst[m]));
  }
  KTX_HIP(hipGetLastError());
  // the FP8 kernels run with every 128x128 block scale = 1 (c = fma(group, 1, c) is a plain fp32 add of the group sums);
  // the row scales are applied by their epilogues
  const std::vector<float> ones((size_t)E * nsc, 1.0f);
  KTX_HIP(hipMemcpy(h->gate_s, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice));
  KTX_HIP(hipMemcpy(h->up_s, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice));
  KTX_HIP(hipMemcpy(h->down_s, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice));
  KTX_HIP(hipMemcpy(h->gate_r, d_gate_scale, (size_t)E * I * sizeof(float), hipMemcpyDeviceToDevice));
  KTX_HIP(hipMemcpy(h->up_r, d_up_scale, (size_t)E * I * sizeof(float), hipMemcpyDeviceToDevice));
  KTX_HIP(hipMemcpy(h->down_r, d_down_scale, (size_t)E * H * sizeof(float), hipMemcpyDeviceToDevice));
  KTX_HIP(hipDeviceSynchronize());
  return 0;
}

extern "C" int ktx_moe_load_rawint4(ktx_moe_t h, const void* d_gate, const void* d_up, const void* d_down,
                                    const void* d_gate_scale, const void* d_up_scale, const void* d_down_scale) {
  KTX_REQUIRE(h && d_gate && d_up && d_down && d_gate_scale && d_up_scale && d_down_scale, "ktx_moe_load_rawint4: null argument");
  KTX_REQUIRE(h->cfg.format == KTX_FMT_RAWINT4, "ktx_moe_load_rawint4: handle was not created with KTX_FMT_RAWINT4");
  KTX_ON_DEVICE(h->cfg.device);
  const int E = h->cfg.expert_num, H = h->cfg.hidden_size, I = h->cfg.intermediate_size;
  const size_t dwords = (size_t)I * H / 8, nsc = (size_t)I * (H / 32);
  for (int e = 0; e < E; e++) {
    const uint8_t* src[3] = {(const uint8_t*)d_gate + (size_t)e * I * H / 2, (const uint8_t*)d_up + (size_t)e * I * H / 2,
                             (const uint8_t*)d_down + (size_t)e * H * I / 2};
    const bf16_t* ssrc[3] = {(const bf16_t*)d_gate_scale + e * nsc, (const bf16_t*)d_up_scale + e * nsc,
                             (const bf16_t*)d_down_scale + e * nsc};
    uint8_t* dst[3] = {h->gate_w + e * h->gu_stride, h->up_w + e * h->gu_stride, h->down_w + e * h->dn_stride};
    bf16_t* sdst[3] = {(bf16_t*)h->gate_s + e * nsc, (bf16_t*)h->up_s + e * nsc, (bf16_t*)h->down_s + e * nsc};
    const int Ns[3] = {I, I, H}, Ks[3] = {H, H, I};
    for (int m = 0; m < 3; m++) {
      hipLaunchKernelGGL(pack_rawint4_kernel, dim3((unsigned)((dwords + 255) / 256)), dim3(256), 0, 0, src[m], Ns[m], Ks[m],
                         reinterpret_cast<uint32_t*>(dst[m]));
      hipLaunchKernelGGL(pack_rawint4_scales_kernel, dim3((unsigned)((nsc + 255) / 256)), dim3(256), 0, 0, ssrc[m], Ns[m],
                         Ks[m], sdst[m]);
    }
  }
  KTX_HIP(hipGetLastError());
  KTX_HIP(hipDeviceSynchronize());
  return 0;
}

extern "C" int ktx_moe_set_expert_mask(ktx_moe_t h, const uint8_t* mask) {
  KTX_REQUIRE(h, "ktx_moe_set_expert_mask: null handle");
  KTX_ON_DEVICE(h->cfg.device);
  if (!mask) {
    if (h->mask) { KTX_HIP(hipFree(h->mask)); h->mask = nullptr; }
    return 0;
  }
  if (!h->mask) KTX_HIP(hipMalloc(&h->mask, h->cfg.expert_num));
  KTX_HIP(hipMemcpy(h->mask, mask, h->cfg.expert_num, hipMemcpyHostToDevice));
  return 0;
}

extern "C" int ktx_moe_debug_ptrs(ktx_moe_t h, const void** act_bf16, const void** down_bf16,
                                  const int32_t** row_of_pair) {
  KTX_REQUIRE(h, "ktx_moe_debug_ptrs: null handle");
  if (act_bf16) *act_bf16 = h->ws->a_buf;
  if (down_bf16) *down_bf16 = h->ws->dn_buf;
  if (row_of_pair) *row_of_pair = h->ws->row_of_pair;
  return 0;
}

template <int WBITS, int MT, int SPC, bool GATE_UP>
static int launch_gemm(const GemmParams& p, int max_tiles, hipStream_t st) {
  constexpr int BUF_BYTES = MT * (SPC * 128 / 16) * 256;
  const size_t lds = 2 * BUF_BYTES + MT * 16 * 8;
  auto kern = moe_gemm_kernel<WBITS, MT, SPC, GATE_UP>;
  const hipError_t attr_err = ktx_set_max_lds(reinterpret_cast<const void*>(kern), (int)lds);
  KTX_HIP(attr_err);
  const dim3 grid((p.N / 16 + 3) / 4, max_tiles);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  KTX_HIP(hipGetLastError());
  return 0;
}

static int g_dbg[32] = {0};  // dev knobs: [0] waves/workgroup override of the decode gate/up kernel, [1] ablation bits
template <int WBITS, bool GATE_UP>
static int launch_gemm_stream(const GemmParams& p, int max_tiles, hipStream_t st) {
  constexpr int MT = 4, D = (WBITS == 4) ? 4 : 2, TOK = MT * 16, CS = TOK * 16;
  const int kch = std::min(p.K, 2048);
  const size_t lds = (size_t)(kch / 16) * CS + TOK * 8;
  auto kern = moe_gemm_stream_kernel<WBITS, MT, D, GATE_UP>;
  const hipError_t attr_err = ktx_set_max_lds(reinterpret_cast<const void*>(kern), 160 * 1024);
  KTX_HIP(attr_err);
  const int strips_per_wg = 8 * (GATE_UP ? 2 : 4);
  hipLaunchKernelGGL(kern, dim3((p.N / 16 + strips_per_wg - 1) / strips_per_wg, max_tiles), dim3(512), lds, st, p, kch);
  KTX_HIP(hipGetLastError());
  return 0;
}

template <int WBITS, bool GATE_UP>
static int launch_gemm_rt(const GemmParams& p, int max_tiles, hipStream_t st) {
  const size_t lds = 2 * (8 * 256 * 16 + 16 * 2048) + 256 * 8;
  auto kern = moe_gemm_rt_kernel<WBITS, GATE_UP>;
  const hipError_t attr_err = ktx_set_max_lds(reinterpret_cast<const void*>(kern), 160 * 1024);
  KTX_HIP(attr_err);
  const int strips_per_wg = GATE_UP ? 8 : 16;
  hipLaunchKernelGGL(kern, dim3((p.N / 16 + strips_per_wg - 1) / strips_per_wg, max_tiles), dim3(512), lds, st, p);
  KTX_HIP(hipGetLastError());
  return 0;
}

template <int WBITS, bool GATE_UP>
static int launch_gemm_mt(int mt, const GemmParams& p, int max_tiles, hipStream_t st) {
  switch (mt) {
    case 1: return launch_gemm<WBITS, 1, 4, GATE_UP>(p, max_tiles, st);
    case 2: return launch_gemm<WBITS, 2, 4, GATE_UP>(p, max_tiles, st);
    default: return launch_gemm<WBITS, 4, 2, GATE_UP>(p, max_tiles, st);   // (512-k chunks, SPC = 4, measured slower)
  }
}

template <bool FP8, bool GATE_UP>
static int launch_gemm_fp(int mt, const FpGemmParams& p, int max_tiles, hipStream_t st) {
  const int nstrips = (p.N + 15) / 16;
#define KTX_FP_LAUNCH(MT, WIDE, PER_WG, NTH)                                                                                  \
  do {                                                                                                                        \
    constexpr size_t lds = 2 * (MT * 32 * 256) + MT * 16 * sizeof(int);                                                       \
    const hipError_t err = ktx_set_max_lds(reinterpret_cast<const void*>(moe_gemm_fp_kernel<FP8, MT, GATE_UP, WIDE>), (int)lds);                        \
    KTX_HIP(err);                                                                                                             \
    hipLaunchKernelGGL((moe_gemm_fp_kernel<FP8, MT, GATE_UP, WIDE>), dim3((nstrips + PER_WG - 1) / PER_WG, max_tiles), dim3(NTH), lds, st, p); \
  } while (0)
  if (mt == 1) KTX_FP_LAUNCH(1, false, 4, 256);
  else if (mt == 2) KTX_FP_LAUNCH(2, false, 4, 256);
  else if (g_dbg[30] == 1) KTX_FP_LAUNCH(4, false, 4, 256);          // A/B: the round-3 configuration (4 strips per workgroup)
  else if (GATE_UP) KTX_FP_LAUNCH(4, true, 8, 512);
  else KTX_FP_LAUNCH(4, true, 16, 512);
#undef KTX_FP_LAUNCH
  KTX_HIP(hipGetLastError());
  return 0;
}

static int forward_fp(ktx_moe_s* h, const int32_t* d_bsz, int qlen, int k, const int64_t* d_expert_ids,
                      const float* d_weights, const void* d_input, void* d_output, int flags, hipStream_t st);
static int forward_rawint4(ktx_moe_s* h, const int32_t* d_bsz, int qlen, int k, const int64_t* d_expert_ids,
                           const float* d_weights, const void* d_input, void* d_output, int flags, hipStream_t st);
static int forward_gguf(ktx_moe_s* h, const int32_t* d_bsz, int qlen, int k, const int64_t* d_expert_ids,
                        const float* d_weights, const void* d_input, void* d_output, int flags, hipStream_t st);

// ---- optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg) -----------------
// Slots: 0 prep, 1 gate/up GEMM, 2 act-quant, 3 down GEMM, 4 combine.  Not graph-capturable; off by default.
static bool g_prof_on = false;
static bool g_force_generic = false;  // tests: route small batches through the grouped (prefill) path too
extern "C" int ktx_debug_force_generic(int on) { g_force_generic = on != 0; return 0; }
extern "C" int ktx_debug_set(int idx, int val) { if (idx >= 0 && idx < 32) g_dbg[idx] = val; return 0; }
extern "C" int ktx_debug_get(int idx) { return idx >= 0 && idx < 32 ? g_dbg[idx] : 0; }
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_ev[5];
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_free;

extern "C" int ktx_profile_enable(int on) {
  g_prof_on = on != 0;
  return 0;
}

// Synchronises the device, then returns per slot the summed elapsed ms and the number of launches since the last call.
extern "C" int ktx_profile_collect(double* ms5, long long* count5) {
  KTX_HIP(hipDeviceSynchronize());
  for (int s = 0; s < 5; s++) {
    double tot = 0.0;
    for (auto& pr : g_prof_ev[s]) {
      float ms = 0.f;
      KTX_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
      tot += ms;
      g_prof_free.push_back(pr);
    }
    if (ms5) ms5[s] = tot;
    if (count5) count5[s] = (long long)g_prof_ev[s].size();
    g_prof_ev[s].clear();
  }
  return 0;
}

static const char* const kProfSlotName[5] = {"moe_prep_kernel (bucket + x-quant)", "moe grouped gate|up GEMM", "moe_actquant_kernel",
                                             "moe grouped down GEMM", "moe_combine_kernel"};
struct ProfScope {
  int slot;
  hipStream_t st;
  std::pair<hipEvent_t, hipEvent_t> ev;
  bool on;
  KtxTimeScope ts;   // the library-wide per-launch log (ktx_prof.hip); the decode kernels label themselves
  ProfScope(int slot_, hipStream_t st_, bool timed = true)
      : slot(slot_), st(st_), on(g_prof_on),
        ts(st_, 0.0, timed && ktx_timing_mode() ? std::string(kProfSlotName[slot_]) : std::string()) {
    if (!on) return;
    if (!g_prof_free.empty()) { ev = g_prof_free.back(); g_prof_free.pop_back(); }
    else { hipEventCreate(&ev.first); hipEventCreate(&ev.second); }
    hipEventRecord(ev.first, st);
  }
  ~ProfScope() {
    if (!on) return;
    hipEventRecord(ev.second, st);
    g_prof_ev[slot].push_back(ev);
  }
};

extern "C" int ktx_moe_forward(ktx_moe_t h, const int32_t* d_bsz, int qlen, int k, const int64_t* d_expert_ids,
                               const float* d_weights, const void* d_input, void* d_output, int incremental,
                               ktx_stream_t stream) {
  return ktx_moe_forward_ex(h, d_bsz, qlen, k, d_expert_ids, d_weights, d_input, d_output,
                            incremental ? KTX_FWD_INCREMENTAL : 0, stream);
}

struct DecSide {   // ktx_moe_forward_side: the dense linear the down kernel carries, and the residual rows
  KtxLinearRaw lin;
  const void* x;
  const void* add2;
};
constexpr int KTX_MOE_NOT_FUSED = -2;   // moe_forward_impl with a side linear: this shape has no combined kernel, nothing was launched

static int moe_forward_impl(ktx_moe_t h, const int32_t* d_bsz, int qlen, int k, const int64_t* d_expert_ids,
                            const float* d_weights, const void* d_input, void* d_output, int flags, ktx_stream_t stream,
                            const DecSide* side);

extern "C" int ktx_moe_forward_ex(ktx_moe_t h, const int32_t* d_bsz, int qlen, int k, const int64_t* d_expert_ids,
                                  const float* d_weights, const void* d_input, void* d_output, int flags,
                                  ktx_stream_t stream) {
  return moe_forward_impl(h, d_bsz, qlen, k, d_expert_ids, d_weights, d_input, d_output, flags, stream, nullptr);
}

extern "C" int ktx_moe_forward_side(ktx_moe_t h, const int32_t* d_bsz, int qlen, int k, const int64_t* d_expert_ids,
                                    const float* d_weights, const void* d_input, void* d_output, ktx_linear_t side_linear,
                                    const void* d_side_x, const void* d_residual, ktx_stream_t stream) {
  KTX_REQUIRE(h && side_linear && d_side_x, "ktx_moe_forward_side: null argument");
  DecSide sd;
  if (int rc = ktx_linear_raw(side_linear, &sd.lin)) return rc;
  KTX_REQUIRE(sd.lin.loaded, "ktx_moe_forward_side: the side linear has no weights loaded");
  KTX_REQUIRE(sd.lin.out_features == h->cfg.hidden_size && sd.lin.batch == 1 && sd.lin.device == h->cfg.device,
              "ktx_moe_forward_side: the side linear must produce hidden_size outputs on the experts' device");
  sd.x = d_side_x; sd.add2 = d_residual;
  const int rc = moe_forward_impl(h, d_bsz, qlen, k, d_expert_ids, d_weights, d_input, d_output, 0, stream, &sd);
  if (rc != KTX_MOE_NOT_FUSED) return rc;
  // no combined kernel for this shape / format: the launches it would have replaced (in-place add: every output element is
  // read and written by the same thread of the linear's epilogue)
  if (int rc2 = moe_forward_impl(h, d_bsz, qlen, k, d_expert_ids, d_weights, d_input, d_output, 0, stream, nullptr)) return rc2;
  ktx_linear_fusion fu{};
  fu.add1 = d_output; fu.add2 = d_residual;
  return ktx_linear_forward_fused(side_linear, d_bsz, qlen, d_side_x, d_output, &fu, stream);
}

static int moe_forward_impl(ktx_moe_t h, const int32_t* d_bsz, int qlen, int k, const int64_t* d_expert_ids,
                            const float* d_weights, const void* d_input, void* d_output, int flags, ktx_stream_t stream,
                            const DecSide* side) {
  const int incremental = (flags & KTX_FWD_INCREMENTAL) ? 1 : 0;
  KTX_REQUIRE(h, "ktx_moe_forward: null handle");
  KTX_REQUIRE(!((flags & KTX_FWD_PARTIAL_F32) && incremental), "ktx_moe_forward_ex: PARTIAL_F32 excludes INCREMENTAL");
  KTX_REQUIRE(qlen > 0 && qlen <= h->cfg.max_len, "ktx_moe_forward: qlen exceeds max_len");
  KTX_REQUIRE(k > 0 && k <= h->cfg.num_experts_per_tok, "ktx_moe_forward: k exceeds num_experts_per_tok");
  KTX_REQUIRE(d_expert_ids && d_weights && d_input && d_output, "ktx_moe_forward: null pointer");
  KTX_ON_DEVICE(h->cfg.device);   // the handle's device, whatever the caller's current one is (restored on return)
  hipStream_t st = (hipStream_t)stream;
  const int E = h->cfg.expert_num, H = h->cfg.hidden_size, I = h->cfg.intermediate_size;
  Workspace* ws = h->ws;
  if (h->cfg.format == KTX_FMT_AMXINT4 || h->cfg.format == KTX_FMT_AMXINT8)
  if (qlen * k <= KTX_DEC_MAX_PAIRS && k <= 8 && H <= 8192 && I <= 2048 && !g_force_generic) {
    DecParams dp;
    dp.d_bsz = d_bsz; dp.qlen = qlen; dp.k = k; dp.E = E; dp.expert_begin = h->cfg.expert_begin; dp.H = H; dp.I = I;
    dp.ids = d_expert_ids; dp.mask = h->mask; dp.x = (const bf16_t*)d_input; dp.weights = d_weights;
    dp.gate_w = h->gate_w; dp.up_w = h->up_w; dp.down_w = h->down_w;
    dp.gate_s = h->gate_s; dp.up_s = h->up_s; dp.down_s = h->down_s;
    dp.gu_stride = h->gu_stride; dp.dn_stride = h->dn_stride; dp.a_buf = ws->a_buf; dp.y = d_output;
    dp.incremental = incremental; dp.partial_f32 = (flags & KTX_FWD_PARTIAL_F32) ? 1 : 0;
    dp.ablate = 0;
    dp.side_w = nullptr; dp.side_sc = nullptr; dp.side_x = nullptr; dp.side_add2 = nullptr; dp.side_nks = 0;
    if (side) {   // W4 g64 without bias, whole 128-input k-steps, at most SIDE_MAX of them per wavefront
      if (side->lin.format != KTX_LIN_W4 || side->lin.group_size != 64 || side->lin.bias || side->lin.in_features % 128 != 0 ||
          side->lin.NKS > 4 * k || g_dbg[14] == 1 ||
          (h->wbits == 8 && (I / 128) % 8 != 0 && (I / 128) % 11 == 0))   // (that instantiation would spill at 128 registers)
        return KTX_MOE_NOT_FUSED;
      dp.side_w = side->lin.w; dp.side_sc = (const bf16_t*)side->lin.sc; dp.side_x = (const bf16_t*)side->x;
      dp.side_add2 = (const bf16_t*)side->add2; dp.side_nks = side->lin.NKS;
    }
    // waves per workgroup of the gate/up kernel: with few (pair, strip) work items use small workgroups so every CU
    // gets work; the x-row register cache needs H <= NW*2048.
    const int strips = I / 16;
    int nw = 4;
    if (H <= 2048 && (strips + 3) / 4 * qlen * k < 512) nw = 1;
    else if (H <= 4096 && (strips + 3) / 4 * qlen * k < 512) nw = 2;
    if (g_dbg[0] == 1 && H <= 2048) nw = 1;
    if (g_dbg[0] == 2 && H <= 4096) nw = 2;
    if (g_dbg[0] == 4) nw = 4;
    const dim3 g1((strips + nw - 1) / nw, qlen * k), g2(H / 16, qlen);
    // two k-slices per strip when a single-token launch would otherwise leave one 4-wave workgroup per CU (DeepSeek-V3 /
    // Kimi-K2: 32 strip groups x 8 pairs = 256 workgroups): knob 10 = 1 turns the split off for A/B timing
    const bool ksplit = h->wbits == 4 && nw == 4 && (H / 128) % 28 == 0 && g1.x * g1.y <= 512 && g_dbg[10] != 1;
    const size_t lds1 = (size_t)H + 128 + 16 + (ksplit ? 8 * sizeof(float) + (size_t)4 * 8 * 64 * sizeof(int) : 4 * sizeof(float));
    const size_t lds2 = (size_t)k * (I + 128) + (size_t)k * 16 * sizeof(float) + (size_t)k * 2 * sizeof(int) +
                        (side ? (size_t)k * 16 * sizeof(float) + 16 + (size_t)dp.side_nks * (256 + 2 * 4 * sizeof(float)) : 0);
    KTX_REQUIRE(lds2 <= 160 * 1024, "ktx_moe_forward: k*I too large for the decode path");
    const int nks1 = H / 128, nks2 = I / 128;
    const int only = g_dbg[2];
#define KTX_LAUNCH_GU(WB, DD, EX)                                                                                   \
    do {                                                                                                            \
      if (nw == 1) hipLaunchKernelGGL((moe_dec_gateup_kernel<WB, DD, 1, EX>), g1, dim3(64), lds1, st, dp);          \
      else if (nw == 2) hipLaunchKernelGGL((moe_dec_gateup_kernel<WB, DD, 2, EX>), g1, dim3(128), lds1, st, dp);    \
      else hipLaunchKernelGGL((moe_dec_gateup_kernel<WB, DD, 4, EX>), g1, dim3(256), lds1, st, dp);                 \
    } while (0)
#define KTX_LAUNCH_DN1(WB, DD, EX, SG, SM)                                                                          \
    do {                                                                                                            \
      const hipError_t err = ktx_set_max_lds(reinterpret_cast<const void*>(moe_dec_down_kernel<WB, DD, EX, SG, SM>), 160 * 1024); \
      KTX_HIP(err);                                                                                                 \
      hipLaunchKernelGGL((moe_dec_down_kernel<WB, DD, EX, SG, SM>), g2, dim3(64 * k), lds2, st, dp);                \
    } while (0)
    // (W8 with the 11-deep ring never carries a side strip — refused above)
#define KTX_LAUNCH_DN(WB, DD, EX)                                                                                   \
    do {                                                                                                            \
      if (side && (dp.side_nks + k - 1) / k <= 2) KTX_LAUNCH_DN1(WB, DD, EX, 64, 2);                                \
      else if (side) KTX_LAUNCH_DN1(WB, DD, EX, 64, 4);                                                             \
      else KTX_LAUNCH_DN1(WB, DD, EX, 0, 2);                                                                        \
    } while (0)
    const double wb = h->wbits / 8.0;
    if (only != 2) {
      ProfScope ps(1, st, false);
      KTX_TIMED(st, qlen * k * (2.0 * I * H * wb + 2.0 * I * 4 + I * 2.0) + qlen * H * 2.0,
                "moe_dec_gateup_kernel<W%d> T=%d k=%d H=%d I=%d", h->wbits, qlen, k, H, I);
      if (ksplit) hipLaunchKernelGGL((moe_dec_gateup_kernel<4, 14, 4, true, 2>), g1, dim3(512), lds1, st, dp);
      else if (h->wbits == 4) {
        if (nks1 % 16 == 0) KTX_LAUNCH_GU(4, 16, true);
        else if (nks1 % 14 == 0) KTX_LAUNCH_GU(4, 14, true);
        else if (nks1 % 11 == 0) KTX_LAUNCH_GU(4, 11, true);
        else KTX_LAUNCH_GU(4, 16, false);
      } else {
        if (nks1 % 8 == 0) KTX_LAUNCH_GU(8, 8, true);
        else if (nks1 % 7 == 0) KTX_LAUNCH_GU(8, 7, true);
        else KTX_LAUNCH_GU(8, 8, false);
      }
    }
    KTX_HIP(hipGetLastError());
    if (only != 1) {
      ProfScope ps(3, st, false);
      KTX_TIMED(st, qlen * k * ((double)H * I * wb + H * 4.0 + I * 2.0) + qlen * H * 2.0 +
                        (side ? (double)H * dp.side_nks * 128 * 0.5625 + qlen * (dp.side_nks * 256.0 + H * 2.0) : 0.0),
                "moe_dec_down_kernel<W%d> T=%d k=%d H=%d I=%d%s", h->wbits, qlen, k, H, I,
                side ? ktx_fmt(" + side W4 %d->%d", dp.side_nks * 128, H).c_str() : "");
      if (h->wbits == 4) {
        if (nks2 % 16 == 0) KTX_LAUNCH_DN(4, 16, true);
        else if (nks2 % 14 == 0) KTX_LAUNCH_DN(4, 14, true);
        else if (nks2 % 11 == 0) KTX_LAUNCH_DN(4, 11, true);
        else KTX_LAUNCH_DN(4, 16, false);
      } else {
        if (nks2 % 8 == 0) KTX_LAUNCH_DN(8, 8, true);
        else if (nks2 % 11 == 0) KTX_LAUNCH_DN(8, 11, true);
        else KTX_LAUNCH_DN(8, 8, false);
      }
    }
    KTX_HIP(hipGetLastError());
    return 0;
  }
  if (side) return KTX_MOE_NOT_FUSED;   // only the AMXINT4 / AMXINT8 decode kernels carry a side linear
  if (h->cfg.format == KTX_FMT_FP8 || h->cfg.format == KTX_FMT_FP8_PERCHANNEL || h->cfg.format == KTX_FMT_BF16)
    return forward_fp(h, d_bsz, qlen, k, d_expert_ids, d_weights, d_input, d_output, flags, st);
  if (h->cfg.format == KTX_FMT_RAWINT4)
    return forward_rawint4(h, d_bsz, qlen, k, d_expert_ids, d_weights, d_input, d_output, flags, st);
  if (h->cfg.format == KTX_FMT_GGUF)
    return forward_gguf(h, d_bsz, qlen, k, d_expert_ids, d_weights, d_input, d_output, flags, st);
  int mt = pick_mt(qlen, k, E);
  const bool use_rt = mt == 4 && g_dbg[4] == 3;   // register-tile kernels (256-row tiles): parity-tested, not yet timed
  if (use_rt) mt = 16;
  const int npairs = qlen * k;
  // 64-row tiles (prompts) of large expert matrices take the streaming kernels: measured 1.35x (gate/up) - 1.55x (down) on
  // DeepSeek-V3-shaped experts (7168 x 2048), a wash on V2-Lite's 2048 x 1408.  Dev knob [4]: 1 = never, 2 = always (tests).
  const bool use_stream = mt == 4 && g_dbg[4] != 1 && (g_dbg[4] == 2 || (size_t)H * I >= ((size_t)4 << 20));
  const int max_tiles = std::min(npairs, E) + npairs / (16 * mt);

  PrepParams pp;
  pp.d_bsz = d_bsz; pp.qlen = qlen; pp.k = k; pp.E = E; pp.expert_begin = h->cfg.expert_begin; pp.H = H;
  pp.rows_per_tile = 16 * mt; pp.ids = d_expert_ids; pp.mask = h->mask; pp.x = (const bf16_t*)d_input;
  pp.x_q = ws->x_q; pp.x_d = ws->x_d; pp.row_of_pair = ws->row_of_pair; pp.src_of_row = ws->src_of_row;
  pp.tiles = ws->tiles; pp.counters = ws->counters;
  {
    ProfScope ps(0, st);
    KTX_HIP(launch_moe_prep(pp, qlen + 1, st));
  }
  KTX_HIP(hipGetLastError());

  GemmParams g1;
  g1.w0 = h->gate_w; g1.w1 = h->up_w; g1.s0 = h->gate_s; g1.s1 = h->up_s; g1.expert_stride = h->gu_stride;
  g1.N = I; g1.K = H; g1.act_q = ws->x_q; g1.act_d = ws->x_d; g1.row_src = ws->src_of_row; g1.tiles = ws->tiles;
  g1.counters = ws->counters; g1.out = ws->a_buf;
  int rc;
  {
    ProfScope ps(1, st);
    if (use_rt) rc = h->wbits == 4 ? launch_gemm_rt<4, true>(g1, max_tiles, st) : launch_gemm_rt<8, true>(g1, max_tiles, st);
    else if (use_stream) rc = h->wbits == 4 ? launch_gemm_stream<4, true>(g1, max_tiles, st) : launch_gemm_stream<8, true>(g1, max_tiles, st);
    else rc = h->wbits == 4 ? launch_gemm_mt<4, true>(mt, g1, max_tiles, st) : launch_gemm_mt<8, true>(mt, g1, max_tiles, st);
  }
  if (rc) return rc;

  {
    ProfScope ps(2, st);
    hipLaunchKernelGGL(moe_actquant_kernel, dim3(npairs), dim3(256), 0, st, ws->a_buf, I, ws->a_q, ws->a_d, ws->counters);
  }
  KTX_HIP(hipGetLastError());

  GemmParams g2;
  g2.w0 = h->down_w; g2.w1 = nullptr; g2.s0 = h->down_s; g2.s1 = nullptr; g2.expert_stride = h->dn_stride;
  g2.N = H; g2.K = I; g2.act_q = ws->a_q; g2.act_d = ws->a_d; g2.row_src = nullptr; g2.tiles = ws->tiles;
  g2.counters = ws->counters; g2.out = ws->dn_buf;
  {
    ProfScope ps(3, st);
    if (use_rt) rc = h->wbits == 4 ? launch_gemm_rt<4, false>(g2, max_tiles, st) : launch_gemm_rt<8, false>(g2, max_tiles, st);
    else if (use_stream) rc = h->wbits == 4 ? launch_gemm_stream<4, false>(g2, max_tiles, st) : launch_gemm_stream<8, false>(g2, max_tiles, st);
    else rc = h->wbits == 4 ? launch_gemm_mt<4, false>(mt, g2, max_tiles, st) : launch_gemm_mt<8, false>(mt, g2, max_tiles, st);
  }
  if (rc) return rc;

  CombineParams cp{};
  cp.d_bsz = d_bsz; cp.qlen = qlen; cp.k = k; cp.H = H; cp.dn = ws->dn_buf; cp.row_of_pair = ws->row_of_pair;
  cp.weights = d_weights; cp.y = (bf16_t*)d_output; cp.incremental = incremental;
  cp.partial_f32 = (flags & KTX_FWD_PARTIAL_F32) ? 1 : 0;
  {
    ProfScope ps(4, st);
    hipLaunchKernelGGL(moe_combine_kernel, dim3((H / 4 + 255) / 256, qlen), dim3(256), 0, st, cp);
  }
  KTX_HIP(hipGetLastError());
  return 0;
}


// FP8 / BF16: bucket -> gate/up GEMM (+SiLU*up) -> down GEMM -> combine; activations stay bf16 (no quantisation launch).
static int forward_fp(ktx_moe_s* h, const int32_t* d_bsz, int qlen, int k, const int64_t* d_expert_ids,
                      const float* d_weights, const void* d_input, void* d_output, int flags, hipStream_t st) {
  Workspace* ws = h->ws;
  const int E = h->cfg.expert_num, H = h->cfg.hidden_size, I = h->cfg.intermediate_size;
  const bool fp8 = h->cfg.format == KTX_FMT_FP8 || h->cfg.format == KTX_FMT_FP8_PERCHANNEL;   // per-channel: block scales = 1 + row scales
  const int mt = std::min(4, pick_mt(qlen, k, E));
  const int npairs = qlen * k;
  const int max_tiles = std::min(npairs, E) + npairs / (16 * mt);

  // ---- decode fast path: two launches (see moe_dec_fp_gateup_kernel) ------------------------------------------------------
  const int nks1 = H / 128, nks2 = I / 128;
  const int d1 = fp8 ? (nks1 % 8 == 0 ? 8 : nks1 % 7 == 0 ? 7 : nks1 % 4 == 0 ? 4 : nks1 % 2 == 0 ? 2 : 1)
                     : (nks1 % 4 == 0 ? 4 : nks1 % 2 == 0 ? 2 : 1);
  const int d2 = fp8 ? (nks2 % 16 == 0 ? 16 : nks2 % 8 == 0 ? 8 : nks2 % 4 == 0 ? 4 : nks2 % 2 == 0 ? 2 : 1)
                     : (nks2 % 8 == 0 ? 8 : nks2 % 4 == 0 ? 4 : nks2 % 2 == 0 ? 2 : 1);
  const size_t lds_dn = (size_t)k * I * 2 + (size_t)k * 16 * sizeof(float) + (size_t)k * 2 * sizeof(int);
  if (npairs <= KTX_DEC_MAX_PAIRS && k <= 8 && H <= 16384 && I <= 4096 && lds_dn <= 160 * 1024 && !g_force_generic) {
    DecFpParams dp;
    dp.d_bsz = d_bsz; dp.qlen = qlen; dp.k = k; dp.E = E; dp.expert_begin = h->cfg.expert_begin; dp.H = H; dp.I = I;
    dp.ids = d_expert_ids; dp.mask = h->mask; dp.x = (const bf16_t*)d_input; dp.weights = d_weights;
    dp.gate_w = h->gate_w; dp.up_w = h->up_w; dp.down_w = h->down_w;
    dp.gate_s = h->gate_s; dp.up_s = h->up_s; dp.down_s = h->down_s;
    dp.gate_r = h->gate_r; dp.up_r = h->up_r; dp.down_r = h->down_r;
    dp.gu_stride = h->gu_stride; dp.dn_stride = h->dn_stride; dp.a_buf = ws->a_buf; dp.y = d_output;
    dp.incremental = (flags & KTX_FWD_INCREMENTAL) ? 1 : 0; dp.partial_f32 = (flags & KTX_FWD_PARTIAL_F32) ? 1 : 0;
    const dim3 g1((I / 16 + 3) / 4, npairs), g2(H / 16, qlen);
    const size_t lds_gu = (size_t)H * 2;
    const double wb = fp8 ? 1.0 : 2.0;
    const int only = g_dbg[2];
#define KTX_FP_GU(F8, DD)                                                                                            \
    do {                                                                                                             \
      const hipError_t err = ktx_set_max_lds(reinterpret_cast<const void*>(moe_dec_fp_gateup_kernel<F8, DD, 4>), 160 * 1024); \
      KTX_HIP(err);                                                                                                  \
      hipLaunchKernelGGL((moe_dec_fp_gateup_kernel<F8, DD, 4>), g1, dim3(256), lds_gu, st, dp);                      \
    } while (0)
#define KTX_FP_DN(F8, DD)                                                                                            \
    do {                                                                                                             \
      const hipError_t err = ktx_set_max_lds(reinterpret_cast<const void*>(moe_dec_fp_down_kernel<F8, DD>), 160 * 1024); \
      KTX_HIP(err);                                                                                                  \
      hipLaunchKernelGGL((moe_dec_fp_down_kernel<F8, DD>), g2, dim3(64 * k), lds_dn, st, dp);                        \
    } while (0)
    if (only != 2) {
      KTX_TIMED(st, npairs * (2.0 * I * H * wb + I * 2.0) + qlen * H * 2.0, "moe_dec_fp_gateup_kernel<%s> T=%d k=%d H=%d I=%d",
                fp8 ? "FP8" : "BF16", qlen, k, H, I);
      if (fp8) {
        switch (d1) { case 8: KTX_FP_GU(true, 8); break; case 7: KTX_FP_GU(true, 7); break; case 4: KTX_FP_GU(true, 4); break;
                      case 2: KTX_FP_GU(true, 2); break; default: KTX_FP_GU(true, 1); break; }
      } else {
        switch (d1) { case 4: KTX_FP_GU(false, 4); break; case 2: KTX_FP_GU(false, 2); break; default: KTX_FP_GU(false, 1); break; }
      }
    }
    KTX_HIP(hipGetLastError());
    if (only != 1) {
      KTX_TIMED(st, npairs * ((double)H * I * wb + I * 2.0) + qlen * H * 2.0, "moe_dec_fp_down_kernel<%s> T=%d k=%d H=%d I=%d",
                fp8 ? "FP8" : "BF16", qlen, k, H, I);
      if (fp8) {
        switch (d2) { case 16: KTX_FP_DN(true, 16); break; case 8: KTX_FP_DN(true, 8); break; case 4: KTX_FP_DN(true, 4); break;
                      case 2: KTX_FP_DN(true, 2); break; default: KTX_FP_DN(true, 1); break; }
      } else {
        switch (d2) { case 8: KTX_FP_DN(false, 8); break; case 4: KTX_FP_DN(false, 4); break; case 2: KTX_FP_DN(false, 2); break;
                      default: KTX_FP_DN(false, 1); break; }
      }
    }
    KTX_HIP(hipGetLastError());
#undef KTX_FP_GU
#undef KTX_FP_DN
    return 0;
  }

  PrepParams pp;
  pp.d_bsz = d_bsz; pp.qlen = qlen; pp.k = k; pp.E = E; pp.expert_begin = h->cfg.expert_begin; pp.H = H;
  pp.rows_per_tile = 16 * mt; pp.ids = d_expert_ids; pp.mask = h->mask; pp.x = (const bf16_t*)d_input;
  pp.x_q = ws->x_q; pp.x_d = ws->x_d; pp.row_of_pair = ws->row_of_pair; pp.src_of_row = ws->src_of_row;
  pp.tiles = ws->tiles; pp.counters = ws->counters;
  {
    ProfScope ps(0, st);
    KTX_HIP(launch_moe_prep(pp, 1, st));  // block 0 only: bucketing, no quantisation
  }
  KTX_HIP(hipGetLastError());

  FpGemmParams g1;
  g1.w0 = h->gate_w; g1.w1 = h->up_w; g1.s0 = h->gate_s; g1.s1 = h->up_s; g1.r0 = h->gate_r; g1.r1 = h->up_r;
  g1.expert_stride = h->gu_stride;
  g1.scale_stride = (size_t)(I / 128) * (H / 128); g1.N = I; g1.K = H; g1.act = (const bf16_t*)d_input;
  g1.row_src = ws->src_of_row; g1.tiles = ws->tiles; g1.counters = ws->counters; g1.out = ws->a_buf;
  int rc;
  {
    ProfScope ps(1, st);
    rc = fp8 ? launch_gemm_fp<true, true>(mt, g1, max_tiles, st) : launch_gemm_fp<false, true>(mt, g1, max_tiles, st);
  }
  if (rc) return rc;
  FpGemmParams g2;
  g2.w0 = h->down_w; g2.w1 = nullptr; g2.s0 = h->down_s; g2.s1 = nullptr; g2.r0 = h->down_r; g2.r1 = nullptr;
  g2.expert_stride = h->dn_stride;
  g2.scale_stride = (size_t)(H / 128) * (I / 128); g2.N = H; g2.K = I; g2.act = ws->a_buf; g2.row_src = nullptr;
  g2.tiles = ws->tiles; g2.counters = ws->counters; g2.out = ws->dn_buf;
  {
    ProfScope ps(3, st);
    rc = fp8 ? launch_gemm_fp<true, false>(mt, g2, max_tiles, st) : launch_gemm_fp<false, false>(mt, g2, max_tiles, st);
  }
  if (rc) return rc;
  CombineParams cp{};
  cp.d_bsz = d_bsz; cp.qlen = qlen; cp.k = k; cp.H = H; cp.dn = ws->dn_buf; cp.row_of_pair = ws->row_of_pair;
  cp.weights = d_weights; cp.y = (bf16_t*)d_output; cp.incremental = (flags & KTX_FWD_INCREMENTAL) ? 1 : 0;
  cp.partial_f32 = (flags & KTX_FWD_PARTIAL_F32) ? 1 : 0;
  {
    ProfScope ps(4, st);
    hipLaunchKernelGGL(moe_combine_kernel, dim3((H / 4 + 255) / 256, qlen), dim3(256), 0, st, cp);
  }
  KTX_HIP(hipGetLastError());
  return 0;
}


template <int NT, bool GATE_UP>
static int launch_rawint4(const RawGemmParams& p, int max_tiles, hipStream_t st) {
  const size_t lds = 4 * (size_t)(p.K + 32) + (size_t)(p.K / 32) * 16;
  const hipError_t err = ktx_set_max_lds(reinterpret_cast<const void*>(moe_rawint4_gemm_kernel<NT, GATE_UP>), 96 * 1024);
  KTX_HIP(err);
  KTX_REQUIRE(lds <= 96 * 1024, "ktx_moe_forward: K too large for the RAWINT4 kernel");
  hipLaunchKernelGGL((moe_rawint4_gemm_kernel<NT, GATE_UP>), dim3((p.N / 16 + 3) / 4, max_tiles), dim3(256), lds, st, p);
  KTX_HIP(hipGetLastError());
  return 0;
}

template <bool GATE_UP>
static int launch_rawint4_chunk(const RawGemmParams& p, int max_tiles, hipStream_t st) {
  constexpr size_t lds = 2 * (32 * (64 * 16 + 16) + 16 * 64 * 4);
  const hipError_t err = ktx_set_max_lds(reinterpret_cast<const void*>(moe_rawint4_chunk_kernel<GATE_UP>), (int)lds);
  KTX_HIP(err);
  const int per_wg = GATE_UP ? 8 : 16;      // strips per workgroup of 8 wavefronts (down: two strips per wavefront)
  hipLaunchKernelGGL((moe_rawint4_chunk_kernel<GATE_UP>), dim3((p.N / 16 + per_wg - 1) / per_wg, max_tiles), dim3(512), lds, st, p);
  KTX_HIP(hipGetLastError());
  return 0;
}

// RAWINT4: bucket (4-row tiles) -> per-group activation quant -> gate/up -> per-group requant -> down -> combine
static int forward_rawint4(ktx_moe_s* h, const int32_t* d_bsz, int qlen, int k, const int64_t* d_expert_ids,
                           const float* d_weights, const void* d_input, void* d_output, int flags, hipStream_t st) {
  Workspace* ws = h->ws;
  const int E = h->cfg.expert_num, H = h->cfg.hidden_size, I = h->cfg.intermediate_size;
  const int npairs = qlen * k;
  const int max_tiles = std::min(npairs, E) + npairs / 4;
  const size_t nsc_gu = (size_t)I * (H / 32), nsc_dn = (size_t)H * (I / 32);

  // ---- decode fast path: two launches (see moe_dec_raw_gateup_kernel) -----------------------------------------------------
  const size_t lds_rdn = (size_t)k * (I + I / 32 * 4 + 16 * 4 + 8);
  if (npairs <= KTX_DEC_MAX_PAIRS && k <= 8 && H <= 8192 && I <= 4096 && lds_rdn <= 160 * 1024 && !g_force_generic) {
    RawDecParams dp;
    dp.d_bsz = d_bsz; dp.qlen = qlen; dp.k = k; dp.E = E; dp.expert_begin = h->cfg.expert_begin; dp.H = H; dp.I = I;
    dp.ids = d_expert_ids; dp.mask = h->mask; dp.x = (const bf16_t*)d_input; dp.weights = d_weights;
    dp.gate_w = h->gate_w; dp.up_w = h->up_w; dp.down_w = h->down_w;
    dp.gate_s = (const bf16_t*)h->gate_s; dp.up_s = (const bf16_t*)h->up_s; dp.down_s = (const bf16_t*)h->down_s;
    dp.gu_stride = h->gu_stride; dp.dn_stride = h->dn_stride; dp.gu_sstride = nsc_gu; dp.dn_sstride = nsc_dn;
    dp.a_buf = ws->a_buf; dp.y = d_output;
    dp.incremental = (flags & KTX_FWD_INCREMENTAL) ? 1 : 0; dp.partial_f32 = (flags & KTX_FWD_PARTIAL_F32) ? 1 : 0;
    const dim3 g1((I / 16 + 3) / 4, npairs), g2(H / 16, qlen);
    const size_t lds_gu = (size_t)H + (size_t)H / 32 * 4;
    const int ns1 = H / 512, ns2 = I / 512;
    const int only = g_dbg[2];
    const double wbytes = 0.5 + 2.0 / 32;   // nibbles + one bf16 scale per 32
    if (only != 2) {
      KTX_TIMED(st, npairs * (2.0 * I * H * wbytes + I * 2.0) + qlen * H * 2.0, "moe_dec_raw_gateup_kernel T=%d k=%d H=%d I=%d",
                qlen, k, H, I);
      if (ns1 % 2 == 0) hipLaunchKernelGGL((moe_dec_raw_gateup_kernel<2, 4>), g1, dim3(256), lds_gu, st, dp);
      else hipLaunchKernelGGL((moe_dec_raw_gateup_kernel<1, 4>), g1, dim3(256), lds_gu, st, dp);
    }
    KTX_HIP(hipGetLastError());
    if (only != 1) {
      KTX_TIMED(st, npairs * ((double)H * I * wbytes + I * 2.0) + qlen * H * 2.0, "moe_dec_raw_down_kernel T=%d k=%d H=%d I=%d",
                qlen, k, H, I);
      if (ns2 % 4 == 0) hipLaunchKernelGGL((moe_dec_raw_down_kernel<4>), g2, dim3(64 * k), lds_rdn, st, dp);
      else if (ns2 % 2 == 0) hipLaunchKernelGGL((moe_dec_raw_down_kernel<2>), g2, dim3(64 * k), lds_rdn, st, dp);
      else hipLaunchKernelGGL((moe_dec_raw_down_kernel<1>), g2, dim3(64 * k), lds_rdn, st, dp);
    }
    KTX_HIP(hipGetLastError());
    return 0;
  }

  // prompt chunks: 64-row tiles through the group-scaled 16x16x32 kernel (moe_rawint4_chunk_kernel: same terms, one fp32 chain per
  // output instead of the reference's sixteen); short batches and dev knob 29 = 1 keep the exact 4-row kernel
  const bool chunk = qlen >= 64 && g_dbg[29] != 1 && !h->exact && H % 512 == 0 && I % 512 == 0;
  PrepParams pp;
  pp.d_bsz = d_bsz; pp.qlen = qlen; pp.k = k; pp.E = E; pp.expert_begin = h->cfg.expert_begin; pp.H = H;
  pp.rows_per_tile = chunk ? 64 : 4; pp.ids = d_expert_ids; pp.mask = h->mask; pp.x = (const bf16_t*)d_input;
  pp.x_q = ws->x_q; pp.x_d = ws->x_d; pp.row_of_pair = ws->row_of_pair; pp.src_of_row = ws->src_of_row;
  pp.tiles = ws->tiles; pp.counters = ws->counters;
  {
    ProfScope ps(0, st);
    KTX_HIP(launch_moe_prep(pp, 1, st));
    hipLaunchKernelGGL(moe_actquant_kgroup_kernel, dim3(qlen), dim3(256), 0, st, (const bf16_t*)d_input, H, ws->x_q, ws->x_d,
                       ws->counters, d_bsz, qlen, 1);
  }
  KTX_HIP(hipGetLastError());
  RawGemmParams g1;
  g1.w0 = h->gate_w; g1.w1 = h->up_w; g1.s0 = (const bf16_t*)h->gate_s; g1.s1 = (const bf16_t*)h->up_s;
  g1.expert_stride = h->gu_stride; g1.scale_stride = nsc_gu; g1.N = I; g1.K = H; g1.act_q = ws->x_q; g1.act_d = ws->x_d;
  g1.row_src = ws->src_of_row; g1.tiles = ws->tiles; g1.counters = ws->counters; g1.out = ws->a_buf;
  int rc;
  {
    ProfScope ps(1, st);
    rc = chunk ? launch_rawint4_chunk<true>(g1, std::min(npairs, E) + npairs / 64, st)
               : qlen == 1 ? launch_rawint4<1, true>(g1, max_tiles, st) : launch_rawint4<4, true>(g1, max_tiles, st);
  }
  if (rc) return rc;
  {
    ProfScope ps(2, st);
    hipLaunchKernelGGL(moe_actquant_kgroup_kernel, dim3(npairs), dim3(256), 0, st, ws->a_buf, I, ws->a_q, ws->a_d, ws->counters,
                       d_bsz, qlen, 0);
  }
  KTX_HIP(hipGetLastError());
  RawGemmParams g2;
  g2.w0 = h->down_w; g2.w1 = nullptr; g2.s0 = (const bf16_t*)h->down_s; g2.s1 = nullptr; g2.expert_stride = h->dn_stride;
  g2.scale_stride = nsc_dn; g2.N = H; g2.K = I; g2.act_q = ws->a_q; g2.act_d = ws->a_d; g2.row_src = nullptr;
  g2.tiles = ws->tiles; g2.counters = ws->counters; g2.out = ws->dn_buf;
  {
    ProfScope ps(3, st);
    rc = chunk ? launch_rawint4_chunk<false>(g2, std::min(npairs, E) + npairs / 64, st)
               : qlen == 1 ? launch_rawint4<1, false>(g2, max_tiles, st) : launch_rawint4<4, false>(g2, max_tiles, st);
  }
  if (rc) return rc;
  CombineParams cp{};
  cp.d_bsz = d_bsz; cp.qlen = qlen; cp.k = k; cp.H = H; cp.dn = ws->dn_buf; cp.row_of_pair = ws->row_of_pair;
  cp.weights = d_weights; cp.y = (bf16_t*)d_output; cp.incremental = (flags & KTX_FWD_INCREMENTAL) ? 1 : 0;
  cp.partial_f32 = (flags & KTX_FWD_PARTIAL_F32) ? 1 : 0;
  {
    ProfScope ps(4, st);
    hipLaunchKernelGGL(moe_combine_kernel, dim3((H / 4 + 255) / 256, qlen), dim3(256), 0, st, cp);
  }
  KTX_HIP(hipGetLastError());
  return 0;
}


// Slot-ordered weighted combine on its own (expert-parallel prefill: the per-pair expert outputs come back over the
// all-to-all and are summed at the token's home rank exactly like the single-GPU path sums them).
extern "C" int ktx_moe_combine(int qlen, int k, int hidden, const void* d_rows, const int32_t* d_row_of_pair,
                               const float* d_weights, void* d_output, int incremental, ktx_stream_t stream) {
  KTX_REQUIRE(d_rows && d_row_of_pair && d_weights && d_output, "ktx_moe_combine: null pointer");
  KTX_REQUIRE(qlen > 0 && k > 0 && hidden > 0 && hidden % 4 == 0, "ktx_moe_combine: bad shape");
  CombineParams cp{};
  cp.d_bsz = nullptr; cp.qlen = qlen; cp.k = k; cp.H = hidden; cp.dn = (const bf16_t*)d_rows; cp.row_of_pair = d_row_of_pair;
  cp.weights = d_weights; cp.y = (bf16_t*)d_output; cp.incremental = incremental ? 1 : 0;
  hipLaunchKernelGGL(moe_combine_kernel, dim3((hidden / 4 + 255) / 256, qlen), dim3(256), 0, (hipStream_t)stream, cp);
  KTX_HIP(hipGetLastError());
  return 0;
}

// merge_results of the reference's NUMA tensor-parallel MoE (operators/amx/moe_base.hpp:749-791): part 0's fp32 row, plus the
// previous bf16 output when incremental, plus parts 1.. in order, one bf16 rounding.  4 elements per thread.
__global__ __launch_bounds__(256) void moe_merge_partials_kernel(int nparts, int qlen, int H, const float* __restrict__ parts,
                                                                 long part_stride, bf16_t* __restrict__ y, int incremental,
                                                                 const int32_t* __restrict__ d_bsz) {
  int T = qlen;
  if (d_bsz) T = min(max(*d_bsz, 0), qlen);
  const int t = blockIdx.y;
  if (t >= T) return;
  const int h = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (h >= H) return;
  const size_t off = (size_t)t * H + h;
  float4 a = *reinterpret_cast<const float4*>(parts + off);
  bf16_t* yp = y + off;
  if (incremental) {
    const uint2 o = *reinterpret_cast<const uint2*>(yp);
    a.x = a.x + bf16_to_f32((bf16_t)(o.x & 0xffffu)); a.y = a.y + bf16_to_f32((bf16_t)(o.x >> 16));
    a.z = a.z + bf16_to_f32((bf16_t)(o.y & 0xffffu)); a.w = a.w + bf16_to_f32((bf16_t)(o.y >> 16));
  }
  for (int i = 1; i < nparts; i++) {
    const float4 b = *reinterpret_cast<const float4*>(parts + (size_t)i * part_stride + off);
    a.x = a.x + b.x; a.y = a.y + b.y; a.z = a.z + b.z; a.w = a.w + b.w;
  }
  *reinterpret_cast<uint2*>(yp) = make_uint2((uint32_t)f32_to_bf16(a.x) | ((uint32_t)f32_to_bf16(a.y) << 16),
                                             (uint32_t)f32_to_bf16(a.z) | ((uint32_t)f32_to_bf16(a.w) << 16));
}

extern "C" int ktx_moe_merge_partials(int nparts, int qlen, int hidden, const float* d_parts, int64_t part_stride, void* d_output,
                                      int incremental, const int32_t* d_bsz, ktx_stream_t stream) {
  KTX_REQUIRE(d_parts && d_output, "ktx_moe_merge_partials: null pointer");
  KTX_REQUIRE(nparts > 0 && qlen > 0 && hidden > 0 && hidden % 4 == 0 && part_stride >= (int64_t)qlen * hidden && part_stride % 4 == 0,
              "ktx_moe_merge_partials: bad shape");
  hipLaunchKernelGGL(moe_merge_partials_kernel, dim3((hidden / 4 + 255) / 256, qlen), dim3(256), 0, (hipStream_t)stream, nparts, qlen,
                     hidden, d_parts, (long)part_stride, (bf16_t*)d_output, incremental ? 1 : 0, d_bsz);
  KTX_HIP(hipGetLastError());
  return 0;
}

template <int WT, int MT, bool GATE_UP>
static int launch_gguf(const GgGemmParams& p, int max_tiles, hipStream_t st) {
  constexpr int TOK = MT * 16, US = TOK * 16 + 16, NBS = gg_nbs(WT);
  const size_t lds = 2 * 16 * US + 2 * NBS * TOK * 4 + 2 * TOK * 4 + TOK * 4 + (WT == GG_IQ1S ? 4096 * 8 : 0);
  const dim3 grid((p.N / 16 + 3) / 4, max_tiles);
  const dim3 grid_f((unsigned)((p.N / 16 + 3) / 4) * (unsigned)((max_tiles + 7) / 8 * 8));   // folded kernels: (XCD, tile, strip group) in one dimension
  if constexpr (gg_foldable(WT)) {   // round 6: the folded-operand kernel; dev knob [21] = 1: gg_block's (the bit-identity test, A/B timing)
    if (g_dbg[21] != 1) {
      const size_t lds_f = 2 * TOK * 256 + 2 * TOK * 4 + TOK * 4 + (WT == GG_Q4K ? 2 * TOK * 16 : WT == GG_IQ1S ? 2 * TOK * 4 + 16384 : 0);
      auto kern = moe_gguf_fold_kernel<WT, MT, GATE_UP>;
      if (lds_f > 64 * 1024) KTX_HIP(ktx_set_max_lds(reinterpret_cast<const void*>(kern), (int)lds_f));
      hipLaunchKernelGGL(kern, grid_f, dim3(256), lds_f, st, p);
      KTX_HIP(hipGetLastError());
      return 0;
    }
  }
  if constexpr (MT > 4) { ktx_fail("launch_gguf: 128-row tiles exist for the folded Q4_K / Q6_K / IQ1_S kernels only"); return 1; }
  else {
  hipLaunchKernelGGL((moe_gguf_gemm_kernel<WT, MT, GATE_UP>), grid, dim3(256), lds, st, p);
  KTX_HIP(hipGetLastError());
  return 0;
  }
}
template <int WT, bool GATE_UP>
static int launch_gguf_mt(int mt, const GgGemmParams& p, int max_tiles, hipStream_t st) {
  switch (mt) {
    case 1: return launch_gguf<WT, 1, GATE_UP>(p, max_tiles, st);
    case 2: return launch_gguf<WT, 2, GATE_UP>(p, max_tiles, st);
    case 8:
      if constexpr (gg_foldable(WT)) return launch_gguf<WT, 8, GATE_UP>(p, max_tiles, st);
      else { ktx_fail("launch_gguf: 128-row tiles exist for the folded Q4_K / Q6_K / IQ1_S kernels only"); return 1; }
    default: return launch_gguf<WT, 4, GATE_UP>(p, max_tiles, st);
  }
}

template <bool GATE_UP>
static int launch_gguf_type(int type, int mt, const GgGemmParams& p, int max_tiles, hipStream_t st) {
  switch (type) {
    case GG_Q2K: return launch_gguf_mt<GG_Q2K, GATE_UP>(mt, p, max_tiles, st);
    case GG_Q3K: return launch_gguf_mt<GG_Q3K, GATE_UP>(mt, p, max_tiles, st);
    case GG_Q4K: return launch_gguf_mt<GG_Q4K, GATE_UP>(mt, p, max_tiles, st);
    case GG_Q5K: return launch_gguf_mt<GG_Q5K, GATE_UP>(mt, p, max_tiles, st);
    case GG_Q6K: return launch_gguf_mt<GG_Q6K, GATE_UP>(mt, p, max_tiles, st);
    case GG_IQ4XS: return launch_gguf_mt<GG_IQ4XS, GATE_UP>(mt, p, max_tiles, st);
    default: return launch_gguf_mt<GG_IQ1S, GATE_UP>(mt, p, max_tiles, st);
  }
}

// GGUF: bucket -> Q8_K(x) -> gate/up (+ act, fp32) -> Q8_K(a) -> down (fp32) -> combine (llamafile/moe.hpp:461-747)
static int forward_gguf(ktx_moe_s* h, const int32_t* d_bsz, int qlen, int k, const int64_t* d_expert_ids,
                        const float* d_weights, const void* d_input, void* d_output, int flags, hipStream_t st) {
  KTX_REQUIRE(h->loaded_gguf, "ktx_moe_forward: GGUF weights not loaded");
  KTX_REQUIRE(h->gg_type[0] == h->gg_type[1], "ktx_moe_forward: gate and up must share one ggml type (one Q8_K input, moe.hpp:284-288)");
  Workspace* ws = h->ws;
  const int E = h->cfg.expert_num, H = h->cfg.hidden_size, I = h->cfg.intermediate_size;
  const int npairs = qlen * k;
  if (gl_known(h->gg_type[0])) {
    // ---- legacy types (ktx_moe_legacy.inc): bucket -> Q8_0(x) -> gate/up (+ act, fp32) -> Q8_0(a) -> down (fp32) -> combine -------------
    const int max_tiles = std::min(npairs, E) + npairs / 16;
    PrepParams pp;
    pp.d_bsz = d_bsz; pp.qlen = qlen; pp.k = k; pp.E = E; pp.expert_begin = h->cfg.expert_begin; pp.H = H;
    pp.rows_per_tile = 16; pp.ids = d_expert_ids; pp.mask = h->mask; pp.x = (const bf16_t*)d_input;
    pp.x_q = ws->x_q; pp.x_d = ws->x_d; pp.row_of_pair = ws->row_of_pair; pp.src_of_row = ws->src_of_row;
    pp.tiles = ws->tiles; pp.counters = ws->counters;
    KTX_HIP(launch_moe_prep(pp, 1, st));
    hipLaunchKernelGGL(q80_quant_kernel<false>, dim3(qlen), dim3(256), 0, st, d_input, H, ws->x_q, ws->x_d, d_bsz, 1, qlen);
    KTX_HIP(hipGetLastError());
    GlGemmParams g1{};
    g1.w0 = h->gate_w; g1.w1 = h->up_w; g1.stride0 = h->gg_stride[0]; g1.stride1 = h->gg_stride[1]; g1.N = I; g1.K = H;
    g1.act_q = ws->x_q; g1.act_d = ws->x_d; g1.row_src = ws->src_of_row; g1.tiles = ws->tiles; g1.counters = ws->counters;
    g1.out = reinterpret_cast<float*>(ws->a_buf);
    {
      KTX_TIMED(st, (double)std::min(npairs, E) * (h->gg_stride[0] + h->gg_stride[1]) + (double)npairs * I * 4.0 + qlen * H * 2.0,
                "moe_legacy_gemm_kernel<%s,gate|up> T=%d k=%d H=%d I=%d", gl_type_name(h->gg_type[0]), qlen, k, H, I);
      if (int rc = launch_legacy_type<true>(h->gg_type[0], g1, max_tiles, st)) return rc;
    }
    hipLaunchKernelGGL(q80_quant_kernel<true>, dim3(npairs), dim3(256), 0, st, (const void*)ws->a_buf, I, ws->a_q, ws->a_d, ws->counters, 0, npairs);
    KTX_HIP(hipGetLastError());
    GlGemmParams g2{};
    g2.w0 = h->down_w; g2.w1 = nullptr; g2.stride0 = h->gg_stride[2]; g2.stride1 = 0; g2.N = H; g2.K = I;
    g2.act_q = ws->a_q; g2.act_d = ws->a_d; g2.row_src = nullptr; g2.tiles = ws->tiles; g2.counters = ws->counters;
    g2.out = reinterpret_cast<float*>(ws->dn_buf);
    {
      KTX_TIMED(st, (double)std::min(npairs, E) * h->gg_stride[2] + (double)npairs * (I + H) * 4.0,
                "moe_legacy_gemm_kernel<%s,down> T=%d k=%d H=%d I=%d", gl_type_name(h->gg_type[2]), qlen, k, H, I);
      if (int rc = launch_legacy_type<false>(h->gg_type[2], g2, max_tiles, st)) return rc;
    }
    CombineParams cp{};
    cp.d_bsz = d_bsz; cp.qlen = qlen; cp.k = k; cp.H = H; cp.dn = ws->dn_buf; cp.row_of_pair = ws->row_of_pair;
    cp.weights = d_weights; cp.y = (bf16_t*)d_output; cp.incremental = (flags & KTX_FWD_INCREMENTAL) ? 1 : 0;
    cp.partial_f32 = (flags & KTX_FWD_PARTIAL_F32) ? 1 : 0; cp.dn_f32 = 1;
    hipLaunchKernelGGL(moe_combine_kernel, dim3((H / 4 + 255) / 256, qlen), dim3(256), 0, st, cp);
    KTX_HIP(hipGetLastError());
    return 0;
  }
  int mt = std::min(4, pick_mt(qlen, k, E));
  // round 6: 128-row tiles for the folded Q4_K / Q6_K kernels once an expert sees ~100 rows — the operand folding of a 256-block is
  // paid once per tile and strip, so twice the rows halve its share (dev knob [22]: 1 = 64-row tiles always, 2 = 128-row tiles always)
  if (gg_foldable(h->gg_type[0]) && gg_foldable(h->gg_type[2]) && h->gg_type[0] != GG_IQ1S && h->gg_type[2] != GG_IQ1S &&   // (IQ1_S: its 16 KiB
      // grid table beside two 32 KiB stages leaves ONE workgroup per CU — measured 1.6x slower than 64-row tiles)
      g_dbg[21] != 1 && g_dbg[22] != 1 &&
      (g_dbg[22] == 2 || (double)npairs / std::max(1, E) >= 96.0))
    mt = 8;
  const int max_tiles = std::min(npairs, E) + npairs / (16 * mt);

  // ---- decode fast path: two launches (ktx_moe_gguf.inc, moe_dec_gguf_gateup_kernel) --------------------------------------
  {
    const int tg = h->gg_type[0], td = h->gg_type[2];
    const int nkb1 = H / 256, nkb2 = I / 256;
    const size_t lds_gu = (size_t)H + (size_t)nkb1 * (gg_nbs(tg) + 1) * 4 + 32 + (tg == GG_IQ1S ? 4096 * 8 : 0);   // (+ the arrival flag)
    const size_t lds_dn = (size_t)k * I + (size_t)k * nkb2 * (gg_nbs(td) + 1) * 4 + (size_t)k * (16 + 2) * 4 + 8 +
                          (td == GG_IQ1S ? 4096 * 8 : 0);
    if (npairs <= KTX_DEC_MAX_PAIRS && k <= 8 && nkb2 <= 64 && lds_gu <= 64 * 1024 && lds_dn <= 64 * 1024 && !g_force_generic) {
      GgDecParams dp;
      dp.d_bsz = d_bsz; dp.qlen = qlen; dp.k = k; dp.E = E; dp.expert_begin = h->cfg.expert_begin; dp.H = H; dp.I = I;
      dp.ids = d_expert_ids; dp.mask = h->mask; dp.x = (const bf16_t*)d_input; dp.weights = d_weights;
      dp.gate_w = h->gate_w; dp.up_w = h->up_w; dp.down_w = h->down_w;
      dp.gate_stride = h->gg_stride[0]; dp.up_stride = h->gg_stride[1]; dp.down_stride = h->gg_stride[2];
      dp.a_buf = reinterpret_cast<float*>(ws->a_buf); dp.y = d_output;
      dp.a_q = reinterpret_cast<uint8_t*>(ws->a_q); dp.a_bs = ws->a_bs; dp.a_d8 = ws->a_d; dp.tickets = ws->counters + 4;
      dp.incremental = (flags & KTX_FWD_INCREMENTAL) ? 1 : 0; dp.partial_f32 = (flags & KTX_FWD_PARTIAL_F32) ? 1 : 0;
      const dim3 g1((I / 16 + 3) / 4, npairs), g2(H / 16, qlen);
      int d1 = nkb1 % 4 == 0 ? 4 : nkb1 % 2 == 0 ? 2 : 1;
      const int d2 = nkb2 % 4 == 0 ? 4 : nkb2 % 2 == 0 ? 2 : 1;
      if ((tg == GG_Q6K || tg == GG_Q5K) && d1 == 4) d1 = 2;   // a 4-deep ring of gate AND up Q6_K / Q5_K tiles (3 planes each) does not fit the registers
      const int only = g_dbg[2];
      const bool ks2 = nkb1 % 2 == 0 && g_dbg[20] != 1;
      auto tname = [](int t) { return gg_type_name(t); };
#define KTX_GG_GU(WT)                                                                                                \
      do {                                                                                                           \
        if constexpr (WT == GG_IQ1S) {   /* 4 strips x 4 k-slices, a slice's tiles all in flight from the start: four wavefronts per SIMD (dev knob 20 = 2: two) */ \
          const size_t lds4 = lds_gu + 4 * 4 * 2 * 16 * 4;                                                           \
          if (nkb1 == 28 && g_dbg[20] == 0) { hipLaunchKernelGGL((moe_dec_gguf_gateup_kernel<WT, 3, 4, 4, 7>), g1, dim3(1024), lds4, st, dp); break; } \
          if (nkb1 == 16 && g_dbg[20] == 0) { hipLaunchKernelGGL((moe_dec_gguf_gateup_kernel<WT, 2, 4, 4, 4>), g1, dim3(1024), lds4, st, dp); break; } \
          if (nkb1 == 8 && g_dbg[20] == 0) { hipLaunchKernelGGL((moe_dec_gguf_gateup_kernel<WT, 2, 4, 4, 2>), g1, dim3(1024), lds4, st, dp); break; } \
        }                                                                                                            \
        if (ks2) {   /* 4 strips x 2 k-slices per workgroup: two wavefronts per SIMD (dev knob 20 = 1: one) */       \
          if (nkb1 % 4 == 0) hipLaunchKernelGGL((moe_dec_gguf_gateup_kernel<WT, 2, 4, 2>), g1, dim3(512), lds_gu + 2 * 4 * 2 * 16 * 4, st, dp); \
          else hipLaunchKernelGGL((moe_dec_gguf_gateup_kernel<WT, 1, 4, 2>), g1, dim3(512), lds_gu + 2 * 4 * 2 * 16 * 4, st, dp); \
          break;                                                                                                     \
        }                                                                                                            \
        if constexpr (WT != GG_Q6K && WT != GG_Q5K) {   /* (never chosen for these, see d1 above: not instantiated either) */ \
          if (d1 == 4) { hipLaunchKernelGGL((moe_dec_gguf_gateup_kernel<WT, 4, 4>), g1, dim3(256), lds_gu, st, dp); break; } \
        }                                                                                                            \
        if (d1 >= 2) hipLaunchKernelGGL((moe_dec_gguf_gateup_kernel<WT, 2, 4>), g1, dim3(256), lds_gu, st, dp); \
        else hipLaunchKernelGGL((moe_dec_gguf_gateup_kernel<WT, 1, 4>), g1, dim3(256), lds_gu, st, dp);              \
      } while (0)
#define KTX_GG_DN(WT)                                                                                                \
      do {                                                                                                           \
        if (d2 == 4) hipLaunchKernelGGL((moe_dec_gguf_down_kernel<WT, 4>), g2, dim3(64 * k), lds_dn, st, dp);        \
        else if (d2 == 2) hipLaunchKernelGGL((moe_dec_gguf_down_kernel<WT, 2>), g2, dim3(64 * k), lds_dn, st, dp);   \
        else hipLaunchKernelGGL((moe_dec_gguf_down_kernel<WT, 1>), g2, dim3(64 * k), lds_dn, st, dp);                \
      } while (0)
      if (only != 2) {
        KTX_TIMED(st, (double)npairs * (h->gg_stride[0] + h->gg_stride[1] + I * 4.0) + qlen * H * 2.0,
                  "moe_dec_gguf_gateup_kernel<%s> T=%d k=%d H=%d I=%d", tname(tg), qlen, k, H, I);
        switch (tg) {
          case GG_Q2K: KTX_GG_GU(GG_Q2K); break;
          case GG_Q3K: KTX_GG_GU(GG_Q3K); break;
          case GG_Q4K: KTX_GG_GU(GG_Q4K); break;
          case GG_Q5K: KTX_GG_GU(GG_Q5K); break;
          case GG_Q6K: KTX_GG_GU(GG_Q6K); break;
          case GG_IQ4XS: KTX_GG_GU(GG_IQ4XS); break;
          default: KTX_GG_GU(GG_IQ1S); break;
        }
      }
      KTX_HIP(hipGetLastError());
      if (only != 1) {
        KTX_TIMED(st, (double)npairs * (h->gg_stride[2] + I * 4.0) + qlen * H * 2.0,
                  "moe_dec_gguf_down_kernel<%s> T=%d k=%d H=%d I=%d", tname(td), qlen, k, H, I);
        switch (td) {
          case GG_Q2K: KTX_GG_DN(GG_Q2K); break;
          case GG_Q3K: KTX_GG_DN(GG_Q3K); break;
          case GG_Q4K: KTX_GG_DN(GG_Q4K); break;
          case GG_Q5K: KTX_GG_DN(GG_Q5K); break;
          case GG_Q6K: KTX_GG_DN(GG_Q6K); break;
          case GG_IQ4XS: KTX_GG_DN(GG_IQ4XS); break;
          default: KTX_GG_DN(GG_IQ1S); break;
        }
      }
      KTX_HIP(hipGetLastError());
#undef KTX_GG_GU
#undef KTX_GG_DN
      return 0;
    }
  }

  uint8_t* x_bsp = reinterpret_cast<uint8_t*>(ws->x_bs) + (size_t)h->cfg.max_len * (H / 16) * sizeof(int16_t);
  uint8_t* a_bsp = reinterpret_cast<uint8_t*>(ws->a_bs) + (size_t)h->max_pairs * (I / 16) * sizeof(int16_t);
  int32_t* x_b256 = reinterpret_cast<int32_t*>(x_bsp + (size_t)h->cfg.max_len * (H / 256) * 16);
  int32_t* a_b256 = reinterpret_cast<int32_t*>(a_bsp + (size_t)h->max_pairs * (I / 256) * 16);
  PrepParams pp;
  pp.d_bsz = d_bsz; pp.qlen = qlen; pp.k = k; pp.E = E; pp.expert_begin = h->cfg.expert_begin; pp.H = H;
  pp.rows_per_tile = 16 * mt; pp.ids = d_expert_ids; pp.mask = h->mask; pp.x = (const bf16_t*)d_input;
  pp.x_q = ws->x_q; pp.x_d = ws->x_d; pp.row_of_pair = ws->row_of_pair; pp.src_of_row = ws->src_of_row;
  pp.tiles = ws->tiles; pp.counters = ws->counters;
  {
    ProfScope ps(0, st);
    KTX_HIP(launch_moe_prep(pp, 1, st));
    hipLaunchKernelGGL(q8k_quant_kernel<false>, dim3(qlen), dim3(256), 0, st, d_input, H, ws->x_q, ws->x_d, ws->x_bs, x_bsp, x_b256, d_bsz, 1, qlen);
  }
  KTX_HIP(hipGetLastError());
  GgGemmParams g1{};
  g1.w0 = h->gate_w; g1.w1 = h->up_w; g1.stride0 = h->gg_stride[0]; g1.stride1 = h->gg_stride[1]; g1.N = I; g1.K = H;
  g1.act_q = ws->x_q; g1.act_d = ws->x_d; g1.act_bs = ws->x_bs; g1.act_bsp = x_bsp; g1.act_b256 = x_b256; g1.row_src = ws->src_of_row; g1.tiles = ws->tiles;
  g1.counters = ws->counters; g1.out = reinterpret_cast<float*>(ws->a_buf); g1.dbg = g_dbg[24];
  int rc;
  {
    ProfScope ps(1, st);
    rc = launch_gguf_type<true>(h->gg_type[0], mt, g1, max_tiles, st);
  }
  if (rc) return rc;
  {
    ProfScope ps(2, st);
    hipLaunchKernelGGL(q8k_quant_kernel<true>, dim3(npairs), dim3(256), 0, st, (const void*)ws->a_buf, I, ws->a_q, ws->a_d, ws->a_bs,
                       a_bsp, a_b256, ws->counters, 0, npairs);
  }
  KTX_HIP(hipGetLastError());
  GgGemmParams g2{};
  g2.w0 = h->down_w; g2.w1 = nullptr; g2.stride0 = h->gg_stride[2]; g2.stride1 = 0; g2.N = H; g2.K = I;
  g2.act_q = ws->a_q; g2.act_d = ws->a_d; g2.act_bs = ws->a_bs; g2.act_bsp = a_bsp; g2.act_b256 = a_b256; g2.row_src = nullptr; g2.tiles = ws->tiles;
  g2.counters = ws->counters; g2.out = reinterpret_cast<float*>(ws->dn_buf); g2.dbg = g_dbg[24];
  {
    ProfScope ps(3, st);
    rc = launch_gguf_type<false>(h->gg_type[2], mt, g2, max_tiles, st);
  }
  if (rc) return rc;
  CombineParams cp{};
  cp.d_bsz = d_bsz; cp.qlen = qlen; cp.k = k; cp.H = H; cp.dn = ws->dn_buf; cp.row_of_pair = ws->row_of_pair;
  cp.weights = d_weights; cp.y = (bf16_t*)d_output; cp.incremental = (flags & KTX_FWD_INCREMENTAL) ? 1 : 0;
  cp.partial_f32 = (flags & KTX_FWD_PARTIAL_F32) ? 1 : 0; cp.dn_f32 = 1;
  {
    ProfScope ps(4, st);
    hipLaunchKernelGGL(moe_combine_kernel, dim3((H / 4 + 255) / 256, qlen), dim3(256), 0, st, cp);
  }
  KTX_HIP(hipGetLastError());
  return 0;
}

