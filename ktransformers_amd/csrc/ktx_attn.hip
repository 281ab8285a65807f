// ktx_attn.hip — the attention half of a DeepSeek-V3 decoder layer for ONE decode token as ONE persistent launch (C ABI:
// include/ktx_attn.h).  Reference chain: archive/ktransformers/operators/attention.py:349-523 (forward_linux_flashinfer) with
// modeling_deepseek_v3.py:1200-1219 around it.
//
// Why.  As five dependent launches (lin_sk_kernel q_a|kv_a, lin_qb_absorb_kernel, mla_decode_kernel, lin_merge_unabsorb_kernel,
// lin_sk_kernel o_proj) this chain takes 64 us of a 122 us layer for 130 MB of weights (21 us at 6.2 TB/s): every launch pays
// the boundary, a first-touch round trip for its input, a ramp of its weight stream and a tail, and two of them run on 128
// workgroups — half the chip's CUs, each CU pulling ~24 GB/s whatever runs on it.  Here the five stages are
// PHASES of one launch of 2 x heads = 256 workgroups, one per CU:
//   A  input RMSNorm + q_a|kv_a GEMV                      one 16-row strip per workgroup (132 of them)
//   B  q_a_layernorm + q_b rows of a head + RoPE + absorb  TWO workgroups per head, each produces half of the absorbed row.  W4 (round 6):
//      + (one workgroup) kv_a_layernorm, k_pe RoPE, cache append    each computes all of the head's q_nope rows itself (part 1 the rope rows
//                                                          too) — no exchange; FP8: each half of the q_b rows, q_nope pieces swapped
//   C  split-KV attention over the paged latent cache      (32 heads) x (KV split) per workgroup, mla_decode_kernel<2,4>'s tile loop
//   D  merge of the splits + un-absorb                     two workgroups per head: each merges all 512 dims (round 6: no exchange),
//                                                          each produces half of the head's v_dim outputs
//   E  o_proj + residual                                   whole strips dealt to all workgroups
// A phase's weights are requested BEFORE the workgroup starts waiting for the phase's input (they depend on nothing), so the
// weight stream runs across the hand-offs; a phase's input arrives through the device workspace.  The small rows (A -> B, B -> C)
// travel as tagged 8-byte granules {two bf16, tag = the launch's epoch}: the data is the flag, the consumer's lanes re-read the
// granules they stage until every tag matches (round 6).  The large ones (C -> D: fp32 partials, D -> E: every workgroup reads every
// head's row) keep payload + flag: the producer stores write-through (sc1), drains, and sets a per-workgroup flag to the epoch;
// consumers poll exactly the flags they depend on (one wavefront, relaxed device-scope loads) and read the payload with sc1 loads
// (MI355X_MICROARCH.md §inter-workgroup visibility: sc1 stores + sc1 loads need no fence).  The epoch lives in the workspace and is
// advanced by the last workgroup to leave, so a captured graph replays with fresh tags and flags and nothing has to be zeroed.
//
// Arithmetic.  Every phase restates its stand-alone kernel: same k-steps (ktx_w4_step.inc), same fixed summation orders, same
// roundings, same KV split rule (ktx_mla_decode_nsplit) — tests/test_attn_fused_gpu.py holds the outputs of all five phases
// bit-identical to the five-launch path.  Every poll is bounded: a hand-off that does not arrive within KTX_ATTN_SPIN_TICKS sets the
// workspace status word and the launch runs to its end with undefined results instead of hanging the GPU.
#include "ktx_common.h"

#include <cstring>
#include <mutex>
#include <type_traits>

#include "../../include/ktx_attn.h"
#include "../../include/ktx_gate.h"
#include "../../include/ktx_mla.h"
#include "ktx_internal.h"
#include "ktx_prep.inc"

extern "C" int ktx_debug_get(int idx);   // ktx_moe.hip (include/ktx_moe.h)

namespace {

#include "ktx_w4_step.inc"
#include "ktx_gate_dev.inc"   // the router's selection (phase F)

typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef __bf16 av8bf __attribute__((ext_vector_type(8)));
typedef short av4s16 __attribute__((ext_vector_type(4)));

constexpr int NT = 512;                 // threads per workgroup (8 wavefronts)
constexpr int DA = 7;                   // phase A: k-steps per wavefront (hidden = 8 * 7 * 128)
constexpr int NK2 = 6;                  // phase B: k-steps per k-half of q_b (q_lora = 2 * 6 * 128)
constexpr int NOPE = 128, ROPE = 64, LORA = 512, VDIM = 128, QW = NOPE + ROPE;
constexpr int SPH = QW / 16;            // q_b strips per head (12)
constexpr int GRID = 256;               // workgroups of the launch: one per CU, all resident
constexpr int MAXS = 128;               // KV splits at most (64 heads: 2 head groups x 128; 128 heads: 4 x 64)
constexpr int KROW = LORA + ROPE + 8;   // staged latent row: 584 elements (ktx_mla.hip)
constexpr int TILE = 32;
constexpr unsigned long long SPIN_TICKS = 20000000ull;   // 0.2 s of the 100 MHz wall clock

// ---- workspace (one per device) -------------------------------------------------------------------------------------------
// header words
constexpr int W_EPOCH = 0, W_EXIT = 1, W_STATUS = 2, W_TICKET = 3;
struct WsLayout {   // byte offsets from the workspace base
  unsigned fKV, fX, fC, fD, fE;                         // flag arrays (u32 each)
  unsigned gran;                                        // phase F: {epoch, logit} granules (u64 each)
  unsigned qkv, ckv_new, kpe_new, qx, q_lat, q_pe, attn_out, part_ml, part_o;
  unsigned total;
};
__host__ __device__ inline WsLayout ws_layout(int H, int nA) {
  WsLayout L;
  unsigned o = 256;
  auto take = [&](unsigned bytes) { const unsigned r = o; o += (bytes + 255u) & ~255u; return r; };
  const int NWG = GRID, SPG = GRID / (H / 32);   // splits per head group
  L.fKV = take(4); L.fX = take(4 * NWG); L.fC = take(4 * NWG); L.fD = take(4 * NWG); L.fE = take(4 * NWG); L.gran = take(8 * NWG);
  L.qkv = take(4 * 16 * nA); L.ckv_new = take(2 * LORA); L.kpe_new = take(2 * ROPE);   // (qkv, q_lat, q_pe: granules, 4 bytes per bf16)
  L.qx = take(2 * H * NOPE); L.q_lat = take(4 * H * LORA); L.q_pe = take(4 * H * ROPE);
  L.attn_out = take(2 * H * VDIM); L.part_ml = take(4 * H * SPG * 2); L.part_o = take(4u * H * SPG * LORA);
  L.total = o;
  return L;
}

struct AttnParams {
  // phase A
  const uint8_t* wA; const bf16_t* scA; int nksA, nA;
  const bf16_t* x; const bf16_t* in_norm_w; float in_eps; int hidden;
  // phase B
  const uint8_t* wB; const bf16_t* scB; int nksB;
  const uint8_t* wUK; size_t wbsUK;
  const bf16_t* qa_norm_w; float qa_eps; int q_lora;
  const bf16_t* kv_norm_w; float kv_eps;
  const int64_t* pos; const float* inv_freq; float mscale;
  int H;
  // phase C
  bf16_t* ckv; bf16_t* kpe; long long ckv_ts, kpe_ts;
  const int32_t *kv_indptr, *kv_indices, *kv_len;
  int page_size, nsplit; float sm_scale;
  // phase D
  const uint8_t* wUV; size_t wbsUV;
  // phase E
  const uint8_t* wE; const bf16_t* scE; int nksE, nksE_sh, nE, eQ, eR;
  bf16_t* y;
  // workspace
  uint8_t* ws; unsigned ws_bytes;
  unsigned* hstatus;   // host-mapped copy of the status word (pinned): the host reads it every step without touching the device
  int last;
  unsigned long long* stamps;
};

#define AT_STAMP(i) do { if (p.stamps && blockIdx.x == 0 && threadIdx.x == 0) p.stamps[i] = wall_clock64(); } while (0)
#define AT_STAMP7(i) do { if (p.stamps && blockIdx.x == 0 && threadIdx.x == 448) p.stamps[i] = wall_clock64(); } while (0)

// ---- hand-off primitives -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ld_word(const unsigned* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_word(unsigned* q, unsigned v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ws_rsrc(const AttnParams& p) {
  return __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, (int)p.ws_bytes, 0x00020000);
}
// write-through (sc1) 16-byte store / sc1 16-byte load at a byte offset of the workspace
__device__ __forceinline__ void ws_store16(__amdgpu_buffer_rsrc_t r, unsigned off, const uint4& v) {
  const u4v d = {v.x, v.y, v.z, v.w};
  __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)off, 0, 16);
}
__device__ __forceinline__ uint4 ws_load16(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const u4v d = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16);
  return make_uint4(d.x, d.y, d.z, d.w);
}
// every store of this wavefront has left the CU (write-through stores are visible device-wide once acknowledged)
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ONE wavefront waits until flags[idx(k * 64 + lane)] == epoch for every k * 64 + lane < n.  Bounded; a timeout (or another
// workgroup's earlier timeout) sets / sees the status word and returns.
template <class IDX>
__device__ __forceinline__ void poll_flags(const AttnParams& p, const unsigned* flags, int n, unsigned epoch, int code, IDX idx) {
  const int lane = threadIdx.x & 63;
  unsigned* hdr = reinterpret_cast<unsigned*>(p.ws);
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    bool ok = true;
    for (int k = lane; k < n; k += 64) ok = ok && ld_word(flags + idx(k)) == epoch;
    if (__all(ok)) return;
    if (ld_word(hdr + W_STATUS) != 0) return;
    if (wall_clock64() - t0 > SPIN_TICKS) {
      if (lane == 0) {
        st_word(hdr + W_STATUS, (unsigned)code);
        if (p.hstatus) __hip_atomic_store(p.hstatus, (unsigned)code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

// ---- tagged granules (cdna_hip_programming.md Guideline 16, form R2): the data IS the flag ----------------------------------------
// A small hand-off row travels as naturally aligned 8-byte {two bf16, tag = epoch} granules, two per 16-byte sc1 store; the consumer's
// lanes re-read exactly the granules they need until every tag carries this launch's epoch.  Against payload -> drain -> flag on
// the producer and flag poll -> payload load on the consumer this takes one memory round trip off each side of the hop
// (MI355X_MICROARCH.md, rows handoff-1to1 vs handoff-flag: 2.3 vs 3.8 us between streaming CUs).  Element e of a row sits at byte 4 e.
__device__ __forceinline__ void gran_store(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned v0, unsigned v1, unsigned epoch) {
  ws_store16(r, off, make_uint4(v0, epoch, v1, epoch));
}
// ONE wavefront: (re)loads its N 16-byte granule pairs until every tag == epoch; g[k] = {v0, tag, v1, tag}.  Bounded like poll_flags.
template <int N>
__device__ __forceinline__ void gran_sweep(const AttnParams& p, __amdgpu_buffer_rsrc_t r, const unsigned (&off)[N], const bool (&use)[N],
                                           unsigned epoch, int code, uint4 (&g)[N]) {
  unsigned* hdr = reinterpret_cast<unsigned*>(p.ws);
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < N; k++)
      if (use[k]) g[k] = ws_load16(r, off[k]);
#pragma unroll
    for (int k = 0; k < N; k++) ok = ok && (!use[k] || (g[k].y == epoch && g[k].w == epoch));
    if (__all(ok)) return;
    if (ld_word(hdr + W_STATUS) != 0) return;
    if (wall_clock64() - t0 > SPIN_TICKS) {
      if ((threadIdx.x & 63) == 0) {
        st_word(hdr + W_STATUS, (unsigned)code);
        if (p.hstatus) __hip_atomic_store(p.hstatus, (unsigned)code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
__device__ __forceinline__ uint4 gran_join(const uint4& a, const uint4& b) { return make_uint4(a.x, a.z, b.x, b.z); }

// ---- k-steps (restated from ktx_linear_sk.inc / ktx_linear.hip: same expressions, same order) ---------------------------------
// W4 g64, one token row: acc += s_g * ( sum_{k in g} x_k (128 + q_k) - 136 sum_{k in g} x_k )
__device__ __forceinline__ void w4_kstep1(const uint4& w, const uint2& sc, const uint8_t* xb, const float* aux, float& acc) {
  const uint32_t P[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int gi = 0; gi < 2; gi++) {
    v4f tmp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
      const int j = gi * 2 + jj;
      const uint4 xa = *reinterpret_cast<const uint4*>(xb + j * 4 * 16);
      tmp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w4_as_v8bf(xa), w4_as_v8bf(w4_frag(P[j])), tmp, 0, 0, 0);
    }
    const float s = w4_scale(sc, gi);
    acc = fmaf(s, fmaf(-136.f, aux[gi * 4], tmp[0]), acc);
  }
}
// Block-FP8 (e4m3 weights, fp32 scale per 128 x 128 block; activations quantised per 128-k block as act_quant does), one token row:
// lin_step<F_FP8>'s arithmetic (ktx_linear.hip) for row 0 — acc += dot_128 * a_s * b_s (fp8gemm.py:156)
__device__ __forceinline__ long at_lo64(const uint4& u) { return (long)(((uint64_t)u.y << 32) | u.x); }
__device__ __forceinline__ long at_hi64(const uint4& u) { return (long)(((uint64_t)u.w << 32) | u.z); }
__device__ __forceinline__ void fp8_kstep1(const uint4& w0, const uint4& w1, float bs, const uint8_t* xb, float as, float& acc) {
  const uint4 xa0 = *reinterpret_cast<const uint4*>(xb), xa1 = *reinterpret_cast<const uint4*>(xb + 16);
  v4f tmp = {0.f, 0.f, 0.f, 0.f};
  tmp = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(at_lo64(xa0), at_lo64(w0), tmp, 0, 0, 0);
  tmp = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(at_hi64(xa0), at_hi64(w0), tmp, 0, 0, 0);
  tmp = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(at_lo64(xa1), at_lo64(w1), tmp, 0, 0, 0);
  tmp = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(at_hi64(xa1), at_hi64(w1), tmp, 0, 0, 0);
  acc = fmaf(tmp[0] * as, bs, acc);
}
// act_quant of one 16-byte piece (8 bf16) of a 128-k block held by 16 consecutive lanes: returns the 8 e4m3 bytes, `s` = amax / 448
// of the block (lin_dec_body's stage_piece: same expressions)
__device__ __forceinline__ uint2 fp8_quant_piece(const uint4& v, float& s) {
  const uint32_t d[4] = {v.x, v.y, v.z, v.w};
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) am = fmaxf(am, fmaxf(fabsf(__uint_as_float(d[i] << 16)), fabsf(__uint_as_float(d[i] & 0xffff0000u))));
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
  s = am / 448.f;
  float f[8];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float a = __uint_as_float(d[i] << 16), b = __uint_as_float(d[i] & 0xffff0000u);
    f[2 * i] = s > 0.f ? a / s : 0.f;
    f[2 * i + 1] = s > 0.f ? b / s : 0.f;
  }
  uint32_t o0, o1;
  o0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
  o0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], o0, true);
  o1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
  o1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], o1, true);
  return make_uint2(o0, o1);
}
// BF16 tile of four 1 KiB planes, one k-step of 128: plane j contracts k = kc * 32 + j * 8 + e
__device__ __forceinline__ void bf16_kstep(const uint4 (&w)[4], const uint8_t* xb, v4f& acc) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint4 xa = *reinterpret_cast<const uint4*>(xb + j * 16);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w4_as_v8bf(xa), w4_as_v8bf(w[j]), acc, 0, 0, 0);
  }
}
__device__ __forceinline__ uint4 nt_load16(const uint8_t* q) {
  const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(q));
  return make_uint4(v.x, v.y, v.z, v.w);
}
// group sum of a staged 16-byte piece over the G / 8 = 8 consecutive lanes of its scale group (DPP row operations)
__device__ __forceinline__ float group_sum64(const uint4& v) {
  float s = sum8_bf16(v);
  s += ktx_dpp_f<KTX_DPP_QUAD_1032>(s);
  s += ktx_dpp_f<KTX_DPP_QUAD_2301>(s);
  s += ktx_dpp_f<KTX_DPP_ROW_HALF_MIRROR>(s);
  return s;
}
__device__ __forceinline__ float sumsq8(const uint4& v) {
  const uint32_t d4[4] = {v.x, v.y, v.z, v.w};
  float pq = 0.f;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const float a = __uint_as_float(d4[e] << 16), b = __uint_as_float(d4[e] & 0xffff0000u);
    pq += a * a + b * b;
  }
  return pq;
}
__device__ __forceinline__ uint4 norm8(const uint4& v, float r, const uint4& w) {
  return make_uint4(ktx_norm_pk(v.x, r, w.x), ktx_norm_pk(v.y, r, w.y), ktx_norm_pk(v.z, r, w.z), ktx_norm_pk(v.w, r, w.w));
}

// ---- MLA tile helpers (restated from ktx_mla.hip) ---------------------------------------------------------------------------------
__device__ __forceinline__ av8bf as_av8bf(const uint4& u) {
  union { uint4 u; av8bf v; } c;
  c.u = u;
  return c.v;
}
__device__ __forceinline__ av8bf load_v_frag(const bf16_t* base) {
  typedef __attribute__((address_space(3))) av4s16 lds_v4;
  const av4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(base));
  const av4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(base + 4 * KROW));
  union { av4s16 h[2]; av8bf v; } c;
  c.h[0] = lo;
  c.h[1] = hi;
  return c.v;
}
// (the LDS destination is passed as a BYTE ADDRESS computed from the dynamic region's base: a generic pointer whose provenance the
// compiler has lost needs a run-time address-space cast here, which this compiler mis-selects in some phase combinations)
__device__ __forceinline__ void dma_row(const bf16_t* gsrc_lane, uint32_t lds_byte_addr) {
  const uint32_t lds_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc_lane), "s"(lds_addr)
               : "memory");
}

// the row as a SCALAR base + a per-lane byte offset (saddr form): a staged ckv row is one wave-uniform pointer (ktx_mla.hip, round 6)
__device__ __forceinline__ void dma_row_s(const bf16_t* gsrc_row, uint32_t lane_byte_off, uint32_t lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(lane_byte_off), "s"(gsrc_row), "s"(lds_addr)
               : "memory");
}

// n items over m bins: bins < R take Q + 1 (ktx_linear_sk.inc)
__device__ __forceinline__ int split_begin(int Q, int R, int b) { return b * Q + (b < R ? b : R); }
__device__ __forceinline__ int wave_begin(int Gb, int n, int w) {
  w = w < 8 ? w : 8;
  const int qb = n >> 3, rb = n - qb * 8;
  return Gb + w * qb + (w < rb ? w : rb);
}

constexpr int PH_A = KTX_ATTN_PHASE_QKV_A, PH_B = KTX_ATTN_PHASE_QB, PH_C = KTX_ATTN_PHASE_MLA, PH_D = KTX_ATTN_PHASE_MERGE,
              PH_E = KTX_ATTN_PHASE_OPROJ;
// lin_glu of ktx_linear.hip (DeepseekV3MLP between the merged gate|up GEMV and down_proj, bf16 tensor arithmetic)
__device__ __forceinline__ bf16_t glu_bf16(float g, float u) {
  const float gb = bf16_to_f32(f32_to_bf16(g)), ub = bf16_to_f32(f32_to_bf16(u));
  const float sb = bf16_to_f32(f32_to_bf16(gb / (1.0f + expf(-gb))));
  return f32_to_bf16(sb * ub);
}

// =====================================================================================================================================
// FMT = the format of the three quantised projections (q_a|kv_a, q_b, o_proj): KTX_LIN_W4 (g64) or KTX_LIN_FP8 (128 x 128 blocks)
template <int MASK, int FMT>
__global__ __launch_bounds__(NT, 2) void attn_decode_kernel(AttnParams p) {
  static_assert(FMT == KTX_LIN_W4 || FMT == KTX_LIN_FP8, "W4 g64 or block-FP8 projections");
  constexpr bool F8 = FMT == KTX_LIN_FP8;
  constexpr int NQ = F8 ? 2 : 1, TB = NQ * 1024;   // 1 KiB planes per weight tile, bytes per tile
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t smem_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)smem;   // LDS byte address of the dynamic region
  const int w = blockIdx.x, H = p.H, NWG = GRID, NWB = 2 * H;
  // phases B and D: two workgroups per head — (head, half) = (w % H, w / H) for w < 2 H (H = 64 or 128: the other workgroups of a
  // 64-head model sit those two phases out and take part in A, C and E)
  const int h = w & (H - 1), part = w / H;
  const bool doBD = w < NWB;
  const int SPG = GRID / (H >> 5), SPG_SH = H == 128 ? 6 : 7;   // KV splits per head group of 32 heads (64 / 128), its log2
  unsigned* hdr = reinterpret_cast<unsigned*>(p.ws);
  const WsLayout L = ws_layout(H, p.nA);
  const __amdgpu_buffer_rsrc_t rs = ws_rsrc(p);
  unsigned* fKV = reinterpret_cast<unsigned*>(p.ws + L.fKV);
  unsigned* fX = reinterpret_cast<unsigned*>(p.ws + L.fX);
  unsigned* fC = reinterpret_cast<unsigned*>(p.ws + L.fC);
  unsigned* fD = reinterpret_cast<unsigned*>(p.ws + L.fD);
  unsigned* fE = reinterpret_cast<unsigned*>(p.ws + L.fE);
  AT_STAMP(0);
  AT_STAMP7(23);
  const unsigned epoch = ld_word(hdr + W_EPOCH);

  // =========================== requests that depend on nothing =====================================================================
  // ---- phase A: the layer input row + input_layernorm weights (needed first), then the strip's 7 tiles of this wavefront
  constexpr int XRA = 2;                                   // 16-byte pieces per thread: hidden <= 8192
  const int npieceA = p.hidden >> 3;
  const bool doA = (MASK & PH_A) && w < p.nA;
  uint4 xrA[XRA], nwA[XRA];
  uint4 wrA[DA][NQ];
  uint2 srA[DA];
  if constexpr ((MASK & PH_A) != 0) {
#pragma unroll
    for (int i = 0; i < XRA; i++) {
      const int pc = min(tid + i * NT, npieceA - 1);
      xrA[i] = *reinterpret_cast<const uint4*>(p.x + pc * 8);
      nwA[i] = *reinterpret_cast<const uint4*>(p.in_norm_w + pc * 8);
    }
    // The CU's memory pipe is a FIFO shared by its wavefronts: without this barrier a wavefront that runs ahead queues its 40 KiB of
    // weight requests in front of the others' input-row requests, and the row (needed first) arrives behind the whole burst.
    asm volatile("s_barrier" ::: "memory");
    const int sA = min(w, p.nA - 1);
    const long tile0 = (long)sA * p.nksA + wave * DA;
#pragma unroll
    for (int d = 0; d < DA; d++) {
#pragma unroll
      for (int q = 0; q < NQ; q++) wrA[d][q] = nt_load16(p.wA + (tile0 + d) * TB + q * 1024 + lane * 16);
      if constexpr (F8) srA[d] = make_uint2(__float_as_uint(reinterpret_cast<const float*>(p.scA)[(size_t)(sA >> 3) * p.nksA + wave * DA + d]), 0);
      else srA[d] = load_w4_scales<2>(p.scA + ((tile0 + d) * 16 + (lane & 15)) * 2);
    }
  }
  // ---- phase B: q_a_layernorm weights, rope table inputs, the head's q_b tiles (waves 0..5: one strip, both k-halves) and
  // absorb tiles (waves 0..5: one strip, waves 6..7: five strips)
  uint4 nwB = make_uint4(0, 0, 0, 0);
  float ropePos = 0.f, ropeIf = 0.f;
  // one register file for both wave roles: waves 0..5 hold rb[0..11] = their q_b strip (k-half kh, step s_ at kh * 6 + s_) and
  // rb[12..15] = their absorb strip; waves 6..7 hold rb[4 i .. 4 i + 3] = absorb strip i of their five
  // W4 (round 6): NO exchange between the halves — every workgroup of the pair computes ALL eight q_nope strips of its head (it needs
  // the whole q_nope for its 256 absorbed values), the part-1 workgroup the four rope strips too: 16 / 24 half-strips (strip, k-half)
  // dealt 2 / 3 per wavefront, then two absorb strips per wavefront.  rb[6 i .. 6 i + 5] = half-strip slot i, rb[18 + 4 i ..] = absorb
  // strip i.  12.6 MB more q_b bytes per layer (requested behind phase A's k-steps, while HBM idles) for one hand-off less.
  constexpr int ABS0 = F8 ? 2 * NK2 * NQ : 3 * NK2;   // first absorb register (F8: of waves 0..5, behind their q_b tiles)
  constexpr int NRB = F8 ? (ABS0 + 4 > 20 ? ABS0 + 4 : 20) : ABS0 + 8;
  uint4 rb[NRB];
  uint2 sb[F8 ? 2 * NK2 : 3 * NK2];
  const int nperB = part == 0 ? 2 : 3;   // W4: half-strips per wavefront
  auto prefetch_B = [&]() {
    nwB = *reinterpret_cast<const uint4*>(p.qa_norm_w + min(tid, (p.q_lora >> 3) - 1) * 8);
    ropePos = (float)p.pos[0];
    ropeIf = p.inv_freq[tid & (ROPE / 2 - 1)];
    if constexpr (!F8) {
#pragma unroll
      for (int i = 0; i < 3; i++) {
        if (i < nperB) {
          const int hsi = wave * nperB + i;
          const size_t t0 = ((size_t)h * SPH + (hsi >> 1)) * p.nksB + (size_t)(hsi & 1) * NK2;
          const uint8_t* wp = p.wB + t0 * TB + lane * 16;
          const bf16_t* sp = p.scB + (t0 * 16 + (lane & 15)) * 2;
#pragma unroll
          for (int s_ = 0; s_ < NK2; s_++) {
            rb[i * NK2 + s_] = nt_load16(wp + (size_t)s_ * TB);
            sb[i * NK2 + s_] = load_w4_scales<2>(sp + (size_t)s_ * 16 * 2);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const uint8_t* wp2 = p.wUK + (size_t)h * p.wbsUK + (size_t)(part * 16 + wave * 2 + i) * 4096 + lane * 16;
#pragma unroll
        for (int q = 0; q < 4; q++) rb[ABS0 + i * 4 + q] = nt_load16(wp2 + q * 1024);
      }
    } else if (wave < 6) {
      const size_t strip = (size_t)h * SPH + part * 6 + wave;
#pragma unroll
      for (int kh = 0; kh < 2; kh++) {
        const uint8_t* wp = p.wB + (strip * p.nksB + (size_t)kh * NK2) * TB + lane * 16;
        const bf16_t* sp = p.scB + ((strip * p.nksB + (size_t)kh * NK2) * 16 + (lane & 15)) * 2;
        const float* sp8 = reinterpret_cast<const float*>(p.scB) + (strip >> 3) * p.nksB + kh * NK2;
#pragma unroll
        for (int s_ = 0; s_ < NK2; s_++) {
#pragma unroll
          for (int q = 0; q < NQ; q++) rb[(kh * NK2 + s_) * NQ + q] = nt_load16(wp + (size_t)s_ * TB + q * 1024);
          if constexpr (F8) sb[kh * NK2 + s_] = make_uint2(__float_as_uint(sp8[s_]), 0);
          else sb[kh * NK2 + s_] = load_w4_scales<2>(sp + (size_t)s_ * 16 * 2);
        }
      }
      const uint8_t* wp2 = p.wUK + (size_t)h * p.wbsUK + (size_t)(part * 16 + wave) * 4096 + lane * 16;
#pragma unroll
      for (int q = 0; q < 4; q++) rb[ABS0 + q] = nt_load16(wp2 + q * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < 5; i++) {
        const uint8_t* wp2 = p.wUK + (size_t)h * p.wbsUK + (size_t)(part * 16 + 6 + (wave - 6) * 5 + i) * 4096 + lane * 16;
#pragma unroll
        for (int q = 0; q < 4; q++) rb[i * 4 + q] = nt_load16(wp2 + q * 1024);
      }
    }
  };
  // Phase B's 136 KiB are requested behind phase A's k-steps (workgroups with a strip of phase A): issued first they cost the
  // wavefront ~2 us of instruction issue in front of its first wait, and their wave-role branches make the compiler drain the
  // whole queue (vmcnt(0)) at the first consumer behind them (measured with the stamps: the input row was staged 6.4 us after
  // entry with the requests in front, 2.9 us behind; phase A's k-steps ended at 7.0 us with the requests in front of them).
  if constexpr ((MASK & PH_B) != 0) {
    if (!doA && doBD) {
      // (workgroups without a strip of phase A: their 136 KiB each would compete chip-wide with the 7 MB of phase A tiles the step
      // is waiting for — measured: those tiles took 6 us to arrive — so they start ~1.5 us late)
      // (3.4 and 5.4 us were measured in round 6 as well: phase A ends 1.4 us earlier, phase C starts at the same time)
      if constexpr ((MASK & PH_A) != 0) __builtin_amdgcn_s_sleep(56);
      prefetch_B();
    }
  }
  AT_STAMP(18);
  AT_STAMP7(25);

  // =========================== phase A ===================================================================================
  if constexpr ((MASK & PH_A) != 0) {
    if (doA) {
      uint8_t* xs = smem;                                                       // [npiece][16 B]
      float* aux = reinterpret_cast<float*>(smem + (size_t)p.nksA * 16 * 16);   // [nksA * 2][4]
      float* nred = aux + (size_t)p.nksA * 2 * 4;                               // [8][4]
      float* table = nred + 32;                                                 // [8][64]
      bf16_t* ostage = reinterpret_cast<bf16_t*>(table + 8 * 64);               // [16]
#pragma unroll
      for (int i = 0; i < XRA; i++)
        if (!(tid + i * NT < npieceA)) xrA[i] = make_uint4(0, 0, 0, 0);
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < XRA; i++) ss += sumsq8(xrA[i]);
      const float wsum = wave_sum(ss);
      if (lane == 0) nred[wave * 4] = wsum;
      AT_STAMP(16);
      AT_STAMP7(24);
      __syncthreads();
      AT_STAMP(17);
      float tot = nred[0];
#pragma unroll
      for (int v = 1; v < 8; v++) tot += nred[v * 4];
      const float rnorm = 1.0f / sqrtf(tot / (float)p.hidden + p.in_eps);
#pragma unroll
      for (int i = 0; i < XRA; i++) {
        const int pc = tid + i * NT;
        if (pc < p.nksA * 16) {
          const uint4 v = norm8(xrA[i], rnorm, nwA[i]);
          if constexpr (F8) {   // act_quant per 128-k block: 8 e4m3 bytes per piece, the block's scale in aux[k-step]
            float sq;
            const uint2 q8 = fp8_quant_piece(v, sq);
            *reinterpret_cast<uint2*>(xs + (size_t)pc * 8) = q8;
            if ((pc & 15) == 0) aux[pc >> 4] = sq;
          } else {
            *reinterpret_cast<uint4*>(xs + (size_t)pc * 16) = v;
            const float s = group_sum64(v);
            if ((pc & 7) == 0) {
#pragma unroll
              for (int r = 0; r < 4; r++) aux[(pc >> 3) * 4 + r] = s;
            }
          }
        }
      }
      __syncthreads();
      AT_STAMP(1);
      const int kc = lane >> 4;
      const uint8_t* xb0 = xs + kc * (F8 ? 32 : 16);
      {   // (the host admits at most one strip per workgroup: nA <= the grid)
        const int s = w;
        float acc = 0.f;
        const int ks0 = wave * DA;
#pragma unroll
        for (int d = 0; d < DA; d++) {
          if constexpr (F8) fp8_kstep1(wrA[d][0], wrA[d][NQ - 1], __uint_as_float(srA[d].x), xb0 + (size_t)(ks0 + d) * 128, aux[ks0 + d], acc);
          else w4_kstep1(wrA[d][0], srA[d], xb0 + (size_t)(ks0 + d) * 256, aux + (ks0 + d) * 8, acc);
        }
        // phase B's requests go out HERE: nothing older is pending any more, so the wave-role branches inside (whose request counts
        // the compiler cannot line up: it drains the queue at their join) cost nothing, and the replies fly during the hand-off
        if constexpr ((MASK & PH_B) != 0) { if (doBD) prefetch_B(); }
        if (lane < 16) table[wave * 64 + lane] = acc;
        AT_STAMP(19);
        AT_STAMP7(26);
        __syncthreads();
        if (wave == 0) {
          if (lane < 16) {
            float v = 0.f;
#pragma unroll
            for (int u = 0; u < 8; u++) v += table[u * 64 + lane];
            ostage[lane] = f32_to_bf16(v);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (lane < 4) {
            const uint2 v = *reinterpret_cast<const uint2*>(ostage + lane * 4);
            gran_store(rs, L.qkv + s * 64 + lane * 16, v.x, v.y, epoch);
          }
        }
      }
    }
    AT_STAMP(2);
  }

  // =========================== phase B ===================================================================================
  // LDS of phase B (the phase A region is dead for this workgroup once its strip is published; a barrier separates them)
  if constexpr ((MASK & PH_B) != 0) {
   if (doBD) {
    __syncthreads();
    bf16_t* kvraw = reinterpret_cast<bf16_t*>(smem);                              // [576]
    uint8_t* xsB = smem + 1280;                                                   // [q_lora / 8][16 B]
    float* auxB = reinterpret_cast<float*>(xsB + (size_t)(p.q_lora >> 3) * 16);   // [nksB * 2][4]
    float* nredB = auxB + p.nksB * 2 * 4;                                         // [8][4]
    float* red1 = nredB + 32;                                                     // [12][2][16] (F8: [6][2][16])
    float* s_cs = red1 + 12 * 2 * 16;                                             // [64]: cos | sin
    bf16_t* qh = reinterpret_cast<bf16_t*>(s_cs + ROPE);                          // [192] the head's q_b outputs (F8: [96] this half's)
    bf16_t* stage = qh + QW;                                                      // [256] publication staging
    uint8_t* xs2 = reinterpret_cast<uint8_t*>(stage + 256);                       // [16][16 B] the head's q_nope
    float* s_redK = reinterpret_cast<float*>(xs2 + 256);                          // [8] (kv prep)
    bf16_t* qpe_st = reinterpret_cast<bf16_t*>(s_redK + 8);                       // [64] rotated q_pe (W4)

    // ---- the phase A output row [q_a | ckv | k_pe] (granules: every lane waits for the 8 values it stages): q_a pieces -> RMSNorm ->
    // staging; kv pieces -> LDS
    const int npq = p.q_lora >> 3, npall = p.nA * 2;
    uint4 xp = make_uint4(0, 0, 0, 0);
    {
      uint4 g[2];
      const unsigned off[2] = {L.qkv + (unsigned)tid * 32, L.qkv + (unsigned)tid * 32 + 16};
      const bool use[2] = {tid < npall, tid < npall};
      gran_sweep<2>(p, rs, off, use, epoch, 0xA1, g);
      if (tid < npall) xp = gran_join(g[0], g[1]);
    }
    AT_STAMP(3);
    if (tid >= npq && tid < npall) *reinterpret_cast<uint4*>(kvraw + (tid - npq) * 8) = xp;
    {
      const float q = tid < npq ? sumsq8(xp) : 0.f;
      const float wsum = wave_sum(q);
      if (lane == 0) nredB[wave * 4] = wsum;
    }
    if (tid < ROPE / 2) {   // cos / sin of the token's position (mla_prep's table: bf16-rounded, times mscale)
      const float fr = ropePos * ropeIf;
      s_cs[tid] = prep_rbf(cosf(fr) * p.mscale);
      s_cs[ROPE / 2 + tid] = prep_rbf(sinf(fr) * p.mscale);
    }
    __syncthreads();
    {
      float tot = 0.f;
      for (int v = 0; v < 8; v++) tot += nredB[v * 4];
      const float rn = 1.0f / sqrtf(tot / (float)p.q_lora + p.qa_eps);
      if (tid < p.nksB * 16) {
        uint4 v = xp;
        if (tid < npq) v = norm8(xp, rn, nwB);
        else v = make_uint4(0, 0, 0, 0);
        if constexpr (F8) {
          float sq;
          const uint2 q8 = fp8_quant_piece(v, sq);
          *reinterpret_cast<uint2*>(xsB + (size_t)tid * 8) = q8;
          if ((tid & 15) == 0) auxB[tid >> 4] = sq;
        } else {
          *reinterpret_cast<uint4*>(xsB + (size_t)tid * 16) = v;
          const float sm = group_sum64(v);
          if ((tid & 7) == 0) auxB[(tid >> 3) * 4] = sm;
        }
      }
    }
    // ---- one workgroup: kv_a_layernorm + k_pe RoPE (mla_prep_token_block's arithmetic) -> workspace + cache append
    if (w == NWB - 1) {
      float v8[8];
      float ss = 0.f;
      if (tid < LORA / 8) {
        const uint4 raw = *reinterpret_cast<const uint4*>(kvraw + tid * 8);
        const uint32_t d[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int i = 0; i < 4; i++) { v8[2 * i] = __uint_as_float(d[i] << 16); v8[2 * i + 1] = __uint_as_float(d[i] & 0xffff0000u); }
#pragma unroll
        for (int e = 0; e < 8; e++) ss += v8[e] * v8[e];
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
      if (lane == 0) s_redK[wave] = ss;
      __syncthreads();
      float tot = 0.f;
      for (int v = 0; v < 8; v++) tot += s_redK[v];
      // cache row of the new token (StaticCache.update, custom_cache.py:189-195): position kv_len - 1, clamped to the owned pages
      int kl = p.kv_len[0];
      const int pb = p.kv_indptr[0];
      kl = min(kl, (p.kv_indptr[1] - pb) * p.page_size);
      const int app = kl - 1;
      size_t trow = 0;
      if (app >= 0) {
        const int page = p.kv_indices ? p.kv_indices[pb + app / p.page_size] : pb + app / p.page_size;
        trow = (size_t)page * p.page_size + app % p.page_size;
      }
      if (tid < LORA / 8) {
        const float r = 1.0f / sqrtf(tot / (float)LORA + p.kv_eps);
        const uint4 wr = *reinterpret_cast<const uint4*>(p.kv_norm_w + tid * 8);
        const uint32_t wd[4] = {wr.x, wr.y, wr.z, wr.w};
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const float a = __uint_as_float(wd[i] << 16) * prep_rbf(v8[2 * i] * r), b = __uint_as_float(wd[i] & 0xffff0000u) * prep_rbf(v8[2 * i + 1] * r);
          o[i] = prep_rne_bf16(a) | (prep_rne_bf16(b) << 16);
        }
        const uint4 ov = make_uint4(o[0], o[1], o[2], o[3]);
        ws_store16(rs, L.ckv_new + tid * 16, ov);
        if (app >= 0) *reinterpret_cast<uint4*>(p.ckv + trow * p.ckv_ts + tid * 8) = ov;
      }
      if (tid < ROPE / 2) prep_rope_pair(kvraw + LORA, stage, tid, ROPE / 2, s_cs[tid], s_cs[ROPE / 2 + tid]);
      __syncthreads();
      if (tid < ROPE / 8) {
        const uint4 ov = *reinterpret_cast<const uint4*>(stage + tid * 8);
        ws_store16(rs, L.kpe_new + tid * 16, ov);
        if (app >= 0) *reinterpret_cast<uint4*>(p.kpe + trow * p.kpe_ts + tid * 8) = ov;
      }
      drain_stores();
      __syncthreads();
      if (tid == 0) st_word(fKV, epoch);
    }
    __syncthreads();
    AT_STAMP(4);
    // ---- q_b rows of this half: waves 0..5 = one strip each, two k-halves summed in order (lin_qb_absorb_kernel)
    const int kc = lane >> 4;
    if constexpr (!F8) {
      // ---- W4: the head's q_b strips without an exchange (see the register arrays above); each half-strip is lin_qb_absorb_kernel's
      // chain of six k-steps, the two k-halves of a strip summed in its order
      const uint8_t* xb0 = xsB + kc * 16;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        if (i < nperB) {
          const int hsi = wave * nperB + i, kh = hsi & 1;
          float acc = 0.f;
#pragma unroll
          for (int s_ = 0; s_ < NK2; s_++) {
            const int ks = kh * NK2 + s_;
            w4_kstep1(rb[i * NK2 + s_], sb[i * NK2 + s_], xb0 + (size_t)ks * 256, auxB + ks * 8, acc);
          }
          if (lane < 16) red1[hsi * 16 + lane] = acc;   // [(strip * 2 + kh) * 16 + row]
        }
      }
      AT_STAMP(20);
      __syncthreads();
      AT_STAMP(21);
      if (tid < (part == 0 ? NOPE : QW)) {
        const int sih = tid >> 4, f = tid & 15;
        float v = 0.f;
        v += red1[(sih * 2 + 0) * 16 + f];
        v += red1[(sih * 2 + 1) * 16 + f];
        qh[tid] = f32_to_bf16(v);
      }
      __syncthreads();
      if (part == 1 && tid < ROPE / 2) prep_rope_pair(qh + NOPE, qpe_st, tid, ROPE / 2, s_cs[tid], s_cs[ROPE / 2 + tid]);
      AT_STAMP(22);
      AT_STAMP(27);
      AT_STAMP(5);
      {   // absorb: this half's 16 strips of W_UK[h]^T q_nope, two per wavefront, one k-step of 128 each (q_nope = qh[0, 128))
        const uint8_t* xb2 = reinterpret_cast<const uint8_t*>(qh) + kc * 4 * 16;
#pragma unroll
        for (int i = 0; i < 2; i++) {
          v4f acc = {0.f, 0.f, 0.f, 0.f};
          const uint4 wt[4] = {rb[ABS0 + i * 4], rb[ABS0 + i * 4 + 1], rb[ABS0 + i * 4 + 2], rb[ABS0 + i * 4 + 3]};
          bf16_kstep(wt, xb2, acc);
          if (lane < 16) stage[(wave * 2 + i) * 16 + lane] = f32_to_bf16(0.f + acc[0]);
        }
      }
      __syncthreads();
      if (part == 1 && tid >= 64 && tid < 64 + ROPE / 4) {
        const int t = tid - 64;
        const uint2 v = *reinterpret_cast<const uint2*>(qpe_st + t * 4);
        gran_store(rs, L.q_pe + (unsigned)h * ROPE * 4 + t * 16, v.x, v.y, epoch);
      }
    } else {
    if (wave < 6) {
      if constexpr (F8) {   // lin_dec_kernel<FP8> runs q_b's 12 k-steps as ONE k-slice per strip (8 strips per workgroup): one chain
        const uint8_t* xb0 = xsB + kc * 32;
        float acc = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2 * NK2; ks++)
          fp8_kstep1(rb[ks * NQ], rb[ks * NQ + NQ - 1], __uint_as_float(sb[ks].x), xb0 + (size_t)ks * 128, auxB[ks], acc);
        if (lane < 16) { red1[(wave * 2 + 0) * 16 + lane] = acc; red1[(wave * 2 + 1) * 16 + lane] = 0.f; }
      } else {
        const uint8_t* xb0 = xsB + kc * 16;
#pragma unroll
        for (int kh = 0; kh < 2; kh++) {
          float acc = 0.f;
#pragma unroll
          for (int s_ = 0; s_ < NK2; s_++) {
            const int ks = kh * NK2 + s_;
            w4_kstep1(rb[ks], sb[ks], xb0 + (size_t)ks * 256, auxB + ks * 8, acc);
          }
          if (lane < 16) red1[(wave * 2 + kh) * 16 + lane] = acc;
        }
      }
    }
    AT_STAMP(20);
    __syncthreads();
    AT_STAMP(21);
    if (tid < 96) {
      const int sih = tid >> 4, f = tid & 15;
      float v = 0.f;
      v += red1[(sih * 2 + 0) * 16 + f];
      v += red1[(sih * 2 + 1) * 16 + f];
      qh[tid] = f32_to_bf16(v);
    }
    __syncthreads();
    // this half's q values: part 0 = q_nope[0, 96); part 1 = q_nope[96, 128) | q_pe[0, 64)
    if (part == 1 && tid < ROPE / 2) prep_rope_pair(qh + 32, stage, tid, ROPE / 2, s_cs[tid], s_cs[ROPE / 2 + tid]);
    {
      const int npc = part == 0 ? 12 : 4;   // 16-byte pieces of q_nope this half owns
      if (tid < npc) ws_store16(rs, L.qx + (h * NOPE + part * 96) * 2 + tid * 16, *reinterpret_cast<const uint4*>(qh + tid * 8));
    }
    __syncthreads();
    if (part == 1 && tid < ROPE / 4) {
      const uint2 v = *reinterpret_cast<const uint2*>(stage + tid * 4);
      gran_store(rs, L.q_pe + (unsigned)h * ROPE * 4 + tid * 16, v.x, v.y, epoch);
    }
    drain_stores();
    __syncthreads();
    if (tid == 0) st_word(fX + w, epoch);
    AT_STAMP(22);
    const int partner = h + (1 - part) * H;
    if (wave == 7) poll_flags(p, fX, 1, epoch, 0xB1, [partner](int) { return partner; });
    __syncthreads();
    AT_STAMP(27);
    if (tid < 16) *reinterpret_cast<uint4*>(xs2 + tid * 16) = ws_load16(rs, L.qx + h * NOPE * 2 + tid * 16);
    __syncthreads();
    AT_STAMP(5);
    // ---- absorb: this half's 16 strips of W_UK[h]^T q_nope, one k-step of 128 each
    {
      const uint8_t* xb2 = xs2 + kc * 4 * 16;
      if (wave < 6) {
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        const uint4 wt[4] = {rb[ABS0], rb[ABS0 + 1], rb[ABS0 + 2], rb[ABS0 + 3]};
        bf16_kstep(wt, xb2, acc);
        if (lane < 16) stage[wave * 16 + lane] = f32_to_bf16(0.f + acc[0]);
      } else {
#pragma unroll
        for (int i = 0; i < 5; i++) {
          v4f acc = {0.f, 0.f, 0.f, 0.f};
          const uint4 wt[4] = {rb[i * 4], rb[i * 4 + 1], rb[i * 4 + 2], rb[i * 4 + 3]};
          bf16_kstep(wt, xb2, acc);
          const int a = 6 + (wave - 6) * 5 + i;
          if (lane < 16) stage[a * 16 + lane] = f32_to_bf16(0.f + acc[0]);
        }
      }
    }
    __syncthreads();
    }   // (F8: the exchanging form)
    if (tid < 64) {   // this half's 256 absorbed values: 128 granules (no drain, no flag: phase C's lanes wait on the tags)
      const uint2 v = *reinterpret_cast<const uint2*>(stage + tid * 4);
      gran_store(rs, L.q_lat + (unsigned)(h * LORA + part * 256) * 4 + tid * 16, v.x, v.y, epoch);
    }
   }
    AT_STAMP(6);
  }

  // =========================== phase C: split-KV attention ===========================================================================
  // workgroup (hg, split): hg = w / 64 (32 heads), split = w % 64 < nsplit; XCD = w % 8 is the same for the four head groups of a split
  const int hg = w >> SPG_SH, split = w & (SPG - 1);
  const bool doC = (MASK & PH_C) && split < p.nsplit && hg < H / 32;
  // phase D's weights (waves 0..3: one strip of W_UV[h], 4 k-steps x 4 planes) are requested before phase C starts waiting
  // (measured, profiles/r04_a_*: spreading these 64 KiB over the KV tiles of phase C — one k-step per tile, behind the tile's own
  // requests — made the q poll 1.2 us faster and the tile loop 2.4 us slower: every tile's wait is a wait for ALL of the wavefront's
  // requests.  One burst in front of the q poll it is.  Round 6: the whole burst BEHIND the q rows' arrival, the first tile's wait counted:
  // 54.6-54.8 against 54.1-54.6 us per layer — the second tile's requests then queue behind it.)
  uint4 wrD[4][4];
  const uint8_t* wpD = p.wUV + (size_t)h * p.wbsUV + (size_t)(part * 4 + (wave & 3)) * 4 * 4096 + lane * 16;
#define KTX_PF_D(KS)                                                                                   \
  do {                                                                                                 \
    if (wave < 4 && doBD) {                                                                            \
      _Pragma("unroll") for (int q = 0; q < 4; q++) wrD[KS][q] = nt_load16(wpD + (KS) * 4096 + q * 1024); \
    }                                                                                                  \
  } while (0)
  if constexpr ((MASK & PH_D) != 0) { KTX_PF_D(0); KTX_PF_D(1); KTX_PF_D(2); KTX_PF_D(3); }
  if constexpr ((MASK & PH_C) != 0) {
    __syncthreads();
    if (doC) {
      constexpr int HBW = 2, DSPLIT = 4, NWV = 8, NDT = 32 / DSPLIT, NQ = (18 + DSPLIT - 1) / DSPLIT;
      bf16_t* Kt = reinterpret_cast<bf16_t*>(smem);                        // [2][32][584]
      bf16_t* Kp = Kt + 2 * TILE * KROW;                                   // [2][32][64]
      bf16_t* Pt = Kp + 2 * TILE * ROPE;                                   // [8][16][32]
      float* Sx = reinterpret_cast<float*>(Pt + NWV * 16 * TILE);          // [8][8][64]
      const int hbw = wave / DSPLIT, ds = wave % DSPLIT;
      const int head0 = (hg * HBW + hbw) * 16;
      int kl = p.kv_len[0];
      const int page_base = p.kv_indptr[0];
      kl = min(kl, (p.kv_indptr[1] - page_base) * p.page_size);
      const int kv_end = max(kl, 0), app_pos = kl - 1;
      const int ntiles = (kv_end + TILE - 1) / TILE;
      const int t_begin = split, t_end = ntiles, t_step = p.nsplit;
      const bf16_t* app_ckv = reinterpret_cast<const bf16_t*>(p.ws + L.ckv_new);
      const bf16_t* app_kpe = reinterpret_cast<const bf16_t*>(p.ws + L.kpe_new);

      v4f o[NDT];
#pragma unroll
      for (int i = 0; i < NDT; i++) o[i] = v4f{0.f, 0.f, 0.f, 0.f};
      float m_run[4], l_run[4];
#pragma unroll
      for (int r = 0; r < 4; r++) { m_run[r] = -__builtin_inff(); l_run[r] = 0.f; }

      auto stage_tile = [&](int tile, int buf) {   // buf = which of the two staged tiles
        const uint32_t dK = smem_lds + (uint32_t)buf * (TILE * KROW * 2);
        const uint32_t dP = smem_lds + (uint32_t)(2 * TILE * KROW * 2) + (uint32_t)buf * (TILE * ROPE * 2);
        const int tok0 = tile * TILE;
        const int pidx_ = __builtin_amdgcn_readfirstlane(page_base + tok0 / p.page_size);
        const int page0 = __builtin_amdgcn_readfirstlane(p.kv_indices ? p.kv_indices[pidx_] : pidx_);
        const size_t row0 = (size_t)page0 * p.page_size + tok0 % p.page_size;
        const int last = kv_end - 1 - tok0;
        // a whole tile without the newest row: one 64-bit product for the tile, then adds, on the scalar unit (mla_decode_kernel's
        // round-6 fast path; nothing at the 3 tiles per split of a 4 K context, it pays from 8 tiles per split on: DESIGN 4.1.7)
        if (last >= TILE - 1 && (unsigned)(app_pos - tok0) >= (unsigned)TILE) {
          const bf16_t* src = p.ckv + (row0 + wave) * p.ckv_ts;
          const size_t step = (size_t)NWV * p.ckv_ts;
          uint32_t dst = __builtin_amdgcn_readfirstlane(dK + (uint32_t)wave * (KROW * 2));
#pragma unroll
          for (int r = wave; r < TILE; r += NWV, src += step, dst += NWV * KROW * 2) dma_row_s(src, (uint32_t)lane * 16u, dst);
          if (wave < 4) {
            const int r = wave * 8 + (lane >> 3), g = (lane & 7) ^ (lane >> 3);
            dma_row(p.kpe + (row0 + r) * p.kpe_ts + g * 8, dP + (uint32_t)wave * (8 * ROPE * 2));
          }
          return;
        }
#pragma unroll
        for (int r = wave; r < TILE; r += NWV) {
          const int rr = min(r, last);
          const bf16_t* src = p.ckv + (row0 + rr) * p.ckv_ts;
          if (tok0 + rr == app_pos) src = app_ckv;
          dma_row(src + lane * 8, dK + (uint32_t)r * (KROW * 2));
        }
        if (wave < 4) {
          const int r = wave * 8 + (lane >> 3), rr = min(r, last);
          const int g = (lane & 7) ^ (lane >> 3);
          const bf16_t* src = p.kpe + (row0 + rr) * p.kpe_ts;
          if (tok0 + rr == app_pos) src = app_kpe;
          dma_row(src + g * 8, dP + (uint32_t)wave * (8 * ROPE * 2));
        }
      };

      const bool work = t_begin < t_end;
      // the newest row comes from the workspace (phase B's prep workgroup): the split that owns its tile waits for it first
      const bool need_new = work && app_pos >= 0 && (app_pos / TILE) % p.nsplit == split;
      if (need_new) {
        if (wave == 7) poll_flags(p, fKV, 1, epoch, 0xC0, [](int) { return 0; });
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the row is fetched by LDS-DMA (a plain load)
        __syncthreads();
      }
      if (work) stage_tile(t_begin, 0);   // depends on nothing else: in flight while the q rows are awaited
      // (granules: each wavefront waits for exactly the q pieces its lanes hold — no flag sweep, no second round trip)
      av8bf qf[NQ];
      {
        const int hq = head0 + (lane & 15);
        const unsigned qn = L.q_lat + (unsigned)(hq * LORA + (lane >> 4) * 8) * 4;
        const unsigned qr = L.q_pe + (unsigned)(hq * ROPE + (lane >> 4) * 8) * 4;
        uint4 g[2 * NQ];
        unsigned off[2 * NQ];
        bool use[2 * NQ];
#pragma unroll
        for (int j = 0; j < NQ; j++) {
          const int s = ds + j * DSPLIT;   // wave-uniform
          off[2 * j] = s < 16 ? qn + s * 128 : qr + (s - 16) * 128;
          off[2 * j + 1] = off[2 * j] + 16;
          use[2 * j] = use[2 * j + 1] = work && s < 18;
        }
        gran_sweep<2 * NQ>(p, rs, off, use, epoch, 0xC1, g);
#pragma unroll
        for (int j = 0; j < NQ; j++) qf[j] = as_av8bf(use[2 * j] ? gran_join(g[2 * j], g[2 * j + 1]) : make_uint4(0, 0, 0, 0));
      }
      AT_STAMP(7);
      if (work) {
        bf16_t* Pw = Pt + wave * 16 * TILE;
        int cur = 0;
        for (int tile = t_begin; tile < t_end; tile += t_step, cur ^= 1) {
          bf16_t* Kc = Kt + cur * TILE * KROW;
          const bf16_t* Pc = Kp + cur * TILE * ROPE;
          const int tok0 = tile * TILE;
          const int ntok = min(TILE, kv_end - tok0);
          // (staging tiles TWO ahead — three LDS buffers, counted vmcnt waits — was built and measured in round 6: 56.0 vs 56.1 us per
          // layer, the three tiles of a split at 4 K tokens took 6.5 instead of 7.0 us and the q rows arrived 0.5 us later behind the
          // bigger burst: not kept)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (tile + t_step < t_end) stage_tile(tile + t_step, cur ^ 1);

          v4f s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
          const bf16_t* kb0 = Kc + (lane & 15) * KROW + (lane >> 4) * 8;
          const bf16_t* kb1 = kb0 + 16 * KROW;
#pragma unroll
          for (int j = 0; j < NQ; j++) {
            const int s = ds + j * DSPLIT;
            if (s < 16) {
              const av8bf b0 = as_av8bf(*reinterpret_cast<const uint4*>(kb0 + s * 32));
              const av8bf b1 = as_av8bf(*reinterpret_cast<const uint4*>(kb1 + s * 32));
              s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[j], b0, s0, 0, 0, 0);
              s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[j], b1, s1, 0, 0, 0);
            } else if (s < 18) {
              const int row = lane & 15, q = (s - 16) * 4 + (lane >> 4);
              const bf16_t* pr = Pc + row * ROPE + ((q ^ (row & 7)) * 8);
              const av8bf b0 = as_av8bf(*reinterpret_cast<const uint4*>(pr));
              const av8bf b1 = as_av8bf(*reinterpret_cast<const uint4*>(pr + 16 * ROPE));
              s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[j], b0, s0, 0, 0, 0);
              s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[j], b1, s1, 0, 0, 0);
            }
          }
          {   // the head block's partial score tiles meet in LDS; fixed order -> identical S in every wave
            float* sx = Sx + wave * 512 + lane;
#pragma unroll
            for (int r = 0; r < 4; r++) { sx[r * 64] = s0[r]; sx[(4 + r) * 64] = s1[r]; }
            __syncthreads();
            const float* sr = Sx + hbw * DSPLIT * 512 + lane;
#pragma unroll
            for (int r = 0; r < 4; r++) { s0[r] = sr[r * 64]; s1[r] = sr[(4 + r) * 64]; }
#pragma unroll
            for (int d = 1; d < DSPLIT; d++)
#pragma unroll
              for (int r = 0; r < 4; r++) { s0[r] += sr[d * 512 + r * 64]; s1[r] += sr[d * 512 + (4 + r) * 64]; }
          }
          const bool v0 = (lane & 15) < ntok, v1 = 16 + (lane & 15) < ntok;
          // (mla_decode_kernel's round-6 form: the four heads' row reductions interleaved as VOP2-DPP instructions, the O rescale only
          //  where a running maximum moved — same values, same bits)
          float alpha[4], sa[4], sb[4], mx4[4], ps[4], pav[4], pbv[4], mnew[4];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            sa[r] = v0 ? s0[r] * p.sm_scale : -__builtin_inff();
            sb[r] = v1 ? s1[r] * p.sm_scale : -__builtin_inff();
            mx4[r] = fmaxf(sa[r], sb[r]);
          }
          row16_max4(mx4[0], mx4[1], mx4[2], mx4[3]);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            mnew[r] = fmaxf(m_run[r], mx4[r]);
            pav[r] = __expf(sa[r] - mnew[r]);
            pbv[r] = __expf(sb[r] - mnew[r]);
            ps[r] = pav[r] + pbv[r];
          }
          row16_sum4(ps[0], ps[1], ps[2], ps[3]);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float pa = pav[r], pb = pbv[r], m_new = mnew[r], sum = ps[r];
            alpha[r] = __expf(m_run[r] - m_new);
            l_run[r] = l_run[r] * alpha[r] + sum;
            m_run[r] = m_new;
            const int hrow = (lane >> 4) * 4 + r;
            const uint32_t pk = ktx_pk_bf16(pa, pb);
            Pw[hrow * TILE + (lane & 15)] = (bf16_t)(pk & 0xffffu);
            Pw[hrow * TILE + 16 + (lane & 15)] = (bf16_t)(pk >> 16);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          const av8bf pf = as_av8bf(*reinterpret_cast<const uint4*>(Pw + (lane & 15) * TILE + (lane >> 4) * 8));
          const bf16_t* vb = Kc + ((lane >> 4) * 8 + ((lane & 15) >> 2)) * KROW + (lane & 3) * 4 + ds * NDT * 16;
          const bool moved = alpha[0] != 1.f || alpha[1] != 1.f || alpha[2] != 1.f || alpha[3] != 1.f;
          if (__builtin_amdgcn_ballot_w64(moved) != 0ull) {
#pragma unroll
            for (int i = 0; i < NDT; i++)
#pragma unroll
              for (int r = 0; r < 4; r++) o[i][r] *= alpha[r];
          }
#pragma unroll
          for (int i = 0; i < NDT; i++) {
            const av8bf b = load_v_frag(vb + i * 16);
            o[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, b, o[i], 0, 0, 0);
          }
        }
      }
      AT_STAMP(8);
      // ---- partial results (ktx_mla_decode_partials' layout): (m, l) per head and split; the un-normalised O rows leave through
      // LDS so that every store is a whole 16 bytes (write-through)
      float* part_ml = reinterpret_cast<float*>(p.ws + L.part_ml);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int hrow = (lane >> 4) * 4 + r;
        if ((lane & 15) == 0 && ds == 0) {
          const unsigned long long ml = ((unsigned long long)__float_as_uint(l_run[r]) << 32) | __float_as_uint(m_run[r]);
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(part_ml + ((size_t)(head0 + hrow) * p.nsplit + split) * 2), ml,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (work) {
        __syncthreads();   // every wavefront is done with the staged tiles: their LDS becomes the transpose buffer
        constexpr int OROW = 132;   // floats per staged row (128 + pad)
        float* Ot = reinterpret_cast<float*>(smem) + wave * 16 * OROW;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int hrow = (lane >> 4) * 4 + r;
#pragma unroll
          for (int i = 0; i < NDT; i++) Ot[hrow * OROW + i * 16 + (lane & 15)] = o[i][r];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 8; it++) {
          const int idx = it * 64 + lane, row = idx >> 5, c4 = idx & 31;
          const uint4 v = *reinterpret_cast<const uint4*>(Ot + row * OROW + c4 * 4);
          ws_store16(rs, L.part_o + (unsigned)((((head0 + row) * p.nsplit + split) * LORA + ds * 128 + c4 * 4) * 4), v);
        }
      }
      drain_stores();
      __syncthreads();
      if (tid == 0) st_word(fC + w, epoch);
    }
    AT_STAMP(9);
  }

  // =========================== phase D: merge of the splits + un-absorb ========================================================
  // phase E's first ring tiles are requested before phase D starts waiting
  const int GPS_E = p.nksE >> 3;
  int Gb = 0, Ge = 0, gbE = 0, geE = 0;
  // Phase E's register ring is RG groups of 8 k-steps deep.  The first group is requested here (it flies during the C -> D hand-off);
  // the other RG - 1 groups are requested inside phase D once the merge has consumed the splits' partial rows (their 64 registers
  // are free from there on), so they fly during the un-absorb k-steps and the D -> E hand-off — the time this CU's memory pipe
  // otherwise idles.  Phase E then starts with 24 of a wavefront's 16 / 32 k-steps in registers instead of 8.
  // (FP8 tiles are two planes: registers for two groups.  W4 with four groups, the fourth requested with the others or behind phase E's
  // staging requests, measured slower: 58.7 / 57.1 against 56.9 / 55.8 us per layer)
  constexpr int RG = F8 ? 2 : 3;
  uint4 wrE[8 * RG][NQ];
  uint2 srE[8 * RG];
  const long ntileE = (long)p.nE * p.nksE;
  auto load_E = [&](int d, long tile) {
    tile = min(tile, ntileE - 1);   // (a run-ahead slot past the wavefront's range reads a tile nobody uses: no branch around requests)
#pragma unroll
    for (int q = 0; q < NQ; q++) wrE[d][q] = nt_load16(p.wE + tile * TB + q * 1024 + lane * 16);
    if constexpr (F8) {   // fp32 scale of (128-row block, k-step): tile = strip * nksE + ks, nksE a power of two
      const long strip = tile >> p.nksE_sh, ks = tile & (p.nksE - 1);
      srE[d] = make_uint2(__float_as_uint(reinterpret_cast<const float*>(p.scE)[(strip >> 3) * p.nksE + ks]), 0);
    } else {
      srE[d] = load_w4_scales<2>(p.scE + (tile * 16 + (lane & 15)) * 2);
    }
  };
  auto prefetch_E = [&]() {
    const int nwgE = min(NWG, p.nE);
    if (w < nwgE) { Gb = split_begin(p.eQ, p.eR, w) * GPS_E; Ge = split_begin(p.eQ, p.eR, w + 1) * GPS_E; }
    gbE = wave_begin(Gb, Ge - Gb, wave); geE = wave_begin(Gb, Ge - Gb, wave + 1);
    if (geE > gbE) {
#pragma unroll
      for (int d = 0; d < 8; d++) load_E(d, (long)gbE * 8 + d);
    }
  };
  auto prefetch_E2 = [&]() {
    if (geE > gbE + 1) {
#pragma unroll
      for (int d = 8; d < 8 * RG; d++) load_E(d, (long)gbE * 8 + d);
    }
  };
  if constexpr ((MASK & PH_E) != 0) prefetch_E();
  if constexpr ((MASK & PH_E) != 0 && (MASK & PH_D) == 0) prefetch_E2();
  if constexpr ((MASK & PH_E) != 0 && (MASK & PH_D) != 0) { if (!doBD) prefetch_E2(); }
  if constexpr ((MASK & PH_D) != 0) {
   if (doBD) {
    __syncthreads();
    const int S = p.nsplit;
    float* s_w = reinterpret_cast<float*>(smem);              // [MAXS]
    float* s_red = s_w + MAXS;                                // [16]
    float* s_acc = s_red + 16;                                // [8][512]
    bf16_t* stageD = reinterpret_cast<bf16_t*>(s_acc + 8 * LORA);  // [256]
    uint8_t* xsD = reinterpret_cast<uint8_t*>(stageD + 256);       // [64][16 B]
    const int hgD = h >> 5;
    // (every wavefront polling the flags of the splits IT merges and requesting their rows at once: 55.5 against 54.2 us per layer — eight
    //  pollers per CU cost more than the early requests bring)
    if (wave == 7) poll_flags(p, fC, S, epoch, 0xD1, [=](int k) { return hgD * SPG + k; });
    __syncthreads();
    AT_STAMP(10);
    // Every workgroup of the pair merges ALL 512 dims of its head (64 lanes x 8 dims; the pair sits on one XCD, so the partner's
    // second read of the partials is L2-served): the merged row stays in this workgroup's LDS and the round-4 exchange of half
    // rows through the workspace (publish + drain + flag + poll + reload: ~3 us of the chain) is gone.  Same sums in the same order.
    const int sl = wave, dg = lane;   // split lane, dim group (8 dims)
    const size_t base = (size_t)h * S;
    unsigned long long mlraw = 0;
    if (tid < S) mlraw = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p.ws + L.part_ml) + base + tid, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
    const float2 ml = tid < S ? make_float2(__uint_as_float((unsigned)mlraw), __uint_as_float((unsigned)(mlraw >> 32))) : make_float2(0.f, 0.f);
    uint4 fa[8], fb[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int sidx = min(sl + 8 * u, S - 1);
      const unsigned off = L.part_o + (unsigned)(((base + sidx) * LORA + dg * 8) * 4);
      fa[u] = ws_load16(rs, off);
      fb[u] = ws_load16(rs, off + 16);
    }
    float mstar = ml.y > 0.f ? ml.x : -__builtin_inff();
    mstar = wave_max(mstar);
    if (lane == 0) s_red[wave] = mstar;
    __syncthreads();
    mstar = s_red[0];
#pragma unroll
    for (int v = 1; v < 8; v++) mstar = fmaxf(mstar, s_red[v]);
    const float wgt = ml.y > 0.f ? __expf(ml.x - mstar) : 0.f;
    if (tid < S) s_w[tid] = wgt;
    float lsum = wave_sum(ml.y * wgt);
    if (lane == 0) s_red[8 + wave] = lsum;
    __syncthreads();
    lsum = 0.f;
#pragma unroll
    for (int v = 0; v < 8; v++) lsum += s_red[8 + v];
    {
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int sidx = sl + 8 * u;
        const float wv = sidx < S ? s_w[sidx] : 0.f;
        if (wv > 0.f) {   // a dead split's row may be stale memory: selected away, not multiplied by 0
          acc[0] += __uint_as_float(fa[u].x) * wv; acc[1] += __uint_as_float(fa[u].y) * wv; acc[2] += __uint_as_float(fa[u].z) * wv; acc[3] += __uint_as_float(fa[u].w) * wv;
          acc[4] += __uint_as_float(fb[u].x) * wv; acc[5] += __uint_as_float(fb[u].y) * wv; acc[6] += __uint_as_float(fb[u].z) * wv; acc[7] += __uint_as_float(fb[u].w) * wv;
        }
      }
      for (int s0 = sl + 64; s0 < S; s0 += 32) {   // more than 64 splits (64-head models): lin_merge_unabsorb_kernel's continuation, same order
        uint4 va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int sidx = min(s0 + 8 * u, S - 1);
          const unsigned off = L.part_o + (unsigned)(((base + sidx) * LORA + dg * 8) * 4);
          va[u] = ws_load16(rs, off);
          vb[u] = ws_load16(rs, off + 16);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int sidx = s0 + 8 * u;
          const float wv = sidx < S ? s_w[sidx] : 0.f;
          if (wv > 0.f) {
            acc[0] += __uint_as_float(va[u].x) * wv; acc[1] += __uint_as_float(va[u].y) * wv; acc[2] += __uint_as_float(va[u].z) * wv; acc[3] += __uint_as_float(va[u].w) * wv;
            acc[4] += __uint_as_float(vb[u].x) * wv; acc[5] += __uint_as_float(vb[u].y) * wv; acc[6] += __uint_as_float(vb[u].z) * wv; acc[7] += __uint_as_float(vb[u].w) * wv;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 8; q++) s_acc[sl * LORA + dg * 8 + q] = acc[q];
    }
    __syncthreads();
    {
      const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 8; i++) v += s_acc[i * LORA + tid];
      reinterpret_cast<bf16_t*>(xsD)[tid] = f32_to_bf16(v * inv);   // the merged row of head h, all 512 dims (NT == LORA)
    }
    __syncthreads();
    AT_STAMP(11);
    // ---- un-absorb: waves 0..3 = one strip of this half each, the 4 k-steps in order (lin_merge_unabsorb_kernel)
    if (wave < 4) {
      const int kc = lane >> 4;
      const uint8_t* xb0 = xsD + kc * 4 * 16;
      v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ks++) bf16_kstep(wrD[ks], xb0 + (size_t)ks * 256, acc);
      if (lane < 16) stageD[wave * 16 + lane] = f32_to_bf16(0.f + acc[0]);
    }
    __syncthreads();
    // Wavefront 0 publishes the workgroup's attention rows (store, drain, flag: one wavefront, no barrier in between).  Phase E's
    // run-ahead groups go out HERE — the other wavefronts at once, wavefront 0 behind its flag: a wavefront stalls in the issue of
    // 32 requests until the CU's memory pipe has taken them (measured: ~3.5 us when issued inside the merge, on the chain), and
    // from here on nothing of this workgroup is on anybody's chain until the attention rows of ALL heads have arrived.
    if (wave == 0) {
      if (tid < 8) ws_store16(rs, L.attn_out + (unsigned)(h * VDIM + part * 64) * 2 + tid * 16, *reinterpret_cast<const uint4*>(stageD + tid * 8));
      drain_stores();
      if (tid == 0) st_word(fD + w, epoch);
    }
    if constexpr ((MASK & PH_E) != 0) prefetch_E2();
   }
    AT_STAMP(12);
  }

  // =========================== phase E: o_proj + residual ================================================================================
  if constexpr ((MASK & PH_E) != 0) {
    __syncthreads();
    const int nksE = p.nksE;
    uint8_t* xsE = smem;                                                          // [nksE * 16][16 B]
    float* auxE = reinterpret_cast<float*>(smem + (size_t)nksE * 16 * 16);        // [nksE * 2][4]
    float* tableE = auxE + (size_t)nksE * 2 * 4;                                  // [strips of this workgroup][8][64]
    if (wave == 7) poll_flags(p, fD, NWB, epoch, 0xE1, [](int k) { return k; });
    __syncthreads();
    AT_STAMP(13);
    {
      // every piece of the attention row this thread stages is requested BEFORE the first one is used (as a loop `load, use` the
      // compiler waited for each in turn — and with it, replies arriving in order, for every run-ahead tile requested before:
      // four serial round trips, ~3 us of phase E)
      constexpr int XRE = 4;   // 16-byte pieces per thread: heads * v_dim <= 16384
      const int np = nksE * 16;
      uint4 xvE[XRE];
#pragma unroll
      for (int i = 0; i < XRE; i++) xvE[i] = ws_load16(rs, L.attn_out + (unsigned)min(tid + i * NT, np - 1) * 16);
#pragma unroll
      for (int i = 0; i < XRE; i++) {
        const int pc = tid + i * NT;   // (whole wavefronts: np is a multiple of 64)
        if (pc >= np) continue;
        const uint4 v = xvE[i];
        if constexpr (F8) {
          float sq;
          const uint2 q8 = fp8_quant_piece(v, sq);
          *reinterpret_cast<uint2*>(xsE + (size_t)pc * 8) = q8;
          if ((pc & 15) == 0) auxE[pc >> 4] = sq;
        } else {
          *reinterpret_cast<uint4*>(xsE + (size_t)pc * 16) = v;
          const float s = group_sum64(v);
          if ((pc & 7) == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) auxE[(pc >> 3) * 4 + r] = s;
          }
        }
      }
    }
    __syncthreads();
    AT_STAMP(28);
    const int s_first = Gb / GPS_E;
    if (geE > gbE) {
      const int kc = lane >> 4;
      const uint8_t* xb0 = xsE + kc * (F8 ? 32 : 16);
      auto stepE = [&](int d, int ks, float& a) {
        if constexpr (F8) fp8_kstep1(wrE[d][0], wrE[d][NQ - 1], __uint_as_float(srE[d].x), xb0 + (size_t)ks * 128, auxE[ks], a);
        else w4_kstep1(wrE[d][0], srE[d], xb0 + (size_t)ks * 256, auxE + ks * 8, a);
      };
      int strip = gbE / GPS_E, kg = gbE - strip * GPS_E;
      float acc = 0.f;
      auto flush = [&]() {
        if (lane < 16) tableE[((size_t)(strip - s_first) * 8 + wave) * 64 + lane] = acc;
      };
      // groups gbE .. geE - 1 in order.  Stage 1: the first RG groups sit in the ring (requested before / inside phase D) — straight-line
      // code, no requests except group RG's, which goes out into the first ring group as soon as group 0 has been used.  Stage 2
      // (a wavefront with more than RG groups): the round-4 stream through ring group 0, one group of lead.
      const int ngrp = geE - gbE;
      auto end_group = [&](int j) {
        kg++;
        if (kg == GPS_E || j == ngrp - 1) {
          flush();
          acc = 0.f;
          kg = 0;
          strip++;
        }
      };
#pragma unroll
      for (int r = 0; r < RG; r++) {
        if (r < ngrp) {
          const int ks0 = kg * 8;
#pragma unroll
          for (int d = 0; d < 8; d++) {
            stepE(r * 8 + d, ks0 + d, acc);
            __builtin_amdgcn_sched_barrier(0);   // (fences every 2 / 4 / 8 k-steps measured the same: 55.8 - 56.3 us per layer)
          }
          if (r == 0) {   // (requested slot by slot behind each k-step of group 0 instead: 55.2 - 55.5 against 55.1 - 55.2 us per layer)
            if (ngrp > RG) {
#pragma unroll
              for (int d = 0; d < 8; d++) load_E(d, (long)(gbE + RG) * 8 + d);
            }
          }
          end_group(r);
          AT_STAMP(29 + r);
        }
      }
      for (int j = RG; j < ngrp - 1; j++) {
        const int ks0 = kg * 8;
#pragma unroll
        for (int d = 0; d < 8; d++) {
          stepE(d, ks0 + d, acc);
          load_E(d, (long)(gbE + j + 1) * 8 + d);
          __builtin_amdgcn_sched_barrier(0);
        }
        end_group(j);
      }
      if (ngrp > RG) {   // (its own copy of the eight k-steps: sharing one with the loop above lets the compiler lift all eight unpacks)
        const int ks0 = kg * 8;
#pragma unroll
        for (int d = 0; d < 8; d++) {
          stepE(d, ks0 + d, acc);
          __builtin_amdgcn_sched_barrier(0);
        }
        end_group(ngrp - 1);
      }
    }
    __syncthreads();
    AT_STAMP(14);
    const int n_local = Ge > Gb ? (Ge - 1) / GPS_E - s_first + 1 : 0;
    for (int sl = wave; sl < n_local; sl += 8) {
      const int s = s_first + sl, g0 = s * GPS_E, g1 = g0 + GPS_E;
      float v = 0.f;
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int wb = wave_begin(Gb, Ge - Gb, u), we = wave_begin(Gb, Ge - Gb, u + 1);
        if (we > wb && wb < g1 && we > g0) v += tableE[((size_t)sl * 8 + u) * 64 + lane];
      }
      const int n = s * 16 + lane;
      if (lane < 16 && n < p.hidden) {
        bf16_t o = f32_to_bf16(v);
        o = f32_to_bf16(bf16_to_f32(p.x[n]) + bf16_to_f32(o));   // hidden = residual + attn (modeling_deepseek_v3.py:1219)
        p.y[n] = o;
      }
    }
    AT_STAMP(15);
  }

  // =========================== the step's last launch advances the epoch =========================================================================
  if (p.last) {
    __syncthreads();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(hdr + W_EXIT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == gridDim.x - 1) {
        st_word(hdr + W_EXIT, 0u);
        unsigned e = epoch + 1u;
        if (e == 0u) e = 1u;
        st_word(hdr + W_EPOCH, e);
      }
    }
  }
}

// dev / tests (ktx_attn_debug_read): the payload dwords of a granule row
__global__ void attn_ungranule_kernel(const uint2* g, unsigned* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = g[i].x;
}

// =====================================================================================================================================
// host
// =====================================================================================================================================
// One workspace per (device, attention geometry): the layers of a model share it (launches of one stream are ordered), two models of
// different geometry on one device each get their own.  `last` = the geometry of the device's most recent launch (debug reads).
struct DevWs { uint8_t* base = nullptr; size_t bytes = 0; int H = 0, nA = 0; };
constexpr int MAX_DEV = 64, MAX_GEO = 4;
std::mutex g_mu;
DevWs g_ws[MAX_DEV][MAX_GEO];
int g_last[MAX_DEV];
bool g_attr_set[MAX_DEV][128];          // hipFuncSetAttribute is per device (and per instantiation: indexed by the phase mask)
unsigned long long* g_stamps = nullptr;
// Status words the DEVICE writes into pinned host memory when a hand-off gives up (one per device): ktx_attn_status reads them with a
// plain load — no device synchronisation — so a decode loop can afford the check after every token.
unsigned* g_hstatus = nullptr;       // host address of [MAX_DEV] words
unsigned* g_hstatus_dev = nullptr;   // the same memory as the devices see it

// One persistent launch in flight per device: a launch issued on another stream than the device's previous one first waits for an event
// recorded behind that one (eager mode only: inside a stream capture the caller keeps one capture per device).
struct DevOrder { hipEvent_t ev = nullptr; hipStream_t last = nullptr; bool valid = false; };
DevOrder g_order[MAX_DEV];
// held by an eager launch across wait-for-the-previous-launch -> launch -> record (as three separately locked steps two
// host threads on different streams could interleave between the wait and the record, and two 256-workgroup persistent launches
// would overlap — the mutual wait the event exists to prevent)
std::mutex g_launch_mu[MAX_DEV];

int order_before_launch(int dev, hipStream_t st, bool* eager) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
  *eager = cs == hipStreamCaptureStatusNone;
  if (!*eager) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  DevOrder& o = g_order[dev];
  if (o.valid && o.last != st) KTX_HIP(hipStreamWaitEvent(st, o.ev, 0));
  return 0;
}
int order_after_launch(int dev, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_mu);
  DevOrder& o = g_order[dev];
  if (!o.ev) KTX_HIP(hipEventCreateWithFlags(&o.ev, hipEventDisableTiming));
  KTX_HIP(hipEventRecord(o.ev, st));
  o.last = st;
  o.valid = true;
  return 0;
}

int hstatus_for(int dev, unsigned** out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_hstatus) {
    void* h = nullptr;
    KTX_HIP(hipHostMalloc(&h, MAX_DEV * sizeof(unsigned), hipHostMallocMapped | hipHostMallocPortable));
    std::memset(h, 0, MAX_DEV * sizeof(unsigned));
    void* d = nullptr;
    KTX_HIP(hipHostGetDevicePointer(&d, h, 0));
    g_hstatus = (unsigned*)h;
    g_hstatus_dev = (unsigned*)d;
  }
  *out = g_hstatus_dev + dev;
  return 0;
}

int ws_for(int dev, int H, int nA, uint8_t** out, unsigned* bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  KTX_REQUIRE(dev >= 0 && dev < MAX_DEV, "ktx_attn: device index out of range");
  int slot = -1;
  for (int i = 0; i < MAX_GEO; i++) {
    if (g_ws[dev][i].base && g_ws[dev][i].H == H && g_ws[dev][i].nA == nA) { slot = i; break; }
    if (!g_ws[dev][i].base && slot < 0) slot = i;
  }
  KTX_REQUIRE(slot >= 0, "ktx_attn: more attention geometries on one device than workspaces (4)");
  DevWs& d = g_ws[dev][slot];
  if (!d.base) {
    const WsLayout L = ws_layout(H, nA);
    KTX_HIP(hipMalloc((void**)&d.base, L.total));
    KTX_HIP(hipMemset(d.base, 0, L.total));
    const unsigned one = 1;
    KTX_HIP(hipMemcpy(d.base + 4 * W_EPOCH, &one, 4, hipMemcpyHostToDevice));
    d.bytes = L.total; d.H = H; d.nA = nA;
  }
  g_last[dev] = slot;
  *out = d.base;
  *bytes = (unsigned)d.bytes;
  return 0;
}

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int check_args(const ktx_attn_decode_args* a, KtxLinearRaw (&r)[5], int* nsplit_out) {
  KTX_REQUIRE(a, "ktx_attn_decode: null args");
  ktx_linear_t hs[5] = {a->qkv_a, a->q_b, a->q_absorb, a->out_absorb, a->o_proj};
  for (int i = 0; i < 5; i++) {
    KTX_REQUIRE(hs[i], "ktx_attn_decode: null operator handle");
    if (ktx_linear_raw(hs[i], &r[i]) != 0) return -1;
    KTX_REQUIRE(r[i].loaded, "ktx_attn_decode: an operator has no weights loaded");
    KTX_REQUIRE(r[i].device == r[0].device, "ktx_attn_decode: operators on different devices");
    KTX_REQUIRE(r[i].bias == nullptr, "ktx_attn_decode: projections with bias are not covered");
  }
  const int H = a->num_heads;
  KTX_REQUIRE(a->nope_dim == NOPE && a->rope_dim == ROPE && a->kv_lora == LORA && a->v_dim == VDIM, "ktx_attn_decode: nope 128 / rope 64 / kv_lora 512 / v 128 only");
  KTX_REQUIRE(H == 128 || H == 64, "ktx_attn_decode: 64 or 128 heads (two workgroups per head, head groups of 32 over 256 workgroups)");
  KTX_REQUIRE(a->hidden == DA * 8 * 128 && a->q_lora == NK2 * 2 * 128, "ktx_attn_decode: hidden 7168 / q_lora 1536 only");
  const int BF = KTX_LIN_BF16, QF = r[0].format, QG = QF == KTX_LIN_FP8 ? 128 : 64;   // the three quantised projections share one format
  KTX_REQUIRE(QF == KTX_LIN_W4 || QF == KTX_LIN_FP8, "ktx_attn_decode: the projections must be W4 g64 or block-FP8 linears");
  KTX_REQUIRE(r[0].group_size == QG && r[0].batch == 1 && r[0].in_features == a->hidden &&
                  r[0].out_features == a->q_lora + LORA + ROPE, "ktx_attn_decode: qkv_a must be the merged W4 g64 / FP8 q_a|kv_a linear");
  KTX_REQUIRE(r[1].format == QF && r[1].group_size == QG && r[1].batch == 1 && r[1].in_features == a->q_lora &&
                  r[1].out_features == H * QW, "ktx_attn_decode: q_b must be W4 g64 / FP8 [heads * 192, q_lora] like qkv_a");
  KTX_REQUIRE(r[2].format == BF && r[2].batch == H && r[2].in_features == NOPE && r[2].out_features == LORA, "ktx_attn_decode: q_absorb must be BF16 [heads][512, 128]");
  KTX_REQUIRE(r[3].format == BF && r[3].batch == H && r[3].in_features == LORA && r[3].out_features == VDIM, "ktx_attn_decode: out_absorb must be BF16 [heads][128, 512]");
  KTX_REQUIRE(r[4].format == QF && r[4].group_size == QG && r[4].batch == 1 && r[4].in_features == H * VDIM &&
                  r[4].out_features == a->hidden && r[4].NKS % 8 == 0 && (r[4].NKS & (r[4].NKS - 1)) == 0,
              "ktx_attn_decode: o_proj must be W4 g64 / FP8 [hidden, heads * 128] like qkv_a");
  KTX_REQUIRE(a->page_size > 0 && a->page_size % TILE == 0 && a->ckv_token_stride % 8 == 0 && a->kpe_token_stride % 8 == 0,
              "ktx_attn_decode: page_size must be a multiple of 32 and the token strides multiples of 8 elements");
  int ncu = 0;
  KTX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, r[0].device));
  KTX_REQUIRE(ncu >= GRID, "ktx_attn_decode: fewer CUs than workgroups (every workgroup must be resident)");
  ktx_mla_config mc{H, LORA, ROPE, a->page_size, a->sm_scale, 256, a->kv_len_hint};
  const int ns = ktx_mla_decode_nsplit(&mc, 1, (size_t)1 << 40);
  KTX_REQUIRE(ns >= 1 && ns <= GRID / (H / 32), "ktx_attn_decode: the KV split rule asks for a split count this launch does not cover (context too long)");
  *nsplit_out = ns;
  return 0;
}

template <int MASK, int FMT = KTX_LIN_W4>
int launch(const AttnParams& p, int dev, int nwg, hipStream_t st) {
  constexpr size_t LDS = 108 * 1024;
  constexpr int SLOT = MASK + (FMT == KTX_LIN_FP8 ? 64 : 0);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_attr_set[dev][SLOT]) {
      KTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_decode_kernel<MASK, FMT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
      g_attr_set[dev][SLOT] = true;
    }
  }
  hipLaunchKernelGGL((attn_decode_kernel<MASK, FMT>), dim3(nwg), dim3(NT), LDS, st, p);
  KTX_HIP(hipGetLastError());
  return 0;
}

}  // namespace

extern "C" int ktx_attn_debug_stamps(unsigned long long* d_buf) {
  g_stamps = d_buf;
  return 0;
}

extern "C" int ktx_attn_decode_eligible(const ktx_attn_decode_args* a) {
  KtxLinearRaw r[5];
  int ns = 0;
  if (ktx_debug_get(26) == 1) { ktx_fail("ktx_attn_decode: switched off (dev knob 26)"); return 0; }
  return check_args(a, r, &ns) == 0 ? 1 : 0;
}

extern "C" int ktx_attn_decode(const ktx_attn_decode_args* a, ktx_stream_t stream) {
  KtxLinearRaw r[5];
  int nsplit = 0;
  if (check_args(a, r, &nsplit) != 0) return -1;
  KTX_REQUIRE(a->d_x && a->d_y && a->d_in_norm_w && a->d_qa_norm_w && a->d_kv_norm_w && a->d_position && a->d_inv_freq && a->d_ckv &&
                  a->d_k_pe && a->d_kv_indptr && a->d_kv_len, "ktx_attn_decode: null pointer");
  KTX_REQUIRE(a->phases > 0 && a->phases <= KTX_ATTN_PHASE_ALL, "ktx_attn_decode: bad phase mask");
  const int dev = r[0].device, H = a->num_heads, NWG = GRID;
  DeviceGuard guard(dev);
  AttnParams p{};
  p.wA = r[0].w; p.scA = (const bf16_t*)r[0].sc; p.nksA = r[0].NKS; p.nA = r[0].nstrips;
  p.x = (const bf16_t*)a->d_x; p.in_norm_w = (const bf16_t*)a->d_in_norm_w; p.in_eps = a->in_norm_eps; p.hidden = a->hidden;
  p.wB = r[1].w; p.scB = (const bf16_t*)r[1].sc; p.nksB = r[1].NKS;
  p.wUK = r[2].w; p.wbsUK = (size_t)r[2].nstrips * r[2].NKS * 4096;
  p.qa_norm_w = (const bf16_t*)a->d_qa_norm_w; p.qa_eps = a->qa_norm_eps; p.q_lora = a->q_lora;
  p.kv_norm_w = (const bf16_t*)a->d_kv_norm_w; p.kv_eps = a->kv_norm_eps;
  p.pos = a->d_position; p.inv_freq = a->d_inv_freq; p.mscale = a->mscale;
  p.H = H;
  p.ckv = (bf16_t*)a->d_ckv; p.kpe = (bf16_t*)a->d_k_pe; p.ckv_ts = a->ckv_token_stride; p.kpe_ts = a->kpe_token_stride;
  p.kv_indptr = a->d_kv_indptr; p.kv_indices = a->d_kv_indices; p.kv_len = a->d_kv_len;
  p.page_size = a->page_size; p.nsplit = nsplit; p.sm_scale = a->sm_scale;
  p.wUV = r[3].w; p.wbsUV = (size_t)r[3].nstrips * r[3].NKS * 4096;
  p.wE = r[4].w; p.scE = (const bf16_t*)r[4].sc; p.nksE = r[4].NKS; p.nE = r[4].nstrips;
  for (p.nksE_sh = 0; (1 << p.nksE_sh) < p.nksE; p.nksE_sh++) {}
  {
    const int nwgE = std::min(NWG, p.nE);
    p.eQ = p.nE / nwgE; p.eR = p.nE % nwgE;
    KTX_REQUIRE(p.eQ + 1 <= 4, "ktx_attn_decode: o_proj has more strips per workgroup than the LDS table holds");
  }
  p.y = (bf16_t*)a->d_y;
  KTX_REQUIRE(p.nksA == DA * 8 && p.nksB == 2 * NK2 && p.nA <= NWG, "ktx_attn_decode: unexpected tile counts");
  if (ws_for(dev, H, p.nA, &p.ws, &p.ws_bytes) != 0) return -1;
  if (hstatus_for(dev, &p.hstatus) != 0) return -1;
  p.last = a->last ? 1 : 0;
  p.stamps = g_stamps;
  hipStream_t st = (hipStream_t)stream;
  // algorithmic bytes of the phases in this launch (weights as stored + the context's latent rows)
  double bytes = 0;
  auto lin_bytes = [](const KtxLinearRaw& q) {
    const double tile = q.format == KTX_LIN_W4 ? 1024.0 : q.format == KTX_LIN_FP8 ? 2048.0 : 4096.0;
    const double sc = q.format == KTX_LIN_W4 ? (double)q.nstrips * q.NKS * 16 * (128 / q.group_size) * 2
                      : q.format == KTX_LIN_FP8 ? (double)((q.nstrips + 7) / 8) * q.NKS * 4 : 0.0;
    return ((double)q.nstrips * q.NKS * tile + sc) * q.batch;
  };
  if (a->phases & PH_A) bytes += lin_bytes(r[0]);
  if (a->phases & PH_B) bytes += lin_bytes(r[1]) + lin_bytes(r[2]);
  if (a->phases & PH_C) bytes += (double)std::max(a->kv_len_hint, 1) * (LORA + ROPE) * 2.0;
  if (a->phases & PH_D) bytes += lin_bytes(r[3]);
  if (a->phases & PH_E) bytes += lin_bytes(r[4]);
  const bool f8 = r[0].format == KTX_LIN_FP8;
  KTX_TIMED(st, bytes, "attn_decode_kernel<%d%s> H=%d nsplit=%d", a->phases, f8 ? ",FP8" : "", H, nsplit);
  bool eager = false;
  std::unique_lock<std::mutex> launch_lk(g_launch_mu[dev]);   // (a captured launch holds it only for the capture call itself)
  if (order_before_launch(dev, st, &eager) != 0) return -1;
  int rc = -1;
  switch (a->phases) {
#ifdef KTX_ATTN_ONLY_MASK
    case KTX_ATTN_ONLY_MASK: rc = launch<KTX_ATTN_ONLY_MASK>(p, dev, NWG, st); break;
    default: return -1;
#else
    case 31: rc = f8 ? launch<31, KTX_LIN_FP8>(p, dev, NWG, st) : launch<31>(p, dev, NWG, st); break;
    case 1: rc = f8 ? launch<1, KTX_LIN_FP8>(p, dev, NWG, st) : launch<1>(p, dev, NWG, st); break;
    case 2: rc = f8 ? launch<2, KTX_LIN_FP8>(p, dev, NWG, st) : launch<2>(p, dev, NWG, st); break;
    case 4: rc = f8 ? launch<4, KTX_LIN_FP8>(p, dev, NWG, st) : launch<4>(p, dev, NWG, st); break;
    case 8: rc = f8 ? launch<8, KTX_LIN_FP8>(p, dev, NWG, st) : launch<8>(p, dev, NWG, st); break;
    case 16: rc = f8 ? launch<16, KTX_LIN_FP8>(p, dev, NWG, st) : launch<16>(p, dev, NWG, st); break;
    case 3: rc = f8 ? -2 : launch<3>(p, dev, NWG, st); break;
    case 24: rc = f8 ? -2 : launch<24>(p, dev, NWG, st); break;
    case 28: rc = f8 ? -2 : launch<28>(p, dev, NWG, st); break;
    default: return ktx_fail("ktx_attn_decode: this phase subset is not instantiated (31, single phases 1..16, 3, 24, 28)");
#endif
  }
  if (rc == -2) return ktx_fail("ktx_attn_decode: this phase subset is not instantiated for FP8 projections (31 and the single phases are)");
  if (rc == 0 && eager) rc = order_after_launch(dev, st);
  return rc;
}

extern "C" int ktx_attn_status(int device, uint32_t* status_out) {
  KTX_REQUIRE(status_out && device >= 0 && device < MAX_DEV, "ktx_attn_status: bad arguments");
  std::lock_guard<std::mutex> lk(g_mu);
  // the host-mapped word: written by the device at the moment a poll gives up; read here without any device call
  *status_out = g_hstatus ? __atomic_load_n(g_hstatus + device, __ATOMIC_RELAXED) : 0u;
  return 0;
}

extern "C" int ktx_attn_status_any(int* device_out, uint32_t* status_out) {
  KTX_REQUIRE(status_out, "ktx_attn_status_any: bad arguments");
  std::lock_guard<std::mutex> lk(g_mu);
  *status_out = 0;
  if (device_out) *device_out = -1;
  if (!g_hstatus) return 0;
  for (int d = 0; d < MAX_DEV; d++) {
    const unsigned v = __atomic_load_n(g_hstatus + d, __ATOMIC_RELAXED);
    if (v) { *status_out = v; if (device_out) *device_out = d; break; }
  }
  return 0;
}

// dev / tests: copy one of the workspace arrays of the last launch into a device buffer.  which: 0 qkv (A's output row), 1 ckv_new,
// 2 kpe_new, 3 q_lat [H][512], 4 q_pe [H][64], 5 merged rows [H][512], 6 attn_out [H][128], 7 part_ml [H][S][2] fp32, 8 part_o
// [H][S][512] fp32 (S = the split count of the last launch's geometry: pass bytes accordingly), 9 qx [H][128]
extern "C" int ktx_attn_debug_read(int device, int which, void* d_dst, size_t bytes) {
  KTX_REQUIRE(device >= 0 && device < 64 && d_dst, "ktx_attn_debug_read: bad arguments");
  std::lock_guard<std::mutex> lk(g_mu);
  const DevWs& d = g_ws[device][g_last[device]];
  KTX_REQUIRE(d.base, "ktx_attn_debug_read: no workspace on this device yet");
  const WsLayout L = ws_layout(d.H, d.nA);
  const unsigned offs[10] = {L.qkv, L.ckv_new, L.kpe_new, L.q_lat, L.q_pe, 0u, L.attn_out, L.part_ml, L.part_o, L.qx};
  KTX_REQUIRE(which >= 0 && which < 10 && offs[which] + bytes <= d.bytes, "ktx_attn_debug_read: bad array or size");
  KTX_REQUIRE(which != 5, "ktx_attn_debug_read: the merged rows no longer pass through the workspace (each workgroup of a head's pair merges all 512 dims in LDS)");
  DeviceGuard guard(device);
  if (which == 0 || which == 3 || which == 4) {   // granule rows: the payload dwords only
    KTX_REQUIRE(bytes % 4 == 0 && offs[which] + 2 * bytes <= d.bytes, "ktx_attn_debug_read: bad size for a granule row");
    const int n = (int)(bytes / 4);
    hipLaunchKernelGGL(attn_ungranule_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, reinterpret_cast<const uint2*>(d.base + offs[which]),
                       reinterpret_cast<unsigned*>(d_dst), n);
    KTX_HIP(hipGetLastError());
    KTX_HIP(hipDeviceSynchronize());
    return 0;
  }
  KTX_HIP(hipMemcpy(d_dst, d.base + offs[which], bytes, hipMemcpyDeviceToDevice));
  return 0;
}

extern "C" int ktx_attn_reset(int device) {
  KTX_REQUIRE(device >= 0 && device < MAX_DEV, "ktx_attn_reset: bad device");
  std::lock_guard<std::mutex> lk(g_mu);
  DeviceGuard guard(device);
  bool any = false;
  for (int i = 0; i < MAX_GEO; i++) any = any || g_ws[device][i].base;
  if (any) KTX_HIP(hipDeviceSynchronize());
  const unsigned z[2] = {0, 0};
  for (int i = 0; i < MAX_GEO; i++)
    if (g_ws[device][i].base) KTX_HIP(hipMemcpy(g_ws[device][i].base + 4 * W_EXIT, z, 8, hipMemcpyHostToDevice));   // exit counter + status
  if (g_hstatus) __atomic_store_n(g_hstatus + device, 0u, __ATOMIC_RELAXED);
  return 0;
}
