/* ktx_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C (scalar) restatement of the reference's CPU MoE expert forward — the arithmetic contract the
 * HIP path has to reproduce.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (ktransformers_amd/) never does and fails loudly without its HIP
 * extension.
 *
 * Pinning: tests/test_oracle_cpu.py drives this file and the reference's own unmodified kernels
 * (oracle/_ref/libkt_ref.so, built by oracle/Makefile from /root/reference) on the same seeded inputs and
 * requires BIT-EXACT bf16 outputs; tests/golden/ holds vectors generated from the reference build so the
 * same check travels to machines without /root/reference.
 *
 * Every function cites the reference lines (relative to /root/reference/kt-kernel/) it restates.
 * All fp32 arithmetic here is deliberately un-contracted (compile with -ffp-contract=off); where the
 * reference issues an FMA this file calls fmaf() explicitly.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KTXO_FMT_AMXINT4 0
#define KTXO_FMT_AMXINT8 1
#define KTXO_FMT_RAWINT4 2
#define KTXO_FMT_FP8 3
#define KTXO_FMT_FP8PC 5   /* e4m3 weights, one fp32 scale per output row (FP8_PERCHANNEL) */
#define KTXO_FMT_BF16 4

/* ------------------------------------------------------------------------------------------------ */
/* bf16 <-> fp32                                                                                     */
/* ------------------------------------------------------------------------------------------------ */

/* operators/amx/la/utils.hpp:47-52 (avx512_32xbf16_to_32xfp32): bf16 bits << 16. */
float ktxo_bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* operators/amx/la/utils.hpp:14-17: on AVX512-BF16 hosts (this container and the MI355X box's EPYC 9575F) the
 * reference converts with VCVTNE2PS2BF16 = round-to-nearest-even, input denormals treated as zero and results
 * flushed to zero (the instruction ignores MXCSR.DAZ/FTZ and always behaves that way), NaN quieted. */
uint16_t ktxo_f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7f800000u) == 0) return (uint16_t)((u >> 16) & 0x8000u);              /* zero / denormal -> +-0 */
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);     /* NaN -> quiet NaN */
  return (uint16_t)((u + (0x7fffu + ((u >> 16) & 1u))) >> 16);
}

/* _mm512_cvtps_epi32 under default MXCSR = round-to-nearest-even; out of range -> 0x80000000. */
static inline int32_t cvt_rne_i32(float x) {
  if (!(x > -2147483904.0f && x < 2147483648.0f)) return INT32_MIN;
  return (int32_t)lrintf(x); /* FE_TONEAREST is the process default */
}

/* _mm512_cvtsepi32_epi8: signed saturation. */
static inline int8_t sat8(int32_t v) { return (int8_t)(v < -128 ? -128 : (v > 127 ? 127 : v)); }

/* ------------------------------------------------------------------------------------------------ */
/* a6 — activation quantisation, per row int8                                                       */
/* ------------------------------------------------------------------------------------------------ */

/* BufferAImpl::from_mat, operators/amx/la/amx_buffers.hpp:47-98:
 *   d = amax / 127 (fp32), id = d ? 1.0f/d : 0, q = sat8(rne(x * id)). */
void ktxo_quant_act_row(const uint16_t* x, int K, int8_t* q, float* d_out) {
  float amax = 0.0f;
  for (int j = 0; j < K; j++) {
    float a = fabsf(ktxo_bf16_to_f32(x[j]));
    if (a > amax) amax = a;
  }
  float d = amax / 127.0f;
  float id = d ? 1.0f / d : 0.0f;
  for (int j = 0; j < K; j++) q[j] = sat8(cvt_rne_i32(ktxo_bf16_to_f32(x[j]) * id));
  *d_out = d;
}

/* BufferAKGroupImpl::from_mat, operators/amx/la/amx_buffers.hpp:364-419: the same per (row, group of g). */
void ktxo_quant_act_row_kgroup(const uint16_t* x, int K, int g, int8_t* q, float* d_out) {
  for (int kg = 0; kg < K / g; kg++) ktxo_quant_act_row(x + kg * g, g, q + kg * g, d_out + kg);
}

/* ------------------------------------------------------------------------------------------------ */
/* a7 — load-time weight quantisation                                                               */
/* ------------------------------------------------------------------------------------------------ */

/* round_4bit_s8, operators/amx/la/amx_buffers.hpp:527-539: sign-magnitude rounding of an int8 to a multiple
 * of 16: sign(i) * ((|i| + 8) & 0xF0), all in 8-bit arithmetic. */
static inline int8_t round_4bit_s8(int8_t x) {
  uint8_t s = (x & 0x80) ? 0xFF : 0x00;
  uint8_t a = (uint8_t)(x < 0 ? -x : x);
  a = (uint8_t)((a + 8) & 0xF0);
  a = (uint8_t)((a ^ s) - s);
  return (int8_t)a;
}

/* BufferBInt4Impl::_pack_block, operators/amx/la/amx_buffers.hpp:541-627 (AMXINT4):
 *   d[n] = amax_n / 112.0   (double division — the literal is a double — then narrowed to fp32)
 *   i = sat8(rne(w * (1.0f/d))),  q16 = round_4bit_s8(i)  in {-112,...,112 step 16}.
 * Output q16 holds the value the kernel multiplies with (nibble << 4), row-major [N, K]; the reference's
 * tile/VNNI byte layout (same lines) is a pure permutation of these values and is restated separately in
 * ktxo_pack_amxint4_reference_layout(). */
void ktxo_quant_weight_amxint4(const uint16_t* w, int N, int K, int8_t* q16, float* d_out) {
  for (int n = 0; n < N; n++) {
    const uint16_t* row = w + (size_t)n * K;
    float amax = 0.0f;
    for (int j = 0; j < K; j++) {
      float a = fabsf(ktxo_bf16_to_f32(row[j]));
      if (a > amax) amax = a;
    }
    float d = (float)((double)amax / 112.0);
    float id = d ? 1.0f / d : 0.0f;
    for (int j = 0; j < K; j++) q16[(size_t)n * K + j] = round_4bit_s8(sat8(cvt_rne_i32(ktxo_bf16_to_f32(row[j]) * id)));
    d_out[n] = d;
  }
}

/* GemmKernel224Int8::BufferB::_pack_block, operators/amx/la/amx_kernels.hpp:1103-1150 (AMXINT8):
 *   d[n] = amax_n / 127 (fp32), q = sat8(rne(w * (1.0f/d))). */
void ktxo_quant_weight_amxint8(const uint16_t* w, int N, int K, int8_t* q, float* d_out) {
  for (int n = 0; n < N; n++) {
    const uint16_t* row = w + (size_t)n * K;
    float amax = 0.0f;
    for (int j = 0; j < K; j++) {
      float a = fabsf(ktxo_bf16_to_f32(row[j]));
      if (a > amax) amax = a;
    }
    float d = amax / 127.0f;
    float id = d ? 1.0f / d : 0.0f;
    for (int j = 0; j < K; j++) q[(size_t)n * K + j] = sat8(cvt_rne_i32(ktxo_bf16_to_f32(row[j]) * id));
    d_out[n] = d;
  }
}

/* BufferBInt4Impl::to_mat, operators/amx/la/amx_buffers.hpp:683-739: w ~= bf16(float(nibble) * (d * 16.0f)). */
void ktxo_dequant_weight_amxint4(const int8_t* q16, const float* d, int N, int K, uint16_t* w) {
  for (int n = 0; n < N; n++) {
    float vs = d[n] * 16.0f;
    for (int j = 0; j < K; j++) w[(size_t)n * K + j] = ktxo_f32_to_bf16((float)(q16[(size_t)n * K + j] / 16) * vs);
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* a10 — SiLU(gate) * up with the reference's polynomial exp                                        */
/* ------------------------------------------------------------------------------------------------ */

/* amx::exp_avx512, operators/amx/la/amx.hpp:22-45. */
static inline float exp_poly(float x) {
  const float log2e = 1.44269504089f;
  float y = x * log2e;
  int32_t ip = cvt_rne_i32(y);
  float frac = y - (float)ip;
  float p = fmaf(0.0013333558f, frac, 0.0096181291f);
  p = fmaf(p, frac, 0.0555041087f);
  p = fmaf(p, frac, 0.2402265069f);
  p = fmaf(p, frac, 0.6931471805f);
  p = fmaf(p, frac, 0.9999999995f);
  float two_pow_i = ldexpf(1.0f, ip); /* _mm512_scalef_ps(1.0, float(ip)) */
  return two_pow_i * p;
}

/* amx::act_fn (swiglu_limit == 0, swiglu_alpha == 0), operators/amx/la/amx.hpp:47-76:
 *   neg = min(0 - g, 88);  act = g / (1 + exp(neg));  return act * u. */
float ktxo_act_fn(float g, float u) {
  float neg = 0.0f - g;
  if (!(neg <= 88.0f)) neg = 88.0f; /* _mm512_min_ps(neg, 88): returns the second operand on NaN */
  float e = exp_poly(neg);
  float denom = 1.0f + e;
  float act = g / denom;
  return act * u;
}

/* ------------------------------------------------------------------------------------------------ */
/* The expert forward                                                                               */
/* ------------------------------------------------------------------------------------------------ */

typedef struct {
  int fmt;             /* KTXO_FMT_* */
  int E, H, I;
  int group;           /* RAWINT4: k-group size (32); 0 otherwise */
  /* AMXINT4 / AMXINT8: int8 [E][N][K] holding the integer multiplicand (q16 for int4) and fp32 d [E][N]. */
  const int8_t* gate_q; const float* gate_d;
  const int8_t* up_q;   const float* up_d;
  const int8_t* down_q; const float* down_d;
  const uint8_t* gpu_experts_mask; /* optional [E]; common.hpp:256-258 should_skip_expert */
  /* FP8: gate_q/up_q/down_q point at e4m3 bytes [E][N][K], *_d at fp32 scale_inv [E][N/128][K/128].
   * BF16: *_q point at bf16 bits [E][N][K] (as uint16), *_d unused. */
  int dp_even_first; /* test knob: order of the two FMAs inside one VDPBF16PS (0 = odd element first) */
} ktxo_moe;

static inline int skip_expert(const ktxo_moe* m, int64_t id) {
  return id < 0 || id >= m->E || (m->gpu_experts_mask && m->gpu_experts_mask[id]);
}

/* a8 + a9 for one (row, matrix): integer_mat_mul + GemmKernel224Int{4,8}::avx_kernel + apply_scale
 * (operators/amx/la/amx_kernels.hpp:2482-2506,1735-1763,1808-1846) then BufferCImpl::to_mat (amx_buffers.hpp:1716-1732):
 *   acc  = sum_k a_q[k] * w_q[n][k]          exact int32 over the whole K
 *   out  = bf16( (a_d * w_d[n]) * float(acc) ). */
static void gemv_int(const int8_t* aq, float ad, const int8_t* wq, const float* wd, int N, int K, uint16_t* out) {
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; n++) {
    const int8_t* row = wq + (size_t)n * K;
    int32_t acc = 0;
    for (int j = 0; j < K; j++) acc += (int32_t)aq[j] * (int32_t)row[j];
    float s = ad * wd[n];
    out[n] = ktxo_f32_to_bf16(s * (float)acc);
  }
}

/* AMX_MOE_BASE::forward_prefill / forward_decode, operators/amx/moe_base.hpp:197-654, followed by
 * TP_MOE<AMX_MOE_BASE>::merge_results with a single TP part, :749-791.
 *
 * For each token t and slot j (in order j = 0..k-1, skipping should_skip_expert ids):
 *   x_q,x_d = QA(x_t)                                   (a6; identical for every expert => computed once)
 *   g = bf16(GEMM(x_q, Wg_e)), u = bf16(GEMM(x_q, Wu_e)) (a8, a9)
 *   a = bf16(act_fn(f32(g), f32(u)))                    (a10, moe_base.hpp:693-726)
 *   a_q,a_d = QA(a)                                     (a11, moe_base.hpp:378-384)
 *   dn = bf16(GEMM(a_q, Wd_e))
 *   acc_t = fma(f32(dn), w[t][j], acc_t)                (a12, moe_base.hpp:413-436)
 * then y_t = bf16( acc_t (+ f32(y_t_old) if incremental) ).
 *
 * Optional traces (may be NULL): tr_gate/tr_up/tr_act [T*k][I] and tr_down [T*k][H] bf16 per (t, j).
 * Returns 0, or -1 for an unsupported format. */
int ktxo_moe_forward(const ktxo_moe* m, int T, int k, const int64_t* ids, const float* w, const uint16_t* x,
                     uint16_t* y, int incremental, uint16_t* tr_gate, uint16_t* tr_up, uint16_t* tr_act,
                     uint16_t* tr_down) {
  if (m->fmt != KTXO_FMT_AMXINT4 && m->fmt != KTXO_FMT_AMXINT8) return -1;
  const int H = m->H, I = m->I;
  int8_t* xq = (int8_t*)malloc((size_t)H);
  int8_t* aq = (int8_t*)malloc((size_t)I);
  uint16_t* g = (uint16_t*)malloc(sizeof(uint16_t) * I);
  uint16_t* u = (uint16_t*)malloc(sizeof(uint16_t) * I);
  uint16_t* dn = (uint16_t*)malloc(sizeof(uint16_t) * H);
  float* acc = (float*)malloc(sizeof(float) * H);
  for (int t = 0; t < T; t++) {
    float xd;
    ktxo_quant_act_row(x + (size_t)t * H, H, xq, &xd);
    for (int e = 0; e < H; e++) acc[e] = 0.0f;
    for (int j = 0; j < k; j++) {
      int64_t id = ids[(size_t)t * k + j];
      if (skip_expert(m, id)) continue;
      const size_t wo = (size_t)id * I * H;
      gemv_int(xq, xd, m->gate_q + wo, m->gate_d + (size_t)id * I, I, H, g);
      gemv_int(xq, xd, m->up_q + wo, m->up_d + (size_t)id * I, I, H, u);
      if (tr_gate) memcpy(tr_gate + ((size_t)t * k + j) * I, g, sizeof(uint16_t) * I);
      if (tr_up) memcpy(tr_up + ((size_t)t * k + j) * I, u, sizeof(uint16_t) * I);
      for (int i = 0; i < I; i++) g[i] = ktxo_f32_to_bf16(ktxo_act_fn(ktxo_bf16_to_f32(g[i]), ktxo_bf16_to_f32(u[i])));
      if (tr_act) memcpy(tr_act + ((size_t)t * k + j) * I, g, sizeof(uint16_t) * I);
      float ad;
      ktxo_quant_act_row(g, I, aq, &ad);
      gemv_int(aq, ad, m->down_q + wo, m->down_d + (size_t)id * H, H, I, dn);
      if (tr_down) memcpy(tr_down + ((size_t)t * k + j) * H, dn, sizeof(uint16_t) * H);
      const float wt = w[(size_t)t * k + j];
      for (int e = 0; e < H; e++) acc[e] = fmaf(ktxo_bf16_to_f32(dn[e]), wt, acc[e]);
    }
    uint16_t* yt = y + (size_t)t * H;
    for (int e = 0; e < H; e++) {
      float v = acc[e];
      if (incremental) v = v + ktxo_bf16_to_f32(yt[e]);
      yt[e] = ktxo_f32_to_bf16(v);
    }
  }
  free(xq); free(aq); free(g); free(u); free(dn); free(acc);
  return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* FP8 (DeepSeek 128x128 block scales) and BF16 experts: bf16 activations, fp32 FMA chains            */
/* ------------------------------------------------------------------------------------------------ */

/* GemmKernel224FP8::fp8x64_to_bf16x64 byte LUTs (operators/amx/la/amx_raw_kernels.hpp:290-340): plain e4m3 decode
 * (bias 7, denormals exact, 0x7F/0xFF decode to +-480 — no NaN special case); every value is exact in bf16. */
float ktxo_e4m3_to_f32(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + (float)m / 8.0f, e - 7);
  return s ? -f : f;
}

/* One VDPBF16PS lane as issued by avx_kernel(_4) (amx_raw_kernels.hpp:334-505, 154-256): two fp32 FMAs. */
static inline float dpbf16(float c, float a0, float b0, float a1, float b1, int mode) {
  if (mode == 1) { c = fmaf(a0, b0, c); c = fmaf(a1, b1, c); return c; }
  if (mode == 2) {  /* both products added with a single rounding */
    return (float)((double)c + ((double)a0 * (double)b0 + (double)a1 * (double)b1));
  }
  c = fmaf(a1, b1, c); c = fmaf(a0, b0, c);
  return c;
}

/* float_mat_vec_kgroup + avx_kernel + apply_scale_kgroup (amx_raw_kernels.hpp:334-566): per 128-K group a sequential
 * fp32 chain r over the group's k pairs, then c = fma(r, scale_inv[n/128][g], c). */
static void gemv_fp8(const uint16_t* a_bf16, const uint8_t* w, const float* scale, int N, int K, int even_first,
                     uint16_t* out) {
  const int G = 128, kg = K / G;
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; n++) {
    const uint8_t* row = w + (size_t)n * K;
    float c = 0.0f;
    for (int g = 0; g < kg; g++) {
      float r = 0.0f;
      for (int j = g * G; j < (g + 1) * G; j += 2)
        r = dpbf16(r, ktxo_bf16_to_f32(a_bf16[j]), ktxo_e4m3_to_f32(row[j]), ktxo_bf16_to_f32(a_bf16[j + 1]),
                   ktxo_e4m3_to_f32(row[j + 1]), even_first);
      /* apply_scale_kgroup writes mul then add with intrinsics; g++ (-ffp-contract=fast, the GNU default, which is what
       * the reference's CMake build and oracle/_ref both use) contracts the pair into one FMA — verified against _ref. */
      c = fmaf(r, scale[(size_t)(n / G) * kg + g], c);
    }
    out[n] = ktxo_f32_to_bf16(c);
  }
}

/* float_mat_vec_perchannel (amx_raw_kernels.hpp:630-840, fp8-perchannel-moe.hpp:93-108): the e4m3 weights widened to bf16
 * exactly, ONE fp32 chain of bf16-pair products over the whole K (GemmKernel224FP8PerChannel::avx_kernel_4), then
 * apply_scale_perchannel: c = c * scale[n] (a plain multiply), then the bf16 rounding of the output buffer. */
static void gemv_fp8pc(const uint16_t* a_bf16, const uint8_t* w, const float* scale, int N, int K, int even_first,
                       uint16_t* out) {
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; n++) {
    const uint8_t* row = w + (size_t)n * K;
    float c = 0.0f;
    for (int j = 0; j < K; j += 2)
      c = dpbf16(c, ktxo_bf16_to_f32(a_bf16[j]), ktxo_e4m3_to_f32(row[j]), ktxo_bf16_to_f32(a_bf16[j + 1]),
                 ktxo_e4m3_to_f32(row[j + 1]), even_first);
    out[n] = ktxo_f32_to_bf16(c * scale[n]);
  }
}

/* GemmKernel224BF16::avx_kernel(_4) (amx_raw_kernels.hpp:93-256): one fp32 chain over the whole K. */
static void gemv_bf16(const uint16_t* a_bf16, const uint16_t* w, int N, int K, int even_first, uint16_t* out) {
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; n++) {
    const uint16_t* row = w + (size_t)n * K;
    float c = 0.0f;
    for (int j = 0; j < K; j += 2)
      c = dpbf16(c, ktxo_bf16_to_f32(a_bf16[j]), ktxo_bf16_to_f32(row[j]), ktxo_bf16_to_f32(a_bf16[j + 1]),
                 ktxo_bf16_to_f32(row[j + 1]), even_first);
    out[n] = ktxo_f32_to_bf16(c);
  }
}

/* The same frame as ktxo_moe_forward with QA = identity: activations stay bf16 between stages
 * (AMX_FP8_MOE_TP / AMX_BF16_MOE_TP: operators/amx/fp8-moe.hpp:93-108, bf16-moe.hpp; BufferABF16Impl::from_mat copies). */
int ktxo_moe_forward_fp(const ktxo_moe* m, int T, int k, const int64_t* ids, const float* w, const uint16_t* x,
                        uint16_t* y, int incremental) {
  if (m->fmt != KTXO_FMT_FP8 && m->fmt != KTXO_FMT_BF16 && m->fmt != KTXO_FMT_FP8PC) return -1;
  const int H = m->H, I = m->I, ef = m->dp_even_first;
  const int pc = m->fmt == KTXO_FMT_FP8PC;          /* per-channel scales: gate/up [E][I], down [E][H] */
  const int fp8 = m->fmt == KTXO_FMT_FP8 || pc;
  uint16_t* g = (uint16_t*)malloc(sizeof(uint16_t) * I);
  uint16_t* u = (uint16_t*)malloc(sizeof(uint16_t) * I);
  uint16_t* dn = (uint16_t*)malloc(sizeof(uint16_t) * H);
  float* acc = (float*)malloc(sizeof(float) * H);
  const size_t esz = fp8 ? 1 : 2;
  const size_t sgu = (size_t)(I / 128) * (H / 128), sdn = (size_t)(H / 128) * (I / 128);
  for (int t = 0; t < T; t++) {
    const uint16_t* xt = x + (size_t)t * H;
    for (int e = 0; e < H; e++) acc[e] = 0.0f;
    for (int j = 0; j < k; j++) {
      int64_t id = ids[(size_t)t * k + j];
      if (skip_expert(m, id)) continue;
      const size_t wo = (size_t)id * I * H * esz;
      if (pc) {
        gemv_fp8pc(xt, (const uint8_t*)m->gate_q + wo, m->gate_d + id * I, I, H, ef, g);
        gemv_fp8pc(xt, (const uint8_t*)m->up_q + wo, m->up_d + id * I, I, H, ef, u);
      } else if (fp8) {
        gemv_fp8(xt, (const uint8_t*)m->gate_q + wo, m->gate_d + id * sgu, I, H, ef, g);
        gemv_fp8(xt, (const uint8_t*)m->up_q + wo, m->up_d + id * sgu, I, H, ef, u);
      } else {
        gemv_bf16(xt, (const uint16_t*)((const uint8_t*)m->gate_q + wo), I, H, ef, g);
        gemv_bf16(xt, (const uint16_t*)((const uint8_t*)m->up_q + wo), I, H, ef, u);
      }
      for (int i = 0; i < I; i++) g[i] = ktxo_f32_to_bf16(ktxo_act_fn(ktxo_bf16_to_f32(g[i]), ktxo_bf16_to_f32(u[i])));
      if (pc) gemv_fp8pc(g, (const uint8_t*)m->down_q + wo, m->down_d + id * H, H, I, ef, dn);
      else if (fp8) gemv_fp8(g, (const uint8_t*)m->down_q + wo, m->down_d + id * sdn, H, I, ef, dn);
      else gemv_bf16(g, (const uint16_t*)((const uint8_t*)m->down_q + wo), H, I, ef, dn);
      const float wt = w[(size_t)t * k + j];
      for (int e = 0; e < H; e++) acc[e] = fmaf(ktxo_bf16_to_f32(dn[e]), wt, acc[e]);
    }
    uint16_t* yt = y + (size_t)t * H;
    for (int e = 0; e < H; e++) {
      float v = acc[e];
      if (incremental) v = v + ktxo_bf16_to_f32(yt[e]);
      yt[e] = ktxo_f32_to_bf16(v);
    }
  }
  free(g); free(u); free(dn); free(acc);
  return 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* RAWINT4 (Kimi-K2 compressed-tensors int4, group 32): a6', a8'                                     */
/* ------------------------------------------------------------------------------------------------ */

/* GemmKernel224Int4SmallKGroup::integer_mat_vec_kgroup / integer_mat_mat_kgroup
 * (operators/amx/la/amx_kernels.hpp:3385-3455 and :3453-3597) for one (row m, output n):
 *   weights: byte = ((q1+8)<<4) | (q0+8), even k in the low nibble; int8 multiplicand = (nibble ^ 8) << 4 = 16*q
 *   per 64-K block kb and AVX lane L (k = 64kb + 4L .. +3):  dot4 = sum a_q*w8  (exact int32)
 *   lane accumulator  s_L = fma(as[g]*bs[g], float(dot4), s_L),  g = 2kb + (L >= 8)     (g++ contracts add(mul))
 *   result = reduce_add(s_0..s_15) / 16  with _mm512_reduce_add_ps' tree: r_i = s_i + s_{i+8}; t_i = r_i + r_{i+4};
 *            u0 = t0 + t2, u1 = t1 + t3; u0 + u1.
 * use_fma selects the contracted form (what the g++-built reference executes). */
static float rawint4_dot(const int8_t* aq, const float* as, const uint8_t* wrow, const float* bs, int K, int use_fma) {
  float s[16];
  for (int L = 0; L < 16; L++) s[L] = 0.0f;
  for (int kb = 0; kb < K / 64; kb++) {
    for (int L = 0; L < 16; L++) {
      int32_t d = 0;
      for (int i = 0; i < 4; i++) {
        const int kk = kb * 64 + L * 4 + i;
        const uint8_t byte = wrow[kk >> 1];
        const int nib = (kk & 1) ? (byte >> 4) : (byte & 15);
        const int8_t w8 = (int8_t)(((nib ^ 8) << 4) & 0xF0);
        d += (int32_t)aq[kk] * (int32_t)w8;
      }
      const int g = kb * 2 + (L >= 8);
      const float sc = as[g] * bs[g];
      if (use_fma) s[L] = fmaf(sc, (float)d, s[L]);
      else { float t = sc * (float)d; s[L] = s[L] + t; }
    }
  }
  float r[8], t[4];
  for (int i = 0; i < 8; i++) r[i] = s[i] + s[i + 8];
  for (int i = 0; i < 4; i++) t[i] = r[i] + r[i + 4];
  const float u0 = t[0] + t[2], u1 = t[1] + t[3];
  return (u0 + u1) / 16.0f;
}

static void gemv_rawint4(const int8_t* aq, const float* as, const uint8_t* w, const float* bs, int N, int K, int use_fma,
                         uint16_t* out) {
  const int G = K / 32;
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; n++)
    out[n] = ktxo_f32_to_bf16(rawint4_dot(aq, as, w + (size_t)n * (K / 2), bs + (size_t)n * G, K, use_fma));
}

/* AMX_K2_MOE_TP (operators/amx/k2-moe.hpp:88-191) inside the common frame of moe_base.hpp: QA = per (row, 32-group) int8.
 * gate_q/up_q/down_q: packed nibbles [E][N][K/2]; *_d: fp32 scales [E][N][K/32] (bf16 in the checkpoint, widened at load,
 * k2-moe.hpp:172-187). */
int ktxo_moe_forward_rawint4(const ktxo_moe* m, int T, int k, const int64_t* ids, const float* w, const uint16_t* x,
                             uint16_t* y, int incremental) {
  if (m->fmt != KTXO_FMT_RAWINT4 || m->group != 32) return -1;
  const int H = m->H, I = m->I, fm = !m->dp_even_first;   /* knob reused: 0 = contracted (reference build) */
  int8_t* xq = (int8_t*)malloc((size_t)H);
  float* xs = (float*)malloc(sizeof(float) * (H / 32));
  int8_t* aq = (int8_t*)malloc((size_t)I);
  float* as = (float*)malloc(sizeof(float) * (I / 32));
  uint16_t* g = (uint16_t*)malloc(sizeof(uint16_t) * I);
  uint16_t* u = (uint16_t*)malloc(sizeof(uint16_t) * I);
  uint16_t* dn = (uint16_t*)malloc(sizeof(uint16_t) * H);
  float* acc = (float*)malloc(sizeof(float) * H);
  for (int t = 0; t < T; t++) {
    ktxo_quant_act_row_kgroup(x + (size_t)t * H, H, 32, xq, xs);
    for (int e = 0; e < H; e++) acc[e] = 0.0f;
    for (int j = 0; j < k; j++) {
      int64_t id = ids[(size_t)t * k + j];
      if (skip_expert(m, id)) continue;
      const size_t wo = (size_t)id * I * H / 2;
      gemv_rawint4(xq, xs, (const uint8_t*)m->gate_q + wo, m->gate_d + (size_t)id * I * (H / 32), I, H, fm, g);
      gemv_rawint4(xq, xs, (const uint8_t*)m->up_q + wo, m->up_d + (size_t)id * I * (H / 32), I, H, fm, u);
      for (int i = 0; i < I; i++) g[i] = ktxo_f32_to_bf16(ktxo_act_fn(ktxo_bf16_to_f32(g[i]), ktxo_bf16_to_f32(u[i])));
      ktxo_quant_act_row_kgroup(g, I, 32, aq, as);
      gemv_rawint4(aq, as, (const uint8_t*)m->down_q + wo, m->down_d + (size_t)id * H * (I / 32), H, I, fm, dn);
      const float wt = w[(size_t)t * k + j];
      for (int e = 0; e < H; e++) acc[e] = fmaf(ktxo_bf16_to_f32(dn[e]), wt, acc[e]);
    }
    uint16_t* yt = y + (size_t)t * H;
    for (int e = 0; e < H; e++) {
      float v = acc[e];
      if (incremental) v = v + ktxo_bf16_to_f32(yt[e]);
      yt[e] = ktxo_f32_to_bf16(v);
    }
  }
  free(xq); free(xs); free(aq); free(as); free(g); free(u); free(dn); free(acc);
  return 0;
}

/* a5 — token->expert bucketing, operators/amx/moe_base.hpp:208-227: per-expert histogram m_local_num_[e],
 * arrival rank m_local_pos_[t][j] (token-major, slot-minor), compacted list of active experts in ascending id.
 * pos[t*k+j] = -1 for skipped slots.  Returns the number of active experts. */
int ktxo_bucket(int E, const uint8_t* mask, int T, int k, const int64_t* ids, int32_t* num, int32_t* pos,
                int32_t* expert_id_map) {
  for (int e = 0; e < E; e++) num[e] = 0;
  for (int i = 0; i < T; i++)
    for (int j = 0; j < k; j++) {
      int64_t id = ids[(size_t)i * k + j];
      if (id < 0 || id >= E || (mask && mask[id])) { pos[(size_t)i * k + j] = -1; continue; }
      pos[(size_t)i * k + j] = num[id]++;
    }
  int active = 0;
  for (int e = 0; e < E; e++) if (num[e] > 0) expert_id_map[active++] = e;
  return active;
}
