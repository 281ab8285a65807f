/* ktx_oracle_gguf.c — TEST INFRASTRUCTURE ONLY.  Parity: GEMM kernels PINNED against the reference's own code,
 * activation quantiser restated from ggml's published algorithm (*** that part unpinned ***).
 *
 * CPU restatement of the reference's llamafile/GGUF expert path:
 *   LLAMA_MOE_TP::forward_one / forward_many   kt-kernel/operators/llamafile/moe.hpp:271-460, 461-747
 *     bf16 input -> fp32 -> from_float(vec_dot_type = Q8_K) -> llamafile_sgemm(weights x Q8_K) -> fp32
 *     act_fn(g) * u in fp32 with expf (moe.hpp:269, 374) -> from_float(Q8_K) -> down sgemm -> fp32
 *     output[i] += down[i] * w in slot order (moe.hpp:447-449, 715-721) -> bf16 (moe-tp.hpp merge)
 *
 * Where the arithmetic lives and how each piece is pinned:
 *   * weights x Q8_K products (ktxo_vec_dot_q4_K / q6_K / iq1_s): the kernels the reference actually runs are in-tree,
 *     third_party/llamafile/iqk_mul_mat.inc (reached through llamafile_sgemm, tinyblas_cpu_sgemm.inc:331-335).  They are
 *     compiled UNMODIFIED into oracle/_ref/libiqk_ref_{avx2,zen4}.so (oracle/Makefile `iqk`; the empty llama.cpp submodule's
 *     type definitions come from oracle/shim_iqk/) and tests/test_gguf_ref_pin_cpu.py drives both variants and these
 *     functions on the same blocks: identical integer sub-block sums by construction, fp32 results within the stated
 *     re-association bound of the exact value and within 3e-7 of the output scale of each other (median 2-3 ulp) — the
 *     order of the fp32 adds differs (iqk: 8 AVX lanes of partial sums per output, added at the end; here, as in the HIP
 *     kernel, one fma per 256-block: acc = fma(d8*d, isum, acc); acc = fma(-(d8*dmin), msum, acc)).
 *   * block LAYOUTS (de-quantised values): checked against the reference's own numpy dequantize_q4_k / dequantize_q6_k
 *     (tests/golden/make_gguf_golden.py -> tests/golden/gguf_blocks_golden.npz); IQ1_S through the iqk kernel above.
 *   * quantize_row_q8_K (activations -> Q8_K): lives in the un-vendored ggerganov/llama.cpp submodule (third_party/llama.cpp
 *     is empty; pin not recorded in the tree — the reference's numpy dequantisers cite ggml commit
 *     fca1caafea7de9fbd7efc733b9818f9cf2da3050, archive/ktransformers/util/custom_gguf.py:326).  Restated from ggml's
 *     published quantize_row_q8_K_ref; no in-tree code or vector pins it.  The reference's only numeric test for this path
 *     is `diff < 0.5` against torch (kt-kernel/examples/test_moe.py:203-206), which needs the built extension.
 *   * legacy types Q4_0 / Q5_0 / Q8_0 (round 5): ggml pairs them with Q8_0 activations (vec_dot_type).  The products the reference
 *     reaches are iqk's mul_mat_qX_0_q8_0_T (iqk_mul_mat.inc:2201-2213 — Q4_0, Q5_0: pinned through libiqk_ref like the k-quants) and,
 *     for Q8_0 weights (iqk declines Q8_0 x Q8_0, :3228-3236), tinyBLAS_Q0_AVX2 (tinyblas_cpu.h:828-1010; restated from the in-tree
 *     source, not compiled here).  Both keep 8 AVX lanes of per-block terms and add them at the end; here one fma per 32-block in k
 *     order — same integers, another fp32 association (the k-quants' bound).  quantize_row_q8_0 (activations) is ggml's x86
 *     arithmetic (d = fp16(amax/127), q = rne(x * (127/amax))) restated from the published source: *** unpinned *** like q8_K.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define QK_K 256
#define GGML_TYPE_Q2_K 10
#define GGML_TYPE_Q3_K 11
#define GGML_TYPE_Q4_K 12
#define GGML_TYPE_Q5_K 13
#define GGML_TYPE_IQ4_XS 23
#define GGML_TYPE_Q6_K 14
#define GGML_TYPE_IQ1_S 19
#define GGML_TYPE_Q4_0 2
#define GGML_TYPE_Q5_0 6
#define GGML_TYPE_Q8_0 8

/* IQ1_S codebook (format constant), packed 2 bits per weight: see tests/golden/make_iq1s_grid.py */
static const uint16_t iq1s_grid_packed[2048] = {
#include "iq1s_grid.inc"
};

float ktxo_bf16_to_f32(uint16_t h);
uint16_t ktxo_f32_to_bf16(float f);

static float fp16_to_f32(uint16_t h) {
  const uint32_t s = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff;
  uint32_t u;
  if (e == 0) {
    if (m == 0) u = s;
    else {  /* subnormal */
      int sh = 0; uint32_t mm = m;
      while (!(mm & 0x400)) { mm <<= 1; sh++; }
      u = s | ((uint32_t)(113 - sh) << 23) | ((mm & 0x3ff) << 13);
    }
  } else if (e == 31) u = s | 0x7f800000u | (m << 13);
  else u = s | ((e + 112) << 23) | (m << 13);
  float f; memcpy(&f, &u, 4); return f;
}

/* ggml-quants.c nearest_int(): round to nearest even via the 1.5*2^23 trick */
static inline int nearest_int(float fval) {
  float val = fval + 12582912.f;
  int i; memcpy(&i, &val, sizeof(int));
  return (i & 0x007fffff) - 0x00400000;
}

/* ggml-quants.c quantize_row_q8_K_ref: per 256: max = the element of largest magnitude (first on ties),
 * iscale = -127/max, q = min(127, nearest_int(iscale*x)), d = 1/iscale, bsums = sums of 16. */
void ktxo_quantize_row_q8_K(const float* x, int K, int8_t* q, float* d, int16_t* bsums) {
  for (int b = 0; b < K / QK_K; b++, x += QK_K, q += QK_K, bsums += 16) {
    float max = 0, amax = 0;
    for (int j = 0; j < QK_K; j++) { const float ax = fabsf(x[j]); if (ax > amax) { amax = ax; max = x[j]; } }
    if (!amax) { d[b] = 0; memset(q, 0, QK_K); memset(bsums, 0, 32); continue; }
    const float iscale = -127.f / max;
    for (int j = 0; j < QK_K; j++) { int v = nearest_int(iscale * x[j]); q[j] = (int8_t)(v < 127 ? v : 127); }
    for (int j = 0; j < 16; j++) { int s = 0; for (int i = 0; i < 16; i++) s += q[j * 16 + i]; bsums[j] = (int16_t)s; }
    d[b] = 1 / iscale;
  }
}

/* fp32 -> fp16, round to nearest even (what GGML_FP32_TO_FP16 = _cvtss_sh(x, 0) does on the reference's x86 hosts) */
static uint16_t f32_to_fp16(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));   /* inf / nan */
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                      /* rounds to inf (>= 65520) */
  if (x < 0x33000001u) return (uint16_t)sign;                                                   /* <= 2^-25: rounds to zero */
  int e = (int)(x >> 23) - 127;
  uint32_t m = (x & 0x7fffffu) | 0x800000u;
  int shift = e < -14 ? 13 + (-14 - e) : 13;          /* subnormal halves lose more bits */
  const uint32_t half = 1u << (shift - 1), rest = m & ((1u << shift) - 1);
  uint32_t r = m >> shift;
  if (rest > half || (rest == half && (r & 1))) r++;
  if (e < -14) return (uint16_t)(sign | r);           /* (a carry into bit 10 lands on the smallest normal: correct) */
  return (uint16_t)(sign | (((uint32_t)(e + 15) << 10) + (r - 0x400u)));   /* a mantissa carry bumps the exponent */
}

/* ggml-quants.c quantize_row_q8_0 (x86 AVX2 arithmetic): per 32: d = amax / 127 stored as fp16; id = amax ? 127 / amax : 0;
 * q = round-to-nearest-even(x * id).  d[] receives the fp32 VALUE of the stored fp16 (what the dot products multiply with). */
void ktxo_quantize_row_q8_0(const float* x, int K, int8_t* q, float* d) {
  for (int b = 0; b < K / 32; b++, x += 32, q += 32) {
    float amax = 0;
    for (int j = 0; j < 32; j++) { const float ax = fabsf(x[j]); if (ax > amax) amax = ax; }
    const float dd = amax / 127.f;
    const float id = amax != 0.0f ? 127.f / amax : 0.0f;
    d[b] = fp16_to_f32(f32_to_fp16(dd));
    for (int j = 0; j < 32; j++) q[j] = (int8_t)nearest_int(x[j] * id);
  }
}

/* block_q4_0 { fp16 d; uint8 qs[16] } (18 B): element j = (qs[j] & 15) - 8, element j + 16 = (qs[j] >> 4) - 8.
 * One fma per block in k order: acc = fma(d_w * d_x, float(exact int dot), acc). */
float ktxo_vec_dot_q4_0(const uint8_t* wrow, int K, const int8_t* q8, const float* d8) {
  float acc = 0;
  for (int b = 0; b < K / 32; b++, wrow += 18, q8 += 32) {
    uint16_t dh; memcpy(&dh, wrow, 2);
    const uint8_t* qs = wrow + 2;
    int s = 0;
    for (int j = 0; j < 16; j++) s += ((qs[j] & 15) - 8) * q8[j] + ((qs[j] >> 4) - 8) * q8[j + 16];
    acc = fmaf(fp16_to_f32(dh) * d8[b], (float)s, acc);
  }
  return acc;
}
/* block_q5_0 { fp16 d; uint8 qh[4]; uint8 qs[16] } (22 B): element e = ((nibble_e) | (bit e of qh) << 4) - 16 */
float ktxo_vec_dot_q5_0(const uint8_t* wrow, int K, const int8_t* q8, const float* d8) {
  float acc = 0;
  for (int b = 0; b < K / 32; b++, wrow += 22, q8 += 32) {
    uint16_t dh; memcpy(&dh, wrow, 2);
    uint32_t qh; memcpy(&qh, wrow + 2, 4);
    const uint8_t* qs = wrow + 6;
    int s = 0;
    for (int j = 0; j < 16; j++) {
      const int lo = ((qs[j] & 15) | (((qh >> j) & 1) << 4)) - 16, hi = ((qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4)) - 16;
      s += lo * q8[j] + hi * q8[j + 16];
    }
    acc = fmaf(fp16_to_f32(dh) * d8[b], (float)s, acc);
  }
  return acc;
}
/* block_q8_0 { fp16 d; int8 qs[32] } (34 B) */
float ktxo_vec_dot_q8_0(const uint8_t* wrow, int K, const int8_t* q8, const float* d8) {
  float acc = 0;
  for (int b = 0; b < K / 32; b++, wrow += 34, q8 += 32) {
    uint16_t dh; memcpy(&dh, wrow, 2);
    const int8_t* qs = (const int8_t*)(wrow + 2);
    int s = 0;
    for (int j = 0; j < 32; j++) s += qs[j] * q8[j];
    acc = fmaf(fp16_to_f32(dh) * d8[b], (float)s, acc);
  }
  return acc;
}
static int is_legacy(int t) { return t == GGML_TYPE_Q4_0 || t == GGML_TYPE_Q5_0 || t == GGML_TYPE_Q8_0; }
static size_t legacy_row_bytes(int type, int K) { return (size_t)(K / 32) * (type == GGML_TYPE_Q4_0 ? 18 : type == GGML_TYPE_Q5_0 ? 22 : 34); }
static float legacy_vec_dot(int type, const uint8_t* wrow, int K, const int8_t* q8, const float* d8) {
  return type == GGML_TYPE_Q4_0 ? ktxo_vec_dot_q4_0(wrow, K, q8, d8) : type == GGML_TYPE_Q5_0 ? ktxo_vec_dot_q5_0(wrow, K, q8, d8)
                                                                                             : ktxo_vec_dot_q8_0(wrow, K, q8, d8);
}

/* ggml-quants.c get_scale_min_k4 */
static inline void get_scale_min_k4(int j, const uint8_t* q, uint8_t* d, uint8_t* m) {
  if (j < 4) { *d = q[j] & 63; *m = q[j + 4] & 63; }
  else { *d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); *m = (q[j + 4] >> 4) | ((q[j - 0] >> 6) << 4); }
}

/* block_q4_K: { fp16 d, fp16 dmin, uint8 scales[12], uint8 qs[128] } = 144 B (custom_gguf.py:326-343) */
float ktxo_vec_dot_q4_K(const uint8_t* wrow, int K, const int8_t* q8, const float* d8, const int16_t* bs) {
  float acc = 0.f;
  for (int b = 0; b < K / QK_K; b++, wrow += 144, q8 += QK_K, bs += 16) {
    uint16_t dh, mh; memcpy(&dh, wrow, 2); memcpy(&mh, wrow + 2, 2);
    const float d = fp16_to_f32(dh), dmin = fp16_to_f32(mh);
    const uint8_t *sc = wrow + 4, *qs = wrow + 16;
    int32_t isum = 0, msum = 0;
    for (int j = 0; j < 8; j++) {
      uint8_t s, m; get_scale_min_k4(j, sc, &s, &m);
      const uint8_t* q = qs + (j / 2) * 32;
      int32_t dot = 0;
      for (int l = 0; l < 32; l++) dot += (int32_t)((j & 1) ? (q[l] >> 4) : (q[l] & 0xF)) * q8[j * 32 + l];
      isum += (int32_t)s * dot;
      msum += (int32_t)m * ((int32_t)bs[2 * j] + bs[2 * j + 1]);
    }
    acc = fmaf(d8[b] * d, (float)isum, acc);
    acc = fmaf(-(d8[b] * dmin), (float)msum, acc);
  }
  return acc;
}

/* block_q5_K: { fp16 d, fp16 dmin, uint8 scales[12], uint8 qh[32], uint8 qs[128] } = 176 B (custom_gguf.py:356-410
 * dequantize_q5_k): Q4_K's structure with a fifth bit per weight — sub-block j of 32 takes bit j of qh[l].  Same folding as
 * Q4_K.  (No HIP kernel reads this type yet: the product de-quantises such experts to BF16; this
 * restatement and its pin against the reference's iqk kernel are there for the native kernel to be checked against.) */
float ktxo_vec_dot_q5_K(const uint8_t* wrow, int K, const int8_t* q8, const float* d8, const int16_t* bs) {
  float acc = 0.f;
  for (int b = 0; b < K / QK_K; b++, wrow += 176, q8 += QK_K, bs += 16) {
    uint16_t dh, mh; memcpy(&dh, wrow, 2); memcpy(&mh, wrow + 2, 2);
    const float d = fp16_to_f32(dh), dmin = fp16_to_f32(mh);
    const uint8_t *sc = wrow + 4, *qh = wrow + 16, *qs = wrow + 48;
    int32_t isum = 0, msum = 0;
    for (int j = 0; j < 8; j++) {
      uint8_t s, m; get_scale_min_k4(j, sc, &s, &m);
      const uint8_t* q = qs + (j / 2) * 32;
      int32_t dot = 0;
      for (int l = 0; l < 32; l++) {
        const int lo = (j & 1) ? (q[l] >> 4) : (q[l] & 0xF);
        dot += (int32_t)(lo + (((qh[l] >> j) & 1) << 4)) * q8[j * 32 + l];
      }
      isum += (int32_t)s * dot;
      msum += (int32_t)m * ((int32_t)bs[2 * j] + bs[2 * j + 1]);
    }
    acc = fmaf(d8[b] * d, (float)isum, acc);
    acc = fmaf(-(d8[b] * dmin), (float)msum, acc);
  }
  return acc;
}

/* The three remaining Q8_K-activation types the reference's loader knows (custom_gguf.py:225-519).  Like Q5_K: checker only,
 * pinned against the reference's iqk kernels; the product de-quantises such experts to BF16 until native kernels exist.
 *
 * block_q2_K: { uint8 scales[16] (4-bit scale | 4-bit min), uint8 qs[64], fp16 d, fp16 dmin } = 84 B; 16 sub-blocks of 16;
 * element half*128 + shift*32 + l = (qs[half*32 + l] >> 2*shift) & 3. */
float ktxo_vec_dot_q2_K(const uint8_t* wrow, int K, const int8_t* q8, const float* d8, const int16_t* bs) {
  float acc = 0.f;
  for (int b = 0; b < K / QK_K; b++, wrow += 84, q8 += QK_K, bs += 16) {
    const uint8_t *sc = wrow, *qs = wrow + 16;
    uint16_t dh, mh; memcpy(&dh, wrow + 80, 2); memcpy(&mh, wrow + 82, 2);
    const float d = fp16_to_f32(dh), dmin = fp16_to_f32(mh);
    int32_t isum = 0, msum = 0;
    for (int g = 0; g < 16; g++) {
      const int half = g / 8, shift = (g % 8) / 2, l0 = (g & 1) * 16;
      int32_t dot = 0;
      for (int l = 0; l < 16; l++) dot += (int32_t)((qs[half * 32 + l0 + l] >> (2 * shift)) & 3) * q8[g * 16 + l];
      isum += (int32_t)(sc[g] & 0xF) * dot;
      msum += (int32_t)(sc[g] >> 4) * bs[g];
    }
    acc = fmaf(d8[b] * d, (float)isum, acc);
    acc = fmaf(-(d8[b] * dmin), (float)msum, acc);
  }
  return acc;
}

/* block_q3_K: { uint8 hmask[32], uint8 qs[64], uint8 scales[12] (16 six-bit scales, offset 32), fp16 d } = 110 B; element
 * half*128 + shift*32 + l = ((qs[half*32 + l] >> 2*shift) & 3) - (bit (half*4 + shift) of hmask[l] ? 0 : 4). */
float ktxo_vec_dot_q3_K(const uint8_t* wrow, int K, const int8_t* q8, const float* d8, const int16_t* bs) {
  (void)bs;
  float acc = 0.f;
  for (int b = 0; b < K / QK_K; b++, wrow += 110, q8 += QK_K) {
    const uint8_t *hm = wrow, *qs = wrow + 32, *lo = wrow + 96, *hi2 = wrow + 104;
    uint16_t dh; memcpy(&dh, wrow + 108, 2);
    const float d = fp16_to_f32(dh);
    int32_t isum = 0;
    for (int g = 0; g < 16; g++) {
      const int s6 = g < 8 ? (lo[g] & 0xF) | (((hi2[g % 4] >> (2 * (g / 4))) & 3) << 4)
                           : (lo[g - 8] >> 4) | (((hi2[(g - 8) % 4] >> (4 + 2 * ((g - 8) / 4))) & 3) << 4);
      const int half = g / 8, shift = (g % 8) / 2, l0 = (g & 1) * 16;
      int32_t dot = 0;
      for (int l = 0; l < 16; l++) {
        const int q = (int)((qs[half * 32 + l0 + l] >> (2 * shift)) & 3) - (((hm[l0 + l] >> (half * 4 + shift)) & 1) ? 0 : 4);
        dot += q * q8[g * 16 + l];
      }
      isum += (s6 - 32) * dot;
    }
    acc = fmaf(d8[b] * d, (float)isum, acc);
  }
  return acc;
}

/* block_iq4_xs: { fp16 d, uint16 scales_h, uint8 scales_l[4], uint8 qs[128] } = 136 B; 8 sub-blocks of 32 (16 low nibbles,
 * then 16 high nibbles of 16 bytes), values from the 16-entry IQ4_NL table, 6-bit scale - 32. */
float ktxo_vec_dot_iq4_xs(const uint8_t* wrow, int K, const int8_t* q8, const float* d8, const int16_t* bs) {
  (void)bs;
  static const int8_t kv[16] = {-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113};
  float acc = 0.f;
  for (int b = 0; b < K / QK_K; b++, wrow += 136, q8 += QK_K) {
    uint16_t dh, sh; memcpy(&dh, wrow, 2); memcpy(&sh, wrow + 2, 2);
    const float d = fp16_to_f32(dh);
    const uint8_t *sl = wrow + 4, *qs = wrow + 8;
    int32_t isum = 0;
    for (int ib = 0; ib < 8; ib++) {
      const int ls = ((sl[ib / 2] >> (4 * (ib % 2))) & 0xF) | (((sh >> (2 * ib)) & 3) << 4);
      int32_t dot = 0;
      for (int l = 0; l < 16; l++)
        dot += (int32_t)kv[qs[ib * 16 + l] & 0xF] * q8[ib * 32 + l] + (int32_t)kv[qs[ib * 16 + l] >> 4] * q8[ib * 32 + 16 + l];
      isum += (ls - 32) * dot;
    }
    acc = fmaf(d8[b] * d, (float)isum, acc);
  }
  return acc;
}

/* block_q6_K: { uint8 ql[128], uint8 qh[64], int8 scales[16], fp16 d } = 210 B (custom_gguf.py dequantize_q6_k) */
float ktxo_vec_dot_q6_K(const uint8_t* wrow, int K, const int8_t* q8, const float* d8, const int16_t* bs) {
  (void)bs;
  float acc = 0.f;
  for (int b = 0; b < K / QK_K; b++, wrow += 210, q8 += QK_K) {
    const uint8_t *ql = wrow, *qh = wrow + 128;
    const int8_t* sc = (const int8_t*)(wrow + 192);
    uint16_t dh; memcpy(&dh, wrow + 208, 2);
    const float d = fp16_to_f32(dh);
    int32_t isum = 0;
    for (int h = 0; h < 2; h++) {            /* two halves of 128 */
      int32_t dots[8] = {0};
      for (int l = 0; l < 32; l++) {
        const int is = l / 16;
        const int q1 = (int)((ql[h * 64 + l] & 0xF) | (((qh[h * 32 + l] >> 0) & 3) << 4)) - 32;
        const int q2 = (int)((ql[h * 64 + l + 32] & 0xF) | (((qh[h * 32 + l] >> 2) & 3) << 4)) - 32;
        const int q3 = (int)((ql[h * 64 + l] >> 4) | (((qh[h * 32 + l] >> 4) & 3) << 4)) - 32;
        const int q4 = (int)((ql[h * 64 + l + 32] >> 4) | (((qh[h * 32 + l] >> 6) & 3) << 4)) - 32;
        dots[is + 0] += q1 * q8[h * 128 + l];
        dots[is + 2] += q2 * q8[h * 128 + l + 32];
        dots[is + 4] += q3 * q8[h * 128 + l + 64];
        dots[is + 6] += q4 * q8[h * 128 + l + 96];
      }
      for (int j = 0; j < 8; j++) isum += (int32_t)sc[h * 8 + j] * dots[j];
    }
    acc = fmaf(d8[b] * d, (float)isum, acc);
  }
  return acc;
}

/* block_iq1_s: { fp16 d, uint8 qs[32], uint16 qh[8] } = 50 B.  The only in-tree definition of its arithmetic is the
 * reference's AVX2 kernel mul_mat_iq1_s_q8_K (third_party/llamafile/iqk_mul_mat.inc:2689-2770), restated here op for op in
 * scalar form: per 32-weight sub-block ib: scale = 2*((qh[ib]>>12)&7)+1, delta = (qh[ib]&0x8000) ? -9 : -7, the four grid
 * points idx_l = qs[4*ib+l] | (((qh[ib] >> 3*l) & 7) << 8) give 8 unsigned weights (0,1,2) each;
 *   sumi = sum_ib 8*scale*dot(grid_us, q8) + sum_ib scale*delta*bsum32;  acc = fma(d*d8, float(sumi), acc);  result = 0.125*acc
 * i.e. w = d * scale * (g +- 0.125), g in {-1,0,+1}.  (iqk accumulates the 8 AVX lanes of sumi separately in fp32 and adds
 * them at the end; here the lanes are added in int32 first.) */
float ktxo_vec_dot_iq1_s(const uint8_t* wrow, int K, const int8_t* q8, const float* d8, const int16_t* bs) {
  float acc = 0.f;
  for (int b = 0; b < K / QK_K; b++, wrow += 50, q8 += QK_K, bs += 16) {
    uint16_t dh; memcpy(&dh, wrow, 2);
    const float d = fp16_to_f32(dh);
    const uint8_t* qs = wrow + 2;
    uint16_t qh[8]; memcpy(qh, wrow + 34, 16);
    int32_t sumi = 0;
    for (int ib = 0; ib < 8; ib++) {
      const int scale = 2 * ((qh[ib] >> 12) & 7) + 1;
      const int delta = (qh[ib] & 0x8000) ? -9 : -7;
      int32_t dot = 0;
      for (int l = 0; l < 4; l++) {
        const uint16_t g = iq1s_grid_packed[qs[4 * ib + l] | (((qh[ib] >> (3 * l)) & 7) << 8)];
        for (int e = 0; e < 8; e++) dot += (int32_t)((g >> (2 * e)) & 3) * q8[ib * 32 + l * 8 + e];
      }
      sumi += 8 * scale * dot + scale * delta * ((int32_t)bs[2 * ib] + bs[2 * ib + 1]);
    }
    acc = fmaf(d * d8[b], (float)sumi, acc);
  }
  return 0.125f * acc;
}

/* de-quantised values of IQ1_S rows (tests): w = d * scale * (g +- 0.125) */
void ktxo_dequant_iq1_s(const uint8_t* blocks, int nblocks, float* out) {
  for (int b = 0; b < nblocks; b++, blocks += 50, out += QK_K) {
    uint16_t dh; memcpy(&dh, blocks, 2);
    const float d = fp16_to_f32(dh);
    const uint8_t* qs = blocks + 2;
    uint16_t qh[8]; memcpy(qh, blocks + 34, 16);
    for (int ib = 0; ib < 8; ib++) {
      const float dl = d * (float)(2 * ((qh[ib] >> 12) & 7) + 1);
      const float delta = (qh[ib] & 0x8000) ? -0.125f : 0.125f;
      for (int l = 0; l < 4; l++) {
        const uint16_t g = iq1s_grid_packed[qs[4 * ib + l] | (((qh[ib] >> (3 * l)) & 7) << 8)];
        for (int e = 0; e < 8; e++) out[ib * 32 + l * 8 + e] = dl * ((float)((int)((g >> (2 * e)) & 3) - 1) + delta);
      }
    }
  }
}

const uint16_t* ktxo_iq1s_grid(void) { return iq1s_grid_packed; }

static size_t row_bytes(int type, int K) {
  return (size_t)(K / QK_K) * (type == GGML_TYPE_Q4_K ? 144 : type == GGML_TYPE_Q5_K ? 176 : type == GGML_TYPE_Q6_K ? 210 :
                               type == GGML_TYPE_Q2_K ? 84 : type == GGML_TYPE_Q3_K ? 110 : type == GGML_TYPE_IQ4_XS ? 136 : 50);
}

static float vec_dot(int type, const uint8_t* wrow, int K, const int8_t* q8, const float* d8, const int16_t* bs) {
  if (type == GGML_TYPE_IQ1_S) return ktxo_vec_dot_iq1_s(wrow, K, q8, d8, bs);
  if (type == GGML_TYPE_Q5_K) return ktxo_vec_dot_q5_K(wrow, K, q8, d8, bs);
  if (type == GGML_TYPE_Q2_K) return ktxo_vec_dot_q2_K(wrow, K, q8, d8, bs);
  if (type == GGML_TYPE_Q3_K) return ktxo_vec_dot_q3_K(wrow, K, q8, d8, bs);
  if (type == GGML_TYPE_IQ4_XS) return ktxo_vec_dot_iq4_xs(wrow, K, q8, d8, bs);
  return type == GGML_TYPE_Q4_K ? ktxo_vec_dot_q4_K(wrow, K, q8, d8, bs) : ktxo_vec_dot_q6_K(wrow, K, q8, d8, bs);
}

typedef struct {
  int32_t E, H, I, gate_type, up_type, down_type;
  const uint8_t *gate, *up, *down;      /* raw GGUF blocks [E][N][K/256 blocks] */
  const uint8_t* gpu_experts_mask;      /* nullable */
} ktxo_gguf_moe;

/* x bf16 [T][H], ids [T][k], w [T][k] -> y bf16 [T][H]; inter_out (nullable) receives the fp32 intermediate of token 0's
 * slots [k][I] for white-box tests.  Returns 0, or -1 for an unsupported type. */
int ktxo_moe_forward_gguf(const ktxo_gguf_moe* m, int T, int k, const int64_t* ids, const float* w, const uint16_t* x,
                          uint16_t* y, float* inter_out) {
  const int H = m->H, I = m->I;
  const int types[3] = {m->gate_type, m->up_type, m->down_type};
  if (is_legacy(types[0]) && is_legacy(types[1]) && is_legacy(types[2])) {
    /* the same control flow with ggml's partner format of the legacy types: Q8_0 (scale per 32) */
    float* xf = malloc(sizeof(float) * H);
    int8_t* xq = malloc(H); float* xd = malloc(sizeof(float) * (H / 32));
    float* inter = malloc(sizeof(float) * I);
    int8_t* aq = malloc(I); float* ad = malloc(sizeof(float) * (I / 32));
    float* out = malloc(sizeof(float) * H);
    for (int t = 0; t < T; t++) {
      for (int i = 0; i < H; i++) xf[i] = ktxo_bf16_to_f32(x[(size_t)t * H + i]);
      ktxo_quantize_row_q8_0(xf, H, xq, xd);
      for (int i = 0; i < H; i++) out[i] = 0;
      for (int j = 0; j < k; j++) {
        const int64_t e = ids[(size_t)t * k + j];
        if (e < 0 || e >= m->E || (m->gpu_experts_mask && m->gpu_experts_mask[e])) continue;
        const size_t rg = legacy_row_bytes(m->gate_type, H), ru = legacy_row_bytes(m->up_type, H), rd = legacy_row_bytes(m->down_type, I);
        const uint8_t* g = m->gate + (size_t)e * I * rg;
        const uint8_t* u = m->up + (size_t)e * I * ru;
        const uint8_t* dn = m->down + (size_t)e * H * rd;
        for (int i = 0; i < I; i++) {
          const float gv = legacy_vec_dot(m->gate_type, g + (size_t)i * rg, H, xq, xd);
          const float uv = legacy_vec_dot(m->up_type, u + (size_t)i * ru, H, xq, xd);
          inter[i] = (gv / (1.0f + expf(-gv))) * uv;
        }
        if (inter_out && t == 0) memcpy(inter_out + (size_t)j * I, inter, sizeof(float) * I);
        ktxo_quantize_row_q8_0(inter, I, aq, ad);
        const float ew = w[(size_t)t * k + j];
        for (int i = 0; i < H; i++) out[i] += legacy_vec_dot(m->down_type, dn + (size_t)i * rd, I, aq, ad) * ew;
      }
      for (int i = 0; i < H; i++) y[(size_t)t * H + i] = ktxo_f32_to_bf16(out[i]);
    }
    free(xf); free(xq); free(xd); free(inter); free(aq); free(ad); free(out);
    return 0;
  }
  for (int i = 0; i < 3; i++) if (types[i] != GGML_TYPE_Q4_K && types[i] != GGML_TYPE_Q5_K && types[i] != GGML_TYPE_Q6_K && types[i] != GGML_TYPE_IQ1_S &&
                                  types[i] != GGML_TYPE_Q2_K && types[i] != GGML_TYPE_Q3_K && types[i] != GGML_TYPE_IQ4_XS) return -1;
  float* xf = malloc(sizeof(float) * H);
  int8_t* xq = malloc(H); float* xd = malloc(sizeof(float) * (H / QK_K)); int16_t* xbs = malloc(2 * (H / 16));
  float* inter = malloc(sizeof(float) * I);
  int8_t* aq = malloc(I); float* ad = malloc(sizeof(float) * (I / QK_K)); int16_t* abs_ = malloc(2 * (I / 16));
  float* out = malloc(sizeof(float) * H);
  for (int t = 0; t < T; t++) {
    for (int i = 0; i < H; i++) xf[i] = ktxo_bf16_to_f32(x[(size_t)t * H + i]);          /* to_float (moe.hpp:283) */
    ktxo_quantize_row_q8_K(xf, H, xq, xd, xbs);                                           /* from_float(Q8_K) (:286) */
    for (int i = 0; i < H; i++) out[i] = 0;                                               /* moe.hpp:419-421 */
    for (int j = 0; j < k; j++) {
      const int64_t e = ids[(size_t)t * k + j];
      if (e < 0 || e >= m->E || (m->gpu_experts_mask && m->gpu_experts_mask[e])) continue; /* should_skip_expert */
      const uint8_t* g = m->gate + (size_t)e * I * row_bytes(m->gate_type, H);
      const uint8_t* u = m->up + (size_t)e * I * row_bytes(m->up_type, H);
      const uint8_t* dn = m->down + (size_t)e * H * row_bytes(m->down_type, I);
      for (int i = 0; i < I; i++) {
        const float gv = vec_dot(m->gate_type, g + (size_t)i * row_bytes(m->gate_type, H), H, xq, xd, xbs);
        const float uv = vec_dot(m->up_type, u + (size_t)i * row_bytes(m->up_type, H), H, xq, xd, xbs);
        inter[i] = (gv / (1.0f + expf(-gv))) * uv;                                        /* act_fn(g) * u (:269,374) */
      }
      if (inter_out && t == 0) memcpy(inter_out + (size_t)j * I, inter, sizeof(float) * I);
      ktxo_quantize_row_q8_K(inter, I, aq, ad, abs_);                                     /* from_float (:388) */
      const float ew = w[(size_t)t * k + j];
      for (int i = 0; i < H; i++) {
        const float dv = vec_dot(m->down_type, dn + (size_t)i * row_bytes(m->down_type, I), I, aq, ad, abs_);
        out[i] += dv * ew;                                                                /* moe.hpp:447-449 */
      }
    }
    for (int i = 0; i < H; i++) y[(size_t)t * H + i] = ktxo_f32_to_bf16(out[i]);
  }
  free(xf); free(xq); free(xd); free(xbs); free(inter); free(aq); free(ad); free(abs_); free(out);
  return 0;
}
