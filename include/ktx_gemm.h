/* ktx_gemm.h — C ABI of the library's own prompt-sized BF16 GEMM (libktx_hip.so, gfx950).
 *
 * What it replaces.  Prompt-sized calls of the reference's GPU operators end in a dense tensor-core GEMM on bf16 operands:
 *   KLinearMarlin.forward   -> gptq_marlin_gemm(x, marlin_q_w, marlin_s, ...)  archive/ktransformers/operators/linear.py:676-714
 *                              (kt-kernel/cuda/gptq_marlin/gptq_marlin.cu:412 multiplies bf16 activations with bf16((q-8)*s))
 *   kv_b_proj expansion     -> torch.matmul on q_absorb / out_absorb           archive/ktransformers/operators/attention.py:77-194
 *   MoEGate logits          -> F.linear(x.float(), weight.float())             archive/ktransformers/models/modeling_deepseek_v3.py:434-437
 * Rounds 1-2 of this build handed those to a vendor GEMM (torch -> hipBLASLt).  This entry point is the library's own kernel:
 *
 *     Y[b][m][n] = round( sum_k A[b][m][k] * B[b][n][k]  (+ bias[n]) )        ("NT": both operands k-contiguous)
 *
 * bf16 operands, fp32 MFMA accumulation (mfma_f32_16x16x32_bf16), one rounding to bf16 (torch's round-to-nearest-even) or
 * the un-rounded fp32 sums (out_f32 != 0).  Batched with element strides a_bs / b_bs / y_bs (0 = the operand is shared by
 * every batch entry — the latent rows of the kv_b expansion).
 *
 * Constraints (checked; a violation returns non-zero with ktx_last_error set, nothing is launched): K % 64 == 0,
 * lda % 8 == 0, ldb % 8 == 0, A / B / Y 16-byte aligned, N % 8 == 0 (bf16 out) or N % 4 == 0 (fp32 out), ldy likewise.
 * M and N need not be multiples of the 128 x 128 tile.  Device pointers, enqueue-only on `stream`, HIP-graph capturable.
 */
#ifndef KTX_GEMM_H
#define KTX_GEMM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef KTX_STREAM_T_DEFINED
#define KTX_STREAM_T_DEFINED
typedef void* ktx_stream_t; /* hipStream_t */
#endif

typedef struct ktx_gemm_args {
  int32_t M, N, K, batch;
  const void* A; int64_t lda, a_bs;   /* bf16 [batch][M][lda]  */
  const void* B; int64_t ldb, b_bs;   /* bf16 [batch][N][ldb]  */
  void* Y; int64_t ldy, y_bs;         /* bf16 or fp32 [batch][M][ldy] */
  const void* bias;                   /* bf16 [N] or NULL; added in fp32 before the rounding */
  int32_t out_f32;                    /* 0: bf16 output, 1: fp32 output */
  int32_t variant;                    /* tile configuration, 0 = auto.  1: 128 x 128 x 64, one LDS stage (32 KiB, two barriers per
                                         k-step); 2: 128 x 128 x 64, two stages (the next k-step's LDS-DMA in flight under the
                                         MFMAs); 3: 256 x 128 x 32, two stages; 4: 256 x 128 x 64, one stage; 5 (round 6): 256 x 256 x 64,
                                         8 wavefronts, half-tile LDS ring with counted waits, the two wavefronts of a SIMD half a phase
                                         apart (operands of one batch entry < 4 GiB).  All five add the same products in the same
                                         order: bit-identical results (tests, A/B) */
} ktx_gemm_args;

int ktx_gemm_bf16_nt(const ktx_gemm_args* args, ktx_stream_t stream);

/* fp32 [rows][K] -> three bf16 planes [3][rows][K] with w == hi + mid + lo EXACTLY (8 + 8 + 8 mantissa bits, truncation):
 * a bf16 x fp32 product summed in fp32 becomes three bf16 GEMMs whose products are exact in the fp32 accumulator — the
 * router's F.linear(x.float(), weight.float()) on the MFMA units without a fp32 GEMM.  Device pointers. */
int ktx_split_f32_bf16x3(const float* w, int64_t n, void* planes, ktx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
