/* ktx_moe.h — C ABI of the MI355X-native routed-expert MoE forward (libktx_hip.so).
 *
 * Drop-in boundary for the reference's native MoE task protocol (inner seam):
 *
 *   reference (CPU)                                            this library (HBM-resident, gfx950)
 *   ---------------------------------------------------------  -----------------------------------------
 *   MOEConfig field bag   kt-kernel/ext_bindings.cpp:746-831   ktx_moe_config
 *   AMXInt4_MOE(config)   kt-kernel/ext_bindings.cpp:447-517   ktx_moe_create
 *   load_weights_task()   kt-kernel/ext_bindings.cpp:196-219   ktx_moe_load_bf16 / ktx_moe_load_quantized
 *     (online bf16->int4: operators/amx/moe.hpp:352-387;
 *      pre-quantised:     operators/amx/moe.hpp:266-300)
 *   forward_task(bsz_ptr,k,ids,w,in,out,incremental)           ktx_moe_forward
 *                         kt-kernel/ext_bindings.cpp:220-251
 *     -> TP_MOE::forward_binding  operators/moe-tp.hpp:195-199
 *   CPUInfer.submit_with_cuda_stream / sync_with_cuda_stream   (gone: the forward is enqueued on the caller's
 *                         cpu_backend/cpuinfer.h:87-120         hipStream_t; stream order IS the sync)
 *
 * Conventions: every data pointer passed to ktx_moe_forward is a DEVICE pointer on the handle's GPU; the call only
 * enqueues kernels on `stream` (no host sync, no allocation) and is HIP-graph capturable.  `d_bsz`, like the
 * reference's bsz_ptr (operators/moe-tp.hpp:209), is read on the device at execution time so one captured graph
 * serves any batch <= max_len.  All functions return 0 on success, non-zero on error; ktx_last_error() returns the
 * message of the calling thread's last failure (the reference throws std::runtime_error -> Python RuntimeError).
 */
#ifndef KTX_MOE_H
#define KTX_MOE_H
#include <stddef.h>
#include <stdint.h>

#include "ktx_linear.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ktx_moe_s* ktx_moe_t;
#ifndef KTX_STREAM_T_DEFINED
#define KTX_STREAM_T_DEFINED
typedef void* ktx_stream_t; /* hipStream_t */
#endif

/* weight/arithmetic formats = the reference's `method` names (kt-kernel/python/experts.py:316-360) */
enum ktx_moe_format {
  KTX_FMT_AMXINT4 = 0, /* per-row signed int4, d = amax/112; int8 per-row activations   (amx/la/amx_buffers.hpp:498-753) */
  KTX_FMT_AMXINT8 = 1, /* per-row int8,  d = amax/127                                    (amx/la/amx_kernels.hpp:1079-1150) */
  KTX_FMT_RAWINT4 = 2, /* Kimi-K2 compressed-tensors int4, group 32, bf16 scales          (amx/k2-moe.hpp) */
  KTX_FMT_FP8 = 3,     /* DeepSeek e4m3 + 128x128 block scale_inv, bf16 activations       (amx/fp8-moe.hpp) */
  KTX_FMT_BF16 = 4,    /* bf16 weights                                                    (amx/bf16-moe.hpp) */
  KTX_FMT_GGUF = 5,    /* GGUF k-/i-quant blocks (Q4_K / Q6_K / IQ1_S) x Q8_K activations — the llamafile backend (operators/llamafile/moe.hpp) */
  KTX_FMT_FP8_PERCHANNEL = 6, /* e4m3 + one fp32 scale per output row (GLM-4.7-FP8 style), bf16 activations (amx/fp8-perchannel-moe.hpp) */
};

enum ktx_moe_matrix { KTX_MAT_GATE = 0, KTX_MAT_UP = 1, KTX_MAT_DOWN = 2 };

typedef struct ktx_moe_config {
  int32_t expert_num;          /* GeneralMOEConfig::expert_num           (operators/common.hpp:232) */
  int32_t num_experts_per_tok; /* ::num_experts_per_tok                  (:233) */
  int32_t hidden_size;         /* ::hidden_size                          (:234) */
  int32_t intermediate_size;   /* ::intermediate_size                    (:235) */
  int32_t max_len;             /* ::max_len — largest qlen of one forward (:277) */
  int32_t format;              /* enum ktx_moe_format */
  int32_t group_size;          /* QuantConfig::group_size (RAWINT4: 32; FP8: 128) (:222-228) */
  int32_t device;              /* HIP device ordinal */
  int32_t expert_begin;        /* expert parallelism: this handle owns global experts [expert_begin, expert_begin+expert_num) */
  int32_t global_expert_num;   /* routing ids are global; ids outside the owned range are skipped like gpu_experts_mask */
} ktx_moe_config;

const char* ktx_last_error(void);

int ktx_moe_create(const ktx_moe_config* cfg, ktx_moe_t* out);
int ktx_moe_destroy(ktx_moe_t h);

/* Online quantisation on the GPU from bf16 weights, bit-identical to the reference's load_weights() "online quant
 * from bf16" branch (operators/amx/moe.hpp:352-387 -> BufferB::from_mat).  gate/up: [expert_num][I][H], down:
 * [expert_num][H][I], bf16, DEVICE pointers (borrowed for the call; the handle keeps its own packed copy, like the
 * AMX classes do — operators/amx/moe_base.hpp:134-143).  Synchronous. */
int ktx_moe_load_bf16(ktx_moe_t h, const void* d_gate, const void* d_up, const void* d_down);

/* One expert's pre-quantised matrix, HOST pointers: q = int8 [N][K] row-major integer multiplicands (for AMXINT4
 * the value nibble*16, i.e. what BufferBInt4Impl::to_mat inverts, amx/la/amx_buffers.hpp:683-739), scale = fp32 [N].
 * Mirrors the pre-quantised branch of load_weights (operators/amx/moe.hpp:266-300). Synchronous. */
int ktx_moe_load_quantized(ktx_moe_t h, int expert, int which, const int8_t* q, const float* scale);

/* FP8 (DeepSeek block-fp8) experts: e4m3 bytes gate/up [expert_num][I][H], down [expert_num][H][I] and fp32
 * scale_inv [expert_num][N/128][K/128] per matrix, DEVICE pointers — the "native weight" load of AMX_FP8_MOE_TP
 * (operators/amx/fp8-moe.hpp:180-240; layout of BufferBFP8Impl, amx/la/amx_raw_buffers.hpp:285-330).  For KTX_FMT_BF16
 * handles ktx_moe_load_bf16 stores the weights as they are (re-tiled).  Synchronous. */
int ktx_moe_load_fp8(ktx_moe_t h, const void* d_gate, const void* d_up, const void* d_down, const float* d_gate_scale,
                     const float* d_up_scale, const float* d_down_scale);

/* FP8_PERCHANNEL experts: e4m3 bytes as for ktx_moe_load_fp8 and ONE fp32 scale per output row — gate/up scales
 * [expert_num][I], down scales [expert_num][H], DEVICE pointers (load_weights of AMX_FP8_PERCHANNEL_MOE_TP,
 * operators/amx/fp8-perchannel-moe.hpp:508-555).  Arithmetic: fp32 sum of the bf16 products over the whole K, then
 * `* scale[n]`, then the bf16 rounding (float_mat_vec_perchannel, amx/la/amx_raw_kernels.hpp:630-840).  Synchronous. */
int ktx_moe_load_fp8_perchannel(ktx_moe_t h, const void* d_gate, const void* d_up, const void* d_down,
                                const float* d_gate_scale, const float* d_up_scale, const float* d_down_scale);

/* RAWINT4 (Kimi-K2 native / compressed-tensors int4): packed nibbles gate/up [expert_num][I][H/2], down
 * [expert_num][H][I/2] (byte = ((q1+8)<<4)|(q0+8), even k in the low nibble) and bf16 scales [expert_num][N][K/32],
 * DEVICE pointers — what AMX_K2_MOE_TP::load_weights consumes (operators/amx/k2-moe.hpp:124-191;
 * kt-kernel/python/utils/loader.py:683-777 `weight_packed` / `weight_scale`).  Synchronous. */
int ktx_moe_load_rawint4(ktx_moe_t h, const void* d_gate, const void* d_up, const void* d_down, const void* d_gate_scale,
                         const void* d_up_scale, const void* d_down_scale);

/* GGUF experts (the llamafile backend: LLAMA_MOE_TP::load_weights, operators/llamafile/moe.hpp:104-176; the
 * archive engine hands it mmap'ed `blk.N.ffn_{gate,up,down}_exps.weight`, archive/ktransformers/operators/experts.py:
 * 177-224): raw ggml blocks gate/up [expert_num][I][H/B blocks], down [expert_num][H][I/B blocks], DEVICE pointers,
 * with their ggml type ids (MOEConfig gate_type/up_type/down_type).  Two families, by the activation format ggml pairs the
 * weights with (type_traits.vec_dot_type, which is what moe.hpp:284-288,388 quantises the input / the intermediate to):
 *   B = 256, Q8_K activations: Q2_K (10), Q3_K (11), Q4_K (12), Q5_K (13), Q6_K (14), IQ4_XS (23), IQ1_S (19; its only in-tree
 *            definition is the reference's mul_mat_iq1_s_q8_K, third_party/llamafile/iqk_mul_mat.inc:2689-2770);
 *   B = 32,  Q8_0 activations: Q4_0 (2), Q5_0 (6), Q8_0 (8) — iqk_mul_mat.inc:2201-2213 (mul_mat_qX_0_q8_0_T) for the first two,
 *            tinyBLAS_Q0_AVX2 (tinyblas_cpu.h:828-1010) for Q8_0 weights.
 * The three matrices of a handle come from ONE family (the intermediate is quantised once); gate and up must share a type.
 * The handle re-tiles the blocks into its own layout (same bytes).  Synchronous. */
int ktx_moe_load_gguf(ktx_moe_t h, const void* d_gate, const void* d_up, const void* d_down, int gate_type, int up_type,
                      int down_type);

/* The weighted combine of ktx_moe_forward alone (operators/amx/moe_base.hpp:413-436): y[t] = (incremental ? y[t] : 0) +
 * sum_j fma(rows[row_of_pair[t*k+j]], w[t][j], .) in slot order, fp32, one bf16 rounding; row_of_pair < 0 skips the slot.
 * d_rows: bf16 [*][hidden] per-pair expert outputs.  Used by expert-parallel prefill, where the rows arrive over the
 * all-to-all (ktransformers_amd/parallel.py). */
int ktx_moe_combine(int qlen, int k, int hidden, const void* d_rows, const int32_t* d_row_of_pair, const float* d_weights,
                    void* d_output, int incremental, ktx_stream_t stream);

/* should_skip_expert mask (operators/common.hpp:241-258): mask[e] != 0 => expert e contributes nothing. HOST ptr, may be NULL. */
int ktx_moe_set_expert_mask(ktx_moe_t h, const uint8_t* mask);

/* Exact / fast switch of a handle (default 0 = fast).  Every decode-sized call, and every format but RAWINT4 at any size, reproduces
 * the reference's fp32 summation order bit for bit either way.  RAWINT4 (Kimi-K2) prompt chunks of >= 64 tokens are the one place a
 * faster kernel re-associates: moe_rawint4_chunk_kernel adds a row's K/32 group terms (d_a * d_w * int32 dot, the same terms) as ONE
 * fp32 chain, the reference keeps sixteen interleaved chains and a final tree (la/amx_kernels.hpp:3385-3455).  Measured bound of the
 * fast path against the reference on its own test's data (tests/test_moe_gpu.py): |y - ref| <= 2^-7 |ref| + 2^-9 max|ref| (two bf16
 * ulp element-wise), < 5 % of the outputs different, mean error < 1e-3; ~6.7x the exact kernel's prompt rate (16 k vs 2.4 k tok/s at
 * Kimi-K2 dimensions).  exact = 1 keeps the 4-row kernel whose sums are the reference's: bit-identical outputs at any size. */
int ktx_moe_set_exact(ktx_moe_t h, int exact);

/* y[t] = (incremental ? y[t] : 0) + sum_j w[t][j] * Expert_{ids[t][j]}(x[t])   — see DESIGN.md for the exact
 * rounding contract (= SURVEY.md Appendix A).  d_bsz may be NULL (then qlen is used); otherwise min(*d_bsz, qlen)
 * tokens are processed and qlen is only the launch bound.  x, y: bf16 [qlen][H]; ids int64 [qlen][k]; w fp32. */
int ktx_moe_forward(ktx_moe_t h, const int32_t* d_bsz, int qlen, int k, const int64_t* d_expert_ids,
                    const float* d_weights, const void* d_input, void* d_output, int incremental,
                    ktx_stream_t stream);

/* Same, with flags.  KTX_FWD_PARTIAL_F32: d_output is float [qlen][H] and receives the un-rounded fp32 weighted sums of
 * the experts this handle owns (ids outside [expert_begin, expert_begin+expert_num) contribute nothing) — the expert-
 * parallel analogue of the reference's per-NUMA fp32 partials that merge_results sums before the single bf16 rounding
 * (operators/amx/moe_base.hpp:749-791, operators/moe-tp.hpp:201-216). */
enum { KTX_FWD_INCREMENTAL = 1, KTX_FWD_PARTIAL_F32 = 2 };
int ktx_moe_forward_ex(ktx_moe_t h, const int32_t* d_bsz, int qlen, int k, const int64_t* d_expert_ids,
                       const float* d_weights, const void* d_input, void* d_output, int flags, ktx_stream_t stream);

/* merge_results of the reference's NUMA tensor-parallel MoE (operators/amx/moe_base.hpp:749-791; operators/moe-tp.hpp:201-216):
 * a checkpoint converted with threadpool_count = P holds every expert as P parts — gate / up split over intermediate rows, down
 * over K, each part with its own row scales — and TP_MOE runs P complete MoEs of width I / P whose fp32 outputs it adds:
 *     y[t] = bf16( ((part_0[t] + (incremental ? y[t] : 0)) + part_1[t]) + ... + part_{P-1}[t] )       in exactly this order.
 * d_parts: float [nparts][part_stride] holding [qlen][hidden] rows each (the KTX_FWD_PARTIAL_F32 outputs of P handles of
 * intermediate size I / P); d_bsz as in ktx_moe_forward. */
int ktx_moe_merge_partials(int nparts, int qlen, int hidden, const float* d_parts, int64_t part_stride, void* d_output,
                           int incremental, const int32_t* d_bsz, ktx_stream_t stream);

/* Decode step of a whole MoE block's tail (KDeepseekV3MoE.forward + the decoder layer's residual add,
 * archive/ktransformers/operators/experts.py:974-1012, models/modeling_deepseek_v3.py:1225):
 *     y[t] = residual[t] + ( sum_j w[t][j] * Expert_{ids[t][j]}(x[t])  +  side_linear(side_x[t]) )
 * with the reference's bf16 tensor arithmetic: the routed sum rounded to bf16 as ktx_moe_forward gives it, the side linear's
 * output rounded to bf16 as ktx_linear_forward gives it, then two bf16 adds.  side_linear: a ktx_linear_t with
 * out_features == hidden_size — the shared experts' down_proj; side_x: bf16 [qlen][in_features], i.e.
 * act_fn(gate(x)) * up(x) of the shared experts; d_residual: bf16 [qlen][hidden] or NULL.  For the AMXINT4 / AMXINT8 decode
 * path and a W4 g64 side linear the down kernel computes the side strip itself (no launch of its own); every other case runs
 * ktx_moe_forward and ktx_linear_forward_fused (adds in its epilogue) — same result contract, the call never fails for that. */
int ktx_moe_forward_side(ktx_moe_t h, const int32_t* d_bsz, int qlen, int k, const int64_t* d_expert_ids,
                         const float* d_weights, const void* d_input, void* d_output, ktx_linear_t side_linear,
                         const void* d_side_x, const void* d_residual, ktx_stream_t stream);

/* Introspection for tests / bench: bytes of packed expert weights resident in HBM; debug taps (device pointers to
 * the last forward's intermediates in sorted-row order, plus the row of each (t,j) pair). */
size_t ktx_moe_weight_bytes(ktx_moe_t h);
int ktx_moe_debug_ptrs(ktx_moe_t h, const void** act_bf16, const void** down_bf16, const int32_t** row_of_pair);

/* Measurement aid (bench.py roofline leg): when enabled, every ktx_moe_forward brackets each of its kernels with
 * HIP events on the launch stream (slots: 0 prep, 1 gate/up GEMM, 2 act-quant, 3 down GEMM, 4 combine).  collect
 * synchronises the device and returns summed elapsed ms + launch counts since the previous collect.  The reference's
 * analogue is its FORWARD_TIME_PROFILE per-stage timers (operators/amx/moe_base.hpp:200-206,438-452).  Forwards
 * issued while profiling is on are not graph-capturable. */
int ktx_profile_enable(int on);
/* Test hook: non-zero routes every batch through the grouped (bucket + M-tiled GEMM) path, including the small
 * batches that normally take the two-launch decode path, so both implementations can be compared on one input. */
int ktx_debug_force_generic(int on);
/* Development knobs for timing experiments and tests (never set by the product path): idx 0 = waves per workgroup of the
 * decode gate/up kernel (0 = auto), idx 1 = ablation bits (bit0: skip the weight stream; results are then meaningless),
 * idx 2 = run only one decode kernel (1 gate/up, 2 down; bench.py's per-kernel timing), idx 4 = prompt (64-row tile) GEMM
 * implementation: 0 auto (streaming kernels for hidden*intermediate >= 4M), 1 chunk-pipelined only, 2 streaming only,
 * 3 the register-tile kernels (256-row tiles; bit-exact in the tests; timed in round 2 with scripts/stream_ab.py: faster
 * only under uniform routing at V2-Lite shapes, slower under skew and at V3 shapes, hence not selected by default),
 * idx 5 = run only one kernel of ktx_mla_decode* (1 the split-KV kernel, 2 the merge; per-kernel timing), idx 6 / 7 = force
 * the MLA workgroup shape (1, 2, 4 head blocks) / the KV split count, idx 8 = force the strips-per-workgroup split of the
 * decode GEMV (tuning sweeps under scripts/), idx 9 = 2 selects the LDS-DMA ring variant of the W4 decode GEMV (measured
 * slower than the default register ring; kept for tuning), idx 10 = 1 turns the k-slices of the AMXINT4 decode gate/up kernel
 * off, idx 11 = 1 selects the first (>= 384 workgroups) split rule of the decode GEMV (both A/B switches of
 * scripts/ab_decode.py), idx 12 = strips per wavefront of the prompt-sized W4 GEMM (1 = the one-strip kernel, 2 / 4 forced;
 * 0 auto), idx 13 = 1 makes ktx_linear_forward_fused_gate issue its two launches instead of the combined kernel (A/B, tests),
 * idx 14 = 1 makes ktx_moe_forward_side run the side linear as a launch of its own (A/B, tests), idx 15 = 1 makes
 * ktx_linear_qb_absorb_eligible answer no (A/B), idx 22 = 1 runs the non-absorbed prompt attention kernel with its
 * unconstrained register allocation (one wavefront per SIMD; A/B of scripts/mla_prefill_bench.py), idx 25 = experts per
 * router workgroup in ktx_linear_forward_fused_gate's combined kernel: 1 -> 8 (rounds 2-3), 2 -> 2, 0 -> 4 where the grid has room
 * (same logits and selection bit for bit). */
int ktx_debug_set(int idx, int val);
int ktx_debug_get(int idx);
/* Per-launch timing of every kernel of the library (bench.py's per-kernel table; ktx_prof.hip).  mode 1: each launch is
 * bracketed by two HIP events on its stream; mode 2: labels only (rocprofv3 --pmc passes map dispatches to classes with
 * it); 0: off.  collect synchronises the device and writes one line per launch since the previous collect, in launch order:
 * "<kernel and shape>\t<algorithmic bytes>\t<microseconds, -1 in mode 2>\n".  *needed receives the size the text needs; call
 * with buf = NULL to query it.  Launches made while timing is on are not graph-capturable. */
int ktx_timing_enable(int mode);
int ktx_timing_collect(char* buf, size_t cap, size_t* needed);
int ktx_profile_collect(double* ms5, long long* count5);

#ifdef __cplusplus
}
#endif
#endif
