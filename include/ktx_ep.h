/* ktx_ep.h — C ABI of the expert-parallel DECODE exchange over direct peer writes (libktx_hip.so).
 *
 * experts are sharded by id over R ranks (one process per GPU); at decode every rank needs every rank's
 * token row (x, expert ids, routing weights — a few KiB) and every token's home rank needs the R fp32 partial sums
 * [T,H].  The reference's analogue is the NUMA tensor-parallel split of TP_MOE_Common::forward with its fp32
 * merge_results (kt-kernel/operators/moe-tp.hpp:201-246, operators/amx/moe_base.hpp:749-791): every part sees all
 * tokens, the partials are added in fp32 in a FIXED part order, one bf16 rounding at the end.  Same reduce shape here,
 * parts = ranks, order = rank order 0..R-1.
 *
 * Transport: each rank owns one "symmetric" buffer (uncached device memory) that every peer maps (hipIpc handle across
 * processes, or a plain pointer for peers inside one process).  Data travels as 8-byte granules {32-bit payload,
 * 32-bit call tag} written straight into the RECEIVER's buffer; the receiver polls the granules themselves, so there is
 * no separate flag and no fence whose reach over xGMI would have to be trusted: an 8-byte store is single-copy atomic.
 * (The same framing RCCL's LL protocol uses.)  Two launches per MoE layer replace all-gather + reduce-scatter + the
 * concatenate / slice / cast glue around them:
 *
 *   ktx_ep_gather : put my T token rows into every peer's buffer, collect the R*T rows sent to me -> xg, idsg, wg
 *   (local experts on the gathered tokens: ktx_moe_forward_ex(..., KTX_FWD_PARTIAL_F32) -> part fp32 [R*T, H])
 *   ktx_ep_reduce : put part[rows of rank r] into rank r's buffer, collect the R partials of my T tokens,
 *                   out[t] = bf16(((p_0 + p_1) + ...) + p_{R-1})
 *
 * Every rank must make the same sequence of (gather, reduce) calls with the same T (the call tags are counted on the
 * device, so the pair is HIP-graph capturable and replayable).  A poll that sees no data for `spin_seconds`
 * gives up, raises the status word (ktx_ep_status) and lets the kernel finish: a dead peer cannot hang the GPU.
 *
 * DEVICE pointers, enqueue-only on `stream`, 0 = success (ktx_last_error() for the message).
 */
#ifndef KTX_EP_H
#define KTX_EP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef KTX_STREAM_T_DEFINED
#define KTX_STREAM_T_DEFINED
typedef void* ktx_stream_t; /* hipStream_t */
#endif
const char* ktx_last_error(void);

typedef struct ktx_ep_s* ktx_ep_t;

#define KTX_EP_MAX_WORLD 16
#define KTX_EP_HANDLE_BYTES 64 /* sizeof(hipIpcMemHandle_t) */

/* memory kind of the symmetric buffer: 0 = uncached (default), 1 = fine-grained, 2 = ordinary device memory
 * (only valid when all ranks share one GPU, e.g. the single-GPU functional tests) */
int ktx_ep_create(int device, int world, int rank, int max_tokens, int hidden, int topk, int memory_kind, ktx_ep_t* out);
void ktx_ep_destroy(ktx_ep_t ep);

/* this rank's buffer as an inter-process handle (KTX_EP_HANDLE_BYTES bytes) / as a pointer for peers in this process */
int ktx_ep_export(ktx_ep_t ep, void* handle_out);
int ktx_ep_local_ptr(ktx_ep_t ep, void** ptr_out);
/* map peer `peer`'s buffer: from its exported handle (another process) or its pointer (same process; for a peer on another
 * GPU of this process peer access must already be enabled) */
int ktx_ep_import(ktx_ep_t ep, int peer, const void* handle);
int ktx_ep_import_ptr(ktx_ep_t ep, int peer, void* ptr);

/* x bf16 [T,H], ids int64 [T,k], w fp32 [T,k] (this rank's tokens) -> xg bf16 [R*T,H], idsg int64 [R*T,k], wg fp32 [R*T,k],
 * rank r's tokens in rows [r*T, (r+1)*T). */
int ktx_ep_gather(ktx_ep_t ep, int T, const void* d_x, const int64_t* d_ids, const float* d_w, void* d_xg, int64_t* d_idsg,
                  float* d_wg, ktx_stream_t stream);
/* part fp32 [R*T,H] (this rank's experts' contribution to every gathered token) -> out bf16 [T,H] for this rank's tokens */
int ktx_ep_reduce(ktx_ep_t ep, int T, const float* d_part, void* d_out, ktx_stream_t stream);
/* The same reduce for a sequence WITHOUT gathers (one token stream replicated on every rank: each rank's experts see the same
 * rows, `bench.py --strong`): the launch owns its call tag (advanced on the device by its last workgroup) and alternates between
 * two partial regions, so back-to-back calls — also replays of a captured graph — never match granules of an earlier call.
 * Do not interleave with ktx_ep_gather / ktx_ep_reduce pairs inside one step. */
int ktx_ep_reduce_only(ktx_ep_t ep, int T, const float* d_part, void* d_out, ktx_stream_t stream);

/* 0 = healthy; otherwise the code of the first poll that gave up (1 = gather, 2 = reduce).  Synchronises `stream`. */
int ktx_ep_status(ktx_ep_t ep, ktx_stream_t stream, int* status_out);
/* seconds a poll waits before giving up (default 30) */
int ktx_ep_set_spin_seconds(ktx_ep_t ep, double seconds);

#ifdef __cplusplus
}
#endif
#endif
