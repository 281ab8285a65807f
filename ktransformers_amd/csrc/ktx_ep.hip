// ktx_ep.hip — expert-parallel decode exchange over direct peer writes (include/ktx_ep.h).
//
// Reduce shape = the reference's TP_MOE_Common::merge_results (kt-kernel/operators/amx/moe_base.hpp:749-791): fp32 partial
// [T,H] per part, added in part order, one bf16 rounding.  Parts here are ranks (experts sharded by id).
//
// Buffer of rank R (8-byte granules {payload:32, tag:32}):
//   [0, 32)                          header words: gather tag, reduce tag, status
//   gather area  [src][t][i]         i < RW = H/2 + 3k : the token row as 32-bit words (bf16 pairs of x | int64 ids | fp32 w)
//   partial area [src][t][c]         c < H             : fp32 partial sums of this rank's token t computed by rank src
// A granule is valid for call n when its tag equals n; tags count calls on the device (gather bumps the reduce tag, reduce
// bumps the gather tag: the two kernels of a layer alternate on one stream), so a captured graph replays correctly.
#include <algorithm>
#include <cstring>

#include "../../include/ktx_ep.h"
#include "ktx_common.h"

namespace {

constexpr int HDR = 32;          // granules
constexpr int NT = 256;          // threads per workgroup
constexpr int GU = 8;            // granules a thread polls together in the gather

struct EpDev {
  uint64_t* base[KTX_EP_MAX_WORLD];
  int world, rank, maxT, H, k, RW;
  long long spin_ticks;          // wall_clock64 ticks (100 MHz)
};

__device__ __forceinline__ size_t g_off(const EpDev& d, int src, int t) { return HDR + ((size_t)src * d.maxT + t) * d.RW; }
// two partial regions: the gather + reduce pair always uses region 0 (the gather between two reduces orders the ranks); a
// reduce-ONLY sequence (ktx_ep_reduce_only) alternates between them by the parity of its call tag
__device__ __forceinline__ size_t p_off(const EpDev& d, int src, int t, int region = 0) {
  return HDR + (size_t)d.world * d.maxT * d.RW + (((size_t)region * d.world + src) * d.maxT + t) * d.H;
}
__device__ __forceinline__ void put(uint64_t* p, uint32_t data, uint32_t tag) {
  __hip_atomic_store(p, (uint64_t)data | ((uint64_t)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint64_t peek(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint32_t next_tag(uint32_t s) { return s + 1 ? s + 1 : 1; }   // 0 = never written

__global__ __launch_bounds__(NT) void ep_gather_kernel(EpDev d, int T, const uint32_t* __restrict__ x,
                                                       const uint32_t* __restrict__ ids, const uint32_t* __restrict__ w,
                                                       uint32_t* __restrict__ xg, uint32_t* __restrict__ idsg,
                                                       uint32_t* __restrict__ wg) {
  const int p = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
  uint32_t* hdr = reinterpret_cast<uint32_t*>(d.base[d.rank]);
  const uint32_t tag = __hip_atomic_load(&hdr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (p == 0 && t == 0 && tid == 0) hdr[1] = next_tag(hdr[1]);
  const int HW = d.H / 2, IW = 2 * d.k;
  const size_t row = (size_t)p * T + t;
  auto word = [&](int i) -> uint32_t {
    return i < HW ? x[(size_t)t * HW + i] : i < HW + IW ? ids[(size_t)t * IW + (i - HW)] : w[(size_t)t * d.k + (i - HW - IW)];
  };
  auto sink = [&](int i, uint32_t v) {
    if (i < HW) xg[row * HW + i] = v;
    else if (i < HW + IW) idsg[row * IW + (i - HW)] = v;
    else wg[row * d.k + (i - HW - IW)] = v;
  };
  if (p == d.rank) {
    for (int i = tid; i < d.RW; i += NT) sink(i, word(i));
    return;
  }
  uint64_t* dst = d.base[p] + g_off(d, d.rank, t);
  for (int i = tid; i < d.RW; i += NT) put(dst + i, word(i), tag);
  const uint64_t* src = d.base[d.rank] + g_off(d, p, t);
  const long long t0 = wall_clock64();
  for (int i0 = tid; i0 < d.RW; i0 += NT * GU) {
    uint64_t v[GU];
    for (unsigned spins = 1;; ++spins) {
      bool ok = true;
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int i = i0 + u * NT;
        v[u] = i < d.RW ? peek(src + i) : (uint64_t)tag << 32;
        ok &= (uint32_t)(v[u] >> 32) == tag;
      }
      if (ok) break;
      if ((spins & 255) == 0 && wall_clock64() - t0 > d.spin_ticks) {
        atomicCAS(&hdr[2], 0u, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int i = i0 + u * NT;
      if (i < d.RW) sink(i, (uint32_t)v[u]);
    }
  }
}

// grid (world * S, T): workgroup (b, t) first sends sub-range b % S of part[row of rank b / S] to that rank, then reduces
// column slice b of this rank's token t.
// SELF (ktx_ep_reduce_only: one token stream replicated on every rank, no gather in front): the launch owns its call tag — header
// word 4, advanced by the LAST workgroup to leave (word 6 counts the leavers), so every workgroup of the launch has read it
// before it moves — and writes the partial region of the tag's parity: rank A can be at most one call ahead of rank B (it needs
// B's partials of call n + 1 to finish call n + 1, and B sends those only after it has consumed call n), so two regions suffice.
// Without this the reduce re-used the tag the last gather left behind and summed whatever granules the previous call had left
// (ADVICE r3, high).
template <bool SELF>
__global__ __launch_bounds__(NT) void ep_reduce_kernel(EpDev d, int T, int S, const float* __restrict__ part,
                                                       bf16_t* __restrict__ out) {
  const int b = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
  uint32_t* hdr = reinterpret_cast<uint32_t*>(d.base[d.rank]);
  const uint32_t tag = __hip_atomic_load(&hdr[SELF ? 4 : 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int region = SELF ? (int)(tag & 1u) : 0;
  if (!SELF && b == 0 && t == 0 && tid == 0) hdr[0] = next_tag(hdr[0]);
  const int p = b / S, s = b % S;
  if (p != d.rank) {
    const int cw = (d.H + S - 1) / S, lo = s * cw, hi = min(d.H, lo + cw);
    const float* srow = part + ((size_t)p * T + t) * d.H;
    uint64_t* dst = d.base[p] + p_off(d, d.rank, t, region);
    for (int c = lo + tid; c < hi; c += NT) put(dst + c, __float_as_uint(srow[c]), tag);
  }
  const int nsl = d.world * S, sw = (d.H + nsl - 1) / nsl, lo = b * sw, hi = min(d.H, lo + sw);
  const float* mine = part + ((size_t)d.rank * T + t) * d.H;
  const uint64_t* rbase = d.base[d.rank];
  const long long t0 = wall_clock64();
  for (int c = lo + tid; c < hi; c += NT) {
    uint64_t v[KTX_EP_MAX_WORLD];
    for (unsigned spins = 1;; ++spins) {
      bool ok = true;
#pragma unroll
      for (int r = 0; r < KTX_EP_MAX_WORLD; ++r) {
        if (r < d.world && r != d.rank) {
          v[r] = peek(rbase + p_off(d, r, t, region) + c);
          ok &= (uint32_t)(v[r] >> 32) == tag;
        }
      }
      if (ok) break;
      if ((spins & 255) == 0 && wall_clock64() - t0 > d.spin_ticks) {
        atomicCAS(&hdr[2], 0u, 2u);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    float acc = 0.0f;
#pragma unroll
    for (int r = 0; r < KTX_EP_MAX_WORLD; ++r) {
      if (r < d.world) {
        const float pr = r == d.rank ? mine[c] : __uint_as_float((uint32_t)v[r]);
        acc = r == 0 ? pr : acc + pr;      // ((p_0 + p_1) + p_2) + ... in rank order
      }
    }
    out[(size_t)t * d.H + c] = f32_to_bf16(acc);
  }
  if constexpr (SELF) {
    __syncthreads();
    if (tid == 0) {
      const uint32_t left = __hip_atomic_fetch_add(&hdr[6], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (left == gridDim.x * gridDim.y - 1) {
        __hip_atomic_store(&hdr[6], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&hdr[4], next_tag(tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

struct DevGuard {
  int prev = -1;
  hipError_t err;
  explicit DevGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) err = hipSetDevice(dev);
  }
  ~DevGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

}  // namespace

struct ktx_ep_s {
  int device = 0, world = 0, rank = 0, maxT = 0, H = 0, k = 0, RW = 0;
  size_t bytes = 0;
  uint64_t* base[KTX_EP_MAX_WORLD] = {};
  bool ipc[KTX_EP_MAX_WORLD] = {};
  double spin_seconds = 30.0;
};

extern "C" {

int ktx_ep_create(int device, int world, int rank, int max_tokens, int hidden, int topk, int memory_kind, ktx_ep_t* out) {
  KTX_REQUIRE(out, "ktx_ep_create: null out");
  KTX_REQUIRE(world >= 1 && world <= KTX_EP_MAX_WORLD && rank >= 0 && rank < world, "ktx_ep_create: bad world / rank");
  KTX_REQUIRE(max_tokens >= 1 && max_tokens <= 64 && hidden > 0 && hidden % 2 == 0 && topk >= 1 && topk <= 64,
              "ktx_ep_create: max_tokens in [1,64], even hidden, topk in [1,64]");
  KTX_REQUIRE(memory_kind >= 0 && memory_kind <= 2, "ktx_ep_create: memory_kind 0 (uncached), 1 (fine-grained) or 2 (plain)");
  DevGuard dg(device);
  KTX_HIP(dg.err);
  ktx_ep_s* ep = new ktx_ep_s;
  ep->device = device; ep->world = world; ep->rank = rank; ep->maxT = max_tokens; ep->H = hidden; ep->k = topk;
  ep->RW = hidden / 2 + 3 * topk;
  ep->bytes = ((size_t)HDR + (size_t)world * max_tokens * ((size_t)ep->RW + 2 * (size_t)hidden)) * sizeof(uint64_t);
  void* p = nullptr;
  hipError_t e = memory_kind == 2 ? hipMalloc(&p, ep->bytes)
                                  : hipExtMallocWithFlags(&p, ep->bytes, memory_kind == 0 ? hipDeviceMallocUncached
                                                                                          : hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    delete ep;
    return ktx_fail(std::string("ktx_ep_create: allocating the symmetric buffer: ") + hipGetErrorString(e));
  }
  const uint32_t hdr[8] = {1u, 0u, 0u, 0u, 1u, 0u, 0u, 0u};   // first gather uses tag 1 and makes the reduce tag 1; word 4 = the reduce-only tag
  e = hipMemset(p, 0, ep->bytes);
  if (e == hipSuccess) e = hipMemcpy(p, hdr, sizeof(hdr), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)hipFree(p);
    delete ep;
    return ktx_fail(std::string("ktx_ep_create: clearing the symmetric buffer: ") + hipGetErrorString(e));
  }
  ep->base[rank] = static_cast<uint64_t*>(p);
  *out = ep;
  return 0;
}

void ktx_ep_destroy(ktx_ep_t ep) {
  if (!ep) return;
  DevGuard dg(ep->device);
  (void)hipDeviceSynchronize();
  for (int r = 0; r < ep->world; ++r)
    if (r != ep->rank && ep->base[r] && ep->ipc[r]) (void)hipIpcCloseMemHandle(ep->base[r]);
  if (ep->base[ep->rank]) (void)hipFree(ep->base[ep->rank]);
  delete ep;
}

int ktx_ep_export(ktx_ep_t ep, void* handle_out) {
  KTX_REQUIRE(ep && handle_out, "ktx_ep_export: null argument");
  static_assert(sizeof(hipIpcMemHandle_t) == KTX_EP_HANDLE_BYTES, "handle size");
  DevGuard dg(ep->device);
  KTX_HIP(dg.err);
  hipIpcMemHandle_t h;
  KTX_HIP(hipIpcGetMemHandle(&h, ep->base[ep->rank]));
  memcpy(handle_out, &h, sizeof(h));
  return 0;
}

int ktx_ep_local_ptr(ktx_ep_t ep, void** ptr_out) {
  KTX_REQUIRE(ep && ptr_out, "ktx_ep_local_ptr: null argument");
  *ptr_out = ep->base[ep->rank];
  return 0;
}

int ktx_ep_import(ktx_ep_t ep, int peer, const void* handle) {
  KTX_REQUIRE(ep && handle, "ktx_ep_import: null argument");
  KTX_REQUIRE(peer >= 0 && peer < ep->world && peer != ep->rank, "ktx_ep_import: peer out of range");
  KTX_REQUIRE(!ep->base[peer], "ktx_ep_import: peer already mapped");
  DevGuard dg(ep->device);
  KTX_HIP(dg.err);
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  void* p = nullptr;
  KTX_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  ep->base[peer] = static_cast<uint64_t*>(p);
  ep->ipc[peer] = true;
  return 0;
}

int ktx_ep_import_ptr(ktx_ep_t ep, int peer, void* ptr) {
  KTX_REQUIRE(ep && ptr, "ktx_ep_import_ptr: null argument");
  KTX_REQUIRE(peer >= 0 && peer < ep->world && peer != ep->rank, "ktx_ep_import_ptr: peer out of range");
  KTX_REQUIRE(!ep->base[peer], "ktx_ep_import_ptr: peer already mapped");
  ep->base[peer] = static_cast<uint64_t*>(ptr);
  return 0;
}

int ktx_ep_set_spin_seconds(ktx_ep_t ep, double seconds) {
  KTX_REQUIRE(ep && seconds > 0 && seconds <= 600, "ktx_ep_set_spin_seconds: (0, 600]");
  ep->spin_seconds = seconds;
  return 0;
}

static int ep_dev(ktx_ep_t ep, int T, EpDev* d, const char* who) {
  KTX_REQUIRE(ep, std::string(who) + ": null handle");
  KTX_REQUIRE(T >= 1 && T <= ep->maxT, std::string(who) + ": T exceeds max_tokens");
  for (int r = 0; r < ep->world; ++r) KTX_REQUIRE(ep->base[r], std::string(who) + ": a peer's buffer is not mapped yet");
  for (int r = 0; r < KTX_EP_MAX_WORLD; ++r) d->base[r] = r < ep->world ? ep->base[r] : nullptr;
  d->world = ep->world; d->rank = ep->rank; d->maxT = ep->maxT; d->H = ep->H; d->k = ep->k; d->RW = ep->RW;
  d->spin_ticks = (long long)(ep->spin_seconds * 1e8);
  return 0;
}

int ktx_ep_gather(ktx_ep_t ep, int T, const void* d_x, const int64_t* d_ids, const float* d_w, void* d_xg, int64_t* d_idsg,
                  float* d_wg, ktx_stream_t stream) {
  EpDev d;
  if (int rc = ep_dev(ep, T, &d, "ktx_ep_gather")) return rc;
  KTX_REQUIRE(d_x && d_ids && d_w && d_xg && d_idsg && d_wg, "ktx_ep_gather: null pointer");
  KTX_REQUIRE((((uintptr_t)d_x | (uintptr_t)d_xg | (uintptr_t)d_w | (uintptr_t)d_wg) & 3) == 0 &&
                  (((uintptr_t)d_ids | (uintptr_t)d_idsg) & 7) == 0,
              "ktx_ep_gather: rows must be 4-byte aligned (x, w) / 8-byte aligned (ids): they travel as 32-bit words");
  DevGuard dg(ep->device);
  KTX_HIP(dg.err);
  hipStream_t st = (hipStream_t)stream;
  // put-then-spin: every workgroup of every rank must be resident at once (a queued workgroup's puts would never come).
  // world * T workgroups of NT threads against >= 256 CUs x several per CU: far inside the residency of any gfx950 part;
  // refused loudly beyond it instead of trusted
  KTX_REQUIRE((long)ep->world * T <= 1024, "ktx_ep_gather: world * T exceeds the co-residency the put-then-spin exchange assumes");
  KTX_TIMED(st, (double)ep->world * T * ep->RW * 8.0, "ep_gather_kernel R=%d T=%d H=%d", ep->world, T, ep->H);
  hipLaunchKernelGGL(ep_gather_kernel, dim3(ep->world, T), dim3(NT), 0, st, d, T, (const uint32_t*)d_x, (const uint32_t*)d_ids,
                     (const uint32_t*)d_w, (uint32_t*)d_xg, (uint32_t*)d_idsg, (uint32_t*)d_wg);
  KTX_HIP(hipGetLastError());
  return 0;
}

static int ep_reduce_impl(ktx_ep_t ep, int T, const float* d_part, void* d_out, ktx_stream_t stream, bool self);
int ktx_ep_reduce(ktx_ep_t ep, int T, const float* d_part, void* d_out, ktx_stream_t stream) {
  return ep_reduce_impl(ep, T, d_part, d_out, stream, false);
}
int ktx_ep_reduce_only(ktx_ep_t ep, int T, const float* d_part, void* d_out, ktx_stream_t stream) {
  return ep_reduce_impl(ep, T, d_part, d_out, stream, true);
}
static int ep_reduce_impl(ktx_ep_t ep, int T, const float* d_part, void* d_out, ktx_stream_t stream, bool self) {
  EpDev d;
  if (int rc = ep_dev(ep, T, &d, "ktx_ep_reduce")) return rc;
  KTX_REQUIRE(d_part && d_out, "ktx_ep_reduce: null pointer");
  KTX_REQUIRE(((uintptr_t)d_part & 3) == 0 && ((uintptr_t)d_out & 1) == 0, "ktx_ep_reduce: misaligned pointer");
  DevGuard dg(ep->device);
  KTX_HIP(dg.err);
  hipStream_t st = (hipStream_t)stream;
  // column slices of about 512 (two columns per thread), a whole number of them per rank
  const int S = std::max(1, (ep->H + 512 * ep->world - 1) / (512 * ep->world));
  KTX_REQUIRE((long)ep->world * S * T <= 1024, "ktx_ep_reduce: grid exceeds the co-residency the put-then-spin exchange assumes");
  KTX_TIMED(st, (double)ep->world * T * ep->H * 8.0, "ep_reduce_kernel R=%d T=%d H=%d%s", ep->world, T, ep->H, self ? " (reduce only)" : "");
  if (self) hipLaunchKernelGGL(ep_reduce_kernel<true>, dim3(ep->world * S, T), dim3(NT), 0, st, d, T, S, d_part, (bf16_t*)d_out);
  else hipLaunchKernelGGL(ep_reduce_kernel<false>, dim3(ep->world * S, T), dim3(NT), 0, st, d, T, S, d_part, (bf16_t*)d_out);
  KTX_HIP(hipGetLastError());
  return 0;
}

int ktx_ep_status(ktx_ep_t ep, ktx_stream_t stream, int* status_out) {
  KTX_REQUIRE(ep && status_out, "ktx_ep_status: null argument");
  DevGuard dg(ep->device);
  KTX_HIP(dg.err);
  KTX_HIP(hipStreamSynchronize((hipStream_t)stream));
  uint32_t hdr[3];
  KTX_HIP(hipMemcpy(hdr, ep->base[ep->rank], sizeof(hdr), hipMemcpyDeviceToHost));
  *status_out = (int)hdr[2];
  return 0;
}

}  // extern "C"
