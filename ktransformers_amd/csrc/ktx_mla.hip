// ktx_mla.hip — MLA compressed-KV paged attention (absorbed form) for gfx950.  C ABI in include/ktx_mla.h.
//
// All Hq query heads share one latent K = [ckv(512) | k_pe(64)] and V = ckv, so decode attention is two skinny GEMMs per
// KV tile:  S[heads, tokens] = Q[heads, 576] K^T  and  O[heads, 512] += P[heads, tokens] V[tokens, 512], both on
// v_mfma_f32_16x16x32_bf16 with fp32 softmax statistics (flash-decoding: the KV range is split across workgroups and
// the partial (m, l, O) triples are merged by a second launch).
//
// Workgroup = NWV waves; each wave owns 16 heads (Q fragments and the 16x512 fp32 output tile live in registers), all
// waves share one staged 32-token KV tile in LDS:
//   Kt [32][576+8]  bf16 row-major (pad 16 B), ckv rows land by LDS-DMA (global_load_lds, 1 KiB per wave-instruction)
//        -> B operand of S = Q K^T   : ds_read_b128, token = lane&15
//        -> B operand of O += P V    : ds_read_b64_tr_b16 x2 (hardware transpose: 8 tokens of one dim per lane)
//   Pt [NWV][16][32] bf16            : P re-laid from the MFMA C layout to the A layout (wave-private)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "../../include/ktx_mla.h"
#include "ktx_common.h"

extern "C" int ktx_debug_get(int idx);   // ktx_moe.hip (include/ktx_moe.h)

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

#define MLA_TILE 32
#define MLA_DC 512
#define MLA_DR 64
#define MLA_KROW (MLA_DC + MLA_DR + 8)   // 584 elements = 1168 B (73 x 16 B: rows rotate through the 16-B bank slots)

struct MlaParams {
  const bf16_t *q_nope, *q_pe, *ckv, *k_pe;
  long long ckv_ts, kpe_ts;
  const int32_t *qo_indptr, *kv_indptr, *kv_indices, *kv_len, *d_bsz;
  int batch, total_q, Hq, page_size, nsplit;
  int tiles_shift;   // log2(page_size / 32) when the page size is a power of two (the reference's 64 and 256), else -1: tile -> page without a division
  float sm_scale;
  float* part_o;   // [total_q][Hq][nsplit][512]
  float* part_ml;  // [total_q][Hq][nsplit][2]
  bf16_t* out1;    // nsplit == 1: the split-KV kernel normalises and stores the result itself (no partials, no merge launch)
  float* lse1;
  // optional fused cache append for decode (one new token per request): latent rows [batch][512], [batch][64]
  const bf16_t *app_ckv, *app_kpe;
  bf16_t *ckv_w, *kpe_w;
  long long* dbg;   // tuning aid (scripts/mla_sweep.py): 16 wall-clock stamps per workgroup, NULL in normal use
};
#define MLA_TS(k)                                                                                             \
  do {                                                                                                        \
    if (p.dbg && threadIdx.x == 0)                                                                            \
      p.dbg[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (k)] = wall_clock64(); \
  } while (0)

// the three stamps INSIDE the tile loop (first tile: landed, S done, tile done) cost every tile a compare, an exec save and a branch
// each — compiled in only for the stamp pass of scripts/mla_sweep.py (build with -DKTX_MLA_LOOP_STAMPS)
#ifdef KTX_MLA_LOOP_STAMPS
#define MLA_TS_LOOP(k) do { if (tile == t_begin) MLA_TS(k); } while (0)
#else
#define MLA_TS_LOOP(k) do { } while (0)
#endif

typedef short v4s16 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v8bf as_v8bf(const uint4& u) {
  union { uint4 u; v8bf v; } c;
  c.u = u;
  return c.v;
}

// B operand of O += P V straight from the row-major K tile: two hardware-transposed LDS reads (ds_read_b64_tr_b16).
// Within a 16-lane group, lane i supplies the address of a 4-element chunk (row i/4, cols (i%4)*4..+3) of a
// [4 tokens][16 dims] block and receives column i of it, i.e. 4 consecutive tokens of one dim (probed on gfx950:
// scripts/tr_probe.hip).  base points at [token (lane>>4)*8 + ((lane&15)>>2)][dim0 + (lane&3)*4].
__device__ __forceinline__ v8bf load_v_frag(const bf16_t* base) {
  typedef __attribute__((address_space(3))) v4s16 lds_v4;
  const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(base));
  const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(base + 4 * MLA_KROW));
  union { v4s16 h[2]; v8bf v; } c;
  c.h[0] = lo;
  c.h[1] = hi;
  return c.v;
}

// One LDS-DMA wave-instruction: lane l's 16 bytes at gsrc_lane land at LDS byte address lds_addr + 16 l (M0 carries the
// wave-uniform LDS base).  Issued through inline asm ON PURPOSE: for an LDS-DMA it knows about, the compiler drains every
// pending DMA (s_waitcnt vmcnt(0)) in front of any LDS read it cannot prove disjoint from the destination — here the K / V
// fragment reads of the CURRENT tile while the NEXT tile's rows are in flight to the other buffer — which serialised the
// prefetch with the compute: 3.4 us per tile and workgroup, measured, i.e. one HBM round trip per tile.  Unseen by the
// compiler, completion is waited for explicitly in front of the per-tile barrier; the compiler's own waits for ITS loads can
// only wait longer than needed, never shorter (loads retire in order).
// The LDS destination is a BYTE ADDRESS off the dynamic region's base: a generic pointer costs a run-time address-space cast per use
// (a null check and a select), which this compiler also mis-selects in some code shapes (V_CMP_NE_U32 on src_shared_base: a build error).
__device__ __forceinline__ void mla_dma_row_a(const bf16_t* gsrc_lane, uint32_t lds_byte_addr) {
  const uint32_t lds_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc_lane), "s"(lds_addr)
               : "memory");
}
// The same with the row's address as a SCALAR base + a 32-bit per-lane byte offset (the saddr form): a staged ckv row is one wave-uniform
// pointer, so its arithmetic (page, row, append override) runs on the scalar unit — as 64-bit per-lane pointers it was ~45 of the tile
// loop's ~260 VALU instructions, and the loop is VALU-issue bound (profiles/r06_W_mla_valu_diet_ab.txt).
__device__ __forceinline__ void mla_dma_row_sa(const bf16_t* gsrc_row, uint32_t lane_byte_off, uint32_t lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(lane_byte_off), "s"(gsrc_row), "s"(lds_addr)
               : "memory");
}

// Workgroup = HBW head blocks (16 heads each) x DSPLIT slices: wave (hbw, ds) computes 512/DSPLIT dims of O += P V (which
// divides the fp32 accumulator registers — the 128-VGPR O tile of the undivided form left one wave per SIMD running a
// serial chain of LDS reads and MFMAs, ~7 us per 32-token tile) and every DSPLIT-th 32-wide k-step of S = Q K^T; the DSPLIT
// partial score tiles of a head block meet in LDS and every wave adds them in the same order, so all of them hold the
// same S bit for bit.  (Round 1 had every wave compute the whole S: 36 MFMAs and 36 KiB of LDS reads per wave and tile
// instead of 10 and 10 — measured 3.5-4 us per tile and workgroup at 128 heads.)
// KV split s owns tiles s, s + nsplit, s + 2 nsplit, ...: balanced to within one tile whatever kv_len turns out to be.
template <int HBW, int DSPLIT>
__global__ __launch_bounds__(HBW * DSPLIT * 64) void mla_decode_kernel(MlaParams p) {
  constexpr int NWV = HBW * DSPLIT;
  constexpr int NDT = 32 / DSPLIT;   // 16-dim output tiles per wave
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  bf16_t* Kt = reinterpret_cast<bf16_t*>(smem);                        // [2][32][584] row-major ckv rows (+ pad)
  bf16_t* Kp = Kt + 2 * MLA_TILE * MLA_KROW;                           // [2][32][64] k_pe rows, 16-byte pieces XOR-swizzled
  bf16_t* Pt = Kp + 2 * MLA_TILE * MLA_DR;                             // [NWV][16][32]
  float* Sx = reinterpret_cast<float*>(Pt + NWV * 16 * MLA_TILE);      // [NWV][8][64] partial score tiles
  constexpr int NQ = (18 + DSPLIT - 1) / DSPLIT;                        // k-steps of S per wave: ds, ds + DSPLIT, ...
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x, hb = blockIdx.y, qt = blockIdx.z;
  MLA_TS(0);

  // ---- Q fragments first: A[m = head (lane&15)][k = (lane>>4)*8 .. +7] for this wave's k-steps of 32 (of 16 nope + 2
  // rope).  Their addresses depend on nothing but the block indices, so they are in flight while the request lookup resolves.
  const int hbw = wave / DSPLIT, ds = wave % DSPLIT;
  const int head0 = (hb * HBW + hbw) * 16;
  v8bf qf[NQ];
  {
    const int h = head0 + (lane & 15);
    const bf16_t* qn = p.q_nope + ((size_t)qt * p.Hq + h) * MLA_DC + (lane >> 4) * 8;
    const bf16_t* qr = p.q_pe + ((size_t)qt * p.Hq + h) * MLA_DR + (lane >> 4) * 8;
#pragma unroll
    for (int j = 0; j < NQ; j++) {
      const int s = ds + j * DSPLIT;   // wave-uniform
      qf[j] = as_v8bf(make_uint4(0, 0, 0, 0));
      if (s < 16) qf[j] = as_v8bf(*reinterpret_cast<const uint4*>(qn + s * 32));
      else if (s < 18) qf[j] = as_v8bf(*reinterpret_cast<const uint4*>(qr + (s - 16) * 32));
    }
  }

  // which request owns query token qt?  Every wavefront resolves it on its own with ONE round of independent loads
  // (lane i looks at request i) — the live-request count *d_bsz is fetched in the same round and applied afterwards, not
  // waited for first: right after a kernel boundary each dependent load is another 1.5-3 us (measured with the phase stamps)
  int req = -1, kv_end = 0, page_base = 0, app_pos = -1;
  const int bsz_raw = p.d_bsz ? *p.d_bsz : p.batch;
  for (int b0 = 0; b0 < p.batch && req < 0; b0 += 64) {
    const int i = b0 + lane;
    int q0 = 0x7fffffff, q1 = 0, kl = 0, kp = 0;
    if (i < p.batch) {
      q0 = p.qo_indptr[i]; q1 = p.qo_indptr[i + 1]; kl = p.kv_len[i]; kp = p.kv_indptr[i];
      // never index past the pages the request owns: a sequence that outgrew its cache attends to (and appends within)
      // its last page instead of reading kv_indices / writing HBM out of bounds
      kl = min(kl, (p.kv_indptr[i + 1] - kp) * p.page_size);
    }
    const int B = min(max(bsz_raw, 0), p.batch);
    const unsigned long long hit = __ballot(i < B && qt >= q0 && qt < q1);
    if (hit) {
      const int src = __ffsll((long long)hit) - 1;
      q0 = __builtin_amdgcn_readlane(q0, src); q1 = __builtin_amdgcn_readlane(q1, src);
      kl = __builtin_amdgcn_readlane(kl, src); kp = __builtin_amdgcn_readlane(kp, src);
      req = b0 + src;
      const int qo_len = q1 - q0;
      kv_end = max(kl - qo_len + (qt - q0) + 1, 0);   // causal: positions < kv_end are visible
      page_base = kp;
      // fused cache append (decode: one new token per request): the newest position is taken from the append buffers
      if (p.app_ckv && qo_len == 1) app_pos = kl - 1;
    }
  }
  req = __builtin_amdgcn_readfirstlane(req);
  kv_end = __builtin_amdgcn_readfirstlane(kv_end);
  page_base = __builtin_amdgcn_readfirstlane(page_base);
  app_pos = __builtin_amdgcn_readfirstlane(app_pos);
  const size_t pidx = ((size_t)qt * p.Hq + head0) * p.nsplit + split;  // + head*nsplit per head
  if (req < 0) return;
  MLA_TS(1);

  const int ntiles = (kv_end + MLA_TILE - 1) / MLA_TILE;
  const int t_begin = split, t_end = ntiles, t_step = p.nsplit;   // tiles t_begin, t_begin + t_step, ... < t_end

  v4f o[NDT];
#pragma unroll
  for (int i = 0; i < NDT; i++) o[i] = v4f{0.f, 0.f, 0.f, 0.f};
  float m_run[4], l_run[4];
#pragma unroll
  for (int r = 0; r < 4; r++) { m_run[r] = -__builtin_inff(); l_run[r] = 0.f; }

  if (t_begin < t_end) {
    // The appended token (decode: one new row per request) is taken from the append buffers wherever its position is staged;
    // the workgroup (hb 0) whose split owns its tile also writes it into the cache (StaticCache.update, custom_cache.py:189-195)
    if (app_pos >= 0 && hb == 0 && (app_pos / MLA_TILE) % p.nsplit == split && tid < 72) {
      const int page = p.kv_indices ? p.kv_indices[page_base + app_pos / p.page_size] : page_base + app_pos / p.page_size;
      const size_t trow = (size_t)page * p.page_size + app_pos % p.page_size;
      if (tid < 64) *reinterpret_cast<uint4*>(p.ckv_w + trow * p.ckv_ts + tid * 8) = *reinterpret_cast<const uint4*>(p.app_ckv + (size_t)req * MLA_DC + tid * 8);
      else *reinterpret_cast<uint4*>(p.kpe_w + trow * p.kpe_ts + (tid - 64) * 8) = *reinterpret_cast<const uint4*>(p.app_kpe + (size_t)req * MLA_DR + (tid - 64) * 8);
    }

    // Staging of one 32-token tile, entirely by LDS-DMA (no VGPR destination, nothing the compiler has to wait for inside the
    // loop).  A tile lies inside one page (page_size % 32 == 0 is required by the launcher), whose index is a scalar load.
    // ckv: one 1-KiB row per wave-instruction into Kt[row]; k_pe: eight 128-byte rows per wave-instruction into Kp, 16-byte
    // piece g of row r at position g ^ (r & 7) (the XOR keeps the fragment reads of 16 rows at most 2-way bank-conflicted
    // although the rows are packed).  Rows past the end of the context re-read the last valid row: finite values that the
    // softmax weights them with exactly 0 (their scores are masked), and nothing beyond kv_len is ever touched.
    // (LDS destinations as byte addresses off the dynamic region's base, Kt = smem: see mla_dma_row_a)
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)smem;
    constexpr uint32_t KT_BYTES = MLA_TILE * MLA_KROW * 2, KP_BYTES = MLA_TILE * MLA_DR * 2;
    // The page of a tile is looked up ONE TILE AHEAD of its staging (page_of: the load is issued behind the previous tile's DMA requests
    // and waited for by the loop's own vmcnt(0) in front of the barrier): looked up inside stage() the index was a dependent load in
    // front of every tile's addresses — one exposed L2 round trip per tile with a real page table (the serving seam; the single-request
    // cache passes kv_indices == NULL).
    auto page_of = [&](int tile) -> int {
      const int pidx_ = __builtin_amdgcn_readfirstlane(page_base + (p.tiles_shift >= 0 ? tile >> p.tiles_shift : tile * MLA_TILE / p.page_size));
      return p.kv_indices ? p.kv_indices[pidx_] : pidx_;
    };
    auto stage = [&](int tile, int page_v, int buf) {   // buf = which of the two staged tiles
      const uint32_t dK = lds0 + (uint32_t)buf * KT_BYTES, dP = lds0 + 2 * KT_BYTES + (uint32_t)buf * KP_BYTES;
      const int tok0 = tile * MLA_TILE;
      const int page0 = __builtin_amdgcn_readfirstlane(page_v);
      const size_t row0 = (size_t)page0 * p.page_size + (p.tiles_shift >= 0 ? tok0 & (p.page_size - 1) : tok0 % p.page_size);
      const int last = kv_end - 1 - tok0;   // last valid row of the tile (>= 0: the tile holds a visible token)
      // a whole tile without the appended row (all but one or two tiles of a split): one 64-bit product for the tile, then adds — the
      // general form below spends ~25 scalar instructions per row on clamps, the append override and pointer casts (SQ counters of
      // the loop: as many SALU as VALU instructions, profiles/r06_Z_pmc_mla_decode_128k.txt)
      if (last >= MLA_TILE - 1 && (unsigned)(app_pos - tok0) >= (unsigned)MLA_TILE) {
        const bf16_t* src = p.ckv + (row0 + wave) * p.ckv_ts;
        const size_t step = (size_t)NWV * p.ckv_ts;
        uint32_t dst = dK + (uint32_t)wave * (MLA_KROW * 2);
#pragma unroll
        for (int r = wave; r < MLA_TILE; r += NWV, src += step, dst += NWV * MLA_KROW * 2) mla_dma_row_sa(src, (uint32_t)lane * 16u, dst);
        if (wave < 4) {
          const int r = wave * 8 + (lane >> 3), g = (lane & 7) ^ (lane >> 3);
          mla_dma_row_a(p.k_pe + (row0 + r) * p.kpe_ts + g * 8, dP + (uint32_t)wave * (8 * MLA_DR * 2));
        }
        return;
      }
#pragma unroll
      for (int r = wave; r < MLA_TILE; r += NWV) {   // (every term wave-uniform: scalar arithmetic, mla_dma_row_sa)
        const int rr = min(r, last);
        const bf16_t* src = p.ckv + (row0 + rr) * p.ckv_ts;
        if (tok0 + rr == app_pos) src = p.app_ckv + (size_t)req * MLA_DC;
        mla_dma_row_sa(src, (uint32_t)lane * 16u, dK + (uint32_t)r * (MLA_KROW * 2));
      }
      if (wave < 4) {
        const int r = wave * 8 + (lane >> 3), rr = min(r, last);
        const int g = (lane & 7) ^ (lane >> 3);
        const bf16_t* src = p.k_pe + (row0 + rr) * p.kpe_ts;
        if (tok0 + rr == app_pos) src = p.app_kpe + (size_t)req * MLA_DR;
        mla_dma_row_a(src + g * 8, dP + (uint32_t)wave * (8 * MLA_DR * 2));
      }
    };

    stage(t_begin, page_of(t_begin), 0);
    int page_nx = page_of(min(t_begin + t_step, t_end - 1));   // (clamped: always a tile of this request)
    MLA_TS(2);
    bf16_t* Pw = Pt + wave * 16 * MLA_TILE;

    int cur = 0;
    for (int tile = t_begin; tile < t_end; tile += t_step, cur ^= 1) {
      bf16_t* Kc = Kt + cur * MLA_TILE * MLA_KROW;
      const bf16_t* Pc = Kp + cur * MLA_TILE * MLA_DR;
      const int tok0 = tile * MLA_TILE;
      const int ntok = min(MLA_TILE, kv_end - tok0);
      // One barrier per tile: behind the explicit wait for this wave's LDS-DMA rows of this tile it publishes them and
      // proves every wave is done reading the other buffer, which the next tile's DMA may now overwrite.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      MLA_TS_LOOP(3);
      if (tile + t_step < t_end) {
        stage(tile + t_step, page_nx, cur ^ 1);
        page_nx = page_of(min(tile + 2 * t_step, t_end - 1));
      }

      // ---- S = Q K^T for 2 x 16 tokens: this wave's k-steps -------------------------------------------------------------
      v4f s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
      const bf16_t* kb0 = Kc + (lane & 15) * MLA_KROW + (lane >> 4) * 8;
      const bf16_t* kb1 = kb0 + 16 * MLA_KROW;
#pragma unroll
      for (int j = 0; j < NQ; j++) {
        const int s = ds + j * DSPLIT;
        if (s < 16) {   // wave-uniform
          const v8bf b0 = as_v8bf(*reinterpret_cast<const uint4*>(kb0 + s * 32));
          const v8bf b1 = as_v8bf(*reinterpret_cast<const uint4*>(kb1 + s * 32));
          s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[j], b0, s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[j], b1, s1, 0, 0, 0);
        } else if (s < 18) {   // rope k-step: piece (s-16)*4 + (lane>>4) of row (lane&15) (+16), un-swizzled
          const int row = lane & 15, q = (s - 16) * 4 + (lane >> 4);
          const bf16_t* pr = Pc + row * MLA_DR + ((q ^ (row & 7)) * 8);
          const v8bf b0 = as_v8bf(*reinterpret_cast<const uint4*>(pr));
          const v8bf b1 = as_v8bf(*reinterpret_cast<const uint4*>(pr + 16 * MLA_DR));
          s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[j], b0, s0, 0, 0, 0);
          s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[j], b1, s1, 0, 0, 0);
        }
      }
      if constexpr (DSPLIT > 1) {   // the head block's partial score tiles meet in LDS; fixed order -> identical S in every wave
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
      MLA_TS_LOOP(4);
      // lane holds S[head = (lane>>4)*4 + r][token = tok0 + (lane&15) (+16)]
      const bool v0 = (lane & 15) < ntok, v1 = 16 + (lane & 15) < ntok;
      float alpha[4], sa[4], sb[4], mx[4], ps[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        sa[r] = v0 ? s0[r] * p.sm_scale : -__builtin_inff();
        sb[r] = v1 ? s1[r] * p.sm_scale : -__builtin_inff();
        mx[r] = fmaxf(sa[r], sb[r]);
      }
      row16_max4(mx[0], mx[1], mx[2], mx[3]);   // (the four heads' reductions interleaved: ktx_common.h)
      float pav[4], pbv[4], mnew[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        mnew[r] = fmaxf(m_run[r], mx[r]);        // finite: every processed tile has >= 1 visible token
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
        // P to LDS in [head][token] order for the A-operand re-read (one v_cvt_pk_bf16_f32 for the pair)
        const int hrow = (lane >> 4) * 4 + r;
        const uint32_t pk = ktx_pk_bf16(pa, pb);
        Pw[hrow * MLA_TILE + (lane & 15)] = (bf16_t)(pk & 0xffffu);
        Pw[hrow * MLA_TILE + 16 + (lane & 15)] = (bf16_t)(pk >> 16);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const v8bf pf = as_v8bf(*reinterpret_cast<const uint4*>(Pw + (lane & 15) * MLA_TILE + (lane >> 4) * 8));

      // ---- O = O*alpha + P V ; V fragments by transposed reads of the same K tile ------------------------------------
      const bf16_t* vb = Kc + ((lane >> 4) * 8 + ((lane & 15) >> 2)) * MLA_KROW + (lane & 3) * 4 + ds * NDT * 16;
      // the rescale only where some head's running maximum moved (wave-uniform branch; alpha == 1 exactly otherwise, and o * 1 is o:
      // same bits) — past the first tiles of a long context that is rare, and the 4 * NDT multiplies were a sixth of the loop's VALU
      const bool moved = alpha[0] != 1.f || alpha[1] != 1.f || alpha[2] != 1.f || alpha[3] != 1.f;
      if (__builtin_amdgcn_ballot_w64(moved) != 0ull) {
#pragma unroll
        for (int i = 0; i < NDT; i++)
#pragma unroll
          for (int r = 0; r < 4; r++) o[i][r] *= alpha[r];
      }
#pragma unroll
      for (int i = 0; i < NDT; i++) {
        const v8bf b = load_v_frag(vb + i * 16);
        o[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, b, o[i], 0, 0, 0);
      }
      MLA_TS_LOOP(5);
    }
  }
  MLA_TS(6);

  if (p.out1) {   // one split holds the whole row: the merge kernel's arithmetic for a single partial (weight 1), in place
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int h = head0 + (lane >> 4) * 4 + r;
      const float l = __shfl(l_run[r], lane & 48), m = __shfl(m_run[r], lane & 48);
      const float inv = l > 0.f ? 1.0f / l : 0.f;
      bf16_t* po = p.out1 + ((size_t)qt * p.Hq + h) * MLA_DC + ds * NDT * 16;
#pragma unroll
      for (int i = 0; i < NDT; i++) po[i * 16 + (lane & 15)] = f32_to_bf16(o[i][r] * inv);
      if (p.lse1 && (lane & 15) == 0 && ds == 0)
        p.lse1[(size_t)qt * p.Hq + h] = l > 0.f ? (m + __logf(l)) * 1.44269504089f : -__builtin_inff();
    }
    return;
  }
  // ---- partial results: un-normalised O plus (m, l) per head ------------------------------------------------------------
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int hrow = (lane >> 4) * 4 + r;
    if ((lane & 15) == 0 && ds == 0) {
      float* ml = p.part_ml + (pidx + (size_t)hrow * p.nsplit) * 2;
      ml[0] = m_run[r];
      ml[1] = l_run[r];
    }
    if (t_begin < t_end) {
      float* po = p.part_o + (pidx + (size_t)hrow * p.nsplit) * MLA_DC + ds * NDT * 16;
#pragma unroll
      for (int i = 0; i < NDT; i++) po[i * 16 + (lane & 15)] = o[i][r];
    }
  }
  MLA_TS(7);
}

// merge the KV splits: one 256-thread workgroup per (query token, head, quarter of the 512 output dims) — a workgroup
// pulls cross-XCD data at only ~65 GB/s (MI355X_MICROARCH.md handoff-payload), so the partials of one head are spread
// over four workgroups.  Thread (sl, dg) owns splits sl, sl+16, ... for 8 dims.  EVERY global load of the kernel — the
// (m, l) pairs, the partial outputs, the bounds — is issued before the first use: right after the kernel boundary each of
// them is an L2 miss, and a stats -> weights -> addresses -> loads chain would pay that latency three times over.  A dead
// split's partial may be stale memory, so it is selected away (not multiplied by 0).
template <int NS>   // splits per thread: nsplit <= 16 * NS
__global__ __launch_bounds__(256) void mla_merge_kernel(MlaParams p, bf16_t* out, float* lse) {
  __shared__ float s_w[1024];
  __shared__ float s_red[8];
  __shared__ float s_acc[16][128 + 4];
  const int qt = blockIdx.y, h = blockIdx.x, quarter = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dg = tid & 15, sl = tid >> 4;   // 16 dim groups of 8 dims x 16 split lanes
  const size_t base = ((size_t)qt * p.Hq + h) * p.nsplit;
  // ---- all loads
  int B = p.batch;
  const int bsz_raw = p.d_bsz ? *p.d_bsz : p.batch;
  float2 ml[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int s = tid + i * 256;
    ml[i] = s < p.nsplit ? *reinterpret_cast<const float2*>(p.part_ml + (base + s) * 2) : make_float2(0.f, 0.f);
  }
  float4 va[NS], vb[NS];
  const float* po = p.part_o + base * MLA_DC + quarter * 128 + dg * 8;
#pragma unroll
  for (int u = 0; u < NS; u++) {
    const int s = min(sl + 16 * u, p.nsplit - 1);
    va[u] = *reinterpret_cast<const float4*>(po + (size_t)s * MLA_DC);
    vb[u] = *reinterpret_cast<const float4*>(po + (size_t)s * MLA_DC + 4);
  }
  if (p.d_bsz) B = min(max(bsz_raw, 0), p.batch);
  const int q_total = p.qo_indptr[B];

  // ---- softmax statistics, split-parallel; weights parked in LDS
  float mstar = -__builtin_inff();
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (ml[i].y > 0.f) mstar = fmaxf(mstar, ml[i].x);
  mstar = wave_max(mstar);
  if (lane == 0) s_red[wave] = mstar;
  __syncthreads();
  mstar = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  float lsum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int s = tid + i * 256;
    const float w = ml[i].y > 0.f ? __expf(ml[i].x - mstar) : 0.f;
    if (s < p.nsplit) s_w[s] = w;
    lsum += ml[i].y * w;
  }
  lsum = wave_sum(lsum);
  if (lane == 0) s_red[4 + wave] = lsum;
  __syncthreads();
  lsum = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);

  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < NS; u++) {
    const int s = sl + 16 * u;
    const float w = s < p.nsplit ? s_w[s] : 0.f;
    if (w > 0.f) {
      acc[0] += va[u].x * w; acc[1] += va[u].y * w; acc[2] += va[u].z * w; acc[3] += va[u].w * w;
      acc[4] += vb[u].x * w; acc[5] += vb[u].y * w; acc[6] += vb[u].z * w; acc[7] += vb[u].w * w;
    }
  }
#pragma unroll
  for (int q = 0; q < 8; q++) s_acc[sl][dg * 8 + q] = acc[q];
  __syncthreads();
  if (tid < 64 && qt < q_total) {   // 128 dims: two per thread
    const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
    float v0 = 0.f, v1 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) { v0 += s_acc[i][2 * tid]; v1 += s_acc[i][2 * tid + 1]; }
    const uint32_t w = (uint32_t)f32_to_bf16(v0 * inv) | ((uint32_t)f32_to_bf16(v1 * inv) << 16);
    *reinterpret_cast<uint32_t*>(out + ((size_t)qt * p.Hq + h) * MLA_DC + quarter * 128 + 2 * tid) = w;
    if (lse && tid == 0 && quarter == 0)
      lse[(size_t)qt * p.Hq + h] = lsum > 0.f ? (mstar + __logf(lsum)) * 1.44269504089f : -__builtin_inff();
  }
}

__global__ void mla_cache_append_kernel(bf16_t* cache, long long ts, int page_size, const bf16_t* ckv, const bf16_t* kpe,
                                        const int32_t* page_idx, const int32_t* page_off, const int32_t* ntok,
                                        int max_tokens, int num_pages) {
  int T = max_tokens;
  if (ntok) T = min(max(*ntok, 0), max_tokens);
  const int t = blockIdx.x;
  if (t >= T) return;
  if (num_pages > 0 && ((unsigned)page_idx[t] >= (unsigned)num_pages || (unsigned)page_off[t] >= (unsigned)page_size)) return;
  bf16_t* dst = cache + ((size_t)page_idx[t] * page_size + page_off[t]) * ts;
  const int i = threadIdx.x;  // 72 threads x 16 B = 576 bf16
  if (i < 64) *reinterpret_cast<uint4*>(dst + i * 8) = *reinterpret_cast<const uint4*>(ckv + (size_t)t * MLA_DC + i * 8);
  else if (i < 72) *reinterpret_cast<uint4*>(dst + i * 8) = *reinterpret_cast<const uint4*>(kpe + (size_t)t * MLA_DR + (i - 64) * 8);
}

static long long* g_mla_dbg = nullptr;
extern "C" int ktx_mla_debug_stamps(long long* d_buf) {   // device buffer of >= 16 * workgroups entries, or NULL to stop
  g_mla_dbg = d_buf;
  return 0;
}

extern "C" size_t ktx_mla_workspace_bytes(const ktx_mla_config* cfg, int max_q_tokens) {
  if (!cfg || max_q_tokens <= 0) return 0;
  // one (O[512], m, l) record per (token, head, split).  The launcher never uses more than ~2048 workgroups, so beyond
  // that the split count shrinks as the token count grows; it also clamps the split count to what the buffer holds.
  const size_t per_split = (size_t)max_q_tokens * cfg->num_heads;
  const size_t cap = (size_t)2048 * 64 + per_split;
  const size_t n = std::max(per_split, std::min(per_split * (size_t)std::max(1, cfg->max_splits), cap));
  return n * (MLA_DC + 2) * sizeof(float);
}

extern "C" int ktx_mla_decode(const ktx_mla_config* cfg, const void* d_q_nope, const void* d_q_pe, const void* d_ckv,
                              const void* d_k_pe, int64_t ckv_token_stride, int64_t kpe_token_stride,
                              const int32_t* d_qo_indptr, const int32_t* d_kv_indptr, const int32_t* d_kv_indices,
                              const int32_t* d_kv_len_arr, const int32_t* d_bsz, int batch, int total_q_tokens,
                              void* d_out, float* d_lse, void* d_workspace, size_t workspace_bytes, void* stream) {
  return ktx_mla_decode_append(cfg, d_q_nope, d_q_pe, const_cast<void*>(d_ckv), const_cast<void*>(d_k_pe), ckv_token_stride,
                               kpe_token_stride, d_qo_indptr, d_kv_indptr, d_kv_indices, d_kv_len_arr, d_bsz, batch,
                               total_q_tokens, nullptr, nullptr, d_out, d_lse, d_workspace, workspace_bytes, stream);
}

static int mla_decode_impl(const ktx_mla_config* cfg, const void* d_q_nope, const void* d_q_pe, void* d_ckv,
                           void* d_k_pe, int64_t ckv_token_stride, int64_t kpe_token_stride,
                           const int32_t* d_qo_indptr, const int32_t* d_kv_indptr, const int32_t* d_kv_indices,
                           const int32_t* d_kv_len_arr, const int32_t* d_bsz, int batch, int total_q_tokens,
                           const void* d_new_ckv, const void* d_new_kpe, void* d_out, float* d_lse,
                           void* d_workspace, size_t workspace_bytes, void* stream, int* partials_nsplit);

extern "C" int ktx_mla_decode_append(const ktx_mla_config* cfg, const void* d_q_nope, const void* d_q_pe, void* d_ckv,
                                     void* d_k_pe, int64_t ckv_token_stride, int64_t kpe_token_stride,
                                     const int32_t* d_qo_indptr, const int32_t* d_kv_indptr, const int32_t* d_kv_indices,
                                     const int32_t* d_kv_len_arr, const int32_t* d_bsz, int batch, int total_q_tokens,
                                     const void* d_new_ckv, const void* d_new_kpe, void* d_out, float* d_lse,
                                     void* d_workspace, size_t workspace_bytes, void* stream) {
  KTX_REQUIRE(d_out, "ktx_mla_decode: null pointer");
  return mla_decode_impl(cfg, d_q_nope, d_q_pe, d_ckv, d_k_pe, ckv_token_stride, kpe_token_stride, d_qo_indptr, d_kv_indptr,
                         d_kv_indices, d_kv_len_arr, d_bsz, batch, total_q_tokens, d_new_ckv, d_new_kpe, d_out, d_lse, d_workspace,
                         workspace_bytes, stream, nullptr);
}

extern "C" int ktx_mla_decode_partials(const ktx_mla_config* cfg, const void* d_q_nope, const void* d_q_pe, void* d_ckv,
                                       void* d_k_pe, int64_t ckv_token_stride, int64_t kpe_token_stride,
                                       const int32_t* d_qo_indptr, const int32_t* d_kv_indptr, const int32_t* d_kv_indices,
                                       const int32_t* d_kv_len_arr, const int32_t* d_bsz, int batch, int total_q_tokens,
                                       const void* d_new_ckv, const void* d_new_kpe, void* d_workspace, size_t workspace_bytes,
                                       int* nsplit_out, void* stream) {
  KTX_REQUIRE(nsplit_out, "ktx_mla_decode_partials: null nsplit_out");
  return mla_decode_impl(cfg, d_q_nope, d_q_pe, d_ckv, d_k_pe, ckv_token_stride, kpe_token_stride, d_qo_indptr, d_kv_indptr,
                         d_kv_indices, d_kv_len_arr, d_bsz, batch, total_q_tokens, d_new_ckv, d_new_kpe, nullptr, nullptr, d_workspace,
                         workspace_bytes, stream, nsplit_out);
}

// Workgroup shape and KV split count of a decode call — one place, because the one-launch decode step (ktx_attn.hip) must
// split the context exactly as the stand-alone kernel does to reproduce its partials bit for bit.
static void mla_pick_shape(const ktx_mla_config* cfg, int total_q_tokens, size_t workspace_bytes, int* shape_out, int* nsplit_out,
                           bool one_launch = false) {
  const int Hq = cfg->num_heads;
  // workgroup shape (head blocks x dim slices): decode-sized calls of many-headed models take 2x4 (32 heads share one staged
  // KV tile, 128 output dims per wave), prompts of those models 4x2, everything else 1x4
  const bool decode_sized = total_q_tokens <= 16;
  // long contexts (many tiles per split): 64 heads per workgroup share each staged tile and twice the splits pay off — measured
  // at 128 heads / 128K tokens (scripts/mla_sweep.py --ctx 131072): 4x2 with 128 splits 103 us, 2x4 with 64 splits 157 us
  // (one_launch: the split rule of csrc/ktx_attn.hip's phase C, which is built on the 2x4 shape — it keeps that shape, with its
  // <= 256 / head-blocks splits, past 8192 tokens: deeper splits instead of the 4x2 shape's wider ones)
  const bool long_hint = decode_sized && cfg->kv_len_hint >= 8192 && Hq % 64 == 0;
  const bool long_ctx = long_hint && !one_launch;
  // a decode BATCH (>= 8 query tokens, each with its own context): every head block of a token re-reads that token's latent rows, so
  // 64 heads per workgroup halve the KV traffic (V3 dims, 8 x 4096 tokens: 150 -> 75 MB per layer; batch-of-8 step 14.47 -> 13.91 ms,
  // profiles/r06_zz_bs8_mla_shape.txt)
  const bool batched = decode_sized && total_q_tokens >= 8 && Hq % 64 == 0 && !one_launch;
  int shape = (Hq % 32 == 0 && Hq >= 64 && decode_sized && !long_ctx && !batched) ? 2 : (Hq % 64 == 0) ? 4 : 1;
  {   // tuning knobs (scripts/mla_sweep.py): 6 = force the workgroup shape, 7 = force the split count
    const int fs = ktx_debug_get(6);
    if ((fs == 1) || (fs == 2 && Hq % 32 == 0) || (fs == 4 && Hq % 64 == 0)) shape = fs;
  }
  const int hbw = shape;
  const int hblocks = Hq / (16 * hbw);
  // KV splits: one 32-token tile per workgroup whenever the grid stays under ~2048 workgroups (measured on 16 heads: extra
  // tiles per workgroup cost more than the extra partials cost the merge kernel), bounded by the workspace — and by the
  // partials themselves: every split writes Hq x 514 floats per token that the merge kernel reads back (at 128 heads and 144
  // splits that was 38 MB per layer, 8x the 4.7 MB of latent rows the attention actually needs), so the split count is
  // capped where the partials reach ~16 MB and at about one workgroup per CU
  int nsplit = std::max(1, 2048 / std::max(1, hblocks * total_q_tokens));
  const size_t per_split_bytes = (size_t)total_q_tokens * Hq * (MLA_DC + 2) * sizeof(float);
  nsplit = std::min<int>(nsplit, std::max<size_t>(16, ((size_t)(long_hint ? 34 : 16) << 20) / per_split_bytes));
  if (long_hint) nsplit = std::min(nsplit, 128);
  if (shape == 2) nsplit = std::min(nsplit, std::max(16, 256 / std::max(1, hblocks * total_q_tokens)));
  if (ktx_debug_get(7) > 0) nsplit = ktx_debug_get(7);
  if (cfg->kv_len_hint > 0) {
    // no more splits than tiles — and no more than the deepest split needs: with T tiles dealt round-robin the launch takes
    // ceil(T / nsplit) tiles' time, so the smallest split count with that depth writes the fewest partials for the merge to read
    const int tiles = std::max(1, (cfg->kv_len_hint + MLA_TILE - 1) / MLA_TILE);
    nsplit = std::min(nsplit, tiles);
    const int depth = (tiles + nsplit - 1) / nsplit;
    if (ktx_debug_get(7) <= 0) nsplit = (tiles + depth - 1) / depth;
  }
  // <= 256 splits: the merge kernel keeps splits/16 partial rows per thread in registers (NS = 16 is its largest, spill-free
  // instantiation); longer contexts simply put more 32-token tiles into each split
  nsplit = std::min(nsplit, std::min(256, std::max(1, cfg->max_splits)));
  nsplit = (int)std::min<size_t>((size_t)nsplit, workspace_bytes / ((size_t)total_q_tokens * Hq * (MLA_DC + 2) * sizeof(float)));
  *shape_out = shape;
  *nsplit_out = nsplit;
}
extern "C" int ktx_mla_decode_nsplit(const ktx_mla_config* cfg, int total_q_tokens, size_t workspace_bytes) {   // ktx_internal.h
  int shape = 0, nsplit = 0;
  if (!cfg || total_q_tokens <= 0) return 0;
  // Up to 40 K tokens the one launch wins with deeper 2x4 splits: 61.7 vs 68.4 us per layer at 12 K, 64.1 vs 72.0 at 16 K, 77.9 vs 80.5 at
  // 32 K, 81.7 vs 83.4 at 36 K, even from 40 K on (84.8 / 84.8; 48 K 95.1 vs 89.0 before phase C's staging fast path) — the five launches
  // with the 4x2 shape's 128 splits take over there (profiles/r06_AG_attn_fused_long_ctx.txt, final round-6 kernels; the bound was 12 K
  // before the launch lost 7 us to section 4.1.6).
  const int bound_k = ktx_debug_get(1);   // dev knob 1: the bound in K tokens (scripts/attn_fused_bench.py, ATTN_BOUND_K)
  if (cfg->kv_len_hint >= (bound_k > 0 ? bound_k * 1024 : 40960)) return 0;
  mla_pick_shape(cfg, total_q_tokens, workspace_bytes, &shape, &nsplit, true);
  return shape == 2 ? nsplit : 0;   // the one-launch step is built on the 2x4 workgroup shape only
}

static int mla_decode_impl(const ktx_mla_config* cfg, const void* d_q_nope, const void* d_q_pe, void* d_ckv,
                           void* d_k_pe, int64_t ckv_token_stride, int64_t kpe_token_stride,
                           const int32_t* d_qo_indptr, const int32_t* d_kv_indptr, const int32_t* d_kv_indices,
                           const int32_t* d_kv_len_arr, const int32_t* d_bsz, int batch, int total_q_tokens,
                           const void* d_new_ckv, const void* d_new_kpe, void* d_out, float* d_lse,
                           void* d_workspace, size_t workspace_bytes, void* stream, int* partials_nsplit) {
  KTX_REQUIRE((d_new_ckv == nullptr) == (d_new_kpe == nullptr), "ktx_mla_decode_append: give both new_ckv and new_kpe or neither");
  KTX_REQUIRE(cfg && d_q_nope && d_q_pe && d_ckv && d_k_pe && d_workspace, "ktx_mla_decode: null pointer");
  KTX_REQUIRE(d_qo_indptr && d_kv_indptr && d_kv_len_arr, "ktx_mla_decode: null index array");
  KTX_REQUIRE(cfg->head_dim_ckv == MLA_DC && cfg->head_dim_kpe == MLA_DR, "ktx_mla_decode: only kv_lora_rank 512 + rope 64");
  KTX_REQUIRE(cfg->num_heads > 0 && cfg->num_heads % 16 == 0, "ktx_mla_decode: num_heads must be a multiple of 16");
  KTX_REQUIRE(batch > 0 && total_q_tokens > 0 && cfg->page_size > 0, "ktx_mla_decode: bad sizes");
  KTX_REQUIRE(cfg->page_size % MLA_TILE == 0, "ktx_mla_decode: page_size must be a multiple of 32 (the reference's caches use 64 and 256)");
  KTX_REQUIRE(ckv_token_stride % 8 == 0 && kpe_token_stride % 8 == 0, "ktx_mla_decode: token strides must be multiples of 8 elements");
  hipStream_t st = (hipStream_t)stream;
  const int Hq = cfg->num_heads;
  int shape = 0, nsplit = 0;
  mla_pick_shape(cfg, total_q_tokens, workspace_bytes, &shape, &nsplit);
  const bool wide = shape == 4;
  const int hbw = shape, nwv = shape == 1 ? 4 : 8;
  const int hblocks = Hq / (16 * hbw);
  KTX_REQUIRE(nsplit >= 1, "ktx_mla_decode: workspace too small");
  const size_t need = (size_t)total_q_tokens * Hq * nsplit * (MLA_DC + 2) * sizeof(float);
  KTX_REQUIRE(workspace_bytes >= need, "ktx_mla_decode: workspace too small");
  MlaParams p;
  p.q_nope = (const bf16_t*)d_q_nope; p.q_pe = (const bf16_t*)d_q_pe; p.ckv = (const bf16_t*)d_ckv; p.k_pe = (const bf16_t*)d_k_pe;
  p.ckv_ts = ckv_token_stride; p.kpe_ts = kpe_token_stride;
  p.qo_indptr = d_qo_indptr; p.kv_indptr = d_kv_indptr; p.kv_indices = d_kv_indices; p.kv_len = d_kv_len_arr; p.d_bsz = d_bsz;
  p.batch = batch; p.total_q = total_q_tokens; p.Hq = Hq; p.page_size = cfg->page_size; p.nsplit = nsplit;
  p.tiles_shift = -1;
  for (int sh = 0; sh < 20; sh++)
    if (cfg->page_size == (MLA_TILE << sh)) p.tiles_shift = sh;
  p.sm_scale = cfg->sm_scale;
  p.part_o = (float*)d_workspace;
  p.part_ml = p.part_o + (size_t)total_q_tokens * Hq * nsplit * MLA_DC;
  p.dbg = g_mla_dbg;
  int only = ktx_debug_get(5);   // measurement knob (include/ktx_moe.h): 1 = split-KV kernel only, 2 = merge only
  if (partials_nsplit) { only = 1; *partials_nsplit = nsplit; }   // the caller merges the partials itself (ktx_linear_forward_batched_merge)
  const bool direct = nsplit == 1 && only == 0;
  p.out1 = direct ? (bf16_t*)d_out : nullptr;
  p.lse1 = direct ? d_lse : nullptr;
  p.app_ckv = (const bf16_t*)d_new_ckv; p.app_kpe = (const bf16_t*)d_new_kpe; p.ckv_w = (bf16_t*)d_ckv; p.kpe_w = (bf16_t*)d_k_pe;
  const size_t lds = (size_t)(2 * MLA_TILE * MLA_KROW + 2 * MLA_TILE * MLA_DR + nwv * 16 * MLA_TILE) * sizeof(bf16_t) + (size_t)nwv * 512 * sizeof(float) + 16;
  const dim3 grid(nsplit, hblocks, total_q_tokens);
  // algorithmic bytes: the latent rows of the context (hint) + q / out rows; the split partials are an implementation artefact
  const double kv_bytes = (double)std::max(cfg->kv_len_hint, 1) * (MLA_DC + MLA_DR) * 2.0 * batch;
  if (only == 2) {
  } else if (shape == 2) {
    KTX_TIMED(st, kv_bytes + (double)total_q_tokens * Hq * (MLA_DC + MLA_DR + MLA_DC) * 2.0,
              "mla_decode_kernel<2,4> T=%d Hq=%d nsplit=%d", total_q_tokens, Hq, nsplit);
    const hipError_t e2 = ktx_set_max_lds(reinterpret_cast<const void*>(mla_decode_kernel<2, 4>), 128 * 1024);
    KTX_HIP(e2);
    hipLaunchKernelGGL((mla_decode_kernel<2, 4>), grid, dim3(512), lds, st, p);
  } else if (wide) {
    KTX_TIMED(st, kv_bytes + (double)total_q_tokens * Hq * (MLA_DC + MLA_DR + MLA_DC) * 2.0,
              "mla_decode_kernel<4,2> T=%d Hq=%d nsplit=%d", total_q_tokens, Hq, nsplit);
    const hipError_t e4 = ktx_set_max_lds(reinterpret_cast<const void*>(mla_decode_kernel<4, 2>), 128 * 1024);
    KTX_HIP(e4);
    hipLaunchKernelGGL((mla_decode_kernel<4, 2>), grid, dim3(512), lds, st, p);
  } else {
    KTX_TIMED(st, kv_bytes + (double)total_q_tokens * Hq * (MLA_DC + MLA_DR + MLA_DC) * 2.0,
              "mla_decode_kernel<1,4> T=%d Hq=%d nsplit=%d", total_q_tokens, Hq, nsplit);
    const hipError_t e1 = ktx_set_max_lds(reinterpret_cast<const void*>(mla_decode_kernel<1, 4>), 128 * 1024);
    KTX_HIP(e1);
    hipLaunchKernelGGL((mla_decode_kernel<1, 4>), grid, dim3(256), lds, st, p);
  }
  KTX_HIP(hipGetLastError());
  if (only != 1 && !direct) {
    KTX_TIMED(st, (double)total_q_tokens * Hq * MLA_DC * 2.0, "mla_merge_kernel T=%d Hq=%d nsplit=%d", total_q_tokens, Hq, nsplit);
    const dim3 mg(Hq, total_q_tokens, 4);
    if (p.nsplit <= 32) hipLaunchKernelGGL(mla_merge_kernel<2>, mg, dim3(256), 0, st, p, (bf16_t*)d_out, d_lse);
    else if (p.nsplit <= 64) hipLaunchKernelGGL(mla_merge_kernel<4>, mg, dim3(256), 0, st, p, (bf16_t*)d_out, d_lse);
    else if (p.nsplit <= 144) hipLaunchKernelGGL(mla_merge_kernel<9>, mg, dim3(256), 0, st, p, (bf16_t*)d_out, d_lse);
    else hipLaunchKernelGGL(mla_merge_kernel<16>, mg, dim3(256), 0, st, p, (bf16_t*)d_out, d_lse);
  }
  KTX_HIP(hipGetLastError());
  return 0;
}

// =====================================================================================================
// Non-absorbed prompt attention (KDeepseekV2Attention.forward_chunck / forward_linux_flashinfer prefill branch,
// archive/ktransformers/operators/attention.py:349-523: kv_b_proj expands the latents to per-head K_nope / V and a causal
// flash attention runs over qk dim 192 = 128 nope + 64 rope, v dim 128).  The absorbed kernel above spends 2*(576+512) flop
// per (query, key, head) and gives every query token its own workgroup; here it is 2*(192+128) and a workgroup owns 128
// queries of one head.
//   S^T = K Q^T : A = K tile from LDS (rows = keys), B = Q^T from registers; the C layout then leaves a lane with 4
//   consecutive keys of ONE query (lane & 15), so the online softmax is per lane and P never travels through LDS:
//   O^T += V^T P^T : A = V^T tile from LDS (rows = dims; the caller supplies V already transposed, [head][dim][key] — it is
//   the output of a GEMM either way), B = the lane's own P values.  The k-slot -> key map of that product is
//   slot (kc, e) -> key 16*(2*ks + e/4) + 4*kc + e%4, applied to both operands.
// One workgroup: 4 wavefronts x 2 query tiles of 16 = 128 queries, one head; keys in tiles of 64.
// =====================================================================================================
#define PF_BN 64
#define PF_KROW 200     // 192 + 8 bf16: 400 B rows (25 x 16 B)
#define PF_VROW 72      // 64 + 8 bf16: 144 B rows (9 x 16 B)
struct MlaPrefillParams {
  const bf16_t *q_nope, *q_pe;   // [T][H][128] / [T][H][64] with element strides below
  long long qn_ts, qn_hs, qp_ts, qp_hs;
  const bf16_t* k_nope;          // [H][kv_pad][128]
  const bf16_t* k_pe;            // [kv_len][64], token stride kpe_ts
  long long kpe_ts;
  const bf16_t* v_t;             // [H][128][kv_pad]
  bf16_t* out;                   // [T][H][128]
  int T, H, kv_len, kv_pad;
  float sm_scale;
  int no_skip;                   // dev knob 23 = 1: round 3's unconditional mask and rescale (A/B)
  int nqb;                       // > 0: 1-D XCD-aware grid of nqb query blocks x H heads (H % 8 == 0); 0: grid (query blocks, heads)
};

// one 64-key tile of one head in flight in registers: K_nope (64 keys x 16 pieces, contiguous), k_pe (64 keys x 8 pieces of the
// cache rows; nothing is valid past kv_len), V^T (128 dims x 8 pieces of 8 keys).  Branch-free and by value, so it lives in
// VGPRs (a lambda filling captured arrays under a condition was demoted to scratch memory).
struct PfTile { uint4 k0, k1, k2, k3, r0, r1, v0, v1, v2, v3; };
__device__ __forceinline__ uint4 pf_kpe_piece(const MlaPrefillParams& p, int j0, int idx) {
  const int key = j0 + (idx >> 3);
  const uint4 v = *reinterpret_cast<const uint4*>(p.k_pe + (size_t)min(key, p.kv_len - 1) * p.kpe_ts + (idx & 7) * 8);
  const uint32_t keep = key < p.kv_len ? 0xffffffffu : 0u;
  return make_uint4(v.x & keep, v.y & keep, v.z & keep, v.w & keep);
}
__device__ __forceinline__ PfTile pf_load_tile(const MlaPrefillParams& p, const bf16_t* kn, const bf16_t* vt, int tile, int tid) {
  const int j0 = tile * PF_BN;
  const bf16_t* kb = kn + (size_t)j0 * 128 + (size_t)tid * 8;
  const bf16_t* vb = vt + (size_t)(tid >> 3) * p.kv_pad + j0 + (tid & 7) * 8;
  const size_t vs = (size_t)32 * p.kv_pad;           // 256 threads = 32 dim rows per step
  PfTile t;
  t.k0 = *reinterpret_cast<const uint4*>(kb);
  t.k1 = *reinterpret_cast<const uint4*>(kb + 2048);
  t.k2 = *reinterpret_cast<const uint4*>(kb + 4096);
  t.k3 = *reinterpret_cast<const uint4*>(kb + 6144);
  t.r0 = pf_kpe_piece(p, j0, tid);
  t.r1 = pf_kpe_piece(p, j0, tid + 256);
  t.v0 = *reinterpret_cast<const uint4*>(vb);
  t.v1 = *reinterpret_cast<const uint4*>(vb + vs);
  t.v2 = *reinterpret_cast<const uint4*>(vb + 2 * vs);
  t.v3 = *reinterpret_cast<const uint4*>(vb + 3 * vs);
  return t;
}
__device__ __forceinline__ void pf_store_tile(const PfTile& t, bf16_t* Ks, bf16_t* Vs, int tid) {
  bf16_t* kd = Ks + (tid >> 4) * PF_KROW + (tid & 15) * 8;       // 256 threads = 16 key rows per step
  *reinterpret_cast<uint4*>(kd) = t.k0;
  *reinterpret_cast<uint4*>(kd + 16 * PF_KROW) = t.k1;
  *reinterpret_cast<uint4*>(kd + 32 * PF_KROW) = t.k2;
  *reinterpret_cast<uint4*>(kd + 48 * PF_KROW) = t.k3;
  bf16_t* rd = Ks + (tid >> 3) * PF_KROW + 128 + (tid & 7) * 8;  // 32 key rows per step
  *reinterpret_cast<uint4*>(rd) = t.r0;
  *reinterpret_cast<uint4*>(rd + 32 * PF_KROW) = t.r1;
  bf16_t* vd = Vs + (tid >> 3) * PF_VROW + (tid & 7) * 8;
  *reinterpret_cast<uint4*>(vd) = t.v0;
  *reinterpret_cast<uint4*>(vd + 32 * PF_VROW) = t.v1;
  *reinterpret_cast<uint4*>(vd + 64 * PF_VROW) = t.v2;
  *reinterpret_cast<uint4*>(vd + 96 * PF_VROW) = t.v3;
}

// Reductions over the four 16-lane rows of a wavefront (lanes l, l ^ 16, l ^ 32, l ^ 48).  Round 5: gfx950's row swaps — v_permlane32_swap
// exchanges the upper 32 lanes of one register with the lower 32 of another, v_permlane16_swap the odd rows with the even rows — are
// plain VALU instructions; the ds_bpermute shuffles they replace are LDS-crossbar round trips, four of them in a dependent chain per
// query tile and key tile.  Sum: (x_l + x_{l^32}) + (that of l ^ 16) — fp32 addition is commutative, so every lane of a query gets the
// same bits, and they are the bits the shuffle version produced ((x + shfl16) + shfl32 groups the rows as (r0 + r1) + (r2 + r3) for
// rows 0 / 1 and likewise here after the 32-swap first: see pf_sum_rows).
typedef unsigned pf_v2u __attribute__((ext_vector_type(2)));
typedef float pf_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float pf_max_rows(float x) {
  pf_v2u r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pf_sum_rows(float x) {
  // the shuffle version computed (x + x^16) first, then + the same of l ^ 32: keep that association — rows (r0 + r1) and (r2 + r3)
  // first (16-swap), then across the halves (32-swap)
  pf_v2u r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// MINW = wavefronts per SIMD the register allocation must allow.  Unconstrained (MINW = 1) the compiler takes 340 VGPRs: ONE
// wavefront per SIMD, one workgroup per CU, so the QK MFMAs, the softmax VALU chain, the PV MFMAs and the staging of the next
// tile run strictly one after the other (0.10 of the MFMA peak).  MINW = 2 caps the allocation at 256 (7 values
// spill to scratch) and two workgroups share a CU: one's softmax runs under the other's MFMAs.  Dev knob 22 = 1 selects MINW = 1.
// NU = query tiles of 16 per wavefront: 2 (default: 128 queries per workgroup, every K fragment read feeds two MFMAs) or 1 (dev knob 31:
// 64 queries per workgroup, half the registers -> three wavefronts per SIMD; measured in DESIGN.md 4.2.4)
template <int MINW, int NU>
__global__ __launch_bounds__(256, MINW) void mla_prefill_kernel(MlaPrefillParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);      // [64][PF_KROW]
  bf16_t* Vs = Ks + PF_BN * PF_KROW;                 // [128][PF_VROW]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blockIdx -> (head, query block).  Round 5, XCD-aware (p.nqb > 0: a 1-D grid): workgroup ids are dealt round-robin to the 8 XCDs, so
  // id % 8 picks the XCD; all query blocks of ONE head go to one XCD (head % 8 == XCD) and run there back to back — the head's K / V^T
  // (1 MB at 2048 keys) is fetched into that XCD's L2 once instead of into all eight.  Within a head: heaviest query blocks first
  // (causal: the last block sees the whole context).
  int h, qb;
  if (p.nqb > 0) {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    h = (slot / p.nqb) * 8 + xcd;
    qb = p.nqb - 1 - slot % p.nqb;
  } else {
    h = blockIdx.y;
    qb = (int)gridDim.x - 1 - (int)blockIdx.x;
  }
  const int q0 = qb * (64 * NU) + wave * (16 * NU);
  const int qi = lane & 15, g = lane >> 4;
  const int pos_off = p.kv_len - p.T;                // query t sits at position pos_off + t

  // Q^T fragments: lane (query qi, k-chunk g)
  v8bf qf[NU][6];
  int tq[NU];
#pragma unroll
  for (int u = 0; u < NU; u++) {
    tq[u] = min(q0 + u * 16 + qi, p.T - 1);
    const bf16_t* qn = p.q_nope + (size_t)tq[u] * p.qn_ts + (size_t)h * p.qn_hs + g * 8;
    const bf16_t* qr = p.q_pe + (size_t)tq[u] * p.qp_ts + (size_t)h * p.qp_hs + g * 8;
#pragma unroll
    for (int s = 0; s < 4; s++) qf[u][s] = as_v8bf(*reinterpret_cast<const uint4*>(qn + s * 32));
#pragma unroll
    for (int s = 0; s < 2; s++) qf[u][4 + s] = as_v8bf(*reinterpret_cast<const uint4*>(qr + s * 32));
  }
  v4f o[NU][8];
  float m_run[NU], l_run[NU];
#pragma unroll
  for (int u = 0; u < NU; u++) {
#pragma unroll
    for (int i = 0; i < 8; i++) o[u][i] = v4f{0.f, 0.f, 0.f, 0.f};
    m_run[u] = -__builtin_inff();
    l_run[u] = 0.f;
  }

  const int pos_max = pos_off + min(qb * (64 * NU) + 64 * NU - 1, p.T - 1);
  const int ntiles = min(pos_max, p.kv_len - 1) / PF_BN + 1;
  const bf16_t* kn = p.k_nope + (size_t)h * p.kv_pad * 128;
  const bf16_t* vt = p.v_t + (size_t)h * 128 * p.kv_pad;

  // next tile in registers while the current one is consumed from LDS
  PfTile nxt = pf_load_tile(p, kn, vt, 0, tid);
  for (int tile = 0; tile < ntiles; tile++) {
    __syncthreads();                                 // the previous tile's readers are done
    pf_store_tile(nxt, Ks, Vs, tid);
    __syncthreads();
    nxt = pf_load_tile(p, kn, vt, min(tile + 1, ntiles - 1), tid);   // (unconditional: the last iteration re-reads its tile)
    const int j0 = tile * PF_BN;
    // ---- S^T = K Q^T: st[u][kt][r] = S[query qi of tile u][key j0 + 16*kt + 4*g + r] -------------------------------------
    v4f st[NU][4];
#pragma unroll
    for (int u = 0; u < NU; u++)
#pragma unroll
      for (int kt = 0; kt < 4; kt++) st[u][kt] = v4f{0.f, 0.f, 0.f, 0.f};
    // (k-chunk outermost, round 6: 4 x NU independent accumulators between two MFMAs of one chain — with the key tile outermost
    // every MFMA waited for its predecessor's 8 passes with only the other query tile's MFMA in between; each accumulator still
    // sums its six chunks in the same order)
#pragma unroll
    for (int s = 0; s < 6; s++) {
#pragma unroll
      for (int kt = 0; kt < 4; kt++) {
        const v8bf a = as_v8bf(*reinterpret_cast<const uint4*>(Ks + (kt * 16 + qi) * PF_KROW + g * 8 + s * 32));
#pragma unroll
        for (int u = 0; u < NU; u++) st[u][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[u][s], st[u][kt], 0, 0, 0);
      }
    }
    // ---- online softmax, per query (lane & 15; the 4 key-chunk lanes g of a query agree after the two exchanges) ---------
    // Round 5: the causal mask only exists on a query block's last two key tiles — a tile every query of the wavefront sees whole
    // (wave-uniform test) skips the 32 compares + selects, and the 64 multiplies of `o *= alpha` are skipped whenever no lane's running
    // maximum moved (alpha == 1 exactly: the common case once the first tiles are past).  Same bits either way.
    const bool tile_full = !p.no_skip && j0 + PF_BN - 1 <= min(pos_off + min(q0, p.T - 1), p.kv_len - 1);
    // Round 6 (the kernel was VALU-issue-bound: 5.4 VALU per MFMA): the running maximum is kept on the RAW scores and the softmax
    // scale rides in the exponent's multiply — p = exp2(s * c - m * c), c = sm_scale * log2(e): one packed fma (two scores per
    // instruction) + v_exp instead of multiply, subtract, multiply, v_exp per score; the row sum is accumulated two scores per
    // packed add.  sm_scale > 0, so max(s * scale) = scale * max(s): the same softmax to within fp32 rounding of the exponent
    // (tests: rtol 2^-7 on the bf16 outputs against an fp32 softmax).
    const float cexp = p.sm_scale * 1.44269504088896340736f;
    uint4 pb[NU][2];
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int lim = min(pos_off + tq[u], p.kv_len - 1);       // last visible key of this lane's query
      float mx = -__builtin_inff();
      if (!tile_full) {
        // (the empty asm keeps this a wave-uniform BRANCH: speculated into selects, its 31 compares + selects ran on every tile)
#pragma unroll
        for (int kt = 0; kt < 4; kt++) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int key = j0 + kt * 16 + g * 4 + r;
            st[u][kt][r] = key <= lim ? st[u][kt][r] : -__builtin_inff();
          }
          asm volatile("" : "+v"(st[u][kt]));
        }
      }
#pragma unroll
      for (int kt = 0; kt < 4; kt++)
#pragma unroll
        for (int r = 0; r < 4; r++) mx = fmaxf(mx, st[u][kt][r]);
      mx = pf_max_rows(mx);                                     // over the query's four key-chunk lanes (qi, qi + 16, + 32, + 48)
      const float m_new = fmaxf(m_run[u], mx);                  // finite from tile 0 on: key 0 is visible to every query
      const float alpha = __builtin_amdgcn_exp2f((m_run[u] - m_new) * cexp);
      const pf_v2f mc = {-m_new * cexp, -m_new * cexp}, cc = {cexp, cexp};
      pf_v2f sum2 = {0.f, 0.f};
      float pv[4][4];
#pragma unroll
      for (int kt = 0; kt < 4; kt++)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const pf_v2f e = __builtin_elementwise_fma(pf_v2f{st[u][kt][r], st[u][kt][r + 1]}, cc, mc);
          pv[kt][r] = __builtin_amdgcn_exp2f(e[0]);
          pv[kt][r + 1] = __builtin_amdgcn_exp2f(e[1]);
          sum2 += pf_v2f{pv[kt][r], pv[kt][r + 1]};
        }
      const float sum = pf_sum_rows(sum2[0] + sum2[1]);
      l_run[u] = l_run[u] * alpha + sum;
      m_run[u] = m_new;
      if (p.no_skip || !__all(alpha == 1.0f)) {
        const pf_v2f a2 = {alpha, alpha};
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const pf_v2f t = pf_v2f{o[u][i][r], o[u][i][r + 1]} * a2;
            o[u][i][r] = t[0];
            o[u][i][r + 1] = t[1];
          }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
        pb[u][ks] = make_uint4(ktx_pk_bf16(pv[2 * ks][0], pv[2 * ks][1]), ktx_pk_bf16(pv[2 * ks][2], pv[2 * ks][3]),
                               ktx_pk_bf16(pv[2 * ks + 1][0], pv[2 * ks + 1][1]), ktx_pk_bf16(pv[2 * ks + 1][2], pv[2 * ks + 1][3]));
    }
    // ---- O^T += V^T P^T: o[u][i][r] = O[query qi][dim 16*i + 4*g + r] ----------------------------------------------------
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {                  // (key half outermost: 8 x NU independent accumulators, same order per accumulator)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const bf16_t* vb = Vs + (i * 16 + qi) * PF_VROW + g * 4 + ks * 32;
        const uint2 lo = *reinterpret_cast<const uint2*>(vb), hi = *reinterpret_cast<const uint2*>(vb + 16);
        const v8bf a = as_v8bf(make_uint4(lo.x, lo.y, hi.x, hi.y));
#pragma unroll
        for (int u = 0; u < NU; u++) o[u][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, as_v8bf(pb[u][ks]), o[u][i], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NU; u++) {
    const int t = q0 + u * 16 + qi;
    if (t < p.T) {
      const float inv = l_run[u] > 0.f ? 1.0f / l_run[u] : 0.f;
      bf16_t* op = p.out + ((size_t)t * p.H + h) * 128 + g * 4;
#pragma unroll
      for (int i = 0; i < 8; i++)
        *reinterpret_cast<uint2*>(op + i * 16) = make_uint2(ktx_pk_bf16(o[u][i][0] * inv, o[u][i][1] * inv),
                                                            ktx_pk_bf16(o[u][i][2] * inv, o[u][i][3] * inv));
    }
  }
}

extern "C" int ktx_mla_prefill(int T, int num_heads, int kv_len, int kv_pad, float sm_scale, const void* d_q_nope,
                               int64_t qn_token_stride, int64_t qn_head_stride, const void* d_q_pe, int64_t qp_token_stride,
                               int64_t qp_head_stride, const void* d_k_nope, const void* d_k_pe, int64_t kpe_token_stride,
                               const void* d_v_t, void* d_out, void* stream) {
  KTX_REQUIRE(d_q_nope && d_q_pe && d_k_nope && d_k_pe && d_v_t && d_out, "ktx_mla_prefill: null pointer");
  KTX_REQUIRE(T > 0 && num_heads > 0 && kv_len >= T, "ktx_mla_prefill: need 0 < T <= kv_len (the new tokens are the last T keys)");
  KTX_REQUIRE(kv_pad >= kv_len && kv_pad % PF_BN == 0, "ktx_mla_prefill: kv_pad must be a multiple of 64 covering kv_len");
  KTX_REQUIRE(qn_token_stride % 8 == 0 && qn_head_stride % 8 == 0 && qp_token_stride % 8 == 0 && qp_head_stride % 8 == 0 &&
              kpe_token_stride % 8 == 0, "ktx_mla_prefill: strides must keep 16-byte alignment");
  MlaPrefillParams p;
  p.q_nope = (const bf16_t*)d_q_nope; p.q_pe = (const bf16_t*)d_q_pe;
  p.qn_ts = qn_token_stride; p.qn_hs = qn_head_stride; p.qp_ts = qp_token_stride; p.qp_hs = qp_head_stride;
  p.k_nope = (const bf16_t*)d_k_nope; p.k_pe = (const bf16_t*)d_k_pe; p.kpe_ts = kpe_token_stride;
  p.v_t = (const bf16_t*)d_v_t; p.out = (bf16_t*)d_out;
  p.T = T; p.H = num_heads; p.kv_len = kv_len; p.kv_pad = kv_pad; p.sm_scale = sm_scale;
  p.no_skip = ktx_debug_get(23) == 1 ? 1 : 0;
  const size_t lds = (size_t)(PF_BN * PF_KROW + 128 * PF_VROW) * sizeof(bf16_t);
  hipStream_t st = (hipStream_t)stream;
  // per (query, key, head): 2*(192 + 128) flop over the causal half
  KTX_TIMED(st, 0.0, "mla_prefill_kernel T=%d Hq=%d kv=%d", T, num_heads, kv_len);
  const bool one_tile = ktx_debug_get(31) == 1;                      // dev knob 31 = 1: one query tile per wavefront, three wavefronts per SIMD
  const int nqb = one_tile ? (T + 63) / 64 : (T + 127) / 128;
  const bool xcd = num_heads % 8 == 0 && ktx_debug_get(24) != 1;     // dev knob 24 = 1: the 2-D grid of rounds 2-4 (A/B)
  p.nqb = xcd ? nqb : 0;
  const dim3 grid = xcd ? dim3((unsigned)(nqb * num_heads)) : dim3(nqb, num_heads);
  if (one_tile) hipLaunchKernelGGL((mla_prefill_kernel<3, 1>), grid, dim3(256), lds, st, p);
  else if (ktx_debug_get(22) == 1) hipLaunchKernelGGL((mla_prefill_kernel<1, 2>), grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL((mla_prefill_kernel<2, 2>), grid, dim3(256), lds, st, p);
  KTX_HIP(hipGetLastError());
  return 0;
}

extern "C" int ktx_mla_cache_append(const ktx_mla_config* cfg, void* d_kv_cache, int64_t token_stride,
                                    const void* d_ckv_new, const void* d_kpe_new, const int32_t* d_page_idx,
                                    const int32_t* d_page_offset, const int32_t* d_ntokens, int max_tokens, int num_pages,
                                    void* stream) {
  KTX_REQUIRE(cfg && d_kv_cache && d_ckv_new && d_kpe_new && d_page_idx && d_page_offset && max_tokens > 0,
              "ktx_mla_cache_append: bad argument");
  KTX_REQUIRE(token_stride >= MLA_DC + MLA_DR && token_stride % 8 == 0, "ktx_mla_cache_append: bad token stride");
  hipLaunchKernelGGL(mla_cache_append_kernel, dim3(max_tokens), dim3(128), 0, (hipStream_t)stream, (bf16_t*)d_kv_cache,
                     (long long)token_stride, cfg->page_size, (const bf16_t*)d_ckv_new, (const bf16_t*)d_kpe_new,
                     d_page_idx, d_page_offset, d_ntokens, max_tokens, num_pages);
  KTX_HIP(hipGetLastError());
  return 0;
}
