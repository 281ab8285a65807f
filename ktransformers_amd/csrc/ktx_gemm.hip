// ktx_gemm.hip — the library's own prompt-sized BF16 "NT" GEMM for gfx950 (C ABI: include/ktx_gemm.h).
//
//   Y[b][m][n] = round( sum_k A[b][m][k] * B[b][n][k] (+ bias[n]) )       bf16 operands, fp32 MFMA accumulation
//
// Users: the prompt path of the W4 linears (weights expanded once per call to bf16((q-8)*s), Marlin's own multiplicand —
// kt-kernel/cuda/gptq_marlin/gptq_marlin.cu:412), the kv_b_proj expansion of the non-absorbed MLA prompt kernel
// (archive/ktransformers/operators/attention.py:77-194) and the router logits of large batches (three exact bf16 planes of
// the fp32 gate weight, modeling_deepseek_v3.py:434-437).  Rounds 1-2 sent all three to a vendor GEMM.
//
// Shape of the kernel (MI355X_MICROARCH / cdna_hip_programming section 5, the "128 x 128 tile + LDS-DMA" structure):
//   * workgroup = 256 threads = 4 wavefronts as 2 (M) x 2 (N); tile 128 x 128 x 64; a wavefront owns 64 x 64 outputs =
//     4 x 4 fragments of mfma_f32_16x16x32_bf16 (64 accumulator registers), 32 MFMAs per k-step against 16 ds_read_b128;
//   * both operand tiles travel HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 8 rows of 128 B per
//     wave-instruction, no staging registers, no ds_write pass).  The DMA destination is lane-linear, so the bank-conflict
//     swizzle is applied on the SOURCE side: LDS slot j of tile row R holds the row's 16-byte piece j ^ g(R),
//     g(R) = (R & 7) ^ ((R >> 3) & 1) — a lane group (16 rows x one k-piece) of a fragment read then touches 16 different
//     16-byte slots of the 256-byte bank row, and every global row is still read as one whole 128-byte line;
//   * NBUF = 1: one 32 KiB stage, two barriers per k-step, 3-4 workgroups per CU hide each other's DMA wait (the guide's
//     874 TFLOP/s structure); NBUF = 2: two stages (64 KiB, 2 workgroups per CU), the next k-step's DMA is issued before
//     the MFMAs of the current one and waited for behind them.  The DMA is issued through inline asm so the compiler does
//     not drain it in front of the current stage's fragment reads (see ktx_mla.hip);
//   * blockIdx -> tile: XCD-aware when there are >= 8 column tiles — workgroup ids are dealt round-robin to the 8 XCDs, so
//     id % 8 selects the XCD and the ids of one XCD walk the M tiles of ONE weight column tile: its 16 KiB k-slabs are
//     fetched from HBM once and hit that XCD's L2 for the other M tiles;
//   * epilogue: fragments leave through LDS (16 rows x 64 columns per pass and wavefront, padded rows) so the global
//     stores are whole 16-byte pieces of contiguous rows, not the fragment's 2-byte column scatter.
#include <cstdlib>

#include "ktx_common.h"

#include "../../include/ktx_gemm.h"

namespace {

typedef __bf16 gv8bf __attribute__((ext_vector_type(8)));

__device__ __forceinline__ gv8bf g_as_v8bf(const uint4& u) {
  union { uint4 u; gv8bf v; } c;
  c.u = u;
  return c.v;
}

struct GemmParams {
  const bf16_t* A; long lda, a_bs;
  const bf16_t* B; long ldb, b_bs;
  void* Y; long ldy, y_bs;
  const bf16_t* bias;
  int M, N, K, tm, tn, xcd_map;
};

// one LDS-DMA wave-instruction: lane l's 16 bytes at gsrc_lane land at LDS byte address lds_addr + 16 l (M0 = base)
__device__ __forceinline__ void gemm_dma(const bf16_t* gsrc_lane, uint32_t lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc_lane), "s"(lds_addr)
               : "memory");
}

// Tile configuration.  BK = 64: rows of 128 B, a DMA wave-instruction covers 8 rows, swizzle g(R) = (R & 7) ^ ((R >> 3) & 1).
// BK = 32: rows of 64 B, 16 rows per wave-instruction, g(R) = (R >> 2) & 3 (four rows share a 256-byte bank row).  In both a
// 16-lane group of a fragment read (16 consecutive rows, one k-piece) touches 16 different 16-byte slots of the bank row.
template <int BM_, int BN_, int BK_, int NBUF_>
struct GemmCfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, NBUF = NBUF_;
  static constexpr int ROWB = BK * 2;              // bytes per LDS row
  static constexpr int PPR = ROWB / 16;            // 16-byte pieces per row
  static constexpr int RPC = 64 / PPR;             // rows per DMA chunk (one wave-instruction = 1 KiB)
  static constexpr int NCH = (BM + BN) / RPC;      // chunks per stage: A rows first, then B rows
  static constexpr int CPW = NCH / 4;              // chunks per wavefront
  static constexpr int STAGE = (BM + BN) * ROWB;   // bytes per stage
  static constexpr int WM = BM / 2, WN = BN / 2;   // wavefront tile
  static constexpr int MI = WM / 16, NI = WN / 16; // fragments per wavefront
  static constexpr int KS = BK / 32;               // MFMA k-steps per stage
  static_assert(NCH % 4 == 0 && BM % RPC == 0 && NBUF * STAGE >= 4 * 8192 && NI == 4, "tile configuration");
  __device__ static __forceinline__ int swz(int R) { return BK == 64 ? ((R & 7) ^ ((R >> 3) & 1)) : ((R >> 2) & 3); }
};

template <class C, bool F32OUT>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel(GemmParams p) {   // >= 2 wavefronts per SIMD: <= 256 registers
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // [NBUF][A rows | B rows]
  int mt, nt;
  {
    const int bid = blockIdx.x;
    if (p.xcd_map) {
      const int xcd = bid & 7, idx = bid >> 3;
      nt = (idx / p.tm) * 8 + xcd;
      mt = idx % p.tm;
      if (nt >= p.tn) return;
    } else {
      nt = bid % p.tn;
      mt = bid / p.tn;
    }
  }
  const int bz = blockIdx.y;
  const bf16_t* A = p.A + (size_t)bz * p.a_bs;
  const bf16_t* B = p.B + (size_t)bz * p.b_bs;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int row0 = mt * C::BM, col0 = nt * C::BN;

  // ---- DMA sources: chunk c = wave * CPW + i covers stage rows c * RPC .. (A rows 0 .. BM-1, then B rows); lane = (row in
  // chunk) * PPR + LDS slot; the slot holds the row's piece slot ^ swz(row)
  const bf16_t* gs[C::CPW];
#pragma unroll
  for (int i = 0; i < C::CPW; i++) {
    const int R = (wave * C::CPW + i) * C::RPC + lane / C::PPR;
    const int piece = (lane % C::PPR) ^ C::swz(R);
    gs[i] = R < C::BM ? A + (size_t)min(row0 + R, p.M - 1) * p.lda + piece * 8
                      : B + (size_t)min(col0 + R - C::BM, p.N - 1) * p.ldb + piece * 8;
  }
  const uint32_t lds0 =
      __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(const __attribute__((address_space(3))) void*)smem) + wave * (C::CPW * 1024);
  auto issue = [&](int kt, int buf) {
    const uint32_t base = lds0 + buf * C::STAGE;
    const int ko = kt * C::BK;
#pragma unroll
    for (int i = 0; i < C::CPW; i++) gemm_dma(gs[i] + ko, base + i * 1024);
  };

  v4f acc[C::MI][C::NI];
#pragma unroll
  for (int i = 0; i < C::MI; i++)
#pragma unroll
    for (int j = 0; j < C::NI; j++) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

  const int r = lane & 15, kq = lane >> 4;
  const int g = C::swz(r);                          // fragment rows start at multiples of 16: swz depends on r only
  const int a_off = (wm * C::WM + r) * C::ROWB, b_off = (C::BM + wn * C::WN + r) * C::ROWB;
  auto compute = [&](int buf) {
    const uint8_t* st = smem + buf * C::STAGE;
#pragma unroll
    for (int s = 0; s < C::KS; s++) {
      const int slot = ((s * 4 + kq) ^ g) * 16;
      uint4 af[C::MI], bf[C::NI];
#pragma unroll
      for (int j = 0; j < C::NI; j++) bf[j] = *reinterpret_cast<const uint4*>(st + b_off + j * 16 * C::ROWB + slot);
#pragma unroll
      for (int i = 0; i < C::MI; i++) af[i] = *reinterpret_cast<const uint4*>(st + a_off + i * 16 * C::ROWB + slot);
#pragma unroll
      for (int i = 0; i < C::MI; i++)
#pragma unroll
        for (int j = 0; j < C::NI; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g_as_v8bf(af[i]), g_as_v8bf(bf[j]), acc[i][j], 0, 0, 0);
    }
  };

  const int nk = p.K / C::BK;
  if constexpr (C::NBUF == 1) {
    for (int kt = 0; kt < nk; kt++) {
      issue(kt, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(0);
      __syncthreads();
    }
  } else {
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
      if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);   // that stage was last read in step kt - 1; every wave is past its barrier
      compute(kt & 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- epilogue: every wave is past the last barrier, the stages are free.  16 rows x 64 columns per pass through this
  // wavefront's 8 KiB of LDS, rows padded by 16 B (the four row groups of a fragment land in different banks).
  constexpr int ES = F32OUT ? 4 : 2;
  constexpr int RS = 64 * ES + 16;
  constexpr int PPR = 64 * ES / 16;        // 16-byte pieces per row
  constexpr int PPL = 16 * PPR / 64;       // pieces per lane and pass
  constexpr int EPP = 16 / ES;             // elements per piece
  uint8_t* ep = smem + wave * 8192;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < 4; j++) bv[j] = bf16_to_f32(p.bias[min(col0 + wn * C::WN + j * 16 + r, p.N - 1)]);
  }
  uint8_t* Yb = reinterpret_cast<uint8_t*>(p.Y) + (size_t)bz * p.y_bs * ES;
#pragma unroll
  for (int i = 0; i < C::MI; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float v = acc[i][j][q] + bv[j];
        uint8_t* dst = ep + (kq * 4 + q) * RS + (j * 16 + r) * ES;
        if constexpr (F32OUT) *reinterpret_cast<float*>(dst) = v;
        else *reinterpret_cast<bf16_t*>(dst) = (bf16_t)(ktx_pk_bf16(v, 0.f) & 0xffffu);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < PPL; u++) {
      const int idx = u * 64 + lane;
      const int row = idx / PPR, pc = idx % PPR;
      const uint4 v = *reinterpret_cast<const uint4*>(ep + row * RS + pc * 16);
      const int grow = row0 + wm * C::WM + i * 16 + row, gcol = col0 + wn * C::WN + pc * EPP;
      if (grow < p.M && gcol < p.N) *reinterpret_cast<uint4*>(Yb + ((size_t)grow * p.ldy + gcol) * ES) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// =====================================================================================================================
// Round 6: the 256 x 256 x 64 tile, 8 wavefronts (2 M x 4 N), two-stage LDS ring of half-tiles with COUNTED waits, and the two
// wavefronts of every SIMD running half a phase apart (cdna_hip_programming section 5, "the 256^2 8-phase template": the schedule
// below is this file's own, built to that description).
//   * a wavefront owns 128 x 64 outputs = 8 x 4 fragments (128 accumulator registers); a k-tile (K = 64) is four PHASES of 16
//     MFMAs, one output quadrant (64 x 32) each, in the order (a0,b0) (a0,b1) (a1,b1) (a1,b0) so that consecutive phases share
//     one operand: fragment reads per phase 12 / 4 / 8 / 0 ds_read_b128;
//   * a stage (64 KiB) = four half-tiles of 128 rows x 128 B: A_a0 / A_a1 = the tile rows the wavefronts read in phase 0 / 2
//     (rows wm*128 + [0,64) / + [64,128)), B_b0 / B_b1 = the columns read in phase 0 / 1 (wn*64 + [0,32) / + [32,64)).  One
//     half-tile (2 x global_load_lds_dwordx4 per thread) is issued per phase, for the NEXT k-tile, in the order of its deadlines:
//     A_a0, B_b0, B_b1, A_a1 — each is issued >= 3 phases before its first read and >= 5 phases after the last read of the slot
//     it overwrites, so one `s_waitcnt vmcnt(4)` per phase (two half-tiles stay in flight, never 0 in the steady state) followed
//     by the phase's barrier orders every read behind its data (the reader is always one barrier later than the wait);
//   * each phase = [load segment: fragment reads + DMA issue + counted wait] barrier [16 MFMAs at priority 1] barrier.  Wavefronts
//     4-7 (wm = 1, the second wavefront of every SIMD) run one barrier behind wavefronts 0-3: while one half of the workgroup
//     multiplies, the other reads and issues — the matrix pipe of a SIMD always has exactly one owner;
//   * LDS rows keep the 128-B row swizzle of the kernels above (applied on the DMA source side); epilogue through LDS as above.
struct Gemm256 {
  static constexpr int BM = 256, BN = 256, BK = 64, STAGE = 65536, HALF = 16384;
  __device__ static __forceinline__ int swz(int R) { return (R & 7) ^ ((R >> 3) & 1); }
};

// lane's 16 bytes at base + off land at LDS byte address lds_addr + 16 * lane
__device__ __forceinline__ void gemm_dma_off(const void* base, uint32_t off_lane, uint32_t lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(off_lane), "s"(base), "s"(lds_addr)
               : "memory");
}

template <bool F32OUT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm256_bf16_nt_kernel(GemmParams p) {
  using C = Gemm256;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // [2 stages][A_a0 | A_a1 | B_b0 | B_b1]
  int mt, nt;
  {
    const int bid = blockIdx.x;
    if (p.xcd_map == 2) {
      // supertiles: the 32 workgroups an XCD runs at once (one per CU) form a 4 (M) x 8 (N) block of tiles — an A k-slab is
      // fetched into that XCD's L2 once for 8 column tiles and a B k-slab once for 4 row tiles, where "all M tiles of one column
      // tile" re-read the whole of A from HBM for every column tile (A no longer fits the 256 MiB last-level cache at 8192 rows)
      const int xcd = bid & 7, idx = bid >> 3;
      const int S = (idx >> 5) * 8 + xcd, w = idx & 31;
      const int ns_m = (p.tm + 3) / 4;
      mt = (S % ns_m) * 4 + (w & 3);
      nt = (S / ns_m) * 8 + (w >> 2);
      if (mt >= p.tm || nt >= p.tn) return;
    } else if (p.xcd_map) {
      const int xcd = bid & 7, idx = bid >> 3;
      nt = (idx / p.tm) * 8 + xcd;
      mt = idx % p.tm;
      if (nt >= p.tn) return;
    } else {
      nt = bid % p.tn;
      mt = bid / p.tn;
    }
  }
  const int bz = blockIdx.y;
  const bf16_t* A = p.A + (size_t)bz * p.a_bs;
  const bf16_t* B = p.B + (size_t)bz * p.b_bs;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int row0 = mt * C::BM, col0 = nt * C::BN;

  // ---- DMA sources (byte offsets from A / B; the launcher checks they fit 32 bits): half-tile h, instruction i covers the 8 local
  // rows (wave * 2 + i) * 8 ..; lane = (row in the 8) * 8 + LDS slot; the slot holds the row's 16-byte piece slot ^ swz(row)
  uint32_t offA[2][2], offB[2][2];
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int L = (wave * 2 + i) * 8 + (lane >> 3);
      const int piece = (lane & 7) ^ C::swz(L);
      const int grow = min(row0 + (L >> 6) * 128 + h * 64 + (L & 63), p.M - 1);
      const int gcol = min(col0 + (L >> 5) * 64 + h * 32 + (L & 31), p.N - 1);
      offA[h][i] = (uint32_t)(((size_t)grow * p.lda) * 2 + piece * 16);
      offB[h][i] = (uint32_t)(((size_t)gcol * p.ldb) * 2 + piece * 16);
    }
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(const __attribute__((address_space(3))) void*)smem) + wave * 2048;
  // half-tile slots of a stage: 0 = A_a0, 1 = A_a1, 2 = B_b0, 3 = B_b1
  auto issue = [&](int kt, int slot) {
    const uint32_t dst = lds0 + (kt & 1) * C::STAGE + slot * C::HALF;
    if (slot < 2) {
      gemm_dma_off(A + (size_t)kt * C::BK, offA[slot][0], dst);
      gemm_dma_off(A + (size_t)kt * C::BK, offA[slot][1], dst + 1024);
    } else {
      gemm_dma_off(B + (size_t)kt * C::BK, offB[slot - 2][0], dst);
      gemm_dma_off(B + (size_t)kt * C::BK, offB[slot - 2][1], dst + 1024);
    }
  };

  v4f acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

  const int r = lane & 15, kq = lane >> 4;
  const int g = C::swz(r);
  // fragment read offsets inside a stage: k-step s = pieces s*4 + kq
  const uint32_t so[2] = {(uint32_t)(((0 + kq) ^ g) * 16), (uint32_t)(((4 + kq) ^ g) * 16)};
  const uint32_t a_off = (wm * 64 + r) * 128, b_off = 2 * C::HALF + (wn * 32 + r) * 128;
  uint4 af[4][2], bf0[2][2], bf1[2][2];
  auto read_a = [&](const uint8_t* st, int part) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int s = 0; s < 2; s++) af[i][s] = *reinterpret_cast<const uint4*>(st + part * C::HALF + a_off + i * 2048 + so[s]);
  };
  auto read_b = [&](const uint8_t* st, int part, uint4 (&bf)[2][2]) {
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int s = 0; s < 2; s++) bf[j][s] = *reinterpret_cast<const uint4*>(st + part * C::HALF + b_off + j * 2048 + so[s]);
  };
  auto mma = [&](int ah, int bh, const uint4 (&bf)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          acc[ah * 4 + i][bh * 2 + j] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(g_as_v8bf(af[i][s]), g_as_v8bf(bf[j][s]), acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  // segment boundary: nothing moves across (fragment reads stay behind the barrier that publishes their data, MFMAs behind the
  // reads' segment), then the workgroup barrier
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  const int nk = p.K / C::BK;
  // prologue: k-tile 0 in deadline order, then the first counted wait
  issue(0, 0); issue(0, 2); issue(0, 3); issue(0, 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  bar();
  if (wm == 1) bar();   // the second wavefront of every SIMD runs one segment behind the first
  for (int kt = 0; kt < nk; kt++) {
    const uint8_t* st = smem + (kt & 1) * C::STAGE;
    const bool more = kt + 1 < nk;
    // phase 0: (a0, b0)
    read_a(st, 0);
    read_b(st, 0, bf0);
    if (more) { issue(kt + 1, 0); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
    mma(0, 0, bf0);
    bar();
    // phase 1: (a0, b1)
    read_b(st, 1, bf1);
    if (more) { issue(kt + 1, 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    bar();
    mma(0, 1, bf1);
    bar();
    // phase 2: (a1, b1)
    read_a(st, 1);
    if (more) issue(kt + 1, 3);
    bar();
    mma(1, 1, bf1);
    bar();
    // phase 3: (a1, b0) — operands already in registers
    if (more) { issue(kt + 1, 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    bar();
    mma(1, 0, bf0);
    bar();
  }
  if (wm == 0) bar();   // pair the extra barrier of the lagging half

  // ---- epilogue: 16 rows x 64 columns per pass through this wavefront's 8 KiB of LDS (every wavefront is past the last barrier)
  constexpr int ES = F32OUT ? 4 : 2;
  constexpr int RS = 64 * ES + 16;
  constexpr int PPR = 64 * ES / 16;
  constexpr int PPL = 16 * PPR / 64;
  constexpr int EPP = 16 / ES;
  uint8_t* ep = smem + wave * 8192;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < 4; j++) bv[j] = bf16_to_f32(p.bias[min(col0 + wn * 64 + j * 16 + r, p.N - 1)]);
  }
  uint8_t* Yb = reinterpret_cast<uint8_t*>(p.Y) + (size_t)bz * p.y_bs * ES;
#pragma unroll
  for (int i = 0; i < 8; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float v = acc[i][j][q] + bv[j];
        uint8_t* dst = ep + (kq * 4 + q) * RS + (j * 16 + r) * ES;
        if constexpr (F32OUT) *reinterpret_cast<float*>(dst) = v;
        else *reinterpret_cast<bf16_t*>(dst) = (bf16_t)(ktx_pk_bf16(v, 0.f) & 0xffffu);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < PPL; u++) {
      const int idx = u * 64 + lane;
      const int row = idx / PPR, pc = idx % PPR;
      const uint4 v = *reinterpret_cast<const uint4*>(ep + row * RS + pc * 16);
      const int grow = row0 + wm * 128 + i * 16 + row, gcol = col0 + wn * 64 + pc * EPP;
      if (grow < p.M && gcol < p.N) *reinterpret_cast<uint4*>(Yb + ((size_t)grow * p.ldy + gcol) * ES) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <bool F32OUT>
int launch_gemm256(GemmParams p, int batch, hipStream_t st) {
  auto kern = gemm256_bf16_nt_kernel<F32OUT>;
  constexpr int lds = 2 * Gemm256::STAGE;
  p.tm = (p.M + 255) / 256;
  p.tn = (p.N + 255) / 256;
  p.xcd_map = (p.tm >= 12 && p.tn >= 8) ? 2 : p.tn >= 8 ? 1 : 0;
  long nblk = p.xcd_map ? (long)p.tm * ((p.tn + 7) / 8 * 8) : (long)p.tm * p.tn;
  if (p.xcd_map == 2) {
    const long ns = (long)((p.tm + 3) / 4) * ((p.tn + 7) / 8);   // 4 x 8 supertiles, dealt to the XCDs round-robin
    nblk = (ns + 7) / 8 * 8 * 32;
  }
  KTX_REQUIRE(nblk < (1L << 31), "ktx_gemm_bf16_nt: too many tiles");
  KTX_HIP(ktx_set_max_lds(reinterpret_cast<const void*>(kern), lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)batch), dim3(512), lds, st, p);
  KTX_HIP(hipGetLastError());
  return 0;
}

// fp32 -> three bf16 planes by truncation: w == hi + mid + lo exactly (finite w; 24 mantissa bits = 8 + 8 + 8)
__global__ __launch_bounds__(256) void split_f32_bf16x3_kernel(const float* __restrict__ w, long n, bf16_t* __restrict__ planes) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = w[i];
  const uint32_t hb = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(hb);                 // exact
  const uint32_t mb = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(mb);                // exact, at most 8 significant bits
  planes[i] = (bf16_t)(hb >> 16);
  planes[n + i] = (bf16_t)(mb >> 16);
  planes[2 * n + i] = (bf16_t)(__float_as_uint(r2) >> 16);
}

template <class C, bool F32OUT>
int launch_gemm(GemmParams p, int batch, hipStream_t st) {
  auto kern = gemm_bf16_nt_kernel<C, F32OUT>;
  constexpr int lds = C::NBUF * C::STAGE;
  p.tm = (p.M + C::BM - 1) / C::BM;
  p.tn = (p.N + C::BN - 1) / C::BN;
  p.xcd_map = p.tn >= 8 ? 1 : 0;
  const long nblk = p.xcd_map ? (long)p.tm * ((p.tn + 7) / 8 * 8) : (long)p.tm * p.tn;
  KTX_REQUIRE(nblk < (1L << 31), "ktx_gemm_bf16_nt: too many tiles");
  const dim3 grid((unsigned)nblk, (unsigned)batch);
  if (lds > 48 * 1024) KTX_HIP(ktx_set_max_lds(reinterpret_cast<const void*>(kern), lds));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  KTX_HIP(hipGetLastError());
  return 0;
}

}  // namespace

extern "C" int ktx_gemm_bf16_nt(const ktx_gemm_args* a, ktx_stream_t stream) {
  KTX_REQUIRE(a != nullptr, "ktx_gemm_bf16_nt: null args");
  KTX_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0 && a->batch > 0, "ktx_gemm_bf16_nt: M, N, K, batch must be positive");
  KTX_REQUIRE(a->K % 64 == 0, "ktx_gemm_bf16_nt: K must be a multiple of 64");
  KTX_REQUIRE(a->lda % 8 == 0 && a->ldb % 8 == 0 && a->lda >= a->K && a->ldb >= a->K,
              "ktx_gemm_bf16_nt: lda / ldb must be multiples of 8 elements and >= K");
  const int ealign = a->out_f32 ? 4 : 8;
  KTX_REQUIRE(a->N % ealign == 0 && a->ldy % ealign == 0 && a->ldy >= a->N,
              "ktx_gemm_bf16_nt: N and ldy must be multiples of 16 bytes of output elements, ldy >= N");
  KTX_REQUIRE(a->a_bs % 8 == 0 && a->b_bs % 8 == 0 && a->y_bs % ealign == 0, "ktx_gemm_bf16_nt: batch strides must keep 16-byte alignment");
  KTX_REQUIRE(a->A && a->B && a->Y && ((uintptr_t)a->A % 16 == 0) && ((uintptr_t)a->B % 16 == 0) && ((uintptr_t)a->Y % 16 == 0),
              "ktx_gemm_bf16_nt: A, B, Y must be non-null and 16-byte aligned");
  KTX_REQUIRE(a->variant >= 0 && a->variant <= 5, "ktx_gemm_bf16_nt: variant must be 0 .. 5");
  KTX_REQUIRE(a->batch <= 65535, "ktx_gemm_bf16_nt: batch must be <= 65535");
  GemmParams p;
  p.A = (const bf16_t*)a->A; p.lda = a->lda; p.a_bs = a->a_bs;
  p.B = (const bf16_t*)a->B; p.ldb = a->ldb; p.b_bs = a->b_bs;
  p.Y = a->Y; p.ldy = a->ldy; p.y_bs = a->y_bs;
  p.bias = (const bf16_t*)a->bias;
  p.M = a->M; p.N = a->N; p.K = a->K;
  hipStream_t st = (hipStream_t)stream;
  // auto (measured on the DeepSeek-V3 chunk shapes, scripts/gemm_bench.py / profiles/r03_gemm_bench.txt): the 256 x 128 tile
  // (25 % fewer LDS bytes per MFMA) wins once its grid fills two workgroups per CU; thinner grids keep the 128 x 128 tile with
  // two DMA stages
  int variant = a->variant;
  if (!variant) {
    const long t256 = (long)((a->M + 255) / 256) * ((a->N + 127) / 128) * a->batch;
    variant = (a->M >= 256 && t256 >= 400) ? 4 : 2;
    // round 6: the 256 x 256 ping-pong tile (one workgroup per CU) once its grid covers most of the chip and K is deep enough to
    // amortise its 128-register epilogue
    // (and its 256 x 256 tiles cover the output with little waste: the kv_b expansions of the prompt attention, 128 columns or 128
    // rows per head, measured 0.6x on it)
    const long tm5 = (a->M + 255) / 256, tn5 = (a->N + 255) / 256, t5 = tm5 * tn5 * a->batch;
    const double fill5 = (double)a->M * a->N / ((double)tm5 * tn5 * 65536.0);
    if (t5 >= 192 && fill5 >= 0.85 && a->K >= 1024 && (size_t)a->M * a->lda * 2 < ((size_t)1 << 32) &&
        (size_t)a->N * a->ldb * 2 < ((size_t)1 << 32) && !getenv("KTX_GEMM_NO_PINGPONG"))
      variant = 5;
  }
  const double na = a->a_bs ? a->batch : 1, nb = a->b_bs ? a->batch : 1;   // a shared operand is counted once
  KTX_TIMED(st, ((double)a->M * a->K * na + (double)a->N * a->K * nb) * 2 + (double)a->M * a->N * a->batch * (a->out_f32 ? 4 : 2),
            "gemm_bf16_nt_kernel<%d,%s> %dx%dx%d b%d", variant, a->out_f32 ? "f32" : "bf16", a->M, a->N, a->K, a->batch);
  const bool f = a->out_f32 != 0;
  if (variant == 5) {
    KTX_REQUIRE((size_t)a->M * a->lda * 2 < ((size_t)1 << 32) && (size_t)a->N * a->ldb * 2 < ((size_t)1 << 32),
                "ktx_gemm_bf16_nt: variant 5 addresses an operand of one batch entry with 32-bit byte offsets (< 4 GiB)");
    return f ? launch_gemm256<true>(p, a->batch, st) : launch_gemm256<false>(p, a->batch, st);
  }
  switch (variant) {
    case 1: return f ? launch_gemm<GemmCfg<128, 128, 64, 1>, true>(p, a->batch, st) : launch_gemm<GemmCfg<128, 128, 64, 1>, false>(p, a->batch, st);
    case 2: return f ? launch_gemm<GemmCfg<128, 128, 64, 2>, true>(p, a->batch, st) : launch_gemm<GemmCfg<128, 128, 64, 2>, false>(p, a->batch, st);
    case 3: return f ? launch_gemm<GemmCfg<256, 128, 32, 2>, true>(p, a->batch, st) : launch_gemm<GemmCfg<256, 128, 32, 2>, false>(p, a->batch, st);
    default: return f ? launch_gemm<GemmCfg<256, 128, 64, 1>, true>(p, a->batch, st) : launch_gemm<GemmCfg<256, 128, 64, 1>, false>(p, a->batch, st);
  }
}

extern "C" int ktx_split_f32_bf16x3(const float* w, int64_t n, void* planes, ktx_stream_t stream) {
  KTX_REQUIRE(w && planes && n > 0, "ktx_split_f32_bf16x3: null pointer or empty tensor");
  hipLaunchKernelGGL(split_f32_bf16x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (long)n,
                     (bf16_t*)planes);
  KTX_HIP(hipGetLastError());
  return 0;
}
