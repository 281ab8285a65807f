// TEST INFRASTRUCTURE ONLY — never linked into, imported by, or called from the product path.
//
// Thin extern "C" driver around the reference's OWN, UNMODIFIED CPU MoE kernels, compiled from the
// sources where they lie under /root/reference (see oracle/Makefile; output goes to oracle/_ref/).
// It exists to (1) pin the plain-C restatement in oracle/ktx_oracle.c bit-for-bit, (2) generate the
// golden vectors under tests/golden/, and (3) serve as the `cpu_baseline` ("kind": "reference") leg
// of bench.py on the GPU box's host cores.
//
// Reference entry points driven here:
//   TP_MOE<AMX_MOE_TP<amx::GemmKernel224Int4>>            kt-kernel/operators/amx/moe.hpp:426-518
//   TP_MOE<AMX_MOE_TP<amx::GemmKernel224Int8>>            (same file, int8 instantiation)
//   TP_MOE<AMX_K2_MOE_TP<amx::GemmKernel224Int4SmallKGroup>>  kt-kernel/operators/amx/k2-moe.hpp
//   TP_MOE<AMX_FP8_MOE_TP<amx::GemmKernel224FP8>>         kt-kernel/operators/amx/fp8-moe.hpp
//   TP_MOE<AMX_BF16_MOE_TP<amx::GemmKernel224BF16>>       kt-kernel/operators/amx/bf16-moe.hpp
//   TP_MOE<AMX_FP8_PERCHANNEL_MOE_TP<amx::GemmKernel224FP8PerChannel>>  kt-kernel/operators/amx/fp8-perchannel-moe.hpp
//   forward() protocol                                    kt-kernel/operators/moe-tp.hpp:201-246
//   WorkerPool(WorkerPoolConfig)                          kt-kernel/cpu_backend/worker_pool.h:132-168
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>

#include "operators/amx/bf16-moe.hpp"
#include "operators/amx/fp8-moe.hpp"
#include "operators/amx/fp8-perchannel-moe.hpp"
#include "operators/amx/k2-moe.hpp"
#include "operators/amx/moe.hpp"

namespace {

thread_local std::string g_err;

enum Kind { KIND_INT4 = 0, KIND_INT8 = 1, KIND_K2 = 2, KIND_FP8 = 3, KIND_BF16 = 4, KIND_FP8PC = 5 };

using MoeInt4 = TP_MOE<AMX_MOE_TP<amx::GemmKernel224Int4>>;
using MoeInt8 = TP_MOE<AMX_MOE_TP<amx::GemmKernel224Int8>>;
using MoeK2 = TP_MOE<AMX_K2_MOE_TP<amx::GemmKernel224Int4SmallKGroup>>;
using MoeFP8 = TP_MOE<AMX_FP8_MOE_TP<amx::GemmKernel224FP8>>;
using MoeBF16 = TP_MOE<AMX_BF16_MOE_TP<amx::GemmKernel224BF16>>;
using MoeFP8PC = TP_MOE<AMX_FP8_PERCHANNEL_MOE_TP<amx::GemmKernel224FP8PerChannel>>;

struct Handle {
  int kind;
  GeneralMOEConfig cfg;
  std::unique_ptr<MoeInt4> i4;
  std::unique_ptr<MoeInt8> i8;
  std::unique_ptr<MoeK2> k2;
  std::unique_ptr<MoeFP8> f8;
  std::unique_ptr<MoeBF16> bf;
  std::unique_ptr<MoeFP8PC> pc;
};

template <class F>
int guarded(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  } catch (...) {
    g_err = "unknown C++ exception";
    return -1;
  }
}

}  // namespace

extern "C" {

const char* ktref_last_error() { return g_err.c_str(); }

// One sub-pool (no NUMA tensor-parallel split => no fp32 partial merge) unless
// subpools > 1, in which case I is split across sub-pools exactly like the reference's NUMA-TP.
void* ktref_pool_create(int subpools, int threads_per_subpool) {
  WorkerPool* pool = nullptr;
  int rc = guarded([&] {
    WorkerPoolConfig pc;
    pc.subpool_count = subpools;
    for (int i = 0; i < subpools; i++) {
      pc.subpool_numa_map.push_back(i);
      pc.subpool_thread_count.push_back(threads_per_subpool);
    }
    pool = new WorkerPool(pc);
  });
  return rc ? nullptr : pool;
}

void ktref_pool_destroy(void* pool) { delete (WorkerPool*)pool; }

void* ktref_moe_create(int kind, int expert_num, int k, int hidden, int inter, int max_len, int group_size,
                       void* pool) {
  Handle* h = nullptr;
  int rc = guarded([&] {
    h = new Handle();
    h->kind = kind;
    GeneralMOEConfig c(expert_num, k, hidden, inter);
    c.max_len = max_len;
    c.layer_idx = 0;
    c.pool = (WorkerPool*)pool;
    c.quant_config.group_size = group_size;
    c.quant_config.zero_point = false;
    c.quant_config.bits = (kind == KIND_K2) ? 4 : 8;
    c.quant_config.per_channel = (kind == KIND_FP8PC);
    h->cfg = c;
  });
  if (rc) {
    delete h;
    return nullptr;
  }
  return h;
}

// gate/up: [E, I, H], down: [E, H, I]; bf16 for INT4/INT8/BF16 (online quantisation by the reference's own
// BufferB::from_mat), packed bytes (+ scales) for K2 / FP8.  Pointers are borrowed only for the call.
int ktref_moe_load(void* hv, const void* gate, const void* up, const void* down, const void* gate_scale,
                   const void* up_scale, const void* down_scale) {
  Handle* h = (Handle*)hv;
  return guarded([&] {
    h->cfg.gate_proj = (void*)gate;
    h->cfg.up_proj = (void*)up;
    h->cfg.down_proj = (void*)down;
    h->cfg.gate_scale = (void*)gate_scale;
    h->cfg.up_scale = (void*)up_scale;
    h->cfg.down_scale = (void*)down_scale;
    switch (h->kind) {
      case KIND_INT4:
        h->i4 = std::make_unique<MoeInt4>(h->cfg);
        h->i4->load_weights();
        break;
      case KIND_INT8:
        h->i8 = std::make_unique<MoeInt8>(h->cfg);
        h->i8->load_weights();
        break;
      case KIND_K2:
        h->k2 = std::make_unique<MoeK2>(h->cfg);
        h->k2->load_weights();
        break;
      case KIND_FP8:
        h->f8 = std::make_unique<MoeFP8>(h->cfg);
        h->f8->load_weights();
        break;
      case KIND_BF16:
        h->bf = std::make_unique<MoeBF16>(h->cfg);
        h->bf->load_weights();
        break;
      case KIND_FP8PC:
        h->pc = std::make_unique<MoeFP8PC>(h->cfg);
        h->pc->load_weights();
        break;
      default:
        throw std::runtime_error("bad kind");
    }
  });
}

// input/output: bf16 [qlen, H]; expert_ids int64 [qlen, k]; weights fp32 [qlen, k].
int ktref_moe_forward(void* hv, int qlen, int k, const int64_t* expert_ids, const float* weights, const void* input,
                      void* output, int incremental) {
  Handle* h = (Handle*)hv;
  return guarded([&] {
    MoE_Interface* m = nullptr;
    switch (h->kind) {
      case KIND_INT4: m = h->i4.get(); break;
      case KIND_INT8: m = h->i8.get(); break;
      case KIND_K2: m = h->k2.get(); break;
      case KIND_FP8: m = h->f8.get(); break;
      case KIND_BF16: m = h->bf.get(); break;
      case KIND_FP8PC: m = h->pc.get(); break;
    }
    if (!m) throw std::runtime_error("not loaded");
    m->forward(qlen, k, expert_ids, weights, input, output, incremental != 0);
  });
}

void ktref_moe_destroy(void* hv) { delete (Handle*)hv; }

// ---- weight-format pins: the reference's own quantiser + its own inverse --------------------------------
// AMXINT4 / AMXINT8 per-row quantisation of one [n, k] bf16 matrix (BufferBInt4Impl::from_mat /
// BufferBInt8Impl::from_mat, kt-kernel/operators/amx/la/amx_buffers.hpp:498-753) followed by the reference's own
// dequantiser to_mat (:683-739).  `scales` receives the n fp32 row scales d[n].
int ktref_quant_roundtrip(int kind, int n, int k, const void* src_bf16, void* dst_bf16, float* scales) {
  return guarded([&] {
    if (kind == KIND_INT4) {
      using K = amx::GemmKernel224Int4;
      size_t sz = K::BufferB::required_size(n, k);
      void* buf = std::aligned_alloc(64, (sz + 63) / 64 * 64);
      K::BufferB bb(n, k, buf);
      int nth = K::recommended_nth(n);
      for (int ith = 0; ith < nth; ith++) bb.from_mat((ggml_bf16_t*)src_bf16, ith, nth);
      for (int ith = 0; ith < nth; ith++) bb.to_mat((ggml_bf16_t*)dst_bf16, ith, nth);
      memcpy(scales, bb.d, sizeof(float) * n);
      std::free(buf);
    } else {
      throw std::runtime_error("roundtrip: kind not supported");
    }
  });
}

// The packed bytes themselves: what the reference's converter writes per expert matrix and NUMA part
// ("<layer>.ffn_*_exps.E.numa.N.weight" = BufferB::b, ".scale" = BufferB::d; operators/amx/moe.hpp:103-124, 271-296).
// kind 0 = AMXINT4 (n*k/2 bytes), 1 = AMXINT8 (n*k bytes).  Pins ktransformers_amd/kt_kernel/utils/amx_packed.py.
int ktref_pack_b(int kind, int n, int k, const void* src_bf16, void* packed, float* scales) {
  return guarded([&] {
    auto run = [&](auto tag) {
      using K = decltype(tag);
      size_t sz = K::BufferB::required_size(n, k);
      void* buf = std::aligned_alloc(64, (sz + 63) / 64 * 64);
      {
        typename K::BufferB bb(n, k, buf);
        int nth = K::recommended_nth(n);
        for (int ith = 0; ith < nth; ith++) bb.from_mat((ggml_bf16_t*)src_bf16, ith, nth);
        memcpy(packed, bb.b, sz - sizeof(float) * n);
        memcpy(scales, bb.d, sizeof(float) * n);
      }
      std::free(buf);
    };
    if (kind == KIND_INT4) run(amx::GemmKernel224Int4{});
    else if (kind == KIND_INT8) run(amx::GemmKernel224Int8{});
    else throw std::runtime_error("pack_b: kind not supported");
  });
}

}  // extern "C"
