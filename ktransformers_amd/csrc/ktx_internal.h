// ktx_internal.h — library-internal (C++ linkage, not part of the C ABI) hand-offs between the translation units of
// libktx_hip.so.
#ifndef KTX_INTERNAL_H
#define KTX_INTERNAL_H
#include <stddef.h>
#include <stdint.h>

#include "../../include/ktx_linear.h"

// what another kernel needs to read a dense linear's tiled weights in place (ktx_linear.hip owns the handle)
struct KtxLinearRaw {
  const uint8_t* w;    // W tiles [strip][NKS][tile]
  const void* sc;      // W4: bf16 [strip][NKS][16][128/G]
  const void* bias;    // bf16 [N] or nullptr
  int in_features, out_features, NKS, nstrips, format, group_size, batch, device;
  bool loaded;
};
int ktx_linear_raw(ktx_linear_t h, KtxLinearRaw* out);

// KV split count ktx_mla_decode_partials would take for this call when it runs the 2x4 workgroup shape (0 otherwise): the
// one-launch decode step (ktx_attn.hip) splits the context the same way (ktx_mla.hip owns the rule)
struct ktx_mla_config;
extern "C" int ktx_mla_decode_nsplit(const ktx_mla_config* cfg, int total_q_tokens, size_t workspace_bytes);

#endif
