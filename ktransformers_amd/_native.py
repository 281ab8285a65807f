"""ctypes binding of the C ABI declared in include/ktx_moe.h.

PyTorch is used only for device memory and streams: every pointer crossing this boundary is a raw device pointer
(tensor.data_ptr()) and every launch goes on the caller's current HIP stream, so calls are HIP-graph capturable.
There is NO CPU fallback: if libktx_hip.so is absent the import raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libktx_hip.so")

FMT = {"AMXINT4": 0, "AMXINT8": 1, "RAWINT4": 2, "FP8": 3, "BF16": 4, "GGUF": 5, "FP8_PERCHANNEL": 6}
GGML_TYPE_Q4_K, GGML_TYPE_Q6_K, GGML_TYPE_IQ1_S = 12, 14, 19
GGML_BLOCK_BYTES = {10: 84, 11: 110, 12: 144, 13: 176, 14: 210, 19: 50, 23: 136,   # Q2_K, Q3_K, Q4_K, Q5_K, Q6_K, IQ1_S, IQ4_XS (256 weights per block)
                    2: 18, 6: 22, 8: 34}                                            # Q4_0, Q5_0, Q8_0 (32 weights per block; Q8_0 activations)
GGML_LEGACY_TYPES = (2, 6, 8)


def ggml_block_elems(ty: int) -> int:
    return 32 if ty in GGML_LEGACY_TYPES else 256
MAT_GATE, MAT_UP, MAT_DOWN = 0, 1, 2


class KtxError(RuntimeError):
    """Raised where the reference raises RuntimeError from a C++ std::runtime_error."""


class _GateConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_routed_experts", "hidden_size", "top_k", "n_group", "topk_group", "scoring",
                                         "topk_method", "norm_topk_prob")] + [("routed_scaling_factor", C.c_float)]


class _MlaConfig(C.Structure):
    _fields_ = [("num_heads", C.c_int32), ("head_dim_ckv", C.c_int32), ("head_dim_kpe", C.c_int32),
                ("page_size", C.c_int32), ("sm_scale", C.c_float), ("max_splits", C.c_int32), ("kv_len_hint", C.c_int32)]


class _LinearConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("in_features", "out_features", "format", "group_size", "max_len", "device", "batch")]


class _LinearFusion(C.Structure):
    _fields_ = [("norm_weight", C.c_void_p), ("norm_eps", C.c_float), ("add1", C.c_void_p), ("add1_ld", C.c_int64),
                ("add2", C.c_void_p), ("add2_ld", C.c_int64), ("x_ld", C.c_int64), ("y_ld", C.c_int64), ("glu", C.c_int32),
                ("glu_in", C.c_int32)]


class _GemmArgs(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("batch", C.c_int32),
                ("A", C.c_void_p), ("lda", C.c_int64), ("a_bs", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64), ("b_bs", C.c_int64),
                ("Y", C.c_void_p), ("ldy", C.c_int64), ("y_bs", C.c_int64), ("bias", C.c_void_p), ("out_f32", C.c_int32),
                ("variant", C.c_int32)]


class _AttnDecodeArgs(C.Structure):   # include/ktx_attn.h: ktx_attn_decode_args
    _fields_ = [("qkv_a", C.c_void_p), ("q_b", C.c_void_p), ("q_absorb", C.c_void_p), ("out_absorb", C.c_void_p), ("o_proj", C.c_void_p),
                ("num_heads", C.c_int32), ("nope_dim", C.c_int32), ("rope_dim", C.c_int32), ("kv_lora", C.c_int32), ("v_dim", C.c_int32),
                ("q_lora", C.c_int32), ("hidden", C.c_int32),
                ("d_x", C.c_void_p), ("d_y", C.c_void_p),
                ("d_in_norm_w", C.c_void_p), ("in_norm_eps", C.c_float),
                ("d_qa_norm_w", C.c_void_p), ("qa_norm_eps", C.c_float),
                ("d_kv_norm_w", C.c_void_p), ("kv_norm_eps", C.c_float),
                ("d_position", C.c_void_p), ("d_inv_freq", C.c_void_p), ("mscale", C.c_float),
                ("d_ckv", C.c_void_p), ("d_k_pe", C.c_void_p), ("ckv_token_stride", C.c_int64), ("kpe_token_stride", C.c_int64),
                ("page_size", C.c_int32), ("d_kv_indptr", C.c_void_p), ("d_kv_indices", C.c_void_p), ("d_kv_len", C.c_void_p),
                ("kv_len_hint", C.c_int32), ("sm_scale", C.c_float), ("phases", C.c_int32), ("last", C.c_int32)]


class _MoeConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "expert_num", "num_experts_per_tok", "hidden_size", "intermediate_size", "max_len", "format", "group_size",
        "device", "expert_begin", "global_expert_num")]


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m ktransformers_amd.build` (hipcc, gfx950). "
            "ktransformers_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.ktx_last_error.restype = C.c_char_p
    lib.ktx_moe_create.argtypes = [C.POINTER(_MoeConfig), C.POINTER(C.c_void_p)]
    lib.ktx_moe_destroy.argtypes = [C.c_void_p]
    lib.ktx_moe_load_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ktx_moe_load_quantized.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ktx_moe_load_fp8.argtypes = [C.c_void_p] * 7
    lib.ktx_moe_load_fp8_perchannel.argtypes = [C.c_void_p] * 7
    lib.ktx_moe_load_rawint4.argtypes = [C.c_void_p] * 7
    lib.ktx_moe_load_gguf.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3
    lib.ktx_moe_combine.argtypes = [C.c_int] * 3 + [C.c_void_p] * 4 + [C.c_int, C.c_void_p]
    lib.ktx_moe_set_expert_mask.argtypes = [C.c_void_p, C.c_void_p]
    lib.ktx_moe_merge_partials.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.ktx_moe_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_void_p]
    lib.ktx_moe_forward_ex.argtypes = lib.ktx_moe_forward.argtypes
    lib.ktx_moe_forward_side.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 8
    lib.ktx_moe_set_exact.argtypes = [C.c_void_p, C.c_int]
    lib.ktx_moe_weight_bytes.argtypes = [C.c_void_p]
    lib.ktx_moe_weight_bytes.restype = C.c_size_t
    lib.ktx_moe_debug_ptrs.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    lib.ktx_gate_logits.argtypes = [C.POINTER(_GateConfig), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ktx_gate_select.argtypes = [C.POINTER(_GateConfig), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]
    lib.ktx_gate_forward.argtypes = [C.POINTER(_GateConfig), C.c_void_p, C.c_int] + [C.c_void_p] * 8
    lib.ktx_gate_forward_norm.argtypes = [C.POINTER(_GateConfig), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float] + [C.c_void_p] * 8
    lib.ktx_mla_workspace_bytes.argtypes = [C.POINTER(_MlaConfig), C.c_int]
    lib.ktx_mla_workspace_bytes.restype = C.c_size_t
    lib.ktx_mla_decode.argtypes = [C.POINTER(_MlaConfig)] + [C.c_void_p] * 4 + [C.c_int64, C.c_int64] + [C.c_void_p] * 5 + [
        C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.ktx_mla_decode_append.argtypes = [C.POINTER(_MlaConfig)] + [C.c_void_p] * 4 + [C.c_int64, C.c_int64] + [C.c_void_p] * 5 + [
        C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.ktx_mla_prefill.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                    C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ktx_mla_cache_append.argtypes = [C.POINTER(_MlaConfig), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.ktx_mla_debug_stamps.argtypes = [C.c_void_p]
    lib.ktx_linear_create.argtypes = [C.POINTER(_LinearConfig), C.POINTER(C.c_void_p)]
    lib.ktx_linear_destroy.argtypes = [C.c_void_p]
    lib.ktx_linear_load_bf16.argtypes = [C.c_void_p] * 3
    lib.ktx_linear_load_w4.argtypes = [C.c_void_p] * 4
    lib.ktx_linear_load_w8.argtypes = [C.c_void_p] * 4
    lib.ktx_linear_load_fp8.argtypes = [C.c_void_p] * 4
    lib.ktx_linear_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ktx_linear_forward_batched.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                               C.c_int64, C.c_int64, C.c_void_p]
    lib.ktx_linear_forward_batched_prep.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                                    C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                                    C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_float, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    lib.ktx_linear_forward_fused.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(_LinearFusion),
                                             C.c_void_p]
    lib.ktx_linear_decode_eligible.argtypes = [C.c_void_p, C.c_int]
    lib.ktx_linear_dequant_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.ktx_fp8_act_quant.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.ktx_linear_gemm_fp8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    lib.ktx_mla_decode_partials.argtypes = [C.POINTER(_MlaConfig)] + [C.c_void_p] * 4 + [C.c_int64, C.c_int64] + [C.c_void_p] * 5 + [
        C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.c_void_p]
    lib.ktx_linear_merge_eligible.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.ktx_linear_forward_batched_merge.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                                     C.c_int64, C.c_int64, C.c_void_p]
    lib.ktx_linear_qb_absorb_eligible.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5
    lib.ktx_linear_forward_qb_absorb.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_float] + \
        [C.c_int] * 4 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                         C.c_void_p, C.c_float, C.c_void_p]
    lib.ktx_linear_forward_fused_gate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(_LinearFusion),
                                                  C.POINTER(_GateConfig)] + [C.c_void_p] * 8
    lib.ktx_linear_weight_bytes.argtypes = [C.c_void_p]
    lib.ktx_linear_weight_bytes.restype = C.c_size_t
    lib.ktx_linear_debug_get_w4.argtypes = [C.c_void_p] * 3
    lib.ktx_linear_debug_force_gemm.argtypes = [C.c_int]
    lib.ktx_rmsnorm.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_int,
                                C.c_void_p, C.c_void_p]
    lib.ktx_fused_add_rmsnorm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    lib.ktx_silu_mul.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ktx_argmax_workspace_bytes.argtypes = [C.c_int]
    lib.ktx_argmax_workspace_bytes.restype = C.c_size_t
    lib.ktx_argmax_bf16.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ktx_mla_prep.argtypes = [C.c_int] * 5 + [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_float,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    lib.ktx_ep_create.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_void_p)]
    lib.ktx_ep_destroy.argtypes = [C.c_void_p]
    lib.ktx_ep_destroy.restype = None
    lib.ktx_ep_export.argtypes = [C.c_void_p, C.c_void_p]
    lib.ktx_ep_local_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.ktx_ep_import.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.ktx_ep_import_ptr.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.ktx_ep_gather.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7
    lib.ktx_ep_reduce.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ktx_ep_reduce_only.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ktx_ep_status.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    lib.ktx_ep_set_spin_seconds.argtypes = [C.c_void_p, C.c_double]
    lib.ktx_profile_enable.argtypes = [C.c_int]
    lib.ktx_debug_force_generic.argtypes = [C.c_int]
    lib.ktx_debug_set.argtypes = [C.c_int, C.c_int]
    lib.ktx_debug_get.argtypes = [C.c_int]
    lib.ktx_timing_enable.argtypes = [C.c_int]
    lib.ktx_timing_collect.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.ktx_gemm_bf16_nt.argtypes = [C.POINTER(_GemmArgs), C.c_void_p]
    lib.ktx_split_f32_bf16x3.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.ktx_profile_collect.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    lib.ktx_attn_decode_eligible.argtypes = [C.POINTER(_AttnDecodeArgs)]
    lib.ktx_attn_decode.argtypes = [C.POINTER(_AttnDecodeArgs), C.c_void_p]
    lib.ktx_attn_status.argtypes = [C.c_int, C.POINTER(C.c_uint32)]
    lib.ktx_attn_reset.argtypes = [C.c_int]
    lib.ktx_attn_status_any.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_uint32)]
    lib.ktx_attn_debug_read.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    lib.ktx_attn_debug_stamps.argtypes = [C.c_void_p]
    return lib


# ---- call tracing (measurement support, used by bench.py) ---------------------------------------------------------------
# Every entry point that ENQUEUES work takes the HIP stream as its last argument.  While TRACE is a list, each such call is
# appended to it as (name, args) right before it is made; replay_call() re-issues one on the current stream.  bench.py
# records one decode step during graph capture (all pointers then stay valid for the life of the graph) and replays the
# calls of ONE kernel class alone to time that kernel on distinct layers' weights.  Not used by the product path.
STREAM_CALLS = frozenset((
    "ktx_moe_forward", "ktx_moe_forward_ex", "ktx_moe_combine", "ktx_gate_logits", "ktx_gate_select", "ktx_gate_forward",
    "ktx_gate_forward_norm", "ktx_mla_decode", "ktx_mla_decode_append", "ktx_mla_cache_append", "ktx_mla_prefill", "ktx_linear_forward",
    "ktx_linear_forward_batched", "ktx_linear_forward_batched_prep", "ktx_linear_forward_fused", "ktx_rmsnorm", "ktx_fused_add_rmsnorm", "ktx_silu_mul",
    "ktx_mla_prep", "ktx_argmax", "ktx_gemm_bf16_nt", "ktx_attn_decode", "ktx_fp8_act_quant", "ktx_linear_gemm_fp8"))   # (not ktx_ep_*: a gather replayed without its reduce would desynchronise the call tags)
TRACE: list | None = None
HANDLES: dict = {}      # native handle address -> weakref to the owning MoEHandle / LinearHandle (labels for traced calls)


class _TracedLib:
    def __init__(self, raw):
        object.__setattr__(self, "_raw", raw)
        object.__setattr__(self, "_wrapped", {})

    def __getattr__(self, name):
        fn = getattr(self._raw, name)
        if name not in STREAM_CALLS:
            return fn
        w = self._wrapped.get(name)
        if w is None:
            def w(*args, _fn=fn, _name=name):
                if TRACE is not None:
                    TRACE.append((_name, args))
                return _fn(*args)
            self._wrapped[name] = w
        return w


def replay_call(name: str, args: tuple, device) -> None:
    """Re-issue a traced call on the current stream of `device` (its recorded stream argument is replaced)."""
    check(getattr(lib._raw, name)(*args[:-1], _stream_ptr(device)))


def _register(handle_obj) -> None:
    import weakref
    HANDLES[handle_obj._h.value] = weakref.ref(handle_obj)


lib = _TracedLib(_load())

PROFILE_SLOTS = ("prep", "gate_up_gemm", "act_quant", "down_gemm", "combine")


def timing_enable(mode: int) -> None:
    """Per-launch timing of every library kernel (include/ktx_moe.h): 0 off, 1 HIP events around each launch, 2 labels only."""
    check(lib.ktx_timing_enable(int(mode)))


def timing_collect() -> list:
    """[(label, algorithmic_bytes, microseconds | None)] for every launch since the previous collect, in launch order."""
    need = C.c_size_t(0)
    check(lib.ktx_timing_collect(None, 0, C.byref(need)))
    buf = C.create_string_buffer(max(int(need.value), 1))
    check(lib.ktx_timing_collect(buf, len(buf), C.byref(need)))
    out = []
    for line in buf.value.decode("utf-8", "replace").splitlines():
        label, nbytes, us = line.rsplit("\t", 2)
        out.append((label, float(nbytes), float(us) if float(us) >= 0 else None))
    return out


def profile_enable(on: bool) -> None:
    check(lib.ktx_profile_enable(1 if on else 0))


def force_generic_path(on: bool) -> None:
    check(lib.ktx_debug_force_generic(1 if on else 0))


def profile_collect() -> dict:
    ms = (C.c_double * 5)()
    cnt = (C.c_longlong * 5)()
    check(lib.ktx_profile_collect(ms, cnt))
    return {name: (ms[i], int(cnt[i])) for i, name in enumerate(PROFILE_SLOTS)}


def check(rc: int) -> None:
    if rc != 0:
        raise KtxError(lib.ktx_last_error().decode("utf-8", "replace"))


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class MoEHandle:
    """Owner of one ktx_moe_t: HBM-resident packed experts + workspace for one MoE layer on one GPU."""

    def __init__(self, expert_num: int, num_experts_per_tok: int, hidden_size: int, intermediate_size: int,
                 max_len: int, method: str = "AMXINT4", device: int | torch.device = 0, group_size: int = 0,
                 expert_begin: int = 0, global_expert_num: int = 0):
        dev = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
        if dev.type != "cuda":
            raise KtxError("MoEHandle needs a HIP device; there is no CPU path")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if method not in FMT:
            raise KtxError(f"unknown method {method!r}")
        self.device = dev
        self.E, self.k, self.H, self.I, self.max_len = expert_num, num_experts_per_tok, hidden_size, intermediate_size, max_len
        self.method = method
        cfg = _MoeConfig(expert_num, num_experts_per_tok, hidden_size, intermediate_size, max_len, FMT[method],
                         group_size, dev.index or 0, expert_begin, global_expert_num or expert_num)
        h = C.c_void_p()
        check(lib.ktx_moe_create(C.byref(cfg), C.byref(h)))
        self._h = h
        _register(self)

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib.ktx_moe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights ---------------------------------------------------------------------------------
    def load_bf16(self, gate: torch.Tensor, up: torch.Tensor, down: torch.Tensor) -> None:
        """Online quantisation from bf16 [E,I,H]/[E,I,H]/[E,H,I] device tensors (reference: moe.hpp:352-387)."""
        for t, shape in ((gate, (self.E, self.I, self.H)), (up, (self.E, self.I, self.H)), (down, (self.E, self.H, self.I))):
            if t.dtype != torch.bfloat16 or tuple(t.shape) != shape or not t.is_contiguous() or t.device != self.device:
                raise KtxError(f"load_bf16: expected contiguous bf16 {shape} on {self.device}, got {t.dtype} {tuple(t.shape)} on {t.device}")
        torch.cuda.synchronize(self.device)
        check(lib.ktx_moe_load_bf16(self._h, gate.data_ptr(), up.data_ptr(), down.data_ptr()))

    def load_fp8(self, gate: torch.Tensor, up: torch.Tensor, down: torch.Tensor, gate_scale: torch.Tensor,
                 up_scale: torch.Tensor, down_scale: torch.Tensor) -> None:
        """DeepSeek block-fp8 experts: uint8/float8_e4m3fn [E,I,H]/[E,I,H]/[E,H,I] + fp32 scale_inv [E,N/128,K/128]."""
        ws = []
        for t, shape in ((gate, (self.E, self.I, self.H)), (up, (self.E, self.I, self.H)), (down, (self.E, self.H, self.I))):
            if t.element_size() != 1 or tuple(t.shape) != shape or not t.is_contiguous() or t.device != self.device:
                raise KtxError(f"load_fp8: expected contiguous 1-byte {shape} on {self.device}")
            ws.append(t)
        ss = []
        for t, (n, kk) in ((gate_scale, (self.I, self.H)), (up_scale, (self.I, self.H)), (down_scale, (self.H, self.I))):
            if t.dtype != torch.float32 or tuple(t.shape) != (self.E, n // 128, kk // 128) or not t.is_contiguous() or t.device != self.device:
                raise KtxError("load_fp8: scale_inv must be contiguous fp32 [E, N/128, K/128] on the handle's device")
            ss.append(t)
        torch.cuda.synchronize(self.device)
        check(lib.ktx_moe_load_fp8(self._h, ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), ss[0].data_ptr(),
                                   ss[1].data_ptr(), ss[2].data_ptr()))

    def load_fp8_perchannel(self, gate: torch.Tensor, up: torch.Tensor, down: torch.Tensor, gate_scale: torch.Tensor,
                            up_scale: torch.Tensor, down_scale: torch.Tensor) -> None:
        """FP8_PERCHANNEL experts: uint8/float8_e4m3fn [E,I,H]/[E,I,H]/[E,H,I] + one fp32 scale per output row:
        gate/up [E,I], down [E,H]."""
        ws = []
        for t, shape in ((gate, (self.E, self.I, self.H)), (up, (self.E, self.I, self.H)), (down, (self.E, self.H, self.I))):
            if t.element_size() != 1 or tuple(t.shape) != shape or not t.is_contiguous() or t.device != self.device:
                raise KtxError(f"load_fp8_perchannel: expected contiguous 1-byte {shape} on {self.device}")
            ws.append(t)
        ss = []
        for t, n in ((gate_scale, self.I), (up_scale, self.I), (down_scale, self.H)):
            if t.dtype != torch.float32 or tuple(t.shape) != (self.E, n) or not t.is_contiguous() or t.device != self.device:
                raise KtxError("load_fp8_perchannel: scales must be contiguous fp32 [E, N] on the handle's device")
            ss.append(t)
        torch.cuda.synchronize(self.device)
        check(lib.ktx_moe_load_fp8_perchannel(self._h, ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), ss[0].data_ptr(),
                                              ss[1].data_ptr(), ss[2].data_ptr()))

    def load_rawint4(self, gate: torch.Tensor, up: torch.Tensor, down: torch.Tensor, gate_scale: torch.Tensor,
                     up_scale: torch.Tensor, down_scale: torch.Tensor) -> None:
        """Kimi-K2 native int4: uint8 nibbles [E,I,H/2]/[E,I,H/2]/[E,H,I/2] + bf16 scales [E,N,K/32]."""
        for t, shape in ((gate, (self.E, self.I, self.H // 2)), (up, (self.E, self.I, self.H // 2)),
                         (down, (self.E, self.H, self.I // 2))):
            if t.dtype != torch.uint8 or tuple(t.shape) != shape or not t.is_contiguous() or t.device != self.device:
                raise KtxError(f"load_rawint4: expected contiguous uint8 {shape} on {self.device}")
        for t, (n, kk) in ((gate_scale, (self.I, self.H)), (up_scale, (self.I, self.H)), (down_scale, (self.H, self.I))):
            if t.dtype != torch.bfloat16 or tuple(t.shape) != (self.E, n, kk // 32) or not t.is_contiguous() or t.device != self.device:
                raise KtxError("load_rawint4: scales must be contiguous bf16 [E, N, K/32] on the handle's device")
        torch.cuda.synchronize(self.device)
        check(lib.ktx_moe_load_rawint4(self._h, gate.data_ptr(), up.data_ptr(), down.data_ptr(), gate_scale.data_ptr(),
                                       up_scale.data_ptr(), down_scale.data_ptr()))

    def set_exact(self, exact: bool = True) -> None:
        """ktx_moe_set_exact (include/ktx_moe.h): with exact=True no path of this handle re-associates the reference's fp32 sums — the
        RAWINT4 prompt-chunk kernel (the one fast path that does, within 2 bf16 ulp) is replaced by the exact 4-row kernel."""
        check(lib.ktx_moe_set_exact(self._h, 1 if exact else 0))

    def load_gguf(self, gate: torch.Tensor, up: torch.Tensor, down: torch.Tensor, gate_type: int, up_type: int,
                  down_type: int) -> None:
        """Raw GGUF blocks (uint8 device tensors): gate/up [E, I, H/blk_elems*blk_bytes], down [E, H, I/blk_elems*blk_bytes]; ggml type
        ids.  The three matrices come from one family: k- / i-quants (Q8_K activations) or the legacy Q4_0 / Q5_0 / Q8_0 (Q8_0)."""
        fam = {ty in GGML_LEGACY_TYPES for ty in (gate_type, up_type, down_type)}
        if len(fam) != 1:
            raise KtxError("load_gguf: gate / up / down must all be k- / i-quants or all be legacy types (Q4_0, Q5_0, Q8_0): the "
                           "intermediate is quantised once, to the format ggml pairs the down matrix with")
        for t, n, kdim, ty in ((gate, self.I, self.H, gate_type), (up, self.I, self.H, up_type), (down, self.H, self.I, down_type)):
            if ty not in GGML_BLOCK_BYTES:
                raise KtxError(f"load_gguf: unsupported ggml type {ty} (Q2_K=10, Q3_K=11, Q4_K=12, Q5_K=13, Q6_K=14, IQ1_S=19, IQ4_XS=23; "
                               "Q4_0=2, Q5_0=6, Q8_0=8)")
            shape = (self.E, n, kdim // ggml_block_elems(ty) * GGML_BLOCK_BYTES[ty])
            if t.dtype != torch.uint8 or tuple(t.shape) != shape or not t.is_contiguous() or t.device != self.device:
                raise KtxError(f"load_gguf: expected contiguous uint8 {shape} on {self.device}, got {t.dtype} {tuple(t.shape)} on {t.device}")
        torch.cuda.synchronize(self.device)
        check(lib.ktx_moe_load_gguf(self._h, gate.data_ptr(), up.data_ptr(), down.data_ptr(), gate_type, up_type, down_type))

    def load_quantized(self, expert: int, which: int, q, scale) -> None:
        """One expert matrix from host int8 [N,K] multiplicands + fp32 [N] scales (numpy arrays)."""
        import numpy as np
        q = np.ascontiguousarray(q, dtype=np.int8)
        scale = np.ascontiguousarray(scale, dtype=np.float32)
        check(lib.ktx_moe_load_quantized(self._h, expert, which, q.ctypes.data, scale.ctypes.data))

    def set_expert_mask(self, mask) -> None:
        if mask is None:
            check(lib.ktx_moe_set_expert_mask(self._h, None))
            return
        import numpy as np
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        if m.shape != (self.E,):
            raise KtxError("mask must have shape [expert_num]")
        check(lib.ktx_moe_set_expert_mask(self._h, m.ctypes.data))

    @property
    def weight_bytes(self) -> int:
        return int(lib.ktx_moe_weight_bytes(self._h))

    # -- forward ---------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, expert_ids: torch.Tensor, weights: torch.Tensor, out: torch.Tensor | None = None,
                incremental: bool = False, bsz_tensor: torch.Tensor | None = None) -> torch.Tensor:
        """x bf16 [T,H]; expert_ids int64 [T,k]; weights fp32 [T,k] -> bf16 [T,H].  Enqueues on the current stream."""
        T, k = expert_ids.shape
        if x.dtype != torch.bfloat16 or x.shape != (T, self.H) or not x.is_contiguous():
            raise KtxError(f"forward: x must be contiguous bf16 [{T},{self.H}]")
        if expert_ids.dtype != torch.int64 or not expert_ids.is_contiguous():
            raise KtxError("forward: expert_ids must be contiguous int64 [T,k]")
        if weights.dtype != torch.float32 or weights.shape != (T, k) or not weights.is_contiguous():
            raise KtxError("forward: weights must be contiguous fp32 [T,k]")
        if out is None:
            if incremental:
                raise KtxError("forward: incremental=True needs `out` holding the previous output")
            out = torch.empty_like(x)
        elif out.dtype != torch.bfloat16 or out.shape != x.shape or not out.is_contiguous():
            raise KtxError("forward: out must be contiguous bf16 [T,H]")
        for t in (x, expert_ids, weights, out):
            if t.device != self.device:
                raise KtxError(f"forward: tensor on {t.device}, handle on {self.device}")
        bsz_ptr = None
        if bsz_tensor is not None:
            if bsz_tensor.dtype != torch.int32 or bsz_tensor.device != self.device:
                raise KtxError("forward: bsz_tensor must be int32 on the handle's device")
            bsz_ptr = bsz_tensor.data_ptr()
        check(lib.ktx_moe_forward(self._h, bsz_ptr, T, k, expert_ids.data_ptr(), weights.data_ptr(), x.data_ptr(),
                                  out.data_ptr(), 1 if incremental else 0, _stream_ptr(self.device)))
        return out

    def forward_side(self, x: torch.Tensor, expert_ids: torch.Tensor, weights: torch.Tensor, side: "LinearHandle",
                     side_x: torch.Tensor, residual: torch.Tensor | None = None, out: torch.Tensor | None = None,
                     bsz_tensor: torch.Tensor | None = None) -> torch.Tensor:
        """Tail of a MoE block in one call (ktx_moe_forward_side): residual + (routed experts(x) + side(side_x)) with the
        reference's bf16 tensor adds.  side: the shared experts' down_proj handle (N == hidden), side_x bf16 [T, side.K]."""
        T, k = expert_ids.shape
        if x.dtype != torch.bfloat16 or x.shape != (T, self.H) or not x.is_contiguous():
            raise KtxError(f"forward_side: x must be contiguous bf16 [{T},{self.H}]")
        if expert_ids.dtype != torch.int64 or not expert_ids.is_contiguous():
            raise KtxError("forward_side: expert_ids must be contiguous int64 [T,k]")
        if weights.dtype != torch.float32 or weights.shape != (T, k) or not weights.is_contiguous():
            raise KtxError("forward_side: weights must be contiguous fp32 [T,k]")
        if side.N != self.H or side.batch != 1 or side.device != self.device:
            raise KtxError("forward_side: the side linear must map to hidden_size on the experts' device")
        if side_x.dtype != torch.bfloat16 or side_x.shape != (T, side.K) or not side_x.is_contiguous():
            raise KtxError(f"forward_side: side_x must be contiguous bf16 [{T},{side.K}]")
        if residual is not None and (residual.dtype != torch.bfloat16 or residual.shape != (T, self.H) or not residual.is_contiguous()):
            raise KtxError(f"forward_side: residual must be contiguous bf16 [{T},{self.H}]")
        if out is None:
            out = torch.empty_like(x)
        elif out.dtype != torch.bfloat16 or out.shape != x.shape or not out.is_contiguous():
            raise KtxError("forward_side: out must be contiguous bf16 [T,H]")
        for t in (x, expert_ids, weights, out, side_x) + ((residual,) if residual is not None else ()):
            if t.device != self.device:
                raise KtxError(f"forward_side: tensor on {t.device}, handle on {self.device}")
        bsz_ptr = None
        if bsz_tensor is not None:
            if bsz_tensor.dtype != torch.int32 or bsz_tensor.device != self.device:
                raise KtxError("forward_side: bsz_tensor must be int32 on the handle's device")
            bsz_ptr = bsz_tensor.data_ptr()
        check(lib.ktx_moe_forward_side(self._h, bsz_ptr, T, k, expert_ids.data_ptr(), weights.data_ptr(), x.data_ptr(),
                                       out.data_ptr(), side._h, side_x.data_ptr(),
                                       residual.data_ptr() if residual is not None else None, _stream_ptr(self.device)))
        return out

    def forward_partial(self, x: torch.Tensor, expert_ids: torch.Tensor, weights: torch.Tensor,
                        out: torch.Tensor | None = None) -> torch.Tensor:
        """Expert-parallel leg: fp32 [T,H] un-rounded weighted sums over the experts this handle owns."""
        T, k = expert_ids.shape
        if x.dtype != torch.bfloat16 or x.shape != (T, self.H) or not x.is_contiguous():
            raise KtxError(f"forward_partial: x must be contiguous bf16 [{T},{self.H}]")
        if expert_ids.dtype != torch.int64 or not expert_ids.is_contiguous():
            raise KtxError("forward_partial: expert_ids must be contiguous int64 [T,k]")
        if weights.dtype != torch.float32 or weights.shape != (T, k) or not weights.is_contiguous():
            raise KtxError("forward_partial: weights must be contiguous fp32 [T,k]")
        if out is None:
            out = torch.empty((T, self.H), dtype=torch.float32, device=self.device)
        elif out.dtype != torch.float32 or out.shape != (T, self.H) or not out.is_contiguous():
            raise KtxError("forward_partial: out must be contiguous fp32 [T,H]")
        check(lib.ktx_moe_forward_ex(self._h, None, T, k, expert_ids.data_ptr(), weights.data_ptr(), x.data_ptr(),
                                     out.data_ptr(), 2, _stream_ptr(self.device)))
        return out


def moe_merge_partials(parts: torch.Tensor, out: torch.Tensor, incremental: bool = False, rows: int | None = None) -> torch.Tensor:
    """merge_results of the reference's NUMA tensor-parallel MoE (include/ktx_moe.h, ktx_moe_merge_partials): parts fp32
    [P, >= rows, H] (the forward_partial outputs of P handles of width I / P), out bf16 [rows, H]:
    out = bf16(((parts[0] + (out if incremental)) + parts[1]) + ...)."""
    if parts.dtype != torch.float32 or parts.dim() != 3 or not parts[0].is_contiguous() or parts.device != out.device:
        raise KtxError("moe_merge_partials: parts must be fp32 [P, rows, H] with contiguous parts on the output's device")
    T = out.shape[0] if rows is None else int(rows)
    if out.dtype != torch.bfloat16 or out.dim() != 2 or out.shape[1] != parts.shape[2] or not out.is_contiguous() or T > parts.shape[1]:
        raise KtxError("moe_merge_partials: out must be contiguous bf16 [rows, H]")
    check(lib.ktx_moe_merge_partials(parts.shape[0], T, parts.shape[2], parts.data_ptr(), parts.stride(0), out.data_ptr(),
                                     1 if incremental else 0, None, _stream_ptr(out.device)))
    return out


EP_MEMORY = {"uncached": 0, "finegrained": 1, "plain": 2}
EP_HANDLE_BYTES = 64


class EpExchange:
    """Expert-parallel decode exchange over direct peer writes (include/ktx_ep.h): this rank's symmetric buffer plus the
    mapped buffers of its peers.  gather() and reduce() are the two launches of one MoE layer; every rank makes the same
    sequence of calls with the same T.  Peers in other processes are mapped from export_handle() bytes, peers in this
    process from local_ptr()."""

    def __init__(self, world: int, rank: int, max_tokens: int, hidden: int, topk: int, device, memory: str = "uncached"):
        dev = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
        if dev.index is None:       # "cuda": the process's current device, like every torch allocation
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.world, self.rank, self.max_tokens, self.H, self.k = world, rank, max_tokens, hidden, topk
        self._h = C.c_void_p()
        if memory not in EP_MEMORY:
            raise KtxError(f"EpExchange: memory must be one of {sorted(EP_MEMORY)}, got {memory!r}")
        check(lib.ktx_ep_create(dev.index, world, rank, max_tokens, hidden, topk, EP_MEMORY[memory], C.byref(self._h)))

    def close(self) -> None:
        if self._h:
            lib.ktx_ep_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def export_handle(self) -> bytes:
        buf = C.create_string_buffer(EP_HANDLE_BYTES)
        check(lib.ktx_ep_export(self._h, buf))
        return buf.raw

    def local_ptr(self) -> int:
        p = C.c_void_p()
        check(lib.ktx_ep_local_ptr(self._h, C.byref(p)))
        return int(p.value)

    def import_handle(self, peer: int, handle: bytes) -> None:
        if len(handle) != EP_HANDLE_BYTES:
            raise KtxError(f"EpExchange.import_handle: expected {EP_HANDLE_BYTES} bytes")
        check(lib.ktx_ep_import(self._h, peer, C.create_string_buffer(handle, EP_HANDLE_BYTES)))

    def import_ptr(self, peer: int, ptr: int) -> None:
        check(lib.ktx_ep_import_ptr(self._h, peer, C.c_void_p(ptr)))

    def set_spin_seconds(self, seconds: float) -> None:
        check(lib.ktx_ep_set_spin_seconds(self._h, float(seconds)))

    def gather(self, x: torch.Tensor, ids: torch.Tensor, w: torch.Tensor):
        """x bf16 [T,H], ids int64 [T,k], w fp32 [T,k] -> (xg [R*T,H], idsg [R*T,k], wg [R*T,k]), rank r's rows at r*T."""
        T = x.shape[0]
        if x.dtype != torch.bfloat16 or x.shape != (T, self.H) or not x.is_contiguous():
            raise KtxError(f"EpExchange.gather: x must be contiguous bf16 [T,{self.H}]")
        if ids.dtype != torch.int64 or ids.shape != (T, self.k) or not ids.is_contiguous():
            raise KtxError(f"EpExchange.gather: ids must be contiguous int64 [T,{self.k}]")
        if w.dtype != torch.float32 or w.shape != (T, self.k) or not w.is_contiguous():
            raise KtxError(f"EpExchange.gather: w must be contiguous fp32 [T,{self.k}]")
        n = self.world * T
        xg = torch.empty((n, self.H), dtype=torch.bfloat16, device=self.device)
        idsg = torch.empty((n, self.k), dtype=torch.int64, device=self.device)
        wg = torch.empty((n, self.k), dtype=torch.float32, device=self.device)
        check(lib.ktx_ep_gather(self._h, T, x.data_ptr(), ids.data_ptr(), w.data_ptr(), xg.data_ptr(), idsg.data_ptr(),
                                wg.data_ptr(), _stream_ptr(self.device)))
        return xg, idsg, wg

    def reduce(self, part: torch.Tensor, out: torch.Tensor | None = None, reduce_only: bool = False) -> torch.Tensor:
        """part fp32 [R*T,H] -> bf16 [T,H]: the partials of this rank's tokens added in rank order, rounded once.
        reduce_only: this call is not preceded by a gather (replicated token stream): ktx_ep_reduce_only, which owns its call tag."""
        n = part.shape[0]
        if part.dtype != torch.float32 or part.dim() != 2 or part.shape[1] != self.H or n % self.world or not part.is_contiguous():
            raise KtxError(f"EpExchange.reduce: part must be contiguous fp32 [{self.world}*T,{self.H}]")
        T = n // self.world
        if out is None:
            out = torch.empty((T, self.H), dtype=torch.bfloat16, device=self.device)
        elif out.dtype != torch.bfloat16 or out.shape != (T, self.H) or not out.is_contiguous():
            raise KtxError("EpExchange.reduce: out must be contiguous bf16 [T,H]")
        fn = lib.ktx_ep_reduce_only if reduce_only else lib.ktx_ep_reduce
        check(fn(self._h, T, part.data_ptr(), out.data_ptr(), _stream_ptr(self.device)))
        return out

    def status(self) -> int:
        """0 = healthy, 1 / 2 = a gather / reduce poll gave up waiting for a peer.  Synchronises the current stream."""
        st = C.c_int(0)
        check(lib.ktx_ep_status(self._h, _stream_ptr(self.device), C.byref(st)))
        return int(st.value)


GATE_SCORING = {"sigmoid": 0, "softmax": 1}
GATE_TOPK = {"greedy": 0, "group_limited_greedy": 1, "noaux_tc": 2}


def moe_combine(rows: torch.Tensor, row_of_pair: torch.Tensor, weights: torch.Tensor, out: torch.Tensor | None = None,
                incremental: bool = False) -> torch.Tensor:
    """rows bf16 [R, H]; row_of_pair int32 [T, k] (-1 = skipped slot); weights fp32 [T, k] -> bf16 [T, H] (ktx_moe_combine)."""
    T, k = row_of_pair.shape
    H = rows.shape[1]
    if rows.dtype != torch.bfloat16 or row_of_pair.dtype != torch.int32 or weights.dtype != torch.float32:
        raise KtxError("moe_combine: rows bf16, row_of_pair int32, weights fp32")
    rows, rp, w = rows.contiguous(), row_of_pair.contiguous(), weights.contiguous()
    if out is None:
        out = torch.empty((T, H), dtype=torch.bfloat16, device=rows.device)
    check(lib.ktx_moe_combine(T, k, H, rows.data_ptr(), rp.data_ptr(), w.data_ptr(), out.data_ptr(), 1 if incremental else 0,
                              _stream_ptr(rows.device)))
    return out


LIN_FMT = {"BF16": 0, "W4": 1, "FP8": 2, "W8": 3}


class LinearHandle:
    """Owner of one ktx_linear_t (include/ktx_linear.h): tiled weights of one dense linear on one GPU."""

    def __init__(self, in_features: int, out_features: int, fmt: str = "W4", group_size: int = 64, max_len: int = 4096,
                 device: int | torch.device = 0, batch: int = 1):
        dev = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
        if dev.type != "cuda":
            raise KtxError("LinearHandle needs a HIP device; there is no CPU path")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if fmt not in LIN_FMT:
            raise KtxError(f"unknown linear format {fmt!r}")
        self.device, self.fmt, self.K, self.N, self.max_len = dev, fmt, in_features, out_features, max_len
        self.group_size = {"BF16": 0, "FP8": 128}.get(fmt, group_size)
        self.batch = max(int(batch), 1)
        cfg = _LinearConfig(in_features, out_features, LIN_FMT[fmt], self.group_size, max_len, dev.index or 0, self.batch)
        h = C.c_void_p()
        check(lib.ktx_linear_create(C.byref(cfg), C.byref(h)))
        self._h = h
        _register(self)

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib.ktx_linear_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, t: torch.Tensor, dtype, shape, what: str) -> torch.Tensor:
        if t.dtype != dtype or tuple(t.shape) != tuple(shape) or t.device != self.device:
            raise KtxError(f"{what}: expected {dtype} {tuple(shape)} on {self.device}, got {t.dtype} {tuple(t.shape)} on {t.device}")
        return t.contiguous()

    def _bias(self, bias):
        self._bias_t = None if bias is None else self._chk(bias, torch.bfloat16, (self.N,), "bias")
        return self._bias_t

    # ---- prompt-sized W4 calls: de-quantise once (Marlin's rounding), then the library's own BF16 GEMM (csrc/ktx_gemm.hip;
    # KTX_VENDOR_GEMM=1 sends the product to torch's F.linear instead — the A/B switch of scripts/gemm_bench.py) ------------
    PROMPT_MIN_T = 512          # below this the hand-written W4 GEMM (no scratch pass) is faster
    _DEQ_SCRATCH: dict = {}

    def dequant_bf16(self, out: torch.Tensor | None = None) -> torch.Tensor:
        """bf16 [N, K] = bf16((q - 8) * s), the weights gptq_marlin_gemm multiplies with (ktx_linear_dequant_bf16)."""
        if self.fmt != "W4" or self.batch != 1:
            raise KtxError("dequant_bf16: W4 handles only")
        if out is None:
            out = torch.empty((self.N, self.K), dtype=torch.bfloat16, device=self.device)
        check(lib.ktx_linear_dequant_bf16(self._h, out.data_ptr(), out.stride(0), _stream_ptr(self.device)))
        return out

    def _prompt_forward(self, x2: torch.Tensor, out, add1, add2, glu: bool) -> torch.Tensor:
        need = self.N * self.K
        buf = LinearHandle._DEQ_SCRATCH.get(self.device)
        if buf is None or buf.numel() < need:
            buf = LinearHandle._DEQ_SCRATCH[self.device] = torch.empty(need, dtype=torch.bfloat16, device=self.device)
        w = self.dequant_bf16(buf[:need].view(self.N, self.K))
        if os.environ.get("KTX_VENDOR_GEMM"):
            y = torch.nn.functional.linear(x2, w, getattr(self, "_bias_t", None))
        else:
            y = gemm_bf16_nt(x2, w, bias=getattr(self, "_bias_t", None))
        if glu:   # rows are interleaved per 16-row strip as [8 gate | 8 up]
            T = y.shape[0]
            y = silu_mul(y.view(T, self.N // 16, 2, 8).permute(0, 2, 1, 3).reshape(T, self.N))
        for a in (add1, add2):   # the epilogue adds of the decoder layer, torch's bf16 tensor arithmetic
            if a is not None:
                y = a.reshape(y.shape) + y
        if out is not None:
            out.copy_(y)
            return out
        return y

    # ---- prompt-sized FP8 calls: act_quant once per call + the block-scaled fp8 GEMM on the handle's own tiles
    # (csrc/ktx_linear_fp8gemm.inc; bit-identical to ktx_linear_forward; KTX_FP8_PROMPT_KERNEL=1 keeps the strip kernel for A/B) ----
    FP8_PROMPT_MIN_T = 128

    def _prompt_forward_fp8(self, x2: torch.Tensor, out, add1, add2, glu: bool) -> torch.Tensor:
        T = x2.shape[0]
        tp = (T + 127) // 128 * 128
        q = torch.empty((T, self.K), dtype=torch.uint8, device=self.device)
        st = torch.zeros((self.K // 128, tp), dtype=torch.float32, device=self.device)   # (act_quant writes T columns: the pad rows' scales are 0, not garbage)
        y = torch.empty((T, self.N), dtype=torch.bfloat16, device=self.device)
        sp = _stream_ptr(self.device)
        check(lib.ktx_fp8_act_quant(x2.data_ptr(), x2.stride(0), T, self.K, q.data_ptr(), st.data_ptr(), tp, sp))
        check(lib.ktx_linear_gemm_fp8(self._h, q.data_ptr(), st.data_ptr(), tp, T, y.data_ptr(), self.N, sp))
        if glu:   # rows are interleaved per 16-row strip as [8 gate | 8 up]
            y = silu_mul(y.view(T, self.N // 16, 2, 8).permute(0, 2, 1, 3).reshape(T, self.N))
        for a in (add1, add2):   # the epilogue adds of the decoder layer, torch's bf16 tensor arithmetic
            if a is not None:
                y = a.reshape(y.shape) + y
        if out is not None:
            out.copy_(y)
            return out
        return y

    def load_bf16(self, weight: torch.Tensor, bias: torch.Tensor | None = None) -> None:
        """weight: bf16 [out, in] (nn.Linear layout). W4 handles quantise with quantize_weights' arithmetic."""
        shape = (self.N, self.K) if self.batch == 1 else (self.batch, self.N, self.K)
        w = self._chk(weight, torch.bfloat16, shape, "load_bf16")
        b = self._bias(bias)
        torch.cuda.synchronize(self.device)
        check(lib.ktx_linear_load_bf16(self._h, w.data_ptr(), b.data_ptr() if b is not None else None))

    def load_w4(self, q: torch.Tensor, s: torch.Tensor, bias: torch.Tensor | None = None) -> None:
        """q: uint8 [in, out] in 0..15, s: bf16 [in/group, out] — the (q_w, s) of quantize_weights."""
        q = self._chk(q, torch.uint8, (self.K, self.N), "load_w4 q")
        s = self._chk(s, torch.bfloat16, (self.K // self.group_size, self.N), "load_w4 s")
        b = self._bias(bias)
        torch.cuda.synchronize(self.device)
        check(lib.ktx_linear_load_w4(self._h, q.data_ptr(), s.data_ptr(), b.data_ptr() if b is not None else None))

    def load_w8(self, q: torch.Tensor, s: torch.Tensor, bias: torch.Tensor | None = None) -> None:
        """q: uint8 [in, out] in 0..255, s: bf16 [in/group, out] — the (q_w, s) of quantize_weights(w, 8, group)."""
        q = self._chk(q, torch.uint8, (self.K, self.N), "load_w8 q")
        s = self._chk(s, torch.bfloat16, (self.K // self.group_size, self.N), "load_w8 s")
        b = self._bias(bias)
        torch.cuda.synchronize(self.device)
        check(lib.ktx_linear_load_w8(self._h, q.data_ptr(), s.data_ptr(), b.data_ptr() if b is not None else None))

    def load_fp8(self, weight: torch.Tensor, scale_inv: torch.Tensor, bias: torch.Tensor | None = None) -> None:
        """weight: float8_e4m3fn (or its uint8 bytes) [out, in]; scale_inv fp32 [ceil(out/128), ceil(in/128)]."""
        if weight.dtype == torch.float8_e4m3fn:
            weight = weight.view(torch.uint8)
        w = self._chk(weight, torch.uint8, (self.N, self.K), "load_fp8 weight")
        sc = self._chk(scale_inv, torch.float32, ((self.N + 127) // 128, (self.K + 127) // 128), "load_fp8 scale_inv")
        b = self._bias(bias)
        torch.cuda.synchronize(self.device)
        check(lib.ktx_linear_load_fp8(self._h, w.data_ptr(), sc.data_ptr(), b.data_ptr() if b is not None else None))

    def weight_bytes(self) -> int:
        return int(lib.ktx_linear_weight_bytes(self._h))

    def debug_get_w4(self):
        import numpy as np
        q = np.empty((self.K, self.N), dtype=np.uint8)
        s = np.empty((self.K // self.group_size, self.N), dtype=np.uint16)
        check(lib.ktx_linear_debug_get_w4(self._h, q.ctypes.data, s.ctypes.data))
        return q, s

    def decode_eligible(self, T: int) -> bool:
        return bool(lib.ktx_linear_decode_eligible(self._h, int(T)))

    def forward(self, x: torch.Tensor, bsz_tensor: torch.Tensor | None = None, out: torch.Tensor | None = None,
                norm: tuple | None = None, add1: torch.Tensor | None = None, add2: torch.Tensor | None = None,
                glu: bool = False, glu_in: bool = False) -> torch.Tensor:
        """x: bf16 [..., in] -> bf16 [..., out] on the current stream.  Optional fusions (include/ktx_linear.h,
        ktx_linear_fusion): norm = (weight bf16 [in], eps) applies RMSNorm to x inside the kernel (falls back to a separate
        ktx_rmsnorm launch where the decode kernel does not run); add1 / add2 = bf16 [..., out] tensors added in that order;
        glu_in: x is bf16 [..., 2 * in] = [gate | up] and the linear reads silu_mul(x) — inside the block-fp8 decode kernel,
        as a separate ktx_silu_mul launch otherwise."""
        if glu_in:
            return self._forward_glu_in(x, bsz_tensor, out, norm, add1, add2, glu)
        if x.dtype != torch.bfloat16 or x.shape[-1] != self.K or x.device != self.device:
            raise KtxError(f"forward: expected bf16 [..., {self.K}] on {self.device}, got {x.dtype} {tuple(x.shape)} on {x.device}")
        x2 = x if x.dim() == 2 else x.reshape(-1, self.K)
        strided = x2.dim() == 2 and x2.stride(1) == 1 and x2.stride(0) % 8 == 0 and x2.stride(0) != self.K
        if not x2.is_contiguous() and not strided:
            x2 = x2.contiguous()
            strided = False
        T = x2.shape[0]
        n_out = self.N // 2 if glu else self.N
        # (the GEMM behind the prompt paths wants K % 64 == 0, N % 8 == 0 and 16-byte aligned rows: other shapes stay on the strip
        # kernels, which take any out_features — ADVICE r3)
        gemm_ok = (self.batch == 1 and bsz_tensor is None and self.N % 8 == 0 and self.K % 64 == 0 and x2.data_ptr() % 16 == 0
                   and (x2.stride(0) * 2) % 16 == 0 and not torch.cuda.is_current_stream_capturing())
        if (self.fmt == "W4" and gemm_ok and T >= self.PROMPT_MIN_T and not os.environ.get("KTX_W4_PROMPT_KERNEL")):
            if norm is not None:
                x2 = rmsnorm(x2.contiguous(), norm[0], norm[1], native_rounding=True)
            return self._prompt_forward(x2, out, add1, add2, glu).reshape(*x.shape[:-1], n_out)
        if (self.fmt == "FP8" and gemm_ok and T >= self.FP8_PROMPT_MIN_T and not os.environ.get("KTX_FP8_PROMPT_KERNEL")):
            if norm is not None:
                x2 = rmsnorm(x2.contiguous(), norm[0], norm[1], native_rounding=True)
            return self._prompt_forward_fp8(x2, out, add1, add2, glu).reshape(*x.shape[:-1], n_out)
        if out is None:
            out = torch.empty((T, n_out), dtype=torch.bfloat16, device=self.device) if bsz_tensor is None else \
                torch.zeros((T, n_out), dtype=torch.bfloat16, device=self.device)
        bsz = None
        if bsz_tensor is not None:
            if bsz_tensor.dtype != torch.int32 or bsz_tensor.device != self.device:
                raise KtxError("forward: bsz_tensor must be int32 on the handle's device")
            bsz = bsz_tensor.data_ptr()
        if norm is None and add1 is None and add2 is None and not strided and not glu:
            check(lib.ktx_linear_forward(self._h, bsz, T, x2.data_ptr(), out.data_ptr(), _stream_ptr(self.device)))
            return out.reshape(*x.shape[:-1], self.N)
        fu = _LinearFusion(None, 0.0, None, 0, None, 0, x2.stride(0) if strided else 0, 0, 1 if glu else 0)
        keep = []
        if norm is not None:
            nw, eps = norm
            if self.decode_eligible(T):
                fu.norm_weight, fu.norm_eps = nw.data_ptr(), float(eps)
            else:
                x2 = rmsnorm(x2, nw, eps, native_rounding=True)
                fu.x_ld = 0
            keep.append(nw)
        for name, a in (("add1", add1), ("add2", add2)):
            if a is not None:
                a2 = a.reshape(-1, self.N)
                if a2.dtype != torch.bfloat16 or a2.shape[0] != T or a2.stride(1) != 1 or a2.device != self.device:
                    raise KtxError(f"forward: {name} must be bf16 [{T}, {self.N}] on {self.device}")
                setattr(fu, name, a2.data_ptr())
                setattr(fu, name + "_ld", a2.stride(0))
                keep.append(a2)
        check(lib.ktx_linear_forward_fused(self._h, bsz, T, x2.data_ptr(), out.data_ptr(), C.byref(fu), _stream_ptr(self.device)))
        return out.reshape(*x.shape[:-1], n_out)

    def _forward_glu_in(self, x, bsz_tensor, out, norm, add1, add2, glu):
        if norm is not None or glu:
            raise KtxError("forward: glu_in does not combine with norm / glu")
        if x.dtype != torch.bfloat16 or x.shape[-1] != 2 * self.K or x.device != self.device:
            raise KtxError(f"forward: glu_in expects bf16 [..., {2 * self.K}] on {self.device}, got {x.dtype} {tuple(x.shape)} on {x.device}")
        x2 = x.reshape(-1, 2 * self.K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        T = x2.shape[0]
        if self.fmt != "FP8" or not self.decode_eligible(T) or self.batch != 1 or os.environ.get("KTX_LINEAR_SEPARATE_SILU_MUL"):
            return self.forward(silu_mul(x2, bsz_tensor), bsz_tensor, out, None, add1, add2).reshape(*x.shape[:-1], self.N)
        if out is None:
            out = torch.empty((T, self.N), dtype=torch.bfloat16, device=self.device) if bsz_tensor is None else \
                torch.zeros((T, self.N), dtype=torch.bfloat16, device=self.device)
        bsz = None
        if bsz_tensor is not None:
            if bsz_tensor.dtype != torch.int32 or bsz_tensor.device != self.device:
                raise KtxError("forward: bsz_tensor must be int32 on the handle's device")
            bsz = bsz_tensor.data_ptr()
        fu = _LinearFusion(None, 0.0, None, 0, None, 0, 0, 0, 0, 1)
        keep = []
        for name, a in (("add1", add1), ("add2", add2)):
            if a is not None:
                a2 = a.reshape(-1, self.N)
                if a2.dtype != torch.bfloat16 or a2.shape[0] != T or a2.stride(1) != 1 or a2.device != self.device:
                    raise KtxError(f"forward: {name} must be bf16 [{T}, {self.N}] on {self.device}")
                setattr(fu, name, a2.data_ptr())
                setattr(fu, name + "_ld", a2.stride(0))
                keep.append(a2)
        check(lib.ktx_linear_forward_fused(self._h, bsz, T, x2.data_ptr(), out.data_ptr(), C.byref(fu), _stream_ptr(self.device)))
        return out.reshape(*x.shape[:-1], self.N)

    def forward_batched(self, x: torch.Tensor, out: torch.Tensor | None = None, bsz_tensor: torch.Tensor | None = None) -> torch.Tensor:
        """x: bf16 [T, batch, in] (any row / batch strides that are multiples of 8, unit stride along `in`) ->
        bf16 [T, batch, out]: y[t, b] = x[t, b] @ W[b]^T — the per-head absorb products of MLA."""
        if x.dtype != torch.bfloat16 or x.dim() != 3 or x.shape[1] != self.batch or x.shape[2] != self.K or x.stride(2) != 1:
            raise KtxError(f"forward_batched: expected bf16 [T, {self.batch}, {self.K}] with unit inner stride, got {tuple(x.shape)}")
        T = x.shape[0]
        if out is None:
            out = torch.empty((T, self.batch, self.N), dtype=torch.bfloat16, device=self.device)
        bsz = bsz_tensor.data_ptr() if bsz_tensor is not None else None
        check(lib.ktx_linear_forward_batched(self._h, bsz, T, x.data_ptr(), x.stride(0), x.stride(1), out.data_ptr(),
                                             out.stride(0), out.stride(1), _stream_ptr(self.device)))
        return out


def merge_and_unabsorb(oabs: "LinearHandle", partials: tuple, T: int, num_heads: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """Decode step: merge of the MLA KV splits + the per-head un-absorb products in one launch
    (ktx_linear_forward_batched_merge).  partials = MLAWrapper.run_partials(...) = (workspace, nsplit).
    Returns bf16 [T, heads, v_head_dim]."""
    ws, nsplit = partials
    if out is None:
        out = torch.empty((T, num_heads, oabs.N), dtype=torch.bfloat16, device=ws.device)
    part_o = ws.data_ptr()
    part_ml = part_o + T * num_heads * nsplit * oabs.K * 4
    check(lib.ktx_linear_forward_batched_merge(oabs._h, T, part_o, part_ml, nsplit, num_heads, out.data_ptr(), out.stride(0),
                                               out.stride(1), _stream_ptr(ws.device)))
    return out


def absorb_and_prep(qabs: "LinearHandle", q: torch.Tensor, kv: torch.Tensor, kv_norm_weight: torch.Tensor, eps: float,
                    positions: torch.Tensor, inv_freq: torch.Tensor, mscale: float, num_heads: int, nope_dim: int, rope_dim: int,
                    kv_lora: int):
    """Decode step (T <= 4): the per-head q-absorb products AND mla_prep in one launch (ktx_linear_forward_batched_prep).
    q: bf16 [T, H*(nope+rope)] (unit inner stride), kv: bf16 [T, kv_lora+rope].  Returns (q_nope [T,H,lora], q_pe, ckv, k_pe)."""
    T, dev = positions.numel(), positions.device
    _bf16_rows(q, "absorb_and_prep q")
    _bf16_rows(kv, "absorb_and_prep kv")
    q = q.reshape(T, -1)
    kv = kv.reshape(T, -1)
    if positions.dtype != torch.int64 or inv_freq.dtype != torch.float32:
        raise KtxError("absorb_and_prep: positions must be int64 and inv_freq fp32")
    if qabs.batch != num_heads or qabs.K != nope_dim:
        raise KtxError("absorb_and_prep: the handle is not the per-head q-absorb linear of this attention")
    q3 = q.unflatten(1, (num_heads, nope_dim + rope_dim))
    q_nope = torch.empty((T, num_heads, qabs.N), dtype=torch.bfloat16, device=dev)
    q_pe = torch.empty((T, num_heads, rope_dim), dtype=torch.bfloat16, device=dev)
    ckv = torch.empty((T, kv_lora), dtype=torch.bfloat16, device=dev)
    kpe = torch.empty((T, rope_dim), dtype=torch.bfloat16, device=dev)
    check(lib.ktx_linear_forward_batched_prep(qabs._h, T, q3.data_ptr(), q3.stride(0), q3.stride(1), q_nope.data_ptr(),
                                              q_nope.stride(0), q_nope.stride(1), num_heads, nope_dim, rope_dim, kv_lora,
                                              q.data_ptr(), q.stride(0), q_pe.data_ptr(), kv.data_ptr(), kv.stride(0),
                                              kv_norm_weight.data_ptr(), float(eps), ckv.data_ptr(), kpe.data_ptr(),
                                              positions.data_ptr(), inv_freq.data_ptr(), float(mscale), _stream_ptr(dev)))
    return q_nope, q_pe, ckv, kpe


def qb_absorb_eligible(q_b: "LinearHandle", qabs: "LinearHandle", T: int, num_heads: int, nope_dim: int, rope_dim: int,
                       kv_lora: int) -> bool:
    return bool(lib.ktx_linear_qb_absorb_eligible(q_b._h, qabs._h, int(T), num_heads, nope_dim, rope_dim, kv_lora))


def qb_absorb_and_prep(q_b: "LinearHandle", qabs: "LinearHandle", q_a: torch.Tensor, q_a_norm: tuple, kv: torch.Tensor,
                       kv_norm_weight: torch.Tensor, eps: float, positions: torch.Tensor, inv_freq: torch.Tensor, mscale: float,
                       num_heads: int, nope_dim: int, rope_dim: int, kv_lora: int):
    """Decode step (T <= 4), one launch (ktx_linear_forward_qb_absorb): q_b_proj(q_a_layernorm(q_a)) per head, the q-absorb
    products of its nope part, RoPE of its rope part, and the kv half of mla_prep.  q_a: bf16 [T, q_lora] (unit inner stride,
    any 16-byte aligned row stride), kv: bf16 [T, kv_lora + rope].  Returns (q_nope [T,H,lora], q_pe [T,H,rope], ckv, k_pe)."""
    T, dev = positions.numel(), positions.device
    _bf16_rows(q_a, "qb_absorb_and_prep q_a")
    _bf16_rows(kv, "qb_absorb_and_prep kv")
    q_a = q_a.reshape(T, -1)
    kv = kv.reshape(T, -1)
    if q_a.stride(1) != 1 or kv.stride(1) != 1 or q_a.shape[1] != q_b.K:
        raise KtxError("qb_absorb_and_prep: q_a must be [T, q_lora] with unit inner stride")
    if positions.dtype != torch.int64 or inv_freq.dtype != torch.float32:
        raise KtxError("qb_absorb_and_prep: positions must be int64 and inv_freq fp32")
    nw, neps = q_a_norm
    q_nope = torch.empty((T, num_heads, kv_lora), dtype=torch.bfloat16, device=dev)
    q_pe = torch.empty((T, num_heads, rope_dim), dtype=torch.bfloat16, device=dev)
    ckv = torch.empty((T, kv_lora), dtype=torch.bfloat16, device=dev)
    kpe = torch.empty((T, rope_dim), dtype=torch.bfloat16, device=dev)
    check(lib.ktx_linear_forward_qb_absorb(q_b._h, qabs._h, T, q_a.data_ptr(), q_a.stride(0), nw.data_ptr(), float(neps), num_heads,
                                           nope_dim, rope_dim, kv_lora, q_nope.data_ptr(), q_pe.data_ptr(), kv.data_ptr(),
                                           kv.stride(0), kv_norm_weight.data_ptr(), float(eps), ckv.data_ptr(), kpe.data_ptr(),
                                           positions.data_ptr(), inv_freq.data_ptr(), float(mscale), _stream_ptr(dev)))
    return q_nope, q_pe, ckv, kpe


ATTN_PHASE_ALL = 31
ATTN_ARRAYS = {"qkv": 0, "ckv_new": 1, "kpe_new": 2, "q_lat": 3, "q_pe": 4, "attn_out": 6, "part_ml": 7, "part_o": 8, "qx": 9}   # (5, the merged rows, no longer pass through the workspace)


def attn_decode_args(qkv_a: "LinearHandle", q_b: "LinearHandle", qabs: "LinearHandle", oabs: "LinearHandle", o_proj: "LinearHandle",
                     x: torch.Tensor, out: torch.Tensor, in_norm: tuple, qa_norm: tuple, kv_norm: tuple, position: torch.Tensor,
                     inv_freq: torch.Tensor, mscale: float, num_heads: int, nope: int, rope: int, kv_lora: int, v_dim: int,
                     ckv_pages: torch.Tensor, kpe_pages: torch.Tensor, page_size: int, kv_indptr: torch.Tensor,
                     kv_indices: torch.Tensor | None, kv_len: torch.Tensor, kv_len_hint: int, sm_scale: float,
                     phases: int = ATTN_PHASE_ALL, last: bool = True) -> _AttnDecodeArgs:
    """Arguments of the one-launch MLA decode step (include/ktx_attn.h).  x / out: bf16 [hidden] rows; the norm tuples are
    (bf16 weight, eps); position int64 [1]; kv_len int32 [1] = context length including the new token; the cache views as in
    MLAWrapper.run.  The caller keeps every tensor alive until the launch is enqueued."""
    for t, what in ((x, "x"), (out, "out"), (in_norm[0], "input norm"), (qa_norm[0], "q_a norm"), (kv_norm[0], "kv_a norm")):
        if t.dtype != torch.bfloat16 or not t.is_contiguous():
            raise KtxError(f"attn_decode: {what} must be contiguous bf16")
    if position.dtype != torch.int64 or inv_freq.dtype != torch.float32 or kv_len.dtype != torch.int32 or kv_indptr.dtype != torch.int32:
        raise KtxError("attn_decode: position int64, inv_freq fp32, kv_len / kv_indptr int32")
    if ckv_pages.stride(-1) != 1 or kpe_pages.stride(-1) != 1 or ckv_pages.dtype != torch.bfloat16:
        raise KtxError("attn_decode: kv tensors must be bf16 with unit inner stride")
    ckv_ts, kpe_ts = ckv_pages.stride(-2), kpe_pages.stride(-2)
    if ckv_pages.stride(0) != ckv_ts * page_size or kpe_pages.stride(0) != kpe_ts * page_size:
        raise KtxError("attn_decode: pages must be contiguous runs of page_size tokens")
    a = _AttnDecodeArgs(qkv_a._h, q_b._h, qabs._h, oabs._h, o_proj._h, num_heads, nope, rope, kv_lora, v_dim, q_b.K, x.numel(),
                        x.data_ptr(), out.data_ptr(), in_norm[0].data_ptr(), float(in_norm[1]), qa_norm[0].data_ptr(),
                        float(qa_norm[1]), kv_norm[0].data_ptr(), float(kv_norm[1]), position.data_ptr(), inv_freq.data_ptr(),
                        float(mscale), ckv_pages.data_ptr(), kpe_pages.data_ptr(), ckv_ts, kpe_ts, page_size, kv_indptr.data_ptr(),
                        kv_indices.data_ptr() if kv_indices is not None else None, kv_len.data_ptr(), int(kv_len_hint),
                        float(sm_scale), int(phases), 1 if last else 0)
    return a


def attn_decode_eligible(args: _AttnDecodeArgs) -> bool:
    return bool(lib.ktx_attn_decode_eligible(C.byref(args)))


def attn_decode(args: _AttnDecodeArgs, device, phases: int | None = None, last: bool | None = None) -> None:
    """Enqueue the launch (or, with `phases`, one launch of a split chain: the caller issues the subsets in phase order and marks
    the final one `last`)."""
    if phases is not None:
        args.phases = int(phases)
    if last is not None:
        args.last = 1 if last else 0
    check(lib.ktx_attn_decode(C.byref(args), _stream_ptr(device)))


def attn_status(device) -> int:
    dev = torch.device(device)
    st = C.c_uint32(0)
    check(lib.ktx_attn_status(dev.index if dev.index is not None else torch.cuda.current_device(), C.byref(st)))
    return int(st.value)


def attn_status_any() -> tuple[int, int]:
    """(device, status) of the first device whose one-launch attention step saw a hand-off give up, (-1, 0) if none.  A host load
    of a pinned word the device writes: cheap enough to call after every decode step."""
    dev, st = C.c_int(-1), C.c_uint32(0)
    check(lib.ktx_attn_status_any(C.byref(dev), C.byref(st)))
    return int(dev.value), int(st.value)


def attn_reset(device) -> None:
    """Synchronise `device`, clear its status word and re-arm the workspaces after a timed-out hand-off."""
    dev = torch.device(device)
    check(lib.ktx_attn_reset(dev.index if dev.index is not None else torch.cuda.current_device()))


def attn_debug_read(device, name: str, shape, dtype=torch.bfloat16) -> torch.Tensor:
    dev = torch.device(device)
    out = torch.empty(shape, dtype=dtype, device=dev)
    torch.cuda.synchronize(dev)
    check(lib.ktx_attn_debug_read(dev.index if dev.index is not None else torch.cuda.current_device(), ATTN_ARRAYS[name], out.data_ptr(),
                                  out.numel() * out.element_size()))
    return out


def linear_force_gemm(on: bool) -> None:
    check(lib.ktx_linear_debug_force_gemm(1 if on else 0))


class GateHandle:
    """Router parameters of one MoE layer + the two-launch HIP router (include/ktx_gate.h)."""

    LOGITS_HIP_MAX_T = 64  # above this the logits come from the prompt-sized MFMA GEMM (csrc/ktx_gemm.hip)

    def __init__(self, n_routed_experts: int, hidden_size: int, top_k: int, n_group: int = 1, topk_group: int = 1,
                 scoring_func: str = "sigmoid", topk_method: str = "noaux_tc", norm_topk_prob: bool = True,
                 routed_scaling_factor: float = 1.0):
        if scoring_func not in GATE_SCORING:
            raise KtxError(f"insupportable scoring function for MoE gating: {scoring_func}")
        if topk_method not in GATE_TOPK:
            raise KtxError(f"insupportable TopK function for MoE gating: {topk_method}")
        self.cfg = _GateConfig(n_routed_experts, hidden_size, top_k, max(1, n_group or 1), max(1, topk_group or 1),
                               GATE_SCORING[scoring_func], GATE_TOPK[topk_method], 1 if norm_topk_prob else 0,
                               float(routed_scaling_factor))
        self.E, self.H, self.k = n_routed_experts, hidden_size, top_k
        self._counters: dict = {}

    def _planes_of(self, weight: torch.Tensor) -> torch.Tensor:
        """bf16 operand(s) of the large-batch logits GEMM: a bf16 weight as it is; an fp32 weight as its three exact bf16 planes
        [3E, H] (split once per weight tensor and version)."""
        if weight.dtype == torch.bfloat16:
            return weight.contiguous()
        try:
            ver = weight._version
        except RuntimeError:      # inference-mode tensors do not track a version: the storage address has to do
            ver = -1
        key = (weight.data_ptr(), ver, weight.device)
        hit = getattr(self, "_planes", None)
        if hit is None or hit[0] != key:
            hit = self._planes = (key, split_f32_bf16x3(weight.to(torch.float32)))
        return hit[1]

    def forward(self, x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None,
                bsz_tensor: torch.Tensor | None = None, norm: tuple | None = None):
        """x bf16 [T,H]; weight [E,H] (bf16 for the HIP GEMV; any float dtype for the library GEMM); bias fp32 [E]|None
        -> (topk_idx int64 [T,k], topk_weight fp32 [T,k]).  norm = (weight bf16 [H], eps): x is the un-normalised hidden
        state, RMSNorm runs inside the router launch and the normalised rows are returned as a third value."""
        T = x.shape[0]
        dev = x.device
        st = _stream_ptr(dev)
        bsz_ptr = bsz_tensor.data_ptr() if bsz_tensor is not None else None
        if norm is not None:
            fusable = (T <= self.LOGITS_HIP_MAX_T and weight.dtype == torch.bfloat16 and x.dtype == torch.bfloat16
                       and self.H <= 8192)
            if not fusable:
                xn = rmsnorm(x, norm[0], norm[1], native_rounding=True)
                return (*self.forward(xn, weight, bias, bsz_tensor), xn)
            logits = torch.empty((T, self.E), dtype=torch.float32, device=dev)
            idx = torch.empty((T, self.k), dtype=torch.int64, device=dev)
            wt = torch.empty((T, self.k), dtype=torch.float32, device=dev)
            xn = torch.empty((T, self.H), dtype=torch.bfloat16, device=dev)
            cnt = self._counters.get(dev)
            if cnt is None:
                cnt = self._counters[dev] = torch.zeros(self.LOGITS_HIP_MAX_T, dtype=torch.int32, device=dev)
            xc, wc = x.contiguous(), weight.contiguous()
            b = bias.to(device=dev, dtype=torch.float32).contiguous() if bias is not None else None
            check(lib.ktx_gate_forward_norm(C.byref(self.cfg), bsz_ptr, T, xc.data_ptr(), norm[0].data_ptr(), float(norm[1]),
                                            xn.data_ptr(), wc.data_ptr(), b.data_ptr() if b is not None else None,
                                            logits.data_ptr(), cnt.data_ptr(), idx.data_ptr(), wt.data_ptr(), st))
            return idx, wt, xn
        if T <= self.LOGITS_HIP_MAX_T and weight.dtype == torch.bfloat16 and x.dtype == torch.bfloat16:
            # one launch: logits GEMV, then the last workgroup of each token selects (include/ktx_gate.h)
            logits = torch.empty((T, self.E), dtype=torch.float32, device=dev)
            idx = torch.empty((T, self.k), dtype=torch.int64, device=dev)
            wt = torch.empty((T, self.k), dtype=torch.float32, device=dev)
            cnt = self._counters.get(dev)
            if cnt is None:
                cnt = self._counters[dev] = torch.zeros(self.LOGITS_HIP_MAX_T, dtype=torch.int32, device=dev)
            xc, wc = x.contiguous(), weight.contiguous()
            b = bias.to(device=dev, dtype=torch.float32).contiguous() if bias is not None else None
            check(lib.ktx_gate_forward(C.byref(self.cfg), bsz_ptr, T, xc.data_ptr(), wc.data_ptr(),
                                       b.data_ptr() if b is not None else None, logits.data_ptr(), cnt.data_ptr(),
                                       idx.data_ptr(), wt.data_ptr(), st))
            return idx, wt
        # large batches: the reference's F.linear(x.float(), weight.float()) (modeling_deepseek_v3.py:434-437).  A bf16 x and an
        # fp32 weight: the weight is split ONCE into three bf16 planes that add up to it exactly, so every product of the three
        # plane GEMMs is exact in the fp32 accumulator and the logits are fp32-GEMM grade on the MFMA units (csrc/ktx_gemm.hip;
        # planes summed smallest first).  Anything else (fp32 activations, odd shapes) keeps the torch expression.
        logits = None
        if (x.dtype == torch.bfloat16 and weight.dim() == 2 and weight.dtype in (torch.bfloat16, torch.float32)
                and weight.shape[1] % 64 == 0 and weight.shape[0] % 4 == 0 and not os.environ.get("KTX_VENDOR_GEMM")):
            planes = self._planes_of(weight)
            xg = x.reshape(T, -1)
            if xg.stride(1) != 1 or xg.stride(0) % 8 or xg.data_ptr() % 16:      # the GEMM reads 16-byte pieces of whole rows
                xg = xg.contiguous()
            # E columns are only 2-6 output tiles wide (a 2048 x 256 problem is 32 workgroups on 256 CUs): the k range is cut into S
            # slices run as the GEMM's batch dimension (strided views, nothing is copied) and the S fp32 partial logits are added in
            # slice order — round 5: 107 -> ~30 us per layer of a 2048-token DeepSeek-V3 chunk.  Every product is still exact in
            # its fp32 accumulator; only the association of the fp32 sum over k changes (the reference leaves it to the vendor GEMM).
            K_ = xg.shape[1]
            tiles = ((T + 127) // 128) * ((planes.shape[0] + 127) // 128)
            S = 1
            if tiles < 128 and not os.environ.get("KTX_GATE_NO_SPLITK"):
                for cand in (8, 7, 16, 14, 4, 2):
                    if (K_ // 64) % cand == 0 and tiles * cand <= 512:
                        S = cand
                        break
            if S > 1:
                ks = K_ // S
                l3 = gemm_bf16_nt(xg.as_strided((S, T, ks), (ks, xg.stride(0), 1)),
                                  planes.as_strided((S, planes.shape[0], ks), (ks, planes.stride(0), 1)), out_f32=True)
                l3 = l3.sum(dim=0)       # one reduction launch; a fixed order for a fixed shape
            else:
                l3 = gemm_bf16_nt(xg, planes, out_f32=True)
            E_ = weight.shape[0]
            logits = l3 if planes.shape[0] == E_ else ((l3[:, 2 * E_:] + l3[:, E_:2 * E_]) + l3[:, :E_]).contiguous()
        if logits is None:
            logits = torch.nn.functional.linear(x.to(torch.float32), weight.to(torch.float32)).contiguous()
        idx = torch.empty((T, self.k), dtype=torch.int64, device=dev)
        w = torch.empty((T, self.k), dtype=torch.float32, device=dev)
        b = None
        if bias is not None:
            b = bias.to(device=dev, dtype=torch.float32).contiguous()
        check(lib.ktx_gate_select(C.byref(self.cfg), bsz_ptr, T, logits.data_ptr(), b.data_ptr() if b is not None else None,
                                  idx.data_ptr(), w.data_ptr(), st))
        return idx, w


def gate_with_linear(gate: "GateHandle", lin: "LinearHandle", x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None,
                     norm: tuple, glu: bool = True, bsz_tensor: torch.Tensor | None = None):
    """Decode step of a MoE block: the router (GateHandle.forward with norm=) and the shared experts' merged gate|up GEMV
    (LinearHandle.forward with norm=, glu=) on the same un-normalised row block x bf16 [T, H], in ONE launch where a combined
    kernel exists (ktx_linear_forward_fused_gate; otherwise the library issues the two launches itself).
    Returns (topk_idx int64 [T,k], topk_weight fp32 [T,k], xn bf16 [T,H], y bf16 [T, N/2 if glu else N])."""
    T, dev = x.shape[0], x.device
    if x.dtype != torch.bfloat16 or x.dim() != 2 or x.shape[1] != gate.H or lin.K != gate.H or not x.is_contiguous():
        raise KtxError("gate_with_linear: x must be contiguous bf16 [T, hidden] and both operators must read hidden-sized rows")
    if weight.dtype != torch.bfloat16 or T > GateHandle.LOGITS_HIP_MAX_T:
        raise KtxError("gate_with_linear: bf16 router weights and a decode-sized batch only")
    nw, eps = norm
    n_out = lin.N // 2 if glu else lin.N
    logits = torch.empty((T, gate.E), dtype=torch.float32, device=dev)
    idx = torch.empty((T, gate.k), dtype=torch.int64, device=dev)
    wt = torch.empty((T, gate.k), dtype=torch.float32, device=dev)
    xn = torch.empty((T, gate.H), dtype=torch.bfloat16, device=dev)
    y = torch.empty((T, n_out), dtype=torch.bfloat16, device=dev) if bsz_tensor is None else \
        torch.zeros((T, n_out), dtype=torch.bfloat16, device=dev)
    cnt = gate._counters.get(dev)
    if cnt is None:
        cnt = gate._counters[dev] = torch.zeros(GateHandle.LOGITS_HIP_MAX_T, dtype=torch.int32, device=dev)
    wc = weight.contiguous()
    b = bias.to(device=dev, dtype=torch.float32).contiguous() if bias is not None else None
    fu = _LinearFusion(nw.data_ptr(), float(eps), None, 0, None, 0, 0, 0, 1 if glu else 0)
    check(lib.ktx_linear_forward_fused_gate(lin._h, bsz_tensor.data_ptr() if bsz_tensor is not None else None, T, x.data_ptr(),
                                            y.data_ptr(), C.byref(fu), C.byref(gate.cfg), wc.data_ptr(),
                                            b.data_ptr() if b is not None else None, logits.data_ptr(), cnt.data_ptr(),
                                            idx.data_ptr(), wt.data_ptr(), xn.data_ptr(), _stream_ptr(dev)))
    return idx, wt, xn, y


class MLAWrapper:
    """plan()/run() facade with the reference's MLAWrapper signature (archive/ktransformers/operators/
    flashinfer_wrapper.py:78-161) over ktx_mla_decode.  plan() only records the (device) index arrays: there is no host
    planning step, so a captured graph stays valid when their contents change."""

    MAX_GRAPH_HEADS = 128   # a graph-captured wrapper's workspace is sized for this many heads up front (DeepSeek-V3 / R1: 128, K2: 64)

    def __init__(self, max_batch_size: int, max_pages: int, use_cuda_graph: bool = True, device="cuda",
                 max_q_tokens: int | None = None, max_splits: int = 256):
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        if self.device.type != "cuda":
            raise KtxError("MLAWrapper needs a HIP device; there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.max_batch_size, self.max_pages, self.max_splits = max_batch_size, max_pages, max_splits
        self.max_q_tokens = max_q_tokens or max(256, max_batch_size)
        self.use_cuda_graph = bool(use_cuda_graph)
        self.cfg = None
        self.workspace = None
        if max_batch_size == 1:  # single-request defaults like the reference wrapper (:88-93)
            self.qo_indptr_buf = torch.arange(0, 2, dtype=torch.int32, device=self.device)
            self.kv_indptr_buf = torch.tensor([0, max_pages], dtype=torch.int32, device=self.device)
            self.kv_indices_buf = torch.arange(0, max_pages, dtype=torch.int32, device=self.device)
            self.batch_size_tensor_buf = torch.tensor([1], dtype=torch.int32, device=self.device)
        self.need_plan = True

    def plan(self, qo_indptr, kv_indptr, kv_indices, kv_len_arr, bsz_tensor, num_heads, head_dim_ckv, head_dim_kpe,
             page_size, sm_scale, q_data_type=torch.bfloat16, kv_data_type=torch.bfloat16, max_kv_len: int = 0,
             identity_pages: bool = False):
        """identity_pages: the page table is the identity (request r owns pages kv_indptr[r] .. kv_indptr[r+1]-1 in order, as
        in the single-request StaticCache): the kernel then skips the page-table load (include/ktx_mla.h)."""
        if q_data_type != torch.bfloat16 or kv_data_type != torch.bfloat16:
            raise KtxError("MLAWrapper: bf16 q/kv only")
        self.qo_indptr = qo_indptr if qo_indptr is not None else self.qo_indptr_buf
        self.kv_indptr = kv_indptr if kv_indptr is not None else self.kv_indptr_buf
        self.kv_indices = None if identity_pages else (kv_indices if kv_indices is not None else self.kv_indices_buf)
        self.bsz_tensor = bsz_tensor if bsz_tensor is not None else getattr(self, "batch_size_tensor_buf", None)
        self.kv_len_arr = kv_len_arr
        self.cfg = _MlaConfig(num_heads, head_dim_ckv, head_dim_kpe, page_size, float(sm_scale), self.max_splits,
                              int(max_kv_len))
        need = int(lib.ktx_mla_workspace_bytes(C.byref(self.cfg), self.max_q_tokens))
        if self.workspace is None:
            first = need
            if self.use_cuda_graph and self.max_q_tokens <= 4:   # the shared decode wrapper: allocated ONCE, for the largest head count a model on this device may bring (graphs keep its address)
                big = _MlaConfig(max(num_heads, self.MAX_GRAPH_HEADS), head_dim_ckv, head_dim_kpe, page_size, float(sm_scale),
                                 self.max_splits, int(max_kv_len))
                first = max(need, int(lib.ktx_mla_workspace_bytes(C.byref(big), self.max_q_tokens)))
            self.workspace = torch.empty(first, dtype=torch.uint8, device=self.device)
        elif self.workspace.numel() < need:
            # captured HIP graphs hold this buffer's address (operators/attention.py shares one decode wrapper per device): a
            # later plan that needs more (more heads: a second model on the device) must not free it under them
            if self.use_cuda_graph:
                raise KtxError(f"MLAWrapper.plan: the workspace ({self.workspace.numel()} B) is too small for this configuration "
                               f"({need} B) and cannot grow: captured graphs reference it; create another wrapper")
            self.workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        self.need_plan = False

    def run(self, q_nope, q_pe, ckv, k_pe, return_lse: bool = False, new_ckv=None, new_kpe=None, partials: bool = False):
        """new_ckv [batch,512] / new_kpe [batch,64]: fuse StaticCache.update of the current decode token into the launch
        (the kernel reads position kv_len-1 from these buffers and stores it into the cache pages)."""
        if self.cfg is None:
            raise KtxError("MLAWrapper.run before plan()")
        T, Hq, dc = q_nope.shape
        if T > self.max_q_tokens:
            raise KtxError(f"MLAWrapper.run: {T} query tokens exceed max_q_tokens={self.max_q_tokens}")
        for t in (q_nope, q_pe):
            if t.dtype != torch.bfloat16 or not t.is_contiguous():
                raise KtxError("MLAWrapper.run: q tensors must be contiguous bf16")
        if ckv.stride(-1) != 1 or k_pe.stride(-1) != 1 or ckv.dtype != torch.bfloat16:
            raise KtxError("MLAWrapper.run: kv tensors must be bf16 with unit inner stride")
        ckv_ts, kpe_ts = ckv.stride(-2), k_pe.stride(-2)
        if ckv.stride(0) != ckv_ts * self.cfg.page_size or k_pe.stride(0) != kpe_ts * self.cfg.page_size:
            raise KtxError("MLAWrapper.run: pages must be contiguous runs of page_size tokens")
        out = torch.empty((T, Hq, dc), dtype=torch.bfloat16, device=q_nope.device)
        lse = torch.empty((T, Hq), dtype=torch.float32, device=q_nope.device) if return_lse else None
        batch = self.qo_indptr.numel() - 1
        if (new_ckv is None) != (new_kpe is None):
            raise KtxError("MLAWrapper.run: pass both new_ckv and new_kpe or neither")
        if new_ckv is not None and (new_ckv.dtype != torch.bfloat16 or not new_ckv.is_contiguous() or not new_kpe.is_contiguous()):
            raise KtxError("MLAWrapper.run: new_ckv/new_kpe must be contiguous bf16")
        if partials:
            ns = C.c_int(0)
            check(lib.ktx_mla_decode_partials(C.byref(self.cfg), q_nope.data_ptr(), q_pe.data_ptr(), ckv.data_ptr(), k_pe.data_ptr(),
                                              ckv_ts, kpe_ts, self.qo_indptr.data_ptr(), self.kv_indptr.data_ptr(),
                                              self.kv_indices.data_ptr() if self.kv_indices is not None else None,
                                              self.kv_len_arr.data_ptr(),
                                              self.bsz_tensor.data_ptr() if self.bsz_tensor is not None else None, batch, T,
                                              new_ckv.data_ptr() if new_ckv is not None else None,
                                              new_kpe.data_ptr() if new_kpe is not None else None, self.workspace.data_ptr(),
                                              self.workspace.numel(), C.byref(ns), _stream_ptr(q_nope.device)))
            return self.workspace, int(ns.value)
        check(lib.ktx_mla_decode_append(C.byref(self.cfg), q_nope.data_ptr(), q_pe.data_ptr(), ckv.data_ptr(), k_pe.data_ptr(),
                                        ckv_ts, kpe_ts, self.qo_indptr.data_ptr(), self.kv_indptr.data_ptr(),
                                        self.kv_indices.data_ptr() if self.kv_indices is not None else None, self.kv_len_arr.data_ptr(),
                                        self.bsz_tensor.data_ptr() if self.bsz_tensor is not None else None, batch, T,
                                        new_ckv.data_ptr() if new_ckv is not None else None,
                                        new_kpe.data_ptr() if new_kpe is not None else None,
                                        out.data_ptr(), lse.data_ptr() if lse is not None else None, self.workspace.data_ptr(),
                                        self.workspace.numel(), _stream_ptr(q_nope.device)))
        return (out, lse) if return_lse else out

    def run_partials(self, q_nope, q_pe, ckv, k_pe, new_ckv=None, new_kpe=None):
        """run() without its merge launch (ktx_mla_decode_partials): returns (workspace, nsplit) for merge_and_unabsorb, which
        folds the merge into the un-absorb products of a decode step.  The workspace is this wrapper's: consume it before the
        next run on the same wrapper (the layers of a model run serially on one stream)."""
        return self.run(q_nope, q_pe, ckv, k_pe, new_ckv=new_ckv, new_kpe=new_kpe, partials=True)


def mla_cache_append(kv_cache: torch.Tensor, ckv_new: torch.Tensor, kpe_new: torch.Tensor, page_idx: torch.Tensor,
                     page_offset: torch.Tensor, ntokens: torch.Tensor | None = None) -> None:
    """kv_cache bf16 [pages, page_size, (1,) 576]; scatter T rows (StaticCache.update, custom_cache.py:189-195)."""
    page_size = kv_cache.shape[1]
    ts = kv_cache.stride(1)
    cfg = _MlaConfig(16, 512, 64, page_size, 1.0, 1, 0)
    T = ckv_new.shape[0]
    # keep the converted temporaries alive until the launch is enqueued (stream-ordered allocator reuse is then safe)
    c, r = ckv_new.contiguous(), kpe_new.contiguous()
    pi, po = page_idx.to(torch.int32).contiguous(), page_offset.to(torch.int32).contiguous()
    check(lib.ktx_mla_cache_append(C.byref(cfg), kv_cache.data_ptr(), ts, c.data_ptr(), r.data_ptr(), pi.data_ptr(),
                                   po.data_ptr(), ntokens.data_ptr() if ntokens is not None else None, T,
                                   int(kv_cache.shape[0]), _stream_ptr(kv_cache.device)))


def mla_prefill(q_nope: torch.Tensor, q_pe: torch.Tensor, k_nope: torch.Tensor, k_pe: torch.Tensor, v_t: torch.Tensor,
                kv_len: int, sm_scale: float) -> torch.Tensor:
    """Causal non-absorbed prompt attention (include/ktx_mla.h, ktx_mla_prefill).  q_nope [T,H,128] / q_pe [T,H,64] (strided
    views allowed, unit inner stride), k_nope [H,kv_pad,128] contiguous, k_pe [kv_len(+),64] rows with a token stride,
    v_t [H,128,kv_pad] contiguous -> bf16 [T,H,128]; the T queries are the last T of the kv_len keys."""
    T, H, _ = q_nope.shape
    kv_pad = k_nope.shape[1]
    for t, what in ((q_nope, "q_nope"), (q_pe, "q_pe"), (k_nope, "k_nope"), (k_pe, "k_pe"), (v_t, "v_t")):
        _bf16_rows(t, f"mla_prefill {what}")
    if q_nope.shape[2] != 128 or q_pe.shape[2] != 64 or k_nope.shape != (H, kv_pad, 128) or v_t.shape != (H, 128, kv_pad) \
            or not k_nope.is_contiguous() or not v_t.is_contiguous() or k_pe.shape[-1] != 64 or k_pe.dim() != 2:
        raise KtxError("mla_prefill: bad operand shapes")
    out = torch.empty((T, H, 128), dtype=torch.bfloat16, device=q_nope.device)
    check(lib.ktx_mla_prefill(T, H, int(kv_len), kv_pad, float(sm_scale), q_nope.data_ptr(), q_nope.stride(0), q_nope.stride(1),
                              q_pe.data_ptr(), q_pe.stride(0), q_pe.stride(1), k_nope.data_ptr(), k_pe.data_ptr(), k_pe.stride(0),
                              v_t.data_ptr(), out.data_ptr(), _stream_ptr(q_nope.device)))
    return out


# ---- small fused ops (include/ktx_ops.h) ---------------------------------------------------------------------------
def _bf16_rows(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dtype != torch.bfloat16 or not t.is_cuda or t.stride(-1) != 1:
        raise KtxError(f"{what}: expected a bf16 device tensor with unit inner stride, got {t.dtype} on {t.device}")
    return t


def gemm_bf16_nt(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor | None = None, bias: torch.Tensor | None = None,
                 out_f32: bool = False, variant: int = 0) -> torch.Tensor:
    """The library's own prompt-sized GEMM (include/ktx_gemm.h): out[..., m, n] = sum_k a[..., m, k] * b[..., n, k] (+ bias[n]),
    bf16 operands, fp32 accumulation, bf16 (or fp32) result.  a: [M, K] or [batch, M, K]; b: [N, K] or [batch, N, K]; a 2-D
    operand is shared by every batch entry of a 3-D one.  Rows must be k-contiguous; row / batch strides may be anything
    that keeps 16-byte alignment."""
    if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16 or a.device != b.device or a.device.type != "cuda":
        raise KtxError(f"gemm_bf16_nt: bf16 operands on one HIP device expected, got {a.dtype} {a.device} / {b.dtype} {b.device}")
    if a.dim() not in (2, 3) or b.dim() not in (2, 3) or a.shape[-1] != b.shape[-1]:
        raise KtxError(f"gemm_bf16_nt: shapes {tuple(a.shape)} x {tuple(b.shape)}^T do not multiply")
    if a.stride(-1) != 1:
        a = a.contiguous()
    if b.stride(-1) != 1:
        b = b.contiguous()
    batch = a.shape[0] if a.dim() == 3 else (b.shape[0] if b.dim() == 3 else 1)
    if (a.dim() == 3 and a.shape[0] != batch) or (b.dim() == 3 and b.shape[0] != batch):
        raise KtxError("gemm_bf16_nt: batch sizes differ")
    M, N, K = a.shape[-2], b.shape[-2], a.shape[-1]
    odt = torch.float32 if out_f32 else torch.bfloat16
    batched = a.dim() == 3 or b.dim() == 3
    if out is None:
        out = torch.empty((batch, M, N) if batched else (M, N), dtype=odt, device=a.device)
    elif out.dtype != odt or out.device != a.device or tuple(out.shape) != ((batch, M, N) if batched else (M, N)) or out.stride(-1) != 1:
        raise KtxError(f"gemm_bf16_nt: out must be {odt} {(batch, M, N) if batched else (M, N)} with unit inner stride")
    if bias is not None and (bias.dtype != torch.bfloat16 or tuple(bias.shape) != (N,) or bias.device != a.device or not bias.is_contiguous()):
        raise KtxError(f"gemm_bf16_nt: bias must be contiguous bf16 [{N}] on {a.device}")
    args = _GemmArgs(M, N, K, batch, a.data_ptr(), a.stride(-2), a.stride(0) if a.dim() == 3 else 0,
                     b.data_ptr(), b.stride(-2), b.stride(0) if b.dim() == 3 else 0,
                     out.data_ptr(), out.stride(-2), out.stride(0) if batched else 0,
                     bias.data_ptr() if bias is not None else None, 1 if out_f32 else 0, int(variant))
    check(lib.ktx_gemm_bf16_nt(C.byref(args), _stream_ptr(a.device)))
    return out


def split_f32_bf16x3(w: torch.Tensor) -> torch.Tensor:
    """fp32 [rows, K] -> bf16 [3 * rows, K]: planes hi | mid | lo with w == hi + mid + lo exactly (include/ktx_gemm.h)."""
    if w.dtype != torch.float32 or w.dim() != 2 or w.device.type != "cuda":
        raise KtxError("split_f32_bf16x3: fp32 [rows, K] on a HIP device expected")
    w = w.contiguous()
    out = torch.empty((3 * w.shape[0], w.shape[1]), dtype=torch.bfloat16, device=w.device)
    check(lib.ktx_split_f32_bf16x3(w.data_ptr(), w.numel(), out.data_ptr(), _stream_ptr(w.device)))
    return out


def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, native_rounding: bool = True,
            bsz_tensor: torch.Tensor | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """RMSNorm over the last dim of a bf16 [..., dim] tensor (layernorm.py:79-87 when native_rounding)."""
    _bf16_rows(x, "rmsnorm")
    dim = x.shape[-1]
    x2 = x.reshape(-1, dim)
    if out is None:
        out = torch.empty((x2.shape[0], dim), dtype=torch.bfloat16, device=x.device)
    o2 = out.reshape(-1, dim)
    check(lib.ktx_rmsnorm(x2.data_ptr(), x2.stride(0), weight.data_ptr(), o2.data_ptr(), o2.stride(0), x2.shape[0], dim,
                          float(eps), 1 if native_rounding else 0, bsz_tensor.data_ptr() if bsz_tensor is not None else None,
                          _stream_ptr(x.device)))
    return out.reshape(x.shape)


def fused_add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                      bsz_tensor: torch.Tensor | None = None) -> None:
    """In place: residual += x; x = rmsnorm(residual) (flashinfer.norm.fused_add_rmsnorm, layernorm.py:69)."""
    _bf16_rows(x, "fused_add_rmsnorm")
    dim = x.shape[-1]
    if not (x.is_contiguous() and residual.is_contiguous()) or residual.shape != x.shape or residual.dtype != torch.bfloat16:
        raise KtxError("fused_add_rmsnorm: x and residual must be contiguous bf16 tensors of the same shape")
    check(lib.ktx_fused_add_rmsnorm(x.data_ptr(), residual.data_ptr(), weight.data_ptr(), x.numel() // dim, dim, float(eps),
                                    bsz_tensor.data_ptr() if bsz_tensor is not None else None, _stream_ptr(x.device)))


def silu_mul(gate_up: torch.Tensor, bsz_tensor: torch.Tensor | None = None) -> torch.Tensor:
    """gate_up: bf16 [T, 2*I] = [gate | up] -> bf16 [T, I] = silu(gate) * up."""
    _bf16_rows(gate_up, "silu_mul")
    g2 = gate_up.reshape(-1, gate_up.shape[-1])
    inter = g2.shape[1] // 2
    out = torch.empty((g2.shape[0], inter), dtype=torch.bfloat16, device=g2.device)
    check(lib.ktx_silu_mul(g2.data_ptr(), g2.stride(0), out.data_ptr(), g2.shape[0], inter,
                           bsz_tensor.data_ptr() if bsz_tensor is not None else None, _stream_ptr(g2.device)))
    return out.reshape(*gate_up.shape[:-1], inter)


_ARGMAX_WS: dict = {}


def argmax_bf16(logits: torch.Tensor) -> torch.Tensor:
    """Greedy sampling in one launch: logits bf16 [..., n] (unit inner stride) -> int64 [...] index of each row's first maximum
    (ktx_argmax_bf16).  The workspace (arrival counters, zeroed once) is kept per device and row count."""
    _bf16_rows(logits, "argmax_bf16")
    x = logits.reshape(-1, logits.shape[-1])
    rows, n = x.shape
    if x.stride(1) != 1 or (rows > 1 and x.stride(0) % 8 != 0) or x.data_ptr() % 16 != 0:
        if rows > 1 and n % 8 != 0:                       # pad the rows to 16-byte boundaries
            xp = torch.full((rows, (n + 7) // 8 * 8), float("-inf"), dtype=torch.bfloat16, device=x.device)
            xp[:, :n] = x
            x = xp
        else:
            x = x.contiguous()
    key = (x.device, rows)
    ws = _ARGMAX_WS.get(key)
    if ws is None:
        ws = _ARGMAX_WS[key] = torch.zeros(int(lib.ktx_argmax_workspace_bytes(rows)), dtype=torch.uint8, device=x.device)
    out = torch.empty((rows,), dtype=torch.int64, device=x.device)
    check(lib.ktx_argmax_bf16(x.data_ptr(), x.stride(0), rows, n, out.data_ptr(), ws.data_ptr(), _stream_ptr(x.device)))
    return out.reshape(logits.shape[:-1])


def mla_prep(q: torch.Tensor | None, kv: torch.Tensor | None, kv_norm_weight: torch.Tensor | None, eps: float,
             positions: torch.Tensor, inv_freq: torch.Tensor, mscale: float, num_heads: int, nope_dim: int, rope_dim: int,
             kv_lora: int):
    """q: bf16 [T, num_heads*(nope+rope)] (or None), kv: bf16 [T, kv_lora+rope] (or None), positions int64 [T].
    Returns (q_pe [T,H,rope] | None, ckv [T,kv_lora] | None, k_pe [T,rope] | None)."""
    T = positions.numel()
    dev = positions.device
    if positions.dtype != torch.int64 or inv_freq.dtype != torch.float32:
        raise KtxError("mla_prep: positions must be int64 and inv_freq fp32")
    q_pe = ckv = kpe = None
    if q is not None:
        _bf16_rows(q, "mla_prep q")
        q = q.reshape(T, -1)
        q_pe = torch.empty((T, num_heads, rope_dim), dtype=torch.bfloat16, device=dev)
    if kv is not None:
        _bf16_rows(kv, "mla_prep kv")
        kv = kv.reshape(T, -1)
        ckv = torch.empty((T, kv_lora), dtype=torch.bfloat16, device=dev)
        kpe = torch.empty((T, rope_dim), dtype=torch.bfloat16, device=dev)
    check(lib.ktx_mla_prep(T, num_heads, nope_dim, rope_dim, kv_lora, q.data_ptr() if q is not None else None,
                           q.stride(0) if q is not None else 0, q_pe.data_ptr() if q is not None else None,
                           kv.data_ptr() if kv is not None else None, kv.stride(0) if kv is not None else 0,
                           kv_norm_weight.data_ptr() if kv is not None else None, float(eps),
                           ckv.data_ptr() if kv is not None else None, kpe.data_ptr() if kv is not None else None,
                           positions.data_ptr(), inv_freq.data_ptr(), float(mscale), _stream_ptr(dev)))
    return q_pe, ckv, kpe
