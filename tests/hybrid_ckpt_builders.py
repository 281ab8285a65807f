"""A tiny DeepSeek-V3-style source pair — an fp8 safetensors checkpoint (HF names, e4m3 weights + weight_scale_inv, per-expert
tensors) and a GGUF file holding the same model's experts as Q4_K / Q6_K blocks — plus the HYBRID directory the reference's
archive/merge_tensors/merge_safetensor_gguf.py makes of the two (all tensors under GGUF names; experts as raw ggml blocks +
ggml_type).  The golden maker builds the hybrid directory with the REFERENCE's own script and checks that `write_hybrid`
produces the same files; the test only needs `write_hybrid`."""
import os

import numpy as np
import torch
from safetensors.torch import save_file

from helpers import write_gguf
from oracle.gguf_ref import quantize_q4_k, quantize_q6_k

E, H, I, HEADS = 4, 256, 256, 2


def _fp8(shape, g):
    return (torch.randn(shape, generator=g) * 0.5).to(torch.float8_e4m3fn)


def source_tensors():
    """HF-named fp8 checkpoint of 2 layers (0 dense, 1 MoE)."""
    g = torch.Generator().manual_seed(4)
    t = {"model.embed_tokens.weight": torch.randn(32, H, generator=g).to(torch.bfloat16),
         "model.norm.weight": torch.randn(H, generator=g).to(torch.bfloat16),
         "lm_head.weight": torch.randn(32, H, generator=g).to(torch.bfloat16)}
    for L in (0, 1):
        p = f"model.layers.{L}."
        t[p + "input_layernorm.weight"] = torch.randn(H, generator=g).to(torch.bfloat16)
        t[p + "post_attention_layernorm.weight"] = torch.randn(H, generator=g).to(torch.bfloat16)
        for name, (n, k) in (("self_attn.q_a_proj", (128, H)), ("self_attn.kv_a_proj_with_mqa", (128, H)),
                             ("self_attn.o_proj", (H, 256))):
            t[p + name + ".weight"] = _fp8((n, k), g)
            t[p + name + ".weight_scale_inv"] = (torch.rand(((n + 127) // 128, (k + 127) // 128), generator=g) + 0.5).float()
        t[p + "self_attn.kv_b_proj.weight"] = torch.randn(128, 128, generator=g).to(torch.bfloat16)
    for proj, (n, k) in (("gate_proj", (I, H)), ("up_proj", (I, H)), ("down_proj", (H, I))):
        t[f"model.layers.0.mlp.{proj}.weight"] = _fp8((n, k), g)
        t[f"model.layers.0.mlp.{proj}.weight_scale_inv"] = (torch.rand((n // 128, k // 128), generator=g) + 0.5).float()
        t[f"model.layers.1.mlp.shared_experts.{proj}.weight"] = _fp8((n, k), g)
        t[f"model.layers.1.mlp.shared_experts.{proj}.weight_scale_inv"] = (torch.rand((n // 128, k // 128), generator=g) + 0.5).float()
        for e in range(E):
            t[f"model.layers.1.mlp.experts.{e}.{proj}.weight"] = _fp8((n, k), g)
            t[f"model.layers.1.mlp.experts.{e}.{proj}.weight_scale_inv"] = (torch.rand((n // 128, k // 128), generator=g) + 0.5).float()
    t["model.layers.1.mlp.gate.weight"] = torch.randn(E, H, generator=g).to(torch.bfloat16)
    t["model.layers.1.mlp.gate.e_score_correction_bias"] = torch.randn(E, generator=g).float()
    return t


def gguf_experts():
    rng = np.random.default_rng(9)
    return {"gate": quantize_q4_k((rng.standard_normal((E, I, H)) / 10).astype(np.float32)),
            "up": quantize_q4_k((rng.standard_normal((E, I, H)) / 10).astype(np.float32)),
            "down": quantize_q6_k((rng.standard_normal((E, H, I)) / 10).astype(np.float32))}


def write_sources(st_dir, gguf_dir):
    os.makedirs(st_dir, exist_ok=True)
    os.makedirs(gguf_dir, exist_ok=True)
    save_file({k: v.contiguous() for k, v in source_tensors().items()}, os.path.join(st_dir, "model-00001-of-00001.safetensors"))
    ex = gguf_experts()
    write_gguf(os.path.join(gguf_dir, "toy.gguf"), {
        "blk.1.ffn_gate_exps.weight": (12, [H, I, E], ex["gate"].tobytes()),
        "blk.1.ffn_up_exps.weight": (12, [H, I, E], ex["up"].tobytes()),
        "blk.1.ffn_down_exps.weight": (14, [I, H, E], ex["down"].tobytes())}, {"deepseek2.expert_count": E})


def write_hybrid(out_dir):
    """What merge_safetensor_gguf.py writes for the sources above: one shard for the non-layer tensors, one per layer."""
    from ktransformers_amd.util.gguf_loader import translate_name_to_gguf

    def tr(name):
        name = translate_name_to_gguf(name)
        for a, b in ((".up_proj.", ".ffn_up_exps."), (".down_proj.", ".ffn_down_exps."), (".gate_proj.", ".ffn_gate_exps."),
                     (".ffn_gate_inp.e_score_correction_bias", ".exp_probs_b.bias")):
            name = name.replace(a, b)
        return name

    os.makedirs(out_dir, exist_ok=True)
    src, ex = source_tensors(), gguf_experts()
    shards = {0: {}, 1: {}, 2: {}}
    for k, v in src.items():
        if ".mlp.experts." in k:
            continue
        shard = 0 if ".layers." not in k else 1 + int(k.split(".")[2])
        shards[shard][tr(k)] = v
    for proj, ty in (("gate", 12), ("up", 12), ("down", 14)):
        shards[2][f"blk.1.ffn_{proj}_exps.weight"] = torch.from_numpy(np.frombuffer(ex[proj].tobytes(), dtype=np.uint8).copy())
        shards[2][f"blk.1.ffn_{proj}_exps.ggml_type"] = torch.tensor(ty)
    for i, t in shards.items():
        save_file({k: v.contiguous() for k, v in t.items()}, os.path.join(out_dir, f"model-{i:05}-of-00002.safetensors"))


QUERIES = ["model.embed_tokens.weight", "lm_head.weight", "model.norm.weight", "model.layers.0.input_layernorm.weight",
           "model.layers.1.self_attn.q_a_proj.weight", "model.layers.1.self_attn.q_a_proj.weight_scale_inv",
           "model.layers.1.self_attn.kv_b_proj.weight", "model.layers.0.mlp.down_proj.weight",
           "model.layers.1.mlp.shared_experts.up_proj.weight", "model.layers.1.mlp.shared_experts.up_proj.weight_scale_inv",
           "model.layers.1.mlp.gate.weight", "blk.1.attn_q_a.weight", "model.layers.1.mlp.experts.0.up_proj.weight",
           "model.layers.7.mlp.gate.weight"]


def digest(t):
    import hashlib
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(t))
    t = t.contiguous()
    raw = t.view(torch.uint8) if t.dim() else t.reshape(1).view(torch.uint8)
    return {"shape": list(t.shape), "dtype": str(t.dtype), "sha": hashlib.sha256(raw.numpy().tobytes()).hexdigest()[:16]}


def probe(loader):
    """What a loader (the reference's SafeTensorLoader or ours) answers about a hybrid directory."""
    out = {"has": {q: bool(loader.has_tensor(q)) for q in QUERIES}, "tensors": {}}
    for q in QUERIES:
        if out["has"][q]:
            out["tensors"][q] = digest(loader.load_tensor(q))
    ex = loader.load_experts("model.layers.1.mlp.experts")
    out["experts"] = {k: (int(v) if k.endswith("_type") else digest(v)) for k, v in sorted(ex.items())}
    gate = loader.load_gate("model.layers.1.mlp.gate")
    out["gate"] = {k: (None if v is None else digest(v)) for k, v in sorted(gate.items())}
    return out
