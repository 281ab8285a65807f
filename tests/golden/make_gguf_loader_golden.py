"""Read the toy GGUF file of tests/test_gguf_loader_cpu.py with the REFERENCE's own GGUFLoader
(archive/ktransformers/util/custom_loader.py:278-526, CUDA-only extension stubbed) and record what it reports.

    python tests/golden/make_gguf_loader_golden.py
"""
import hashlib
import json
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, "/root/reference/archive")
sys.modules["KTransformersOps"] = types.ModuleType("KTransformersOps")

from ktransformers.util.custom_gguf import translate_name_to_gguf  # noqa: E402
from ktransformers.util.custom_loader import GGUFLoader  # noqa: E402
from test_gguf_loader_cpu import build_file  # noqa: E402

with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "toy.gguf")
    build_file(path)
    ld = GGUFLoader(d)
    out = {"sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(),
           "tensors": {n: {"ggml_type": int(t["ggml_type"]), "shape": [int(x) for x in t["shape"]], "offset": int(t["offset"])}
                       for n, t in ld.tensor_info.items()},
           "meta": {"deepseek2.expert_count": ld.gguf_file_meta["deepseek2.expert_count"]},
           "names": {hf: translate_name_to_gguf(hf) for hf in (
               "model.layers.1.input_layernorm.weight", "model.layers.3.self_attn.kv_a_proj_with_mqa.weight",
               "model.layers.3.self_attn.q_b_proj.weight", "model.layers.5.mlp.experts", "model.layers.5.mlp.gate.weight",
               "model.layers.5.mlp.gate.e_score_correction_bias", "model.layers.5.mlp.shared_experts.up_proj.weight",
               "model.layers.0.mlp.down_proj.weight", "lm_head.weight", "model.embed_tokens.weight", "model.norm.weight",
               "model.layers.2.block_sparse_moe.experts.3.w1.weight", "model.layers.2.block_sparse_moe.gate.weight")}}
    import numpy as np
    import torch
    v = ld.load_gguf_tensor("blk.1.ffn_down_exps.weight", device="cpu", target_dtype=torch.float32)
    out["down_sum"] = float(np.float64(v.double().sum()))
json.dump(out, open(os.path.join(HERE, "gguf_loader_golden.json"), "w"), indent=1)
print(out["sha256"], len(out["tensors"]), out["names"])
