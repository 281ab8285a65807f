"""llama-architecture GGUF files (Mixtral's included) hold attn_q / attn_k with each head's rows interleaved; the REFERENCE's
GGUFLoader.load_gguf_tensor undoes it (archive/ktransformers/util/custom_loader.py:507-517).  This reads the toy file of
tests/test_gguf_loader_cpu.py::build_llama_file with the reference loader and records the row order it returns.

    python tests/golden/make_gguf_llama_golden.py
"""
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

import torch  # noqa: E402
from ktransformers.util.custom_loader import GGUFLoader  # noqa: E402
from test_gguf_loader_cpu import build_llama_file  # noqa: E402

with tempfile.TemporaryDirectory() as d:
    build_llama_file(os.path.join(d, "toy.gguf"))
    ld = GGUFLoader(d)
    out = {}
    for name in ("blk.0.attn_q.weight", "blk.0.attn_k.weight", "blk.0.attn_v.weight"):
        v = ld.load_gguf_tensor(name, device="cpu", target_dtype=torch.float32)
        out[name] = {"shape": list(v.shape), "row_ids": [int(x) for x in v[:, 0].tolist()]}   # column 0 of row r holds r
json.dump(out, open(os.path.join(HERE, "gguf_llama_golden.json"), "w"))
print({k: v["row_ids"][:12] for k, v in out.items()})
