"""Generates tests/golden/hybrid_loader_golden.json with the REFERENCE's own code: the hybrid (fp8 linears + GGUF experts)
directory is written by archive/merge_tensors/merge_safetensor_gguf.py (combine_tensor_sources + write_combined_tensor) from
the toy sources of tests/hybrid_ckpt_builders.py and read back by the reference's SafeTensorLoader
(archive/ktransformers/util/custom_loader.py:44-275).  Recorded: the files' tensor inventory (names, shapes, dtypes, hashes) —
tests/hybrid_ckpt_builders.write_hybrid must reproduce it — and the loader's answers (has_tensor / load_tensor / load_experts /
load_gate).  Run in the build container only."""
import contextlib
import importlib.util
import io
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

import hybrid_ckpt_builders as B  # noqa: E402
from ktransformers.util.custom_loader import SafeTensorLoader  # noqa: E402
from safetensors import safe_open  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_merge", "/root/reference/archive/merge_tensors/merge_safetensor_gguf.py")
merge = importlib.util.module_from_spec(spec)
spec.loader.exec_module(merge)


def inventory(folder):
    inv = {}
    for f in sorted(os.listdir(folder)):
        with safe_open(os.path.join(folder, f), framework="pt") as h:
            for k in h.keys():
                inv[k] = B.digest(h.get_tensor(k))
    return inv


with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(io.StringIO()):
    st, gg, out_ref, out_ours = (os.path.join(d, x) for x in ("st", "gguf", "hybrid_ref", "hybrid_ours"))
    B.write_sources(st, gg)
    tmap, gl = merge.combine_tensor_sources(st, gg)
    merge.write_combined_tensor(tmap, out_ref, gl)
    B.write_hybrid(out_ours)
    inv_ref, inv_ours = inventory(out_ref), inventory(out_ours)
    assert inv_ref == inv_ours, {k: (inv_ref.get(k), inv_ours.get(k)) for k in set(inv_ref) | set(inv_ours) if inv_ref.get(k) != inv_ours.get(k)}
    gold = {"inventory": inv_ref, "probe": B.probe(SafeTensorLoader(out_ref))}
json.dump(gold, open(os.path.join(HERE, "hybrid_loader_golden.json"), "w"), indent=1, sort_keys=True)
print(len(gold["inventory"]), "tensors;", gold["probe"]["has"], gold["probe"]["experts"])
