"""Generates tests/golden/kt_loader_golden.json: what the REFERENCE's kt_kernel loaders
(/root/reference/kt-kernel/python/utils/loader.py, imported here; the absent pip package `gguf`, needed only by its GGUFLoader,
is stubbed) return for the synthetic checkpoints of tests/kt_ckpt_builders.py.  Run in the build container only."""
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
import kt_ckpt_builders as B  # noqa: E402

stub = types.ModuleType("gguf.gguf_reader")
stub.GGUFReader = object
sys.modules.setdefault("gguf", types.ModuleType("gguf"))
sys.modules["gguf.gguf_reader"] = stub
spec = importlib.util.spec_from_file_location("ref_kt_loader", "/root/reference/kt-kernel/python/utils/loader.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
for name in B.CASES:
    with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(io.StringIO()):
        out[name] = B.run_case(ref, name, d)
json.dump(out, open(os.path.join(HERE, "kt_loader_golden.json"), "w"), indent=1, sort_keys=True)
print({k: ("raises " + v["raises"]) if "raises" in v else sorted(v["result"]) for k, v in out.items()})
