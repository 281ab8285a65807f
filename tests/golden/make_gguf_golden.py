"""Generate tests/golden/gguf_blocks_golden.npz: random (but valid) Q4_K / Q6_K blocks and the values the REFERENCE's own
numpy dequantisers (archive/ktransformers/util/custom_gguf.py:326-343 dequantize_q4_k, dequantize_q6_k) give for them.
custom_gguf.py imports CUDA-only extensions at module import; they are stubbed (only the numpy functions are used).

    python tests/golden/make_gguf_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

for name in ("KTransformersOps", "ktransformers", "ktransformers.util", "ktransformers.util.custom_loader"):
    sys.modules.setdefault(name, types.ModuleType(name))
REF = "/root/reference/archive/ktransformers/util/custom_gguf.py"
src = open(REF).read()
ns = {"__name__": "ref_custom_gguf"}
# keep only the pure-numpy part: drop imports that need the built extensions
keep = []
for line in src.splitlines():
    if line.startswith(("import KTransformersOps", "from ktransformers", "import ktransformers")):
        continue
    keep.append(line)
exec(compile("\n".join(keep), REF, "exec"), ns)

rng = np.random.default_rng(20260921)
out = {}
nb = 24
q4 = rng.integers(0, 256, (nb, 144), dtype=np.uint8)
q4[:, 0:2] = (rng.random(nb).astype(np.float16) * np.float16(0.01)).view(np.uint8).reshape(nb, 2)
q4[:, 2:4] = (rng.random(nb).astype(np.float16) * np.float16(0.01)).view(np.uint8).reshape(nb, 2)
out["q4k_blocks"] = q4
out["q4k_values"] = np.asarray(ns["dequantize_q4_k"](q4.tobytes()), dtype=np.float32).reshape(nb, 256)
q6 = rng.integers(0, 256, (nb, 210), dtype=np.uint8)
q6[:, 208:210] = ((rng.random(nb).astype(np.float16) - np.float16(0.5)) * np.float16(0.01)).view(np.uint8).reshape(nb, 2)
out["q6k_blocks"] = q6
out["q6k_values"] = np.asarray(ns["dequantize_q6_k"](q6.tobytes()), dtype=np.float32).reshape(nb, 256)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gguf_blocks_golden.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path))
