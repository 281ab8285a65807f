"""Generate tests/golden/gguf_blocks_golden.npz: random (but valid) blocks of every ggml type the reference's loader
de-quantises, and the values the REFERENCE's own numpy dequantisers give for them (archive/ktransformers/util/
custom_gguf.py:225-572: dequantize_q2_k, _q3_k, _q4_k, _q5_k, _q6_k, _iq4_xs, _q4_0, _q5_0, _q8_0).
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
# the other types the reference's loader reads (keys t<ggml type id>_blocks / _values); fp16 fields small and finite
rng2 = np.random.default_rng(20260922)
for t, nbytes, f16_cols, fn in ((2, 18, (0,), "dequantize_q4_0"), (6, 22, (0,), "dequantize_q5_0"), (8, 34, (0,), "dequantize_q8_0"),
                               (10, 84, (80, 82), "dequantize_q2_k"), (11, 110, (108,), "dequantize_q3_k"),
                               (13, 176, (0, 2), "dequantize_q5_k"), (23, 136, (0,), "dequantize_iq4_xs")):
    blk = rng2.integers(0, 256, (nb, nbytes), dtype=np.uint8)
    for c in f16_cols:
        blk[:, c:c + 2] = ((rng2.random(nb).astype(np.float16) - np.float16(0.5)) * np.float16(0.02)).view(np.uint8).reshape(nb, 2)
    out[f"t{t}_blocks"] = blk
    out[f"t{t}_values"] = np.asarray(ns[fn](blk.tobytes()), dtype=np.float32).reshape(nb, -1)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gguf_blocks_golden.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path))
