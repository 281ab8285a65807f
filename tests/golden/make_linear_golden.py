"""Generate tests/golden/linear_w4_golden.npz by importing the REFERENCE's own quantiser
(/root/reference/archive/ktransformers/ktransformers_ext/operators/custom_marlin/quantize/utils/quant_utils.py — pure
torch, loaded by file path so none of the CUDA-only packages around it are imported).

Run in the build container:   python tests/golden/make_linear_golden.py
Stored: bf16 weights [N,K] (nn.Linear layout, raw bits), and for each group size the reference's q_w [K,N] and s [K/g,N].
"""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/archive/ktransformers/ktransformers_ext/operators/custom_marlin/quantize/utils/quant_utils.py"
spec = importlib.util.spec_from_file_location("ref_quant_utils", REF)
qu = importlib.util.module_from_spec(spec)
spec.loader.exec_module(qu)

out = {}
g = torch.Generator().manual_seed(20260921)
for name, (N, K) in {"a": (64, 256), "b": (48, 384)}.items():
    w = (torch.randn(N, K, generator=g) / 10).to(torch.bfloat16)
    w[3, 128:192] = 0            # an all-zero group (s = 0 -> NaN -> q = 0 in the reference)
    w[5, 7] = 3.0                # an outlier
    out[f"{name}_w"] = w.view(torch.uint16).numpy()
    for G in (32, 64, 128):
        # KLinearMarlin.load: weight = w.view(out, in).T ; marlin_quantize(weight, 4, G, False) (linear.py:645,664)
        q, s, _, _ = qu.quantize_weights(w.T.contiguous(), 4, G, False)
        out[f"{name}_q{G}"] = q.numpy().astype(np.uint8)
        out[f"{name}_s{G}"] = s.view(torch.uint16).numpy()
    # 8 bits (linear.py:608, quant_utils.py:5), with and without the simulated act_order: q_w [K,N] in 0..255, s, and for
    # act_order the stored row order rand_perm (the quantised values and scales are computed BEFORE the permutation)
    for G, act in ((64, False), (128, True)):
        torch.manual_seed(7)
        q, s, g_idx, perm = qu.quantize_weights(w.T.contiguous(), 8, G, act)
        tag = f"{name}_8b{G}{'_act' if act else ''}"
        out[tag + "_q"] = q.numpy().astype(np.uint8)
        out[tag + "_s"] = s.view(torch.uint16).numpy()
        if act:
            out[tag + "_perm"] = perm.numpy().astype(np.int64)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "linear_w4_golden.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path))
