"""Generates tests/golden/kt_wrapper_golden.npz: outputs of the REFERENCE's own `generate_gpu_experts_masks` and
`BaseMoEWrapper.select_deferred_experts` (kt-kernel/python/experts_base.py:21-72, 347-375) on seeded inputs.  The module
itself cannot be imported here (it needs the compiled kt_kernel_ext), so the two functions are lifted from its source with
`ast` and executed as they are.  Run in the build container only."""
import ast
import os
import types

import numpy as np
import torch

SRC = "/root/reference/kt-kernel/python/experts_base.py"
tree = ast.parse(open(SRC).read())
ns = {"torch": torch, "Tuple": tuple, "Optional": object, "List": list, "Dict": dict}
for node in tree.body:
    if isinstance(node, ast.FunctionDef) and node.name == "generate_gpu_experts_masks":
        exec(compile(ast.Module([node], []), SRC, "exec"), ns)
    if isinstance(node, ast.ClassDef) and node.name == "BaseMoEWrapper":
        for item in node.body:
            if isinstance(item, ast.FunctionDef) and item.name == "select_deferred_experts":
                item.returns = None
                exec(compile(ast.Module([item], []), SRC, "exec"), ns)

out = {}
g = torch.Generator().manual_seed(0)
freq = torch.rand(5, 16, generator=g)
out["freq"] = freq.numpy()
for n in (0, 3, 17, 200):
    out[f"mask_{n}"] = ns["generate_gpu_experts_masks"](freq, n).numpy()
self = types.SimpleNamespace(num_experts=16)
ids = torch.stack([torch.randperm(16, generator=g)[:6] for _ in range(7)])
scores = torch.rand(7, 6, generator=g)
out["ids"], out["scores"] = ids.numpy(), scores.numpy()
for pk in (0, 2, 6, 9):
    imm, dfr = ns["select_deferred_experts"](self, ids, scores, pk)
    out[f"imm_{pk}"], out[f"def_{pk}"] = imm.numpy(), dfr.numpy()
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kt_wrapper_golden.npz"), **out)
print({k: v.shape for k, v in out.items()})
