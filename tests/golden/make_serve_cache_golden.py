"""Generates tests/golden/serve_cache_golden.npz: the REFERENCE's KDeepSeekV3Cache.get_page_table
(archive/ktransformers/models/custom_cache.py:446-463, lifted from the source with `ast` — the module imports the compiled
sched_ext) on seeded scheduler states.  Run in the build container only."""
import ast
import os
import types

import numpy as np
import torch

SRC = "/root/reference/archive/ktransformers/models/custom_cache.py"
ns = {"torch": torch}
for node in ast.parse(open(SRC).read()).body:
    if isinstance(node, ast.ClassDef) and node.name == "KDeepSeekV3Cache":
        for item in node.body:
            if isinstance(item, ast.FunctionDef) and item.name == "get_page_table":
                item.returns = None
                for a in item.args.args:
                    a.annotation = None
                exec(compile(ast.Module([item], []), SRC, "exec"), ns)

out = {}
g = torch.Generator().manual_seed(0)
for case, page_size in enumerate((16, 64, 256)):
    nreq = 5
    qlens = torch.randint(0, 6, (nreq,), generator=g)
    qlens[0] = 5
    q_indptr = torch.cat([torch.zeros(1, dtype=torch.long), qlens.cumsum(0)])
    npages = torch.randint(1, 5, (nreq,), generator=g)
    kv_indptr = torch.cat([torch.zeros(1, dtype=torch.long), npages.cumsum(0)])
    kv_indices = torch.randperm(64, generator=g)[: int(kv_indptr[-1])].to(torch.int32)
    T = int(q_indptr[-1]) + 3                                    # three padding tokens past the scheduled ones
    pos = torch.randint(0, 5 * page_size, (T,), generator=g)    # some positions lie beyond the request's pages
    for b in (int(q_indptr[-1]), max(int(q_indptr[-1]) - 2, 1), T):
        pi, po = ns["get_page_table"](types.SimpleNamespace(page_size=page_size), pos, q_indptr, kv_indptr, kv_indices, torch.tensor([b]))
        key = f"c{case}_b{b}"
        out[key + "_in"] = np.array([page_size, b])
        for n, v in (("pos", pos), ("q_indptr", q_indptr), ("kv_indptr", kv_indptr), ("kv_indices", kv_indices), ("page_idx", pi), ("page_offset", po)):
            out[f"{key}_{n}"] = v.numpy()
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "serve_cache_golden.npz"), **out)
print(sorted(k for k in out if k.endswith("_in")))
