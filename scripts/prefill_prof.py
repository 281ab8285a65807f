"""One prompt chunk of the resident V3-shape model (bench.py's prefill leg), isolated for rocprofv3: the model is built, then the
chunk runs REPS times between two marker dispatches (a torch.cumsum — the only scan kernel of the process), so the post-processor
(scripts/prefill_prof_summary.py) can cut the load-time kernels away and say which share of the chunk runs inside libktx_hip.so.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python scripts/prefill_prof.py [layers] [tokens]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
REPS = 4
dev = torch.device("cuda", 0)
args = argparse.Namespace(no_graph=True, ctx=4096)
mr = bench.ModelDecodeRunner(bench.WORKLOADS["v3-int4"], layers, dev, 4096, 64, seed=0, use_graph=False)
from ktransformers_amd.util.generate import set_inference_mode  # noqa: E402
from ktransformers_amd.util.utils import InferenceState  # noqa: E402

set_inference_mode(mr.model, InferenceState.PREFILL)
ids = torch.randint(0, mr.cfg.vocab_size, (1, T), device=dev)
pos = torch.arange(T, device=dev).unsqueeze(0)
marker = torch.ones(12345, device=dev)
with torch.no_grad():
    mr.cache.past_tokens = [0] * mr.cfg.num_hidden_layers
    mr.model(ids, pos, mr.cache, pos[0], last_token_only=True)          # warm-up (scratch allocations, first-use paths)
    torch.cuda.synchronize()
    marker.cumsum(0)
    t0 = time.perf_counter()
    for _ in range(REPS):
        mr.cache.past_tokens = [0] * mr.cfg.num_hidden_layers
        mr.model(ids, pos, mr.cache, pos[0], last_token_only=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / REPS
    marker.cumsum(0)
    torch.cuda.synchronize()
print(f"prefill_prof: {layers} layers, {T} tokens, {REPS} chunks, {dt * 1e3:.3f} ms per chunk (host clock, under the profiler)")
