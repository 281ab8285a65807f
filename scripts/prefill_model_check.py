"""Dev measurement: prompt processing (TTFT) of the whole injected V2-Lite model for a few prompt lengths."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ktransformers_amd.util.generate import set_inference_mode
from ktransformers_amd.util.utils import InferenceState

dev = torch.device("cuda", 0)
mr = bench.ModelDecodeRunner(dev, 8, 16, use_graph=False)
model, cache = mr.model, mr.cache
from ktransformers_amd.models.custom_cache import StaticCache
cache = StaticCache(mr.cfg, 1, 8192, str(dev), torch.bfloat16)
set_inference_mode(model, InferenceState.PREFILL)
for T in (128, 512, 2048):
    ids = torch.randint(0, 100000, (1, T), device=dev)
    pos = torch.arange(T, device=dev)[None]
    for rep in range(2):
        cache.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            logits = model(ids, pos, cache, pos[0], last_token_only=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"prefill T={T}: {dt*1e3:.1f} ms  -> {T/dt:.0f} tok/s")
