"""Phase stamps of the one-launch attention kernel INSIDE the whole-model decode graph (the last layer's launches of a
replayed step): what an isolated chain cannot show on the slower class of boxes.  python scripts/model_fused_stamps.py [layers=32]"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ktransformers_amd import _native as N  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
print(subprocess.run("/opt/rocm/bin/rocm-smi --showuniqueid --showdriverversion | grep 'Unique\\|Driver'", shell=True, capture_output=True, text=True).stdout.strip())
sa = torch.zeros(64, dtype=torch.int64, device=dev)
N.lib.ktx_attn_debug_stamps(sa.data_ptr())
wl = bench.WORKLOADS["v3-int4"]
mr = bench.ModelDecodeRunner(wl, L, dev, 4096, 64)
import time
for _ in range(10):
    mr.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    mr.step()
torch.cuda.synchronize()
print(f"{L} layers, graph step {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms (with the stamp stores)")
ta = sa.tolist()
an = ["entry", "A staged", "A done", "B: A's row polled", "B: q_a staged", "B: q_nope exchanged", "B done", "C: q polled", "C tiles done",
      "C done", "D: partials polled", "D: merged exchanged", "D done", "E: attn rows polled", "E streamed", "E done"]
print("attention launch of the last layer, workgroup 0 (us from entry):")
for i, nme in enumerate(an):
    if ta[i]:
        print(f"  {nme:24s} {(ta[i] - ta[0]) * 0.01:7.2f}")
