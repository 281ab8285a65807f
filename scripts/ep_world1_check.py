"""Dev check on the single-GPU box: the whole-model bench step with expert parallelism switched on over a ONE-rank RCCL
group — exercises the all-gather / reduce-scatter inside the captured decode graph (the 8-GPU run is the driver's)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29876")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from ktransformers_amd.parallel import enable_expert_parallel
import bench

enable_expert_parallel()
mr = bench.ModelDecodeRunner(torch.device("cuda", 0), 4096, 200)
print("graph_ok", mr.graph_ok)
for i in range(10):
    mr.step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(100):
    mr.step(i)
torch.cuda.synchronize()
print("EP(world=1) whole-model tok/s", 100 / (time.perf_counter() - t0))
dist.destroy_process_group()
