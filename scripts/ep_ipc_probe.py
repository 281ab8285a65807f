"""Dev probe: tests/ep_peer_worker.py (2 processes sharing cuda:0) once per memory kind of the symmetric buffer, plus the
per-layer cost of the two exchange launches at V3 width.  Prints one line per configuration."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WORKER = os.path.join(ROOT, "tests", "ep_peer_worker.py")

for i, memory in enumerate(("uncached", "finegrained", "plain")):
    with tempfile.TemporaryDirectory() as td:
        outs = [os.path.join(td, f"r{r}.json") for r in range(2)]
        t0 = time.time()
        procs = [subprocess.Popen([sys.executable, WORKER, str(r), "2", str(29900 + i), outs[r], memory],
                                  env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT) for r in range(2)]
        for p in procs:
            try:
                p.communicate(timeout=150)
            except subprocess.TimeoutExpired:
                p.kill()
                p.communicate()
        for o in outs:
            r = json.load(open(o)) if os.path.exists(o) else {"error": "no result"}
            r.pop("trace", None)
            print(f"memory={memory} {time.time() - t0:.1f}s {json.dumps(r)}", flush=True)

# cost of the two launches with both ranks in this process (one stream each), V3 width, one token per rank
import torch  # noqa: E402
from ktransformers_amd._native import EpExchange  # noqa: E402

for world in (2, 3):
    H, k, T = 7168, 8, 1
    exs = [EpExchange(world, r, 4, H, k, 0, "uncached") for r in range(world)]
    for a in exs:
        for b in exs:
            if a is not b:
                a.import_ptr(b.rank, b.local_ptr())
    st = [torch.cuda.Stream() for _ in range(world)]
    x = torch.randn(T, H, device="cuda").to(torch.bfloat16)
    ids = torch.zeros(T, k, dtype=torch.int64, device="cuda")
    w = torch.rand(T, k, device="cuda")
    part = torch.randn(world * T, H, device="cuda")

    def step():
        for r in range(world):
            with torch.cuda.stream(st[r]):
                exs[r].gather(x, ids, w)
        for r in range(world):
            with torch.cuda.stream(st[r]):
                exs[r].reduce(part)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"in-process world={world} H={H}: {dt * 1e6:.1f} us per (gather + reduce) pair, host-issued, status "
          f"{[e.status() for e in exs]}", flush=True)
    for e in exs:
        e.close()
