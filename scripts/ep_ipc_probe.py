"""Dev probe: tests/ep_peer_worker.py (WORLD processes sharing cuda:0, default 2) once per memory kind of the symmetric
buffer.  Prints each rank's result line (bit-exactness flags, status word, per-layer cost of the exchange in a graph)."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WORKER = os.path.join(ROOT, "tests", "ep_peer_worker.py")
WORLD = int(sys.argv[1]) if len(sys.argv) > 1 else 2

KINDS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["uncached", "finegrained", "plain"]
for i, memory in enumerate(KINDS):
    with tempfile.TemporaryDirectory() as td:
        outs = [os.path.join(td, f"r{r}.json") for r in range(WORLD)]
        t0 = time.time()
        procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(WORLD), str(29900 + i), outs[r], memory],
                                  env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT) for r in range(WORLD)]
        logs = []
        for p in procs:
            try:
                logs.append(p.communicate(timeout=150)[0])
            except subprocess.TimeoutExpired:
                p.kill()
                logs.append(b"TIMEOUT " + p.communicate()[0])
        for r, (o, lg) in enumerate(zip(outs, logs)):
            if not os.path.exists(o):
                print(f"--- rank {r} left no result; its output:\n" + lg.decode("utf-8", "replace")[:3000], flush=True)
        for o in outs:
            r = json.load(open(o)) if os.path.exists(o) else {"error": "no result"}
            r.pop("trace", None)
            print(f"memory={memory} {time.time() - t0:.1f}s {json.dumps(r)}", flush=True)
