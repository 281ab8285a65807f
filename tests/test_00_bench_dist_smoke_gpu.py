"""GPU: the N > 1 code path of bench.py — process group, experts sharded E/N per rank, the peer-write decode exchange, max-over-ranks
timing, rank 0's compact line — run as EIGHT processes on the ONE GPU of the test box (`--dist-backend gloo`: the ranks share
cuda:0 and map each other's exchange buffers through inter-process handles exactly as on an 8-GPU node; RCCL itself refuses
several ranks on one device, so the rendezvous / status reductions go through gloo).  It de-risks the first real
`torchrun --nproc-per-node 8 bench.py --gpus 8`: launch line, env handling, sharding arithmetic, exchange tags
over many steps and layers, graph capture per rank, the final line.  DeepSeek-V2-Lite dimensions, three layers (1 dense + 2 MoE: the exchange runs twice per step; 64 experts = 8 per rank) — the eight
ranks share ONE GPU's memory, and the synthetic weight source builds every expert before a rank keeps its share; the one-launch attention is switched off
(`KTX_ATTN_SEPARATE=1`): a persistent launch needs every CU, and eight of them on one GPU would wait for each other.

The file sorts FIRST in the suite on purpose: the eight ranks then share the GPU with a parent process that has not opened a HIP
context yet.  Run behind the other GPU tests (whose streams and graphs keep hardware queues open in the parent) the nine contexts
oversubscribe the hardware queues, the scheduler time-slices spinning poll kernels, and the exchange crawls — seen twice in full-suite
runs, never alone (6 s).  Every attempt is bounded and kills its whole process group."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, tmp_path, attempt, limit_s=150):
    """One launch of the 8-rank bench; returns (line | None, why).  Bounded: the whole process GROUP (torchrun + its ranks) is killed after
    limit_s, so a rank the box never schedules cannot hold the suite (or the GPU) hostage."""
    import signal
    port = 29900 + (os.getpid() + 97 * attempt) % 1500
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", KTX_ATTN_SEPARATE="1", OMP_NUM_THREADS="1", KTX_EP_SPIN_SECONDS="30")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2",
           "--workload", "v2lite-int4", "--layers", "3", "--ctx", "256", "--windows", "0", "--dist-backend", "gloo", "--no-cpu-baseline", "--no-prefill",
           "--no-kernels", "--no-secondary"]
    proc = subprocess.Popen(cmd, cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        stdout, stderr = proc.communicate(timeout=limit_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        stdout, stderr = proc.communicate()
        return None, f"no line within {limit_s} s (ranks killed)"
    # the ranks' own tracebacks come first in stderr, torchrun's summary last: show the head of the first one and the tail
    first = stderr.find("Traceback (most recent call last)")
    tail = (stdout[-1500:] + "\n--- stderr (first traceback) ---\n" + (stderr[first:first + 4000] if first >= 0 else "")
            + "\n--- stderr (tail) ---\n" + stderr[-1500:])
    log_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(log_dir):
        with open(os.path.join(log_dir, f"bench_dist_smoke_{world}_try{attempt}.stderr"), "w") as f:
            f.write(stderr)
    assert proc.returncode == 0, tail
    lines = [ln for ln in stdout.strip().splitlines() if ln.startswith("{")]
    assert lines, tail
    assert len(lines[-1]) < 4096
    line = json.loads(lines[-1])
    return line, (line.get("error") or "")


@pytest.mark.parametrize("world", [8])
def test_bench_runs_as_n_processes_on_one_gpu(world, tmp_path):
    line, why = _run(world, tmp_path, 0)
    if line is None or not line.get("value"):
        line2, why2 = _run(world, tmp_path, 1)
        if line2 is not None:
            line, why = line2, why2
    if line is None or not line.get("value"):
        # Eight processes TIME-SLICE one GPU here: a rank whose queues the hardware scheduler does not run for the poll bound (30 s in
        # this test) makes a peer's gather give up — bench.py then voids its number, as it must — or no line appears in time at all.
        # That is this box's scheduling of nine contexts, not the protocol (tests/test_ep_peer_gpu.py holds it bit-exact with 2 and 4
        # ranks): seen in full-suite runs on busy boxes, never alone (alone: 6 s for the whole test).  Both attempts are bounded.
        if line is not None:
            assert line["n_gpus"] == world and line["config"]["parallelism"] == f"ep{world}"
            # a line that CAME BACK must say why it carries no number: only the bounded poll timeout of the time-sliced exchange is an
            # accepted reason to skip — any other void (a protocol error, a crash of a rank) fails the suite
            if line["config"].get("ep_transport_status") in (None, 0):
                pytest.fail(f"bench.py --gpus {world} returned a line without a value for a reason other than the exchange's poll bound: {line}")
        pytest.skip(f"eight ranks time-slicing one GPU did not finish the exchange twice: {why}")
    assert line["n_gpus"] == world and line["steps"] == 6 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert line["value"] > 0 and abs(line["value"] - world * 1e3 / line["ms_per_step"]) / line["value"] < 1e-2
    cfg = line["config"]
    assert cfg["parallelism"] == f"ep{world}" and cfg["dist_backend"] == "gloo" and cfg["rccl_ranks"] == 0
    assert cfg.get("ep_transport_status") == 0, cfg                       # no poll of the peer-write exchange gave up
    assert str(cfg.get("ep_transport", "")).startswith("peer writes"), cfg   # the transport an 8-GPU node takes, not the collective fallback
