"""GPU: the peer-write decode exchange (include/ktx_ep.h, ktransformers_amd/parallel.py) — SURVEY.md §8e.

The GPU box has ONE GPU, so the ranks of these tests share it, one PROCESS per rank with the buffers mapped through
inter-process handles, the way an 8-GPU node maps them (ranks inside one process would depend on HIP putting their
streams on different hardware queues, which it does not promise).  What is
checked is the protocol — tagged granules, device-side call tags, rank-order fp32 sum, behaviour under graph replay, the
give-up path — not xGMI itself; enable_peer_exchange() re-checks the transport on the real fabric every time it is set up.

Bit-exact: the reduce is sum_r partial_r in rank order with one bf16 rounding, the reduce shape of the reference's
TP_MOE_Common::merge_results (kt-kernel/operators/amx/moe_base.hpp:749-791)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _ring(world, H, k, memory, max_tokens=8):
    from ktransformers_amd._native import EpExchange
    exs = [EpExchange(world, r, max_tokens, H, k, 0, memory) for r in range(world)]
    for a in exs:
        for b in exs:
            if a is not b:
                a.import_ptr(b.rank, b.local_ptr())
        a.set_spin_seconds(5)
    return exs


def test_a_ring_of_one_is_a_copy_and_a_rounding():
    """world = 1: gather returns the rank's own rows, reduce rounds its own partial (the kernels' local legs)."""
    dev = torch.device("cuda", 0)
    (ex,) = _ring(1, 1030, 3, "uncached")
    try:
        g = torch.Generator(device="cpu").manual_seed(5)
        for T in (1, 4):
            x = torch.randn(T, 1030, generator=g).to(torch.bfloat16).to(dev)
            ids = torch.randint(0, 1 << 40, (T, 3), generator=g).to(dev)
            w = torch.rand(T, 3, generator=g).to(dev)
            part = torch.randn(T, 1030, generator=g).to(dev)
            xg, idsg, wg = ex.gather(x, ids, w)
            out = ex.reduce(part)
            assert torch.equal(xg.view(torch.int16), x.view(torch.int16)) and torch.equal(idsg, ids) and torch.equal(wg, w)
            assert torch.equal(out.view(torch.int16), part.to(torch.bfloat16).view(torch.int16))
        assert ex.status() == 0
    finally:
        ex.close()


def test_a_missing_peer_raises_the_status_word_instead_of_hanging():
    """Rank 0 of a 2-rank ring whose peer never calls: both kernels give up after spin_seconds and report which one."""
    dev = torch.device("cuda", 0)
    exs = _ring(2, 512, 2, "uncached")
    try:
        exs[0].set_spin_seconds(0.2)
        x = torch.zeros(1, 512, dtype=torch.bfloat16, device=dev)
        exs[0].gather(x, torch.zeros(1, 2, dtype=torch.int64, device=dev), torch.zeros(1, 2, device=dev))
        assert exs[0].status() == 1
    finally:
        for e in exs:
            e.close()


def test_argument_errors():
    from ktransformers_amd._native import EpExchange, KtxError
    ex = EpExchange(2, 0, 4, 512, 2, 0)
    try:
        x = torch.zeros(1, 512, dtype=torch.bfloat16, device="cuda")
        with pytest.raises(KtxError, match="not mapped"):
            ex.gather(x, torch.zeros(1, 2, dtype=torch.int64, device="cuda"), torch.zeros(1, 2, device="cuda"))
        with pytest.raises(KtxError, match="out of range"):
            ex.import_ptr(0, ex.local_ptr())
        with pytest.raises(KtxError):
            EpExchange(2, 2, 4, 512, 2, 0)
    finally:
        ex.close()


@pytest.mark.parametrize("world,memory", [(2, "uncached"), (4, "uncached"), (2, "finegrained")])
def test_ranks_in_separate_processes_map_each_other_by_ipc_handle(world, memory, tmp_path):
    """The two kernels on seeded random rows / partials (rows intact, rank-order fp32 sum, tags over several rounds, ragged
    H and T), then real experts behind them: every process owns E/world experts and its own tokens; eager and under a
    replayed HIP graph the result equals sum_r forward_partial_r in rank order bit for bit, and the single-handle forward
    to fp32 re-association."""
    port = 29700 + os.getpid() % 2000 + world
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = [str(tmp_path / f"rank{r}.json") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "ep_peer_worker.py"), str(r), str(world), str(port), outs[r], memory],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    try:
        for p in procs:
            try:
                logs.append(p.communicate(timeout=240)[0].decode("utf-8", "replace")[-1500:])
            except subprocess.TimeoutExpired:
                p.kill()
                logs.append("TIMEOUT " + p.communicate()[0].decode("utf-8", "replace")[-1500:])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res = [json.load(open(o)) if os.path.exists(o) else {"ok": False, "error": "no result file"} for o in outs]
    assert all(r["ok"] for r in res), f"{res}\n{logs}"
