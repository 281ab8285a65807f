"""GPU: the peer-write decode exchange (include/ktx_ep.h, ktransformers_amd/parallel.py) — SURVEY.md §8e.

The GPU box has ONE GPU, so the ranks of these tests share it: inside one process (buffers mapped by pointer, one stream per
rank) and across processes (buffers mapped through inter-process handles, the way an 8-GPU node maps them).  What is
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


@pytest.mark.parametrize("memory", ["uncached", "finegrained"])
@pytest.mark.parametrize("world,T,H,k", [(2, 1, 7168, 8), (2, 4, 2048, 6), (3, 2, 1030, 3)])
def test_ranks_in_one_process(world, T, H, k, memory):
    dev = torch.device("cuda", 0)
    exs = _ring(world, H, k, memory)
    streams = [torch.cuda.Stream(dev) for _ in range(world)]
    g = torch.Generator(device="cpu").manual_seed(world * 100 + T)
    try:
        for rnd in range(3):      # the call tags advance on the device
            x = [torch.randn(T, H, generator=g).to(torch.bfloat16).to(dev) for _ in range(world)]
            ids = [torch.randint(0, 1 << 40, (T, k), generator=g).to(dev) for _ in range(world)]
            w = [torch.rand(T, k, generator=g).to(dev) for _ in range(world)]
            part = [torch.randn(world * T, H, generator=g).to(dev) for _ in range(world)]
            torch.cuda.synchronize()
            got, outs = [None] * world, [None] * world
            for r in range(world):
                with torch.cuda.stream(streams[r]):
                    got[r] = exs[r].gather(x[r], ids[r], w[r])
            for r in range(world):
                with torch.cuda.stream(streams[r]):
                    outs[r] = exs[r].reduce(part[r])
            torch.cuda.synchronize()
            for r in range(world):
                assert exs[r].status() == 0, "a poll gave up waiting for its peer"
                xg, idsg, wg = got[r]
                assert torch.equal(xg.view(torch.int16), torch.cat(x).view(torch.int16))
                assert torch.equal(idsg, torch.cat(ids)) and torch.equal(wg, torch.cat(w))
                acc = part[0][r * T:(r + 1) * T]
                for q in range(1, world):
                    acc = acc + part[q][r * T:(r + 1) * T]
                assert torch.equal(outs[r].view(torch.int16), acc.to(torch.bfloat16).view(torch.int16)), \
                    "reduce is not the rank-order fp32 sum rounded once"
    finally:
        for e in exs:
            e.close()


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


@pytest.mark.parametrize("world", [2])
def test_ranks_in_separate_processes_map_each_other_by_ipc_handle(world, tmp_path):
    """Real experts behind it: every process owns E/world experts and its own tokens; eager and under a replayed HIP graph
    the result equals sum_r forward_partial_r in rank order bit for bit, and the single-handle forward to fp32 re-association."""
    port = 29700 + os.getpid() % 2000
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = [str(tmp_path / f"rank{r}.json") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "ep_peer_worker.py"), str(r), str(world), str(port), outs[r]],
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
