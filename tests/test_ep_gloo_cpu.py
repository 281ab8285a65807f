"""Expert-parallel choreography on CPU: world_size 2 and 4, gloo.  Each rank owns half the experts and its own tokens; the
local-expert compute is stood in by the oracle (tests only), so this exercises the all-gather / partial / reduce-scatter
path of ktransformers_amd/parallel.py against the single-process oracle result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import bf16_to_f32, make_case

E, K, H, I, T = 8, 2, 256, 128, 3


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from ktransformers_amd.parallel import ep_decode_forward, expert_range
    from oracle.oracle import FMT_AMXINT4, Oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle()
    c = make_case(11, E, K, H, I, T * world)
    moe = o.make_moe(FMT_AMXINT4, c["gate"], c["up"], c["down"])
    begin, cnt = expert_range(E, world, rank)
    mask = np.ones(E, np.uint8)
    mask[begin:begin + cnt] = 0           # everything this rank does NOT own is skipped
    moe_local = dict(moe, mask=mask)

    def local_partial(xg, idsg, wg):
        # fp32 partial = un-rounded weighted sum over owned experts; the oracle rounds to bf16 at the end, so feed it
        # one slot at a time and accumulate the (exactly representable) per-slot products in fp32 like the kernel does
        xs = xg.view(torch.int16).numpy().view(np.uint16)
        ids, w = idsg.numpy(), wg.numpy()
        acc = np.zeros((xs.shape[0], H), np.float32)
        for j in range(ids.shape[1]):
            one = o.moe_forward(moe_local, ids[:, j:j + 1], np.ones((xs.shape[0], 1), np.float32), xs)
            acc = np.float32(bf16_to_f32(one) * w[:, j:j + 1] + acc)
        return torch.from_numpy(acc)

    sl = slice(rank * T, (rank + 1) * T)
    x = torch.from_numpy(c["x"][sl].view(np.int16).copy()).view(torch.bfloat16)
    y = ep_decode_forward(local_partial, x, torch.from_numpy(c["ids"][sl]), torch.from_numpy(c["w"][sl]))
    full = o.moe_forward(moe, c["ids"], c["w"], c["x"])
    q.put((rank, y.view(torch.int16).numpy().view(np.uint16).copy(), full[sl].copy()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_expert_parallel_decode_matches_single_process(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, want in res:
        a, b = bf16_to_f32(got), bf16_to_f32(want)
        # cross-rank fp32 summation order differs from slot order: <= 1 bf16 ulp
        assert np.all(np.abs(a - b) <= np.abs(b) * 2.0 ** -7 + 1e-5 * np.abs(b).max()), f"rank {rank}"
        assert (got != want).mean() < 0.05


def _replicated_worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from ktransformers_amd import parallel
    from oracle.oracle import FMT_AMXINT4, Oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle()
    c = make_case(31, E, K, H, I, T)
    moe = o.make_moe(FMT_AMXINT4, c["gate"], c["up"], c["down"])
    begin, cnt = parallel.expert_range(E, world, rank)
    mask = np.ones(E, np.uint8)
    mask[begin:begin + cnt] = 0
    moe_local = dict(moe, mask=mask)

    def local_partial(xg, idsg, wg):
        xs = xg.view(torch.int16).numpy().view(np.uint16)
        ids, w = idsg.numpy(), wg.numpy()
        assert xs.shape[0] == T, "strong scaling: no gather, the rank sees only the one stream's rows"
        acc = np.zeros((xs.shape[0], H), np.float32)
        for j in range(ids.shape[1]):
            one = o.moe_forward(moe_local, ids[:, j:j + 1], np.ones((xs.shape[0], 1), np.float32), xs)
            acc = np.float32(bf16_to_f32(one) * w[:, j:j + 1] + acc)
        return torch.from_numpy(acc)

    parallel.set_replicated_input(True)
    x = torch.from_numpy(c["x"].view(np.int16).copy()).view(torch.bfloat16)
    y = parallel.ep_decode_forward(local_partial, x, torch.from_numpy(c["ids"]), torch.from_numpy(c["w"]))
    full = o.moe_forward(moe, c["ids"], c["w"], c["x"])
    q.put((rank, y.view(torch.int16).numpy().view(np.uint16).copy(), full.copy()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_strong_scaling_decode_one_stream_over_all_ranks(world):
    """parallel.set_replicated_input: every rank holds the same rows, runs only its own experts on them, and the fp32 parts
    are added in rank order on every rank — the ranks' outputs are IDENTICAL bits (replicas cannot drift) and within one
    bf16 ulp of the single-process forward."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 40 + world
    procs = [ctx.Process(target=_replicated_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, want in res:
        assert np.array_equal(got, res[0][1]), f"rank {rank} differs from rank 0"
        a, b = bf16_to_f32(got), bf16_to_f32(want)
        assert np.all(np.abs(a - b) <= np.abs(b) * 2.0 ** -7 + 1e-5 * np.abs(b).max()), f"rank {rank}"


def _peer_refused_worker(rank, world, port, q):
    from ktransformers_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        parallel.enable_peer_exchange(256, 2, 4, "cuda:0")
        q.put((rank, "no error"))
    except RuntimeError as e:
        q.put((rank, str(e)))
    assert parallel.EP_STATE["exchange"] is None
    dist.barrier()            # the ranks are still in step with each other
    dist.destroy_process_group()


def test_peer_exchange_is_refused_on_every_rank_alike_where_there_is_no_gpu():
    """enable_peer_exchange is a collective: when a rank cannot set the transport up (here: no HIP device at all) EVERY
    rank raises the same error and the collectives stay the transport — never one rank polling for peers that fell back."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 17
    procs = [ctx.Process(target=_peer_refused_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1] and "failed on rank(s) 0:" in res[0] and "1:" in res[0], res


def test_expert_range():
    from ktransformers_amd.parallel import expert_range
    assert expert_range(256, 8, 3) == (96, 32)
    with pytest.raises(ValueError):
        expert_range(10, 4, 0)


def _prefill_worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from ktransformers_amd.parallel import ep_prefill_forward, expert_range
    from oracle.oracle import FMT_AMXINT4, Oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle()
    Tl = 5 + rank                                   # ragged: ranks hold different token counts
    total = sum(5 + r for r in range(world))
    c = make_case(23, E, K, H, I, total, invalid_ids=True)
    moe = o.make_moe(FMT_AMXINT4, c["gate"], c["up"], c["down"])
    begin, cnt = expert_range(E, world, rank)
    seen = []

    def local_rows(rows, eids):
        e = eids.numpy()
        assert ((e >= begin) & (e < begin + cnt)).all(), "a row reached a rank that does not own its expert"
        seen.append(len(e))
        xs = rows.view(torch.int16).numpy().view(np.uint16)
        y = o.moe_forward(moe, e.reshape(-1, 1), np.ones((len(e), 1), np.float32), xs)
        return torch.from_numpy(y.view(np.int16).copy()).view(torch.bfloat16)

    def combine(rows, row_of_pair, w):
        r = bf16_to_f32(rows.view(torch.int16).numpy().view(np.uint16)).astype(np.float64)
        rp, wn = row_of_pair.numpy(), w.numpy().astype(np.float64)
        acc = np.zeros((rp.shape[0], H), np.float32)
        for j in range(rp.shape[1]):                # slot order, fp32 FMA (product exact in fp64, one fp32 rounding)
            ok = rp[:, j] >= 0
            acc[ok] = (r[rp[ok, j]] * wn[ok, j:j + 1] + acc[ok].astype(np.float64)).astype(np.float32)
        from oracle.oracle import f32_to_bf16
        return torch.from_numpy(f32_to_bf16(acc).view(np.int16).copy()).view(torch.bfloat16)

    off = sum(5 + r for r in range(rank))
    sl = slice(off, off + Tl)
    x = torch.from_numpy(c["x"][sl].view(np.int16).copy()).view(torch.bfloat16)
    st = {}
    y = ep_prefill_forward(local_rows, combine, x, torch.from_numpy(c["ids"][sl]), torch.from_numpy(c["w"][sl]), E, stats=st)
    full = o.moe_forward(moe, c["ids"], c["w"], c["x"])
    # de-duplicated dispatch: one row per distinct (token, destination rank), counted independently here
    ids_l = c["ids"][sl]
    want_rows = sum(len({int(e) // cnt for e in row if 0 <= e < E}) for row in ids_l)
    want_pairs = int(((ids_l >= 0) & (ids_l < E)).sum())
    assert (st["rows_out"], st["pairs_out"]) == (want_rows, want_pairs), (st, want_rows, want_pairs)
    assert st["pairs_in"] == sum(seen)
    q.put((rank, y.view(torch.int16).numpy().view(np.uint16).copy(), full[sl].copy(), sum(seen), st["rows_out"], st["pairs_out"]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_expert_parallel_prefill_all_to_all_is_bit_identical(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_prefill_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows = sent_rows = sent_pairs = 0
    for rank, got, want, n, ro, po in res:
        assert np.array_equal(got, want), f"rank {rank}: {(got != want).sum()} elements differ"
        rows += n
        sent_rows += ro
        sent_pairs += po
    assert rows > 0 and rows == sent_pairs
    assert sent_rows < sent_pairs, "the seeded routing has tokens naming two experts of one rank: their row must travel once"
