"""One rank of tests/test_ep_peer_gpu.py's multi-process cases: `python ep_peer_worker.py RANK WORLD PORT OUT.json [MEMORY]`.

All ranks share cuda:0 (the GPU box has one GPU): the peers' buffers are mapped through inter-process handles exactly as on
an 8-GPU node, only the fabric under them is the local memory system instead of xGMI.  The rendezvous is gloo over
127.0.0.1; no RCCL is involved in this transport."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from helpers import make_case  # noqa: E402

E, K, H, I, T = 16, 4, 1024, 512, 2


def main():
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    memory = sys.argv[5] if len(sys.argv) > 5 else "uncached"
    res = {"rank": rank, "ok": False, "memory": memory}
    try:
        from ktransformers_amd import parallel
        from ktransformers_amd._native import MoEHandle
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        c = make_case(77, E, K, H, I, T * world * 4)           # 4 rounds of T tokens per rank
        bf = lambda a: torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).to(dev)  # noqa: E731
        shards = []
        for r in range(world):
            b, n = parallel.expert_range(E, world, r)
            h = MoEHandle(n, K, H, I, max_len=8 * world, method="AMXINT4", device=dev, expert_begin=b, global_expert_num=E)
            h.load_bf16(bf(c["gate"][b:b + n]), bf(c["up"][b:b + n]), bf(c["down"][b:b + n]))
            shards.append(h)
        ex = parallel.enable_peer_exchange(H, K, max_tokens=4, device=dev, memory=memory)
        ex.set_spin_seconds(30)
        m = parallel.ExpertParallelMoE(shards[rank])
        xs, ids, ws = bf(c["x"]), torch.from_numpy(c["ids"]).to(dev), torch.from_numpy(c["w"]).to(dev)

        def mine(rnd):
            lo = (rnd * world + rank) * T
            return xs[lo:lo + T].contiguous(), ids[lo:lo + T].contiguous(), ws[lo:lo + T].contiguous()

        def want(rnd):      # partials of MY tokens from every rank's experts, added in rank order, one rounding
            x, i, w = mine(rnd)
            acc = shards[0].forward_partial(x, i, w)
            for r in range(1, world):
                acc = acc + shards[r].forward_partial(x, i, w)
            return acc.to(torch.bfloat16)

        # ---- the two kernels alone on seeded random data every rank can reproduce: rows arrive intact, the reduce is the
        # rank-order fp32 sum rounded once; the call tags advance over the rounds; T and H that do not divide evenly
        raw_ok, keep = [], []
        for (Hr, kr, Tr) in ((7168, 8, 1), (1030, 3, 4)):
            from ktransformers_amd._native import EpExchange
            exr = EpExchange(world, rank, 4, Hr, kr, dev, memory)
            hs = [None] * world
            dist.all_gather_object(hs, exr.export_handle())
            for r in range(world):
                if r != rank:
                    exr.import_handle(r, hs[r])
            dist.barrier()
            exr.set_spin_seconds(30)
            g = torch.Generator(device="cpu").manual_seed(Hr + world)
            for rnd in range(3):
                x = [torch.randn(Tr, Hr, generator=g).to(torch.bfloat16).to(dev) for _ in range(world)]
                i = [torch.randint(0, 1 << 40, (Tr, kr), generator=g).to(dev) for _ in range(world)]
                w = [torch.rand(Tr, kr, generator=g).to(dev) for _ in range(world)]
                part = [torch.randn(world * Tr, Hr, generator=g).to(dev) for _ in range(world)]
                xg, ig, wg = exr.gather(x[rank], i[rank], w[rank])
                out = exr.reduce(part[rank])
                acc = part[0][rank * Tr:(rank + 1) * Tr]
                for q in range(1, world):
                    acc = acc + part[q][rank * Tr:(rank + 1) * Tr]
                raw_ok.append(bool(torch.equal(xg.view(torch.int16), torch.cat(x).view(torch.int16)) and torch.equal(ig, torch.cat(i))
                                   and torch.equal(wg, torch.cat(w))
                                   and torch.equal(out.view(torch.int16), acc.to(torch.bfloat16).view(torch.int16))))
            # reduce-ONLY sequences (the replicated-stream mode: no gather between the calls), eager and as replays of ONE captured
            # launch: every call must add THIS call's partials — a stale granule of the previous call has the right shape and
            # the wrong numbers, so changing data tells
            ro_ok = []
            sp = torch.zeros(world * Tr, Hr, dtype=torch.float32, device=dev)
            so = torch.zeros(Tr, Hr, dtype=torch.bfloat16, device=dev)
            exr.reduce(sp, out=so, reduce_only=True)          # (warm-up outside the capture)
            torch.cuda.synchronize()
            dist.barrier()
            gro = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gro):
                exr.reduce(sp, out=so, reduce_only=True)
            for rnd in range(6):
                part = [torch.randn(world * Tr, Hr, generator=g).to(dev) for _ in range(world)]
                acc = part[0][rank * Tr:(rank + 1) * Tr]
                for q in range(1, world):
                    acc = acc + part[q][rank * Tr:(rank + 1) * Tr]
                if rnd % 2 == 0:
                    out = exr.reduce(part[rank], reduce_only=True)
                else:
                    sp.copy_(part[rank])
                    gro.replay()
                    out = so
                torch.cuda.synchronize()
                ro_ok.append(bool(torch.equal(out.view(torch.int16), acc.to(torch.bfloat16).view(torch.int16))))
            raw_ok.extend(ro_ok)
            raw_ok.append(exr.status() == 0)
            dist.barrier()
            keep.append(exr)     # stays mapped until the process ends: no free / re-allocate / re-map cycle of shared memory
        res["raw_bit_exact"] = raw_ok
        eager_ok = []
        for rnd in range(2):
            y = m.forward(*mine(rnd))
            eager_ok.append(bool(torch.equal(y.view(torch.int16), want(rnd).view(torch.int16))))
        res["eager_bit_exact"] = eager_ok
        # the same two launches under a captured graph, replayed on new inputs (the call tags live on the device)
        sx, si, sw = [t.clone() for t in mine(2)]
        torch.cuda.synchronize()
        dist.barrier()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            sy = m.forward(sx, si, sw)
        graph_ok = []
        for rnd in (2, 3, 2):
            x, i, w = mine(rnd)
            sx.copy_(x); si.copy_(i); sw.copy_(w)
            g.replay()
            torch.cuda.synchronize()
            graph_ok.append(bool(torch.equal(sy.view(torch.int16), want(rnd).view(torch.int16))))
        res["graph_bit_exact"] = graph_ok
        res["status"] = ex.status()
        # against ONE handle that owns every expert (slot order instead of rank order): fp32 re-association only
        full = MoEHandle(E, K, H, I, max_len=8, method="AMXINT4", device=dev)
        full.load_bf16(bf(c["gate"]), bf(c["up"]), bf(c["down"]))
        x, i, w = mine(0)
        y1 = full.forward(x, i, w).float()
        y2 = m.forward(x, i, w).float()
        res["max_rel_vs_single_gpu"] = float(((y1 - y2).abs().max() / y1.abs().max()).item())
        res["status"] = max(res["status"], ex.status())
        dist.barrier()
        # cost of the exchange inside a graph: 24 chained (gather, local experts, reduce) layers vs the local experts alone
        # (all ranks share one GPU here, so this bounds the protocol's launch + poll latency, not xGMI's)
        L = 24
        sx2 = sx.clone()

        def chain(with_exchange):
            y = sx2
            for _ in range(L):
                y = m.forward(y, si, sw) if with_exchange else shards[rank].forward_partial(y, si, sw).to(torch.bfloat16)
            return y

        per_layer = {}
        for name, flag in (("exchange", True), ("local_only", False)):
            chain(flag)
            torch.cuda.synchronize()
            dist.barrier()
            gg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gg):
                chain(flag)
            for _ in range(3):
                gg.replay()
            torch.cuda.synchronize()
            dist.barrier()
            import time
            t0 = time.perf_counter()
            for _ in range(20):
                gg.replay()
            torch.cuda.synchronize()
            per_layer[name] = (time.perf_counter() - t0) / 20 / L * 1e6
            dist.barrier()
        res["us_per_layer"] = per_layer
        res["status"] = max(res["status"], ex.status())
        res["ok"] = all(raw_ok) and all(eager_ok) and all(graph_ok) and res["status"] == 0 and res["max_rel_vs_single_gpu"] < 2 ** -7
    except Exception as e:  # the parent prints this
        import traceback
        res["error"] = f"{type(e).__name__}: {e}"
        res["trace"] = traceback.format_exc()[-1500:]
    with open(out_path, "w") as f:
        json.dump(res, f)
    try:
        torch.cuda.synchronize()
        dist.barrier()      # nobody unmaps a buffer a peer might still be using
    except Exception:
        pass
    sys.stdout.flush()
    os._exit(0 if res["ok"] else 1)     # the driver reclaims the mappings; no destructor ordering games at interpreter exit


if __name__ == "__main__":
    main()
