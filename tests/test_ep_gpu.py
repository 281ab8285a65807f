"""GPU: the expert-parallel wrappers on a single-rank RCCL group (the 8-GPU run belongs to the driver): prefill
(all-to-all-v dispatch + ktx_moe_combine) must reproduce MoEHandle.forward bit for bit; decode (all-gather + fp32
partial + reduce-scatter) to 1 bf16 ulp; the peer-write decode transport wired through the whole model (its multi-rank
behaviour is tests/test_ep_peer_gpu.py's)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from helpers import make_case  # noqa: E402


@pytest.fixture(scope="module")
def group():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    yield None
    dist.destroy_process_group()


def test_prefill_and_decode_wrappers(group):
    from ktransformers_amd import _native as n
    from ktransformers_amd.parallel import ExpertParallelMoE
    E, k, H, I, T = 8, 2, 256, 128, 37
    c = make_case(5, E, k, H, I, T, invalid_ids=True)
    h = n.MoEHandle(E, k, H, I, T * k, "AMXINT4", 0)
    tt = lambda a: torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).cuda()
    h.load_bf16(tt(c["gate"]), tt(c["up"]), tt(c["down"]))
    x, ids, w = tt(c["x"]), torch.from_numpy(c["ids"]).cuda(), torch.from_numpy(c["w"]).cuda()
    ref = h.forward(x, ids, w)
    ep = ExpertParallelMoE(h)
    y = ep.forward_prefill(x, ids, w)
    assert torch.equal(y, ref)
    yd = ep.forward(x[:4], ids[:4], w[:4]).float()
    r4 = ref[:4].float()
    assert torch.all((yd - r4).abs() <= r4.abs() * 2.0 ** -7 + 1e-5 * r4.abs().max())


def test_whole_model_decode_graph_with_rccl_collectives(group):
    """bench.py's N > 1 step on a one-rank RCCL group: the YAML-injected model with its routed experts behind all-gather ->
    fp32 partial -> reduce-scatter, the whole greedy token step — collectives included — captured in ONE HIP graph.
    (a) the capture succeeds, (b) replays give the tokens of the same step run eagerly, (c) the first step's logits agree
    with the single-GPU model's to the fp32-reorder tolerance of the decode combine."""
    import bench
    from ktransformers_amd import parallel
    dev = torch.device("cuda", 0)
    wl = bench.WORKLOADS["v2lite-int4"]

    def runner(ep, graph):
        parallel.enable_expert_parallel(enabled=ep)
        torch.manual_seed(0)                      # the synthetic KV cache; the weights are seeded by tensor name
        return bench.ModelDecodeRunner(wl, 3, dev, 64, 32, seed=0, use_graph=graph)

    def first_logits(mr):
        with torch.no_grad():
            return mr.model(mr.cur.clone(), mr.pos.clone(), mr.cache, mr.pos[0].clone())[0, -1].float()

    try:
        single = runner(False, False)
        want = first_logits(single)
        single.close()
        ep_eager = runner(True, False)
        assert all(l.mlp.experts.generate_experts._ep is not None for l in ep_eager.model.model.layers if hasattr(l.mlp, "experts"))
        got = first_logits(ep_eager)
        assert (got - want).abs().max() <= 2e-2 * want.abs().max(), float((got - want).abs().max() / want.abs().max())
        def restart(mr):                          # both runners decode from the same state
            mr.set_position(64)
            mr.cur.fill_(1)

        restart(ep_eager)
        toks_eager = []
        for _ in range(6):
            ep_eager.step()
            toks_eager.append(int(ep_eager.cur.item()))
        ep_eager.close()
        ep_graph = runner(True, True)
        assert ep_graph.graph_ok, ep_graph.graph_error
        restart(ep_graph)
        toks_graph = []
        for _ in range(6):
            ep_graph.step()
            toks_graph.append(int(ep_graph.cur.item()))
        ep_graph.close()
        assert toks_graph == toks_eager, (toks_graph, toks_eager)
        # the same step with the peer-write transport (include/ktx_ep.h) in place of the two collectives: set up through the
        # process group (handle swap + on-fabric self-check), taken by every KExpertsHIP, captured in the graph; with one
        # rank the reduce is a single rounding, so the tokens are those of the collective path
        ex = parallel.enable_peer_exchange(wl["H"], wl["k"], 16, dev)
        try:
            ep_peer = runner(True, True)
            assert ep_peer.graph_ok, ep_peer.graph_error
            restart(ep_peer)
            toks_peer = []
            for _ in range(6):
                ep_peer.step()
                toks_peer.append(int(ep_peer.cur.item()))
            ep_peer.close()
            assert ex.status() == 0
            assert toks_peer == toks_eager, (toks_peer, toks_eager)
        finally:
            parallel.EP_STATE["exchange"] = None
            ex.close()
    finally:
        parallel.enable_expert_parallel(enabled=False)


def test_bench_takes_its_multi_gpu_branch_on_one_rank():
    """`bench.py --force-dist` with WORLD_SIZE=1: the N > 1 code path of main() — RCCL process group, experts behind the EP
    wrappers, the peer-write transport set up through the group (handle swap, self-check), the load/capture barrier, the timed
    loop with its barriers and MAX over ranks, the transport status in the JSON line — on the one GPU this box has.
    (Its own process: the module fixture above already owns this process's default group.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29900 + os.getpid() % 90), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import tempfile
    tmp = tempfile.mkdtemp()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--workload", "v2lite-int4",
                        "--layers", "3", "--steps", "8", "--warmup", "2", "--no-prefill", "--no-secondary", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=280, cwd=tmp)
    assert r.returncode == 0, r.stderr[-2000:]
    out_lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(out_lines) == 1 and len(out_lines[0]) < 4096          # ONE compact stdout line (round 5) ...
    line = json.loads(out_lines[-1])
    detail = json.load(open(os.path.join(tmp, line["detail"])))      # ... the per-kernel table lives in the detail record beside it
    assert detail["value"] == line["value"] and detail["config"]["ep_transport_status"] == 0
    cfg = line["config"]
    assert line["value"] and line["value"] > 0 and line["n_gpus"] == 1 and line["scaling"] == "weak"
    assert cfg["parallelism"] == "ep1" and cfg["rccl_ranks"] == 1 and cfg["hip_graph"] is True, cfg
    assert cfg["ep_transport"].startswith("peer writes") and cfg["ep_transport_status"] == 0, cfg
    # round 3: the N > 1 line carries the per-kernel table and a roofline too (HIP events on rank 0, all ranks stepping together)
    assert line["roofline"]["bound"] == "hbm" and detail["per_kernel"] and detail["per_kernel"][0]["us_per_step"] > 0
    # ... and --strong is ONE token stream: the same rows on every rank, only the fp32 partials travel
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--strong", "--workload",
                        "v2lite-int4", "--layers", "3", "--steps", "8", "--warmup", "2", "--no-prefill", "--no-secondary",
                        "--no-cpu-baseline", "--no-kernels", "--windows", "0"], env=env, capture_output=True, text=True, timeout=280, cwd=tmp)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["value"] > 0 and line["config"]["ep_transport_status"] == 0
