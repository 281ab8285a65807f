"""The one-launch MoE half of a decode step (csrc/ktx_moe_layer.inc, ktx_moe_layer_decode) against the three-launch path it
restates (ktx_linear_forward_fused_gate = router || shared gate|up, then ktx_moe_forward_side = routed gate/up, routed down +
shared down + adds), at the published DeepSeek-V3 dimensions (256 experts of 2048 x 7168, top-8 in 4 of 8 groups, one shared
expert): selected experts and weights, the shared experts' activations and the layer output must be BIT-IDENTICAL — as one
launch, as three launches of one phase each, and as replays of a captured HIP graph with changing inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu

E, K, H, I = 256, 8, 7168, 2048


def _u(shape, gen, dev, scale):
    return ((torch.rand(shape, generator=gen, device=dev, dtype=torch.float32) * 2 - 1) * scale).to(torch.bfloat16)


@pytest.fixture(scope="module")
def layer():
    from ktransformers_amd._native import GateHandle, LinearHandle, MoEHandle

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    o = {"dev": dev, "gen": g}
    ex = MoEHandle(E, K, H, I, max_len=8, method="AMXINT4", device=0)
    gate_w = torch.empty((E, I, H), dtype=torch.bfloat16, device=dev)
    up_w = torch.empty((E, I, H), dtype=torch.bfloat16, device=dev)
    down_w = torch.empty((E, H, I), dtype=torch.bfloat16, device=dev)
    for e0 in range(0, E, 32):
        gate_w[e0:e0 + 32] = _u((32, I, H), g, dev, 0.02)
        up_w[e0:e0 + 32] = _u((32, I, H), g, dev, 0.02)
        down_w[e0:e0 + 32] = _u((32, H, I), g, dev, 0.04)
    ex.load_bf16(gate_w, up_w, down_w)
    del gate_w, up_w, down_w
    torch.cuda.empty_cache()
    o["experts"] = ex
    o["sgu"] = LinearHandle(H, 2 * I, "W4", 64, 8, dev)
    o["sgu"].load_bf16(_u((2 * I, H), g, dev, 0.02))
    o["sdown"] = LinearHandle(I, H, "W4", 64, 8, dev)
    o["sdown"].load_bf16(_u((H, I), g, dev, 0.04))
    o["gate"] = GateHandle(E, H, K, 8, 4, "sigmoid", "noaux_tc", True, 2.5)
    o["gate_w"] = _u((E, H), g, dev, 0.05)
    o["gate_b"] = ((torch.rand(E, generator=g, device=dev) - 0.5) * 0.2).float().contiguous()
    o["norm_w"] = (1 + _u((H,), g, dev, 0.2).float()).to(torch.bfloat16)
    return o


def _three_launches(o, x):
    from ktransformers_amd._native import gate_with_linear

    idx, wt, xn, act = gate_with_linear(o["gate"], o["sgu"], x, o["gate_w"], o["gate_b"], (o["norm_w"], 1e-6), glu=True)
    y = o["experts"].forward_side(xn, idx, wt, o["sdown"], act, residual=x)
    torch.cuda.synchronize()
    return {"idx": idx.clone(), "wt": wt.clone(), "act": act.clone(), "y": y.clone()}


def _args(o, x, y, idx, wt, phases=7, last=True):
    from ktransformers_amd._native import moe_layer_args

    return moe_layer_args(o["experts"], o["sgu"], o["sdown"], o["gate"], o["gate_w"], o["gate_b"], x.reshape(-1), y.reshape(-1),
                          (o["norm_w"], 1e-6), idx, wt, phases, last)


def _check(o, ref, y, idx, wt, tag):
    from ktransformers_amd._native import moe_layer_debug_read, moe_layer_status

    dev = o["dev"]
    assert moe_layer_status(dev) == 0, f"{tag}: a hand-off timed out"
    act = moe_layer_debug_read(dev, "shared_act", (1, I))
    assert torch.equal(idx.view(-1), ref["idx"].view(-1)), f"{tag}: selected experts differ: {idx.tolist()} vs {ref['idx'].tolist()}"
    assert torch.equal(wt.view(-1), ref["wt"].view(-1)), f"{tag}: routing weights differ"
    bad = int((act.view(torch.int16) != ref["act"].view(torch.int16)).sum())
    assert bad == 0, f"{tag}: {bad} of {I} shared activations differ"
    bad = int((y.view(torch.int16) != ref["y"].view(torch.int16)).sum())
    assert bad == 0, f"{tag}: {bad} of {H} outputs differ from the three-launch path"


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_one_launch_equals_three_launches(layer, seed):
    from ktransformers_amd._native import moe_layer_decode, moe_layer_decode_eligible

    o, dev = layer, layer["dev"]
    o["gen"].manual_seed(100 + seed)
    x = _u((1, H), o["gen"], dev, 1.0 + seed)
    ref = _three_launches(o, x)
    y = torch.zeros((1, H), dtype=torch.bfloat16, device=dev)
    idx = torch.zeros((1, K), dtype=torch.int64, device=dev)
    wt = torch.zeros((1, K), dtype=torch.float32, device=dev)
    a = _args(o, x, y, idx, wt)
    assert moe_layer_decode_eligible(a)
    moe_layer_decode(a, dev)
    torch.cuda.synchronize()
    _check(o, ref, y, idx, wt, f"one launch, seed {seed}")


def test_phase_by_phase_and_masked_experts(layer):
    """The same device code as three launches (and 3 + 4, 1 + 6); then with a routed expert masked out (gpu_experts_mask semantics:
    the slot contributes nothing), which both paths must skip alike."""
    from ktransformers_amd._native import moe_layer_decode

    o, dev = layer, layer["dev"]
    o["gen"].manual_seed(77)
    x = _u((1, H), o["gen"], dev, 1.5)
    ref = _three_launches(o, x)
    for chain in ((1, 2, 4), (3, 4), (1, 6)):
        y = torch.zeros((1, H), dtype=torch.bfloat16, device=dev)
        idx = torch.zeros((1, K), dtype=torch.int64, device=dev)
        wt = torch.zeros((1, K), dtype=torch.float32, device=dev)
        a = _args(o, x, y, idx, wt)
        for i, ph in enumerate(chain):
            moe_layer_decode(a, dev, phases=ph, last=(i == len(chain) - 1))
        torch.cuda.synchronize()
        _check(o, ref, y, idx, wt, f"chain {chain}")
    import numpy as np
    mask = np.zeros(E, dtype=np.uint8)
    mask[int(ref["idx"].view(-1)[2])] = 1
    o["experts"].set_expert_mask(mask)
    try:
        ref2 = _three_launches(o, x)
        assert not torch.equal(ref2["y"], ref["y"])
        y = torch.zeros((1, H), dtype=torch.bfloat16, device=dev)
        idx = torch.zeros((1, K), dtype=torch.int64, device=dev)
        wt = torch.zeros((1, K), dtype=torch.float32, device=dev)
        moe_layer_decode(_args(o, x, y, idx, wt), dev)
        torch.cuda.synchronize()
        _check(o, ref2, y, idx, wt, "masked expert")
    finally:
        o["experts"].set_expert_mask(None)


def test_graph_replay_with_changing_rows(layer):
    from ktransformers_amd._native import moe_layer_decode

    o, dev = layer, layer["dev"]
    x = _u((1, H), o["gen"], dev, 1.0)
    y = torch.zeros((1, H), dtype=torch.bfloat16, device=dev)
    idx = torch.zeros((1, K), dtype=torch.int64, device=dev)
    wt = torch.zeros((1, K), dtype=torch.float32, device=dev)
    a = _args(o, x, y, idx, wt)
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        moe_layer_decode(a, dev)
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        moe_layer_decode(a, dev)
    for step in range(4):
        x.copy_(_u((1, H), o["gen"], dev, 0.5 + step))
        ref = _three_launches(o, x)
        g.replay()
        torch.cuda.synchronize()
        _check(o, ref, y, idx, wt, f"replay {step}")
