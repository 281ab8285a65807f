"""GPU: ONE DeepSeek-V3 decoder layer at the PUBLISHED dimensions — 256 routed experts of 7168 x 2048 (AMXINT4), 128 heads, q_lora
1536, W4-g64 Marlin linears, MLA — injected by the product rule file and compared with the reference's own modules + its own
AMXINT4 cpu_backend kernels on identical weights (tests/golden/make_v3_layer_golden.py; weights rebuilt on the device from
seeds, tests/v3_layer_case.py).  Everything else end to end in this suite runs at toy sizes (hidden 256, 8 experts, 2 heads).

  * the expert block on the rows / expert ids / routing weights the reference's cpu_backend saw: bit-exact, through the grouped
    path (25 rows at once) AND through the two-launch decode path (one row at a time);
  * the router on the reference's router input: identical expert sets for all 25 tokens, weights to 2e-6;
  * the whole layer: a 24-token prompt pass + ONE cached decode step (the all-CU GEMV, q_b + absorb, split-KV MLA decode, merge +
    un-absorb, fused router, decode experts — the launches of the headline benchmark at their real shapes) against the
    reference's fp32-arithmetic run, held to the drift the reference's own bf16 run shows against it."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "v3_layer_golden.npz")
RULES = os.path.join(os.path.dirname(HERE), "ktransformers_amd", "optimize", "optimize_rules", "DeepSeek-V3-Chat.yaml")


@pytest.fixture(scope="module")
def v3_layer():
    from v3_layer_case import CFG, E, L, expert_weight, small_weights
    from ktransformers_amd.models.custom_cache import StaticCache
    from ktransformers_amd.models.modeling_deepseek import DeepseekForCausalLM, make_config
    from ktransformers_amd.optimize.optimize import optimize_and_load
    from ktransformers_amd.util.loader import DictLoader

    dev = "cuda:0"
    state = small_weights(dev)
    for e in range(E):
        for proj in ("gate", "up", "down"):
            state[f"{L}mlp.experts.{e}.{proj}_proj.weight"] = expert_weight(e, proj, dev)
    cfg = make_config(**CFG, architectures=["DeepseekV3ForCausalLM"])
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device("meta"):
            model = DeepseekForCausalLM(cfg)
        optimize_and_load(model, RULES, DictLoader(state), cfg, default_device=dev)
    finally:
        torch.set_default_dtype(torch.float32)
    del state
    torch.cuda.empty_cache()
    mlp = model.model.layers[0].mlp
    assert mlp.experts.generate_experts.method == "AMXINT4"
    assert type(model.model.layers[0].self_attn.orig_module.o_proj.generate_linear).__name__ == "KLinearMarlin"
    cache = StaticCache(cfg, 1, 256, dev, torch.bfloat16)
    return model, cache, np.load(GOLD)


def bf16_rows(bits):
    return torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16).cuda()


def test_expert_block_bit_exact_at_real_dimensions(v3_layer):
    model, _, g = v3_layer
    h = model.model.layers[0].mlp.experts.generate_experts.handle
    x, ids, w, want = bf16_rows(g["moe_x"]), torch.from_numpy(g["moe_ids"]).cuda(), torch.from_numpy(g["moe_w"]).cuda(), g["moe_y"]
    got = h.forward(x.contiguous(), ids.contiguous(), w.contiguous())
    torch.cuda.synchronize()
    got = got.cpu().view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(got, want), f"grouped path: {int((got != want).sum())} of {want.size} bf16 outputs differ"
    for t in range(x.shape[0]):                 # one row at a time: the two-launch decode kernels of the headline benchmark
        y = h.forward(x[t:t + 1].contiguous(), ids[t:t + 1].contiguous(), w[t:t + 1].contiguous())
        torch.cuda.synchronize()
        y = y.cpu().view(torch.int16).numpy().view(np.uint16)
        assert np.array_equal(y[0], want[t]), f"decode path, token {t}: {int((y[0] != want[t]).sum())} outputs differ"


def test_router_picks_the_reference_experts_at_real_dimensions(v3_layer):
    model, _, g = v3_layer
    gate = model.model.layers[0].mlp.gate
    x = bf16_rows(g["moe_x"])
    idx, wt = gate(x[None])
    torch.cuda.synchronize()
    idx, wt = idx.reshape(-1, idx.shape[-1]).cpu().numpy(), wt.reshape(-1, wt.shape[-1]).float().cpu().numpy()
    for t in range(x.shape[0]):
        assert set(idx[t].tolist()) == set(g["moe_ids"][t].tolist()), f"token {t}: routed expert set differs"
        ref = dict(zip(g["moe_ids"][t].tolist(), g["moe_w"][t].tolist()))
        for e, v in zip(idx[t].tolist(), wt[t].tolist()):
            assert abs(v - ref[e]) <= 2e-6 * abs(ref[e]) + 1e-9, (t, e, v, ref[e])


def test_prompt_pass_and_cached_decode_step_track_the_reference(v3_layer):
    from ktransformers_amd.util.generate import set_inference_mode
    from ktransformers_amd.util.utils import InferenceState
    model, cache, g = v3_layer
    T = int(g["t_prompt"])
    ids = torch.from_numpy(g["token_ids"]).cuda()[None]
    ref32 = torch.from_numpy(g["out_f32"])
    ref16 = bf16_rows(g["out_bf16"]).float().cpu()
    got = []
    hook = model.model.layers[0].register_forward_hook(
        lambda m, a, out: got.append((out[0] if isinstance(out, tuple) else out).detach().float().cpu().reshape(-1, ref32.shape[1])))
    try:
        cache.reset()
        with torch.no_grad():
            set_inference_mode(model, InferenceState.PREFILL)
            pos = torch.arange(T, device="cuda")[None]
            model(ids[:, :T], pos, cache, pos[0])
            set_inference_mode(model, InferenceState.GENERATE)
            pos = torch.tensor([[T]], device="cuda")
            model(ids[:, T:T + 1], pos, cache, pos[0])
        torch.cuda.synchronize()
    finally:
        hook.remove()
        set_inference_mode(model, InferenceState.GENERATE)
    out = torch.cat(got)                                          # rows 0..23 from the prompt pass, row 24 from the decode step
    assert tuple(out.shape) == tuple(ref32.shape)
    rel = lambda a: ((a - ref32).norm(dim=1) / ref32.norm(dim=1)).numpy()  # noqa: E731
    ours, theirs = rel(out), rel(ref16)
    # yardstick = the reference's own bf16 run against its fp32 run, token by token.  A routed-expert near-tie that falls the other
    # way moves one token by tens of per cent in EITHER pipeline, so: median within 2.5x, and no more large outliers than the
    # reference's own bf16 run has plus two.
    assert np.median(ours) < 2.5 * np.median(theirs) + 5e-3, (np.round(ours, 4).tolist(), np.round(theirs, 4).tolist())
    big = max(4 * float(np.median(theirs)), 0.05)
    assert int((ours > big).sum()) <= int((theirs > big).sum()) + 2, (np.round(ours, 4).tolist(), np.round(theirs, 4).tolist())
    # the cached decode step (row 24) is as close as the prompt rows are
    assert ours[T] < max(2.5 * float(np.median(theirs)) + 5e-3, 1.5 * float(np.sort(ours[:T])[-3])), (float(ours[T]), np.round(ours, 4).tolist())
