"""GPU, end to end: the YAML-injected model (models/modeling_deepseek.py skeleton + operators) against the reference's own
pure-torch decoder layers run on CPU (tests/golden/make_model_golden.py): same weights, same prompt.

With un-quantised back-ends the two compute the same function in bf16 with different op orders, so:
  * logits: norm-wise <= 2e-2 of the reference's fp32 run (the reference's own bf16 run is 1e-2 away);
  * greedy tokens: identical wherever the reference's top-2 logit margin exceeds 4x the observed logit error;
  * decode (one token per step through the paged cache, and through a captured HIP graph) reproduces the prompt pass.
The quantised back-ends are covered bit-exactly per operator (tests/test_moe_gpu.py, test_linear_gpu.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "model_golden.npz")
CFG = dict(vocab_size=512, hidden_size=128, intermediate_size=256, moe_intermediate_size=128, num_hidden_layers=2,
           num_attention_heads=2, n_shared_experts=1, n_routed_experts=8, num_experts_per_tok=2, first_k_dense_replace=1,
           moe_layer_freq=1, n_group=2, topk_group=1, topk_method="noaux_tc", scoring_func="sigmoid", norm_topk_prob=True,
           routed_scaling_factor=2.5, q_lora_rank=64, kv_lora_rank=512, qk_rope_head_dim=64, qk_nope_head_dim=128,
           v_head_dim=128, max_position_embeddings=4096, rope_theta=10000.0, rms_norm_eps=1e-6, attention_bias=False,
           rope_scaling={"type": "yarn", "factor": 40, "mscale": 1.0, "mscale_all_dim": 1.0,
                         "original_max_position_embeddings": 4096, "beta_fast": 32, "beta_slow": 1},
           architectures=["DeepseekV3ForCausalLM"])


@pytest.fixture(scope="module")
def model_and_gold():
    from ktransformers_amd.models.custom_cache import StaticCache
    from ktransformers_amd.models.modeling_deepseek import DeepseekForCausalLM, make_config
    from ktransformers_amd.optimize.optimize import optimize_and_load
    from ktransformers_amd.util.loader import DictLoader

    g = np.load(GOLD)
    state = {k[2:]: torch.from_numpy(g[k]).view(torch.bfloat16) for k in g.files if k.startswith("w.")}
    cfg = make_config(**CFG)
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device("meta"):
            model = DeepseekForCausalLM(cfg)
        optimize_and_load(model, os.path.join(HERE, "model_rules_bf16.yaml"), DictLoader(state), cfg, default_device="cuda:0")
    finally:
        torch.set_default_dtype(torch.float32)
    cache = StaticCache(cfg, 1, 256, "cuda:0", torch.bfloat16)
    return model, cache, g


def test_prompt_logits_and_greedy_tokens_match_reference(model_and_gold):
    from ktransformers_amd.util.generate import set_inference_mode
    from ktransformers_amd.util.utils import InferenceState
    model, cache, g = model_and_gold
    ids = torch.from_numpy(g["input_ids"]).cuda()[None]
    T = ids.shape[1]
    ref = torch.from_numpy(g["logits_f32"])
    set_inference_mode(model, InferenceState.PREFILL)
    cache.reset()
    pos = torch.arange(T, device="cuda")[None]
    with torch.no_grad():
        logits = model(ids, pos, cache, pos[0])[0].cpu()
    rel = float((logits - ref).norm() / ref.norm())
    assert rel < 2e-2, rel
    err = float((logits - ref).abs().max())
    clear = torch.from_numpy(g["margin_f32"]) > 4 * err
    assert clear.sum() >= T // 2
    assert torch.equal(logits.argmax(-1)[clear], ref.argmax(-1)[clear])

    # decode: feed the same tokens one at a time through the paged cache
    set_inference_mode(model, InferenceState.GENERATE)
    cache.reset()
    outs = []
    with torch.no_grad():
        for t in range(T):
            p = torch.tensor([[t]], device="cuda")
            outs.append(model(ids[:, t:t + 1], p, cache, p[0])[0, 0].cpu())
    dec = torch.stack(outs)
    # greedy_next_token (lm_head -> one-launch argmax on the bf16 logits) picks forward()'s argmax
    cache.reset()
    with torch.no_grad():
        for t in range(T):
            p = torch.tensor([[t]], device="cuda")
            assert int(model.greedy_next_token(ids[:, t:t + 1], p, cache, p[0])[0]) == int(dec[t].argmax())
    assert float((dec - ref).norm() / ref.norm()) < 2e-2
    assert float((dec - logits).norm() / logits.norm()) < 1.5e-2
    assert torch.equal(dec.argmax(-1)[clear], ref.argmax(-1)[clear])


def test_generation_through_a_captured_graph_equals_eager(model_and_gold):
    from ktransformers_amd.util.generate import prefill_and_generate
    model, cache, g = model_and_gold
    ids = torch.from_numpy(g["input_ids"]).cuda()[None]
    cache.reset()
    eager, le = prefill_and_generate(model, ids, cache, max_new_tokens=6, use_cuda_graph=False, return_logits=True)
    cache.reset()
    graph, lg = prefill_and_generate(model, ids, cache, max_new_tokens=6, use_cuda_graph=True, return_logits=True)
    assert torch.equal(le, lg), "graph replay must be bit-identical to eager launches"
    assert torch.equal(eager, graph)
    # first generated token = argmax of the reference's last prompt position (when its margin is clear)
    ref_last = torch.from_numpy(g["logits_f32"])[-1]
    if float(g["margin_f32"][-1]) > 0.1:
        assert int(eager[0]) == int(ref_last.argmax())


def test_serving_variant_block_honours_bsz_tensor(model_and_gold):
    """KDeepseekV3MoEV2.forward(hidden, bsz_tensor, cuda_graph_idx) (experts.py:1172-1213) on the injected MoE block: with all
    rows valid it equals the single-request block, with fewer the valid rows are unchanged."""
    from ktransformers_amd.operators.experts import KDeepseekV3MoEV2
    from ktransformers_amd.util.generate import set_inference_mode
    from ktransformers_amd.util.utils import InferenceState
    model, _, _ = model_and_gold
    set_inference_mode(model, InferenceState.GENERATE)
    blk = model.model.layers[1].mlp
    T, H = 6, CFG["hidden_size"]
    x = (torch.randn(1, T, H, generator=torch.Generator().manual_seed(0)) * 0.5).to(torch.bfloat16).cuda()
    with torch.no_grad():
        full = blk(x).clone()
        allrows = KDeepseekV3MoEV2.forward(blk, x, torch.tensor([T], dtype=torch.int32, device="cuda"), 0).clone()
        part = KDeepseekV3MoEV2.forward(blk, x, torch.tensor([4], dtype=torch.int32, device="cuda"), 0).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(full.float()).all() and full.float().abs().max() > 0
    assert torch.equal(allrows, full)
    assert torch.equal(part[:, :4], full[:, :4])
