"""Host logic of the drop-in boundary: YAML rule matching, injection, attribute fall-through, op registry
(reference: archive/ktransformers/optimize/optimize.py:28-118, operators/base_operator.py:12-63, experts.py:680-757)."""
import os

import pytest
import torch

from toy_model import ToyConfig, ToyModel

from ktransformers_amd.operators.base_operator import BaseInjectedModule
from ktransformers_amd.operators.experts import (EXPERTS_MAP, KDeepseekV3MoE, KExpertsHIP, KTransformersExperts)
from ktransformers_amd.operators.gate import KMoEGate
from ktransformers_amd.optimize.optimize import gen_optimize_config, load_rules, optimize_and_load, resolve_class
from ktransformers_amd.util.loader import DictLoader
from ktransformers_amd.util.utils import InferenceState

RULES = os.path.join(os.path.dirname(__file__), "toy_rules.yaml")


@pytest.fixture()
def injected():
    cfg = ToyConfig()
    with torch.device("meta"):
        model = ToyModel(cfg)
    loader = DictLoader({})
    conf = optimize_and_load(model, RULES, loader, cfg, default_device="cuda:0", load=False)
    return model, conf, cfg


def test_reference_class_paths_resolve_to_the_mirror():
    assert resolve_class("ktransformers.operators.experts.KTransformersExperts") is KTransformersExperts
    assert resolve_class("ktransformers.operators.gate.KMoEGate") is KMoEGate
    with pytest.raises(ImportError):
        resolve_class("ktransformers.operators.nope.Missing")


def test_rules_pick_first_match_and_respect_recursive(injected):
    model, conf, _ = injected
    assert conf["model.layers.1.mlp"]["class"].endswith("KDeepseekV3MoE")
    assert conf["model.layers.0.mlp"]["class"] == "default"          # dense layer: not a ToyMoE
    assert conf["model.layers.1.mlp.experts"]["class"].endswith("KTransformersExperts")
    assert "model.layers.1.mlp.experts.0" not in conf                # recursive: False stops the descent
    assert conf["lm_head"]["class"] == "default" and conf["lm_head"]["kwargs"]["generate_device"] == "cuda"
    assert conf["model.layers.2.mlp.gate"]["kwargs"]["generate_device"] == "cuda:0"


def test_injected_module_types_and_fallthrough(injected):
    model, _, cfg = injected
    mlp = model.model.layers[1].mlp
    assert isinstance(mlp, KDeepseekV3MoE) and isinstance(mlp, BaseInjectedModule)
    assert isinstance(mlp.experts, KTransformersExperts)
    assert isinstance(mlp.gate, KMoEGate)
    assert mlp.key == "model.layers.1.mlp" and mlp.experts.key == "model.layers.1.mlp.experts"
    # attribute fall-through to the replaced module
    assert mlp.gate.top_k == cfg.num_experts_per_tok
    assert mlp.gate.orig_module.weight.shape == (cfg.n_routed_experts, cfg.hidden_size)
    assert mlp.config is cfg
    # writes to unknown attributes land on the original module (base_operator.py:52-56)
    mlp.gate.some_flag = 7
    assert mlp.gate.orig_module.some_flag == 7


def test_experts_registry_and_modes(injected):
    model, _, _ = injected
    ex = model.model.layers[2].mlp.experts
    assert EXPERTS_MAP["KExpertsCPU"] is KExpertsHIP and EXPERTS_MAP["KExpertsTorch"] is KExpertsHIP
    assert isinstance(ex.generate_experts, KExpertsHIP) and ex.prefill_experts is ex.generate_experts
    assert ex.generate_experts.n_routed_experts == 8 and ex.generate_experts.method == "AMXINT4"
    assert ex.mode == InferenceState.UNLOAD
    with pytest.raises(ValueError):
        ex.forward(torch.zeros(1, 256), torch.zeros(1, 2, dtype=torch.long), torch.zeros(1, 2))
    with pytest.raises(ValueError):
        ex.set_inference_mode("bogus")


def test_unknown_backend_is_rejected():
    with pytest.raises(ValueError):
        KExpertsHIP("k", DictLoader({}), ToyConfig(), 8, backend="Q2_K")


def test_kexperts_hip_has_no_cpu_path():
    e = KExpertsHIP("k", DictLoader({}), ToyConfig(), 8, device="cpu", out_device="cpu")
    with pytest.raises(RuntimeError):
        e.load({"gate": torch.zeros(8, 128, 256), "up": torch.zeros(8, 128, 256), "down": torch.zeros(8, 256, 128)})


def test_rule_without_match_keys_raises():
    with pytest.raises(Exception):
        gen_optimize_config(ToyModel(ToyConfig()), {}, [{"match": {}, "replace": {"class": "default"}}])


def test_shipped_rule_files_parse():
    d = os.path.join(os.path.dirname(os.path.dirname(__file__)), "ktransformers_amd", "optimize", "optimize_rules")
    for f in os.listdir(d):
        rules = load_rules(os.path.join(d, f))
        assert isinstance(rules, list) and all("match" in r and "replace" in r for r in rules)


def test_linear_injection_and_registry(injected):
    from ktransformers_amd.operators.linear import (LINEAR_MAP, KLinearFP8, KLinearMarlin, KLinearTorch,
                                                     KTransformersLinear)
    model, conf, cfg = injected
    assert resolve_class("ktransformers.operators.linear.KTransformersLinear") is KTransformersLinear
    lin = model.model.layers[0].mlp.gate_proj
    assert isinstance(lin, KTransformersLinear) and lin.key == "model.layers.0.mlp.gate_proj"
    assert conf["model.layers.0.mlp.down_proj"]["kwargs"]["generate_op"] == "KLinearMarlin"
    assert isinstance(lin.generate_linear, KLinearMarlin) and lin.generate_linear.group_size == 64
    assert lin.prefill_linear is lin.generate_linear            # one quantised copy serves prefill and decode
    assert (lin.in_features, lin.out_features) == (cfg.hidden_size, cfg.intermediate_size)
    assert lin.mode == InferenceState.UNLOAD
    assert LINEAR_MAP["VLinearMarlin"] is KLinearMarlin and LINEAR_MAP["KLinearFP8"] is KLinearFP8
    assert LINEAR_MAP["KLinearCPUInfer"] is KLinearTorch
    with pytest.raises(RuntimeError):
        lin.generate_linear.forward(torch.zeros(1, cfg.hidden_size))     # no silent fallback before load()
    with pytest.raises(AssertionError):
        KTransformersLinear("k", DictLoader({}), cfg, torch.nn.Linear(8, 8), generate_op="Nope")
    with pytest.raises(NotImplementedError):
        KLinearMarlin("k", DictLoader({}), cfg, torch.nn.Linear(8, 8), num_bits=3)          # quant_utils.py:5: 4 or 8
    m8 = KLinearMarlin("k", DictLoader({}), cfg, torch.nn.Linear(8, 8), num_bits=8, act_order=True)
    assert (m8.num_bits, m8.act_order, m8.FMT) == (8, True, "BF16")
    with pytest.raises(ValueError):
        lin.set_inference_mode("bogus")


def test_static_cache_bookkeeping_on_cpu():
    """The parts of StaticCache that need no kernel: page table, length bookkeeping, prefix truncation
    (archive/ktransformers/models/custom_cache.py:45-254)."""
    import types

    import torch

    from ktransformers_amd.models.custom_cache import StaticCache
    cfg = types.SimpleNamespace(max_position_embeddings=4096, kv_lora_rank=512, qk_rope_head_dim=64, num_hidden_layers=3)
    c = StaticCache(cfg, 2, 200, "cpu", torch.bfloat16)
    assert c.page_size == 64 and c.max_pages == 4 and c.key_cache[0].shape == (4, 64, 1, 576) and c.value_cache[0] is None
    assert c.page_table_list[0].tolist() == [[0, 1, 2, 3], [4, 5, 6, 7]] and c.page_table_list[1] is c.page_table_list[0]
    assert c.max_cache_len == 200 and c.get_max_length() == 200 and c.get_max_cache_shape() == 200 and c.max_batch_size == 2
    assert StaticCache(cfg, 1, None, "cpu").max_cache_len == 4096
    c.note_appended(1, 5)
    c.change_seq_length(2)
    assert [c.get_seq_length(i) for i in range(3)] == [2, 7, 2] and c.get_usable_length(9, 1) == 0
    c.key_cache[2].fill_(1)
    c.remove_suffix(70)
    flat = c.key_cache[2].view(-1, 576).float()
    assert bool((flat[:70] == 1).all()) and bool((flat[70:] == 0).all()) and c.get_seq_length(2) == 70
    c.reset()
    assert c.get_seq_length(0) == 0 and float(c.key_cache[2].float().abs().sum()) == 0
    import pytest
    with pytest.raises(ValueError):
        StaticCache(cfg, 1, 64, "cpu", torch.float16)


REF_RULES = "/root/reference/archive/ktransformers/optimize/optimize_rules"


@pytest.mark.skipif(not os.path.isdir(REF_RULES), reason="needs the reference checkout (build container only)")
@pytest.mark.parametrize("rule_file,v3", [("DeepSeek-V3-Chat.yaml", True), ("DeepSeek-V3-Chat-amx.yaml", True),
                                          ("DeepSeek-V3-Chat-fp8-linear-ggml-experts.yaml", True), ("Moonlight-16B-A3B.yaml", True),
                                          ("DeepSeek-V2-Lite-Chat.yaml", False), ("DeepSeek-V2-Chat.yaml", False),
                                          ("DeepSeek-V3-Chat-serve.yaml", True), ("Moonlight-16B-A3B-serve.yaml", True),
                                          ("DeepSeek-V3-Chat-fp8-linear-ggml-experts-serve.yaml", True)])
def test_reference_rule_files_inject_unmodified(rule_file, v3):
    """The drop-in claim at the YAML level: the reference's OWN single-GPU DeepSeek rule files, read where they lie, resolve
    every class to this package's mirrors and inject into the skeleton model (meta device, no weights)."""
    import contextlib
    import io

    from ktransformers_amd.models.modeling_deepseek import DeepseekForCausalLM, make_config
    from ktransformers_amd.operators.attention import KDeepseekV2Attention
    from ktransformers_amd.operators.experts import KDeepseekV2MoE
    from ktransformers_amd.operators.linear import KTransformersLinear
    from ktransformers_amd.operators.models import KDeepseekV2Model
    base = dict(vocab_size=512, hidden_size=128, intermediate_size=256, moe_intermediate_size=128, num_hidden_layers=2,
                num_attention_heads=2, n_shared_experts=1, n_routed_experts=8, num_experts_per_tok=2, first_k_dense_replace=1,
                moe_layer_freq=1, n_group=2, topk_group=1, routed_scaling_factor=2.5, kv_lora_rank=512, qk_rope_head_dim=64,
                qk_nope_head_dim=128, v_head_dim=128, max_position_embeddings=4096, rope_theta=10000.0, rms_norm_eps=1e-6,
                attention_bias=False, rope_scaling={"type": "yarn", "factor": 40, "mscale": 1.0, "mscale_all_dim": 1.0,
                                                    "original_max_position_embeddings": 4096, "beta_fast": 32, "beta_slow": 1})
    if v3:
        base.update(topk_method="noaux_tc", scoring_func="sigmoid", norm_topk_prob=True, q_lora_rank=64, architectures=["DeepseekV3ForCausalLM"])
    else:
        base.update(topk_method="group_limited_greedy", scoring_func="softmax", norm_topk_prob=False, q_lora_rank=None,
                    architectures=["DeepseekV2ForCausalLM"])
    cfg = make_config(**base)
    with torch.device("meta"):
        model = DeepseekForCausalLM(cfg)
    with contextlib.redirect_stdout(io.StringIO()):
        optimize_and_load(model, os.path.join(REF_RULES, rule_file), DictLoader({}), cfg, default_device="cuda:0", load=False)
    moe_layer = model.model.layers[1]
    assert isinstance(model.model, KDeepseekV2Model) and model.model.per_layer_prefill_intput_threshold == 0
    if "serve" in rule_file:   # the balance_serve engine's operator set (bsz_tensor / page-table contracts)
        from ktransformers_amd.operators.balance_serve_attention import flashinfer_attn
        from ktransformers_amd.operators.experts import KDeepseekV3MoEV2, KTransformersExpertsV2
        from ktransformers_amd.operators.layernorm import RMSNorm
        assert isinstance(moe_layer.mlp, KDeepseekV3MoEV2) and isinstance(moe_layer.mlp.experts, KTransformersExpertsV2)
        assert isinstance(moe_layer.self_attn, flashinfer_attn) and isinstance(moe_layer.input_layernorm, RMSNorm)
        return
    assert isinstance(moe_layer.mlp, KDeepseekV3MoE if v3 else KDeepseekV2MoE)
    assert isinstance(moe_layer.mlp.experts, KTransformersExperts)
    assert isinstance(moe_layer.mlp.gate, KMoEGate) == v3     # the V2 rule files leave the router to the model's own module
    assert isinstance(moe_layer.self_attn, KDeepseekV2Attention)
    assert isinstance(moe_layer.self_attn.o_proj, KTransformersLinear)
    assert isinstance(model.model.layers[0].mlp.down_proj, KTransformersLinear)
    assert not isinstance(moe_layer.self_attn.kv_b_proj, KTransformersLinear)     # kept dense for the absorb (rule regex)


def test_serving_cache_page_table_matches_reference():
    """KDeepSeekV3Cache.get_page_table (vectorised here) against the reference's own function on seeded scheduler states
    (tests/golden/serve_cache_golden.npz from tests/golden/make_serve_cache_golden.py)."""
    import types

    import numpy as np

    from ktransformers_amd.models.custom_cache import KDeepSeekV3Cache
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "serve_cache_golden.npz"))
    cfg = types.SimpleNamespace(kv_lora_rank=512, qk_rope_head_dim=64, num_hidden_layers=2)
    keys = sorted(k[:-3] for k in g.files if k.endswith("_in"))
    assert len(keys) == 9
    for key in keys:
        page_size, b = (int(v) for v in g[key + "_in"])
        c = KDeepSeekV3Cache(cfg, page_size=page_size, device="cpu")
        t = lambda n: torch.from_numpy(g[f"{key}_{n}"])
        pi, po = c.get_page_table(t("pos"), t("q_indptr"), t("kv_indptr"), t("kv_indices"), torch.tensor([b]))
        assert torch.equal(pi, t("page_idx")) and torch.equal(po, t("page_offset")), key
    c.allocate(3)
    assert len(c.k_caches) == 2 and c.k_caches[0].shape == (3, 256, 1, 576) and c.max_cache_len == 768
    c.load(types.SimpleNamespace(k_cache=[[torch.zeros(2, 256, 1, 576, dtype=torch.bfloat16)] * 2]))
    assert c.max_cache_len == 512
    assert resolve_class("ktransformers.operators.balance_serve_attention.flashinfer_attn").__name__ == "flashinfer_attn"
    assert resolve_class("ktransformers.models.custom_cache.KDeepSeekV3Cache") is KDeepSeekV3Cache


def test_dict_and_safetensor_loaders(tmp_path):
    """The archive-style weight sources (custom_loader.py:96-250 protocol): per-expert stacking for bf16, block-fp8 and
    compressed-tensors int4 checkpoints, the short `up/gate/down` stems, load_gate, and the lazy safetensors variant."""
    from safetensors.torch import save_file

    from ktransformers_amd.util.loader import SafeTensorLoader
    g = torch.Generator().manual_seed(0)
    E, H, I = 3, 64, 32
    st = {}
    for e in range(E):
        for p, (n, k) in (("gate", (I, H)), ("up", (I, H)), ("down", (H, I))):
            st[f"L.mlp.experts.{e}.{p}_proj.weight"] = torch.randn(n, k, generator=g).to(torch.bfloat16)
            st[f"F.mlp.experts.{e}.{p}_proj.weight"] = torch.randint(0, 255, (n, k), generator=g, dtype=torch.uint8)
            st[f"F.mlp.experts.{e}.{p}_proj.weight_scale_inv"] = torch.rand(1, 1, generator=g)
            st[f"Q.mlp.experts.{e}.{p}_proj.weight_packed"] = torch.randint(0, 255, (n, k // 2), generator=g, dtype=torch.uint8)
            st[f"Q.mlp.experts.{e}.{p}_proj.weight_scale"] = torch.rand(n, k // 32, generator=g).to(torch.bfloat16)
            st[f"S.experts.{e}.{p}.weight"] = torch.randn(n, k, generator=g).to(torch.bfloat16)
    st["L.mlp.gate.weight"] = torch.randn(E, H, generator=g)
    st["L.mlp.gate.e_score_correction_bias"] = torch.randn(E, generator=g)
    save_file(st, str(tmp_path / "m.safetensors"))
    for ld in (DictLoader(st), SafeTensorLoader(str(tmp_path))):
        w = ld.load_experts("L.mlp.experts")
        assert w["gate"].shape == (E, I, H) and w["down"].shape == (E, H, I) and set(w) == {"gate", "up", "down"}
        assert torch.equal(w["up"][2], st["L.mlp.experts.2.up_proj.weight"])
        f = ld.load_experts("F.mlp.experts")
        assert f["gate"].dtype == torch.uint8 and f["down_scale"].shape == (E, 1, 1)
        q = ld.load_experts("Q.mlp.experts")                     # no `.weight` key at all: only weight_packed
        assert q["gate"].shape == (E, I, H // 2) and q["gate_scale"].dtype == torch.bfloat16 and ld.get_expert_count("Q.mlp.experts") == E
        s2 = ld.load_experts("S.experts")
        assert torch.equal(s2["down"][1], st["S.experts.1.down.weight"])
        gate = ld.load_gate("L.mlp.gate")
        assert torch.equal(gate["weight"], st["L.mlp.gate.weight"]) and gate["e_score_correction_bias"].shape == (E,)
        assert ld.load_gate("F.mlp.gate") == {"weight": None, "e_score_correction_bias": None}
        with pytest.raises(ValueError, match="No experts found"):
            ld.load_experts("nope")
        assert ld.has_tensor("L.mlp.gate.weight") and not ld.has_tensor("zzz")


def _mixtral_skeleton():
    from ktransformers_amd.models.modeling_mixtral import MixtralForCausalLM, make_mixtral_config
    cfg = make_mixtral_config(vocab_size=64, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                              num_key_value_heads=2, max_position_embeddings=512, rope_theta=10000.0)
    with torch.device("meta"):
        return cfg, MixtralForCausalLM(cfg)


@pytest.mark.parametrize("rule_file", [os.path.join(REF_RULES, "Mixtral.yaml"),
                                       os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ktransformers_amd",
                                                    "optimize", "optimize_rules", "Mixtral-8x7B.yaml")])
def test_mixtral_rule_files_inject(rule_file):
    """BASELINE config C1's model: the reference's own Mixtral.yaml (read where it lies) and this package's rule file inject into
    the Mixtral skeleton — rotary embedding, every linear of the layers (the router's nn.Linear included, as in the reference),
    lm_head, the sparse MoE block and its experts; the attention core and the norms stay the skeleton's."""
    import contextlib
    import io

    from ktransformers_amd.models.modeling_mixtral import MixtralAttention, MixtralRMSNorm
    from ktransformers_amd.operators.experts import KExpertsHIP, KMistralSparseMoEBlock
    from ktransformers_amd.operators.linear import KTransformersLinear
    from ktransformers_amd.operators.RoPE import RotaryEmbedding
    if not os.path.exists(rule_file):
        pytest.skip("needs the reference checkout (build container only)")
    cfg, model = _mixtral_skeleton()
    with contextlib.redirect_stdout(io.StringIO()):
        optimize_and_load(model, rule_file, DictLoader({}), cfg, default_device="cuda:0", load=False)
    layer = model.model.layers[1]
    assert isinstance(layer.block_sparse_moe, KMistralSparseMoEBlock) and layer.block_sparse_moe.top_k == 2
    ex = layer.block_sparse_moe.experts
    assert isinstance(ex, KTransformersExperts) and isinstance(ex.generate_experts, KExpertsHIP)
    assert ex.generate_experts.method == "GGUF" and ex.generate_experts.n_routed_experts == 8        # llamafile: the reference's default backend
    assert isinstance(layer.block_sparse_moe.gate, KTransformersLinear)
    assert isinstance(layer.self_attn, MixtralAttention) and isinstance(layer.self_attn.rotary_emb, RotaryEmbedding)
    for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
        assert isinstance(getattr(layer.self_attn, n), KTransformersLinear)
    assert isinstance(model.lm_head, KTransformersLinear) and isinstance(layer.input_layernorm, MixtralRMSNorm)
    assert not isinstance(model.model.embed_tokens, KTransformersLinear)
