#!/usr/bin/env python3
"""bench.py — decode / prefill throughput of the MI355X-native quantized-MoE + MLA hot path.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1: launched under torch.distributed.run)
prints ONE JSON line on rank 0.

Workload (BASELINE.json `metric`: DeepSeek-V3 671B int4).  The full model is 327 GB of int4 experts and does not fit one
288 GB GPU, so N=1 runs the LARGEST LAYER SUBSET that fits with room for the bf16 staging of one layer: DeepSeek-V3
dimensions (H=7168, 128 heads, q_lora 1536, kv_lora 512 + rope 64, 256 routed experts top-8 + 1 shared, I=2048, sigmoid
noaux_tc router, vocab 129280), the 3 dense layers + `--layers - 3` of the 58 MoE layers (default 32 layers in all), every
layer with its OWN weights (never re-used), AMXINT4 ("int4") routed experts + W4-g64 (Marlin semantics) linears, all
resident in HBM; synthetic seeded weights.  The same model runs at every N (experts sharded E/N per rank, expert parallel).

One "step" = one greedy decode token (batch 1 per GPU) through the WHOLE YAML-injected decoder stack — every §8(a) row:
embedding, RMSNorm, MLA attention operator (projections, YaRN RoPE, absorb, paged MQA over --ctx cached tokens, cache
append), router, routed + shared experts, lm_head, argmax — replayed as one HIP graph; the sampled token and the position
are fed back inside the graph.  value = tokens/s of the whole job.
N>1: every rank decodes its own token stream (weak scaling); attention / dense parts are replicated, the routed experts are
sharded E/N per rank (expert parallel): per MoE layer all-gather [x, ids, w] -> local experts -> reduce-scatter (RCCL).

Extra fields at N=1:
  per_kernel / roofline : real decode steps are re-run as plain launches with every library kernel bracketed by two HIP
      events on the launch stream (the whole step is enqueued behind L3-flushing traffic, so the kernels run back to back on
      cold caches as in the graph).  `roofline` is the kernel class with the largest measured total per step (not a
      hard-coded name); `traffic` comes from rocprofv3 --pmc child passes lined up with the library's launch log.
  prefill               : a --prefill-tokens prompt chunk through the whole resident model (prefill path of every operator)
  v2lite                : BASELINE.json configs[1] (DeepSeek-V2-Lite, whole model) for continuity with round 1
  cpu_baseline          : the reference's own AVX512 kernels (oracle/_ref) on the host cores, same expert shape
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

YARN = {"type": "yarn", "factor": 40, "original_max_position_embeddings": 4096, "beta_fast": 32, "beta_slow": 1}
MODELS = {
    # field names of DeepseekV3Config (archive/ktransformers/models/configuration_deepseek_v3.py:106-131)
    "v3": dict(vocab_size=129280, hidden_size=7168, intermediate_size=18432, moe_intermediate_size=2048,
               num_attention_heads=128, n_shared_experts=1, n_routed_experts=256, num_experts_per_tok=8,
               first_k_dense_replace=3, n_group=8, topk_group=4, topk_method="noaux_tc", scoring_func="sigmoid",
               norm_topk_prob=True, routed_scaling_factor=2.5, q_lora_rank=1536, kv_lora_rank=512, qk_rope_head_dim=64,
               qk_nope_head_dim=128, v_head_dim=128, max_position_embeddings=163840,
               rope_scaling=dict(YARN, mscale=1.0, mscale_all_dim=1.0), architectures=["DeepseekV3ForCausalLM"]),
    "v2lite": dict(max_position_embeddings=163840, rope_scaling=dict(YARN, mscale=0.707, mscale_all_dim=0.707)),
}
MODELS["k2"] = dict(MODELS["v3"], vocab_size=163840, num_attention_heads=64, n_routed_experts=384, n_group=1, topk_group=1,
                    first_k_dense_replace=1, routed_scaling_factor=2.827,
                    rope_scaling=dict(YARN, factor=32, mscale=1.0, mscale_all_dim=1.0))   # kt-kernel/bench/bench_k2_moe_amx.py:22-30
WORKLOADS = {
    # H/I/E/k/L/method: the routed-expert shape (kernel-level scripts under scripts/ build stand-alone layers from these).
    # linear: format of the dense linears (rule `generate_op`), experts: backend of the routed experts (rule `backend`).
    "v3-int4": dict(model="v3", rules="DeepSeek-V3-Chat.yaml", layers=32, full_layers=61, dense=3,
                    H=7168, I=2048, E=256, k=8, L=8, method="AMXINT4", heads=128, linear="W4", experts="AMXInt4",
                    name="DeepSeek-V3 671B int4",
                    desc="DeepSeek-V3 dims, AMXINT4 routed experts + W4-g64 linears + MLA, layer subset with distinct "
                         "resident weights per layer (the 671B model is 327 GB of int4 experts: > 288 GB)"),
    "v2lite-int4": dict(model="v2lite", rules="DeepSeek-V2-Lite-Chat.yaml", layers=27, full_layers=27, dense=1,
                        H=2048, I=1408, E=64, k=6, L=26, method="AMXINT4", heads=16, linear="W4", experts="AMXInt4",
                        name="DeepSeek-V2-Lite int4",
                        desc="DeepSeek-V2-Lite 16B (whole model), AMXINT4 routed experts + W4-g64 linears + MLA"),
    # BASELINE.json configs[4]: the reference's DeepSeek-V3-Chat-fp8-linear-ggml-experts.yaml combination — GGUF IQ1_S routed
    # experts (llamafile backend) + block-fp8 linears (KLinearFP8) — the one V3-class model that fits one GPU WHOLE
    "r1-iq1s": dict(model="v3", rules="DeepSeek-V3-Chat.yaml", layers=61, full_layers=61, dense=3,
                    H=7168, I=2048, E=256, k=8, L=8, method="GGUF", heads=128, linear="FP8", experts="llamafile", ggml=(19, 19, 19),
                    name="DeepSeek-R1 IQ1_S/fp8 hybrid",
                    desc="DeepSeek-R1 (= V3 dims), WHOLE 61-layer model: GGUF IQ1_S routed experts (llamafile arithmetic) + "
                         "block-fp8 linears (KLinearFP8) + MLA"),
    # configs[2]: block-fp8 experts AND linears; 11.3 GB of experts per MoE layer -> layer subset
    "v3-fp8": dict(model="v3", rules="DeepSeek-V3-Chat.yaml", layers=20, full_layers=61, dense=3,
                   H=7168, I=2048, E=256, k=8, L=8, method="FP8", heads=128, linear="FP8", experts="FP8",
                   name="DeepSeek-V3 671B fp8",
                   desc="DeepSeek-V3 dims, block-fp8 (e4m3, 128x128 scales) routed experts + block-fp8 linears + MLA, layer subset"),
    # configs[3]: Kimi-K2 dims (384 experts, 64 heads), RAWINT4 experts (compressed-tensors int4 g32), W4 linears
    "k2-rawint4": dict(model="k2", rules="DeepSeek-V3-Chat.yaml", layers=22, full_layers=61, dense=1,
                       H=7168, I=2048, E=384, k=8, L=8, method="RAWINT4", heads=64, linear="W4", experts="RAWINT4",
                       name="Kimi-K2 1T int4",
                       desc="Kimi-K2 dims (384 routed experts, 64 heads), RAWINT4 (int4 g32, bf16 scales) routed experts + "
                            "W4-g64 linears + MLA, layer subset (one GPU holds 1/8 of an expert-parallel deployment's layers whole)"),
    # configs[0]: kt-kernel/bench/bench_moe.py — the routed experts of Mixtral-8x7B alone, q4_k_m (Q4_K gate/up, Q6_K down)
    "mixtral-q4km": dict(kind="experts", layers=32, H=4096, I=14336, E=8, k=2, L=32, method="GGUF", ggml=(12, 12, 14),
                         name="Mixtral-8x7B q4_k_m experts",
                         desc="Mixtral-8x7B routed experts only (bench_moe.py's scope): 32 layers of 8 experts, top-2, GGUF q4_k_m "
                              "(Q4_K gate/up, Q6_K down), llamafile arithmetic"),
}
SECONDARY = ("v2lite-int4", "r1-iq1s", "v3-fp8", "k2-rawint4", "mixtral-q4km")
PREFILL_LONG = 8192      # the reference's default prompt chunk (archive/ktransformers/local_chat.py:86, chunk_size = 8192)
# stored bytes per weight (+ scales) of the formats, for the algorithmic-bytes figures
EXPERT_BPW = {"AMXINT4": 0.5, "AMXINT8": 1.0, "RAWINT4": 0.5 + 2 / 32, "FP8": 1.0 + 4 / 16384, "BF16": 2.0}
LINEAR_BPW = {"W4": 0.5 + 2 / 64, "FP8": 1.0 + 4 / 16384, "BF16": 2.0}
GGML_BLOCK = {12: 144, 14: 210, 19: 50}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured by a float4 copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------------------------------
# stand-alone expert layers (kernel-level dev scripts; not part of the default bench line)
# ---------------------------------------------------------------------------------------------------------------------
def build_layers(wl, dev, max_len, expert_begin=0, expert_num=None, seed=0):
    from ktransformers_amd._native import MoEHandle

    H, I, E, k, L = wl["H"], wl["I"], wl["E"], wl["k"], wl["L"]
    e_local = expert_num or E
    layers = []
    g = torch.Generator(device=dev)
    for li in range(L):
        g.manual_seed(seed * 1000 + li)
        h = MoEHandle(e_local, k, H, I, max_len=max_len, method=wl["method"], device=dev.index,
                      expert_begin=expert_begin, global_expert_num=E)
        if wl["method"] == "GGUF":          # raw ggml blocks (random valid blocks: timing only, like bench_moe.py:197-224)
            ty = wl["ggml"]
            h.load_gguf(random_ggml_blocks(e_local, I, H, ty[0], g, dev), random_ggml_blocks(e_local, I, H, ty[1], g, dev),
                        random_ggml_blocks(e_local, H, I, ty[2], g, dev), *ty)
            layers.append(h)
            continue
        # randn/10 bf16 weights (reference tests: test_moe_rawint4_accuracy.py:175-183), quantised by the GPU restatement
        # of the reference quantiser.  All E experts are generated so every EP rank sees the same global weights.
        gate = torch.randn((E, I, H), generator=g, device=dev, dtype=torch.bfloat16).mul_(0.1)
        up = torch.randn((E, I, H), generator=g, device=dev, dtype=torch.bfloat16).mul_(0.1)
        down = torch.randn((E, H, I), generator=g, device=dev, dtype=torch.bfloat16).mul_(0.1)
        sl = slice(expert_begin, expert_begin + e_local)
        h.load_bf16(gate[sl].contiguous(), up[sl].contiguous(), down[sl].contiguous())
        del gate, up, down
        layers.append(h)
    torch.cuda.synchronize(dev)
    return layers


def make_routing(wl, T, nsets, dev, seed):
    """nsets x L distinct routings: randperm(E)[:k] per token, rand weights (reference kernel tests' convention)."""
    E, k, L = wl["E"], wl["k"], wl["L"]
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    scores = torch.rand((nsets, L, T, E), generator=g, device=dev)
    ids = scores.topk(k, dim=-1).indices.to(torch.int64).contiguous()
    w = torch.rand((nsets, L, T, k), generator=g, device=dev, dtype=torch.float32).contiguous()
    return ids, w


class DecodeRunner:
    """One token (or T tokens) through L stand-alone MoE layers; static buffers so the whole step is one HIP graph."""

    def __init__(self, wl, layers, T, dev, nsets=16, seed=1, ep_group=None):
        self.wl, self.layers, self.T, self.dev = wl, layers, T, dev
        self.ids_all, self.w_all = make_routing(wl, T, nsets, dev, seed)
        self.nsets = nsets
        self.ids = self.ids_all[0].clone()
        self.w = self.w_all[0].clone()
        g = torch.Generator(device=dev)
        g.manual_seed(seed + 77)
        self.x = (torch.randn((T, wl["H"]), generator=g, device=dev) / 100).to(torch.bfloat16)
        self.y = [torch.empty_like(self.x) for _ in range(2)]
        self.graph = None

    def set_step(self, i):
        s = i % self.nsets
        self.ids.copy_(self.ids_all[s])
        self.w.copy_(self.w_all[s])

    def step_eager(self):
        for li, h in enumerate(self.layers):
            h.forward(self.x, self.ids[li], self.w[li], out=self.y[li & 1])

    def capture(self):
        self.step_eager()
        torch.cuda.synchronize(self.dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.step_eager()
        torch.cuda.synchronize(self.dev)

    def step(self, i):
        self.set_step(i)
        if self.graph is not None:
            self.graph.replay()
        else:
            self.step_eager()


# ---------------------------------------------------------------------------------------------------------------------
# whole-model decode
# ---------------------------------------------------------------------------------------------------------------------
def random_ggml_blocks(E, N, K, ty, g, dev):
    """[E, N, K/256 * block_bytes] uint8: random code bytes of ggml type `ty` with small positive fp16 super-block scales, so
    that every block is a valid finite block (what bench_moe.py feeds, :197-224, minus its chance of inf scales)."""
    bb = GGML_BLOCK[ty]
    t = torch.randint(0, 256, (E, N, K // 256, bb), generator=g, device=dev, dtype=torch.uint8)
    # scales chosen so the weights come out with randn/10's spread: Q4_K w = d*sc*q - dmin*m (6-bit sc, m; q in 0..15; dmin = 7.5 d
    # centres it), Q6_K w = d*sc*q (int8 sc, q in -32..31), IQ1_S w = d*(2s+1)*(grid +- 1/8)
    lo, span = {12: (0.0004, 0.0002), 14: (0.00007, 0.00003), 19: (0.010, 0.006)}[ty]
    dv = torch.rand((E, N, K // 256), generator=g, device=dev) * span + lo
    d = dv.to(torch.float16).view(torch.uint8)
    off = {12: 0, 14: 208, 19: 0}[ty]
    t[..., off:off + 2] = d.view(E, N, K // 256, 2)
    if ty == 12:
        t[..., 2:4] = (dv * 7.5).to(torch.float16).view(torch.uint8).view(E, N, K // 256, 2)        # dmin
    return t.reshape(E, N, -1).contiguous()


def fp8_block_quant(w):
    """DeepSeek block-fp8 of a bf16 matrix batch [..., N, K] (N, K multiples of 128 up to padding): e4m3 bytes + fp32
    weight_scale_inv [..., ceil(N/128), ceil(K/128)] = amax / 448 per 128x128 block (the checkpoint format KLinearFP8 reads)."""
    *lead, N, K = w.shape
    Np, Kp = (N + 127) // 128 * 128, (K + 127) // 128 * 128
    wf = torch.zeros((*lead, Np, Kp), dtype=torch.float32, device=w.device)
    wf[..., :N, :K] = w.float()
    blk = wf.view(*lead, Np // 128, 128, Kp // 128, 128)
    sc = blk.abs().amax(dim=(-3, -1)).clamp_min(1e-12) / 448.0
    q = (blk / sc[..., :, None, :, None]).view(*lead, Np, Kp)[..., :N, :K].contiguous().to(torch.float8_e4m3fn)
    return q, sc.contiguous()


class RandomLoader:
    """Weight source for the whole-model run: tensors are generated on the device on demand (seeded by their name) with the
    shapes of the meta-device skeleton — there is no network for checkpoints.  Same protocol as util/loader.py.
    `linear` = "W4" (bf16 weights, quantised by the operator) | "FP8" (e4m3 weight + weight_scale_inv, as a DeepSeek fp8
    checkpoint holds them); `experts` = AMXInt4 (bf16, quantised online) | FP8 | RAWINT4 | llamafile (raw ggml blocks)."""

    def __init__(self, shapes, dev, linear="W4", experts="AMXInt4", ggml=None):
        self.shapes, self.dev, self.tensor_device_map = shapes, dev, {}
        self.linear, self.experts, self.ggml = linear, experts, ggml

    def _fp8_linear(self, name):
        return (self.linear == "FP8" and name.startswith("model.layers.") and name.endswith(".weight") and "_proj" in name
                and "kv_b_proj" not in name and ".experts." not in name and len(self.shapes.get(name, ())) == 2)

    def has_tensor(self, name):
        if name.endswith(".weight_scale_inv"):
            return self._fp8_linear(name[:-len("_scale_inv")])
        return name in self.shapes

    def _gen(self, name, shape, scale, mean=0.0):
        import zlib
        g = torch.Generator(device=self.dev)
        g.manual_seed(zlib.crc32(name.encode()))
        t = torch.randn(tuple(shape), generator=g, device=self.dev, dtype=torch.bfloat16).mul_(scale)
        return t.add_(mean) if mean else t

    def load_tensor(self, name, device="cpu"):
        if name.endswith(".weight_scale_inv"):
            return fp8_block_quant(self._gen(name[:-len("_scale_inv")], self.shapes[name[:-len("_scale_inv")]], self.shapes[name[:-len("_scale_inv")]][-1] ** -0.5))[1]
        shape = self.shapes[name]
        if "layernorm" in name or name.endswith("norm.weight"):
            return self._gen(name, shape, 0.1, 1.0)
        if "embed_tokens" in name:
            return self._gen(name, shape, 1.0)
        if name.endswith("e_score_correction_bias"):
            return self._gen(name, shape, 0.1).float()
        w = self._gen(name, shape, shape[-1] ** -0.5)
        return fp8_block_quant(w)[0] if self._fp8_linear(name) else w

    def get_expert_count(self, key):
        n = 0
        while f"{key}.{n}.gate_proj.weight" in self.shapes:
            n += 1
        return n

    def load_experts(self, key, device="cpu"):
        import zlib
        n = self.get_expert_count(key)
        out = {}
        g = torch.Generator(device=self.dev)
        g.manual_seed(zlib.crc32(key.encode()))
        for proj in ("gate", "up", "down"):
            N, K = self.shapes[f"{key}.0.{proj}_proj.weight"]
            if self.experts == "llamafile":
                ty = self.ggml[("gate", "up", "down").index(proj)]
                out[proj], out[f"{proj}_type"] = random_ggml_blocks(n, N, K, ty, g, self.dev), ty
            elif self.experts == "RAWINT4":      # random nibbles, bf16 group scales of randn/10 magnitude
                out[proj] = torch.randint(0, 256, (n, N, K // 2), generator=g, device=self.dev, dtype=torch.uint8)
                out[f"{proj}_scale"] = (torch.rand((n, N, K // 32), generator=g, device=self.dev) * 0.02 + 0.01).to(torch.bfloat16)
            elif self.experts == "FP8":
                q, sc = [], []
                for e0 in range(0, n, 32):        # bounded fp32 staging
                    qq, ss = fp8_block_quant(self._gen(f"{key}.{proj}.{e0}", (min(32, n - e0), N, K), 0.1))
                    q.append(qq.view(torch.uint8)); sc.append(ss)
                out[proj], out[f"{proj}_scale"] = torch.cat(q), torch.cat(sc)
            else:
                out[proj] = self._gen(f"{key}.{proj}", (n, N, K), 0.1)          # randn/10 like the reference's MoE tests
        return out


def rules_for(wl):
    """The product rule file, with the two format choices of the workload written into it (what a user edits in the reference's
    rule files too: `generate_op` of the layer linears, `backend` of the routed experts).  Returns a path."""
    import yaml
    src = os.path.join(ROOT, "ktransformers_amd", "optimize", "optimize_rules", wl["rules"])
    if wl.get("linear", "W4") == "W4" and wl.get("experts", "AMXInt4") == "AMXInt4":
        return src
    rules = yaml.safe_load(open(src))
    for r in rules:
        kw = r.get("replace", {}).get("kwargs", {})
        name = r.get("match", {}).get("name", "")
        if kw.get("generate_op") == "KLinearMarlin" and name.startswith("^model\\.layers") and wl["linear"] == "FP8":
            kw["generate_op"] = "KLinearFP8"
        if "backend" in kw:
            kw["backend"] = wl["experts"]
    fd, path = tempfile.mkstemp(prefix="ktx_rules_", suffix=".yaml", dir="/tmp")
    with os.fdopen(fd, "w") as f:
        yaml.safe_dump(rules, f)
    return path


class GreedyFeedbackStep(torch.nn.Module):
    """logits -> argmax (greedy sampling, utils.py:485-494 with do_sample=False) -> the token and the positions are written
    back into the step's own input buffers, so a captured replay IS one whole decode step."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, cur_token, position_ids, past_key_values, cache_position):
        if hasattr(self.model, "greedy_next_token") and not os.environ.get("KTX_TORCH_ARGMAX"):
            nxt = self.model.greedy_next_token(cur_token, position_ids, past_key_values, cache_position).view(1, 1)
        else:
            logits = self.model(cur_token, position_ids, past_key_values, cache_position)
            nxt = logits[0, -1].argmax(dim=-1).view(1, 1)
        cur_token.copy_(nxt)
        position_ids.add_(1)
        cache_position.add_(1)
        return nxt


class ModelDecodeRunner:
    """Whole-model greedy decode through the YAML-injected operators, one HIP graph per token."""

    def __init__(self, wl, n_layers, dev, ctx, max_new, seed=0, use_graph=True, trace=None):
        from ktransformers_amd.models.custom_cache import StaticCache
        from ktransformers_amd.models.modeling_deepseek import DeepseekForCausalLM, make_config
        from ktransformers_amd.optimize.optimize import optimize_and_load
        from ktransformers_amd.util.generate import CUDAGraphRunner, set_inference_mode
        from ktransformers_amd.util.utils import InferenceState

        cfg = make_config(**dict(MODELS[wl["model"]], num_hidden_layers=n_layers))
        self.cfg, self.dev, self.wl, self.ctx = cfg, dev, wl, ctx
        torch.set_default_dtype(torch.bfloat16)
        try:
            with torch.device("meta"):
                model = DeepseekForCausalLM(cfg)
            shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
            rules = rules_for(wl)
            import contextlib
            import io
            with contextlib.redirect_stdout(io.StringIO()):                   # "Injecting ..." lines
                optimize_and_load(model, rules, RandomLoader(shapes, dev, wl.get("linear", "W4"), wl.get("experts", "AMXInt4"),
                                                             wl.get("ggml")), cfg, default_device=str(dev))
        finally:
            torch.set_default_dtype(torch.float32)
        set_inference_mode(model, InferenceState.GENERATE)
        self.model = model
        self.cache = StaticCache(cfg, 1, ctx + max_new + 64, str(dev), torch.bfloat16)
        for kc in self.cache.key_cache:
            kc.normal_()
        self.step_mod = GreedyFeedbackStep(model)
        self.seed = seed
        self.runner, self.graph_ok, self.graph_error = None, False, None
        self.capture(use_graph)

    def capture(self, use_graph=True):
        """(Re-)capture the decode step: launch-time choices (tuning knobs) are baked into the graph at this point."""
        from ktransformers_amd.util.generate import CUDAGraphRunner

        dev, ctx = self.dev, self.ctx
        self.runner, self.graph_ok, self.graph_error = None, False, None
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # ranks finish loading seconds apart; the first warm-up step already exchanges with the peers
            torch.cuda.synchronize(dev)
            torch.distributed.barrier()
        self.cache.past_tokens = [ctx] * self.cfg.num_hidden_layers
        self.pos = torch.tensor([[ctx]], device=dev, dtype=torch.long)
        self.cur = torch.tensor([[1 + 17 * self.seed]], device=dev, dtype=torch.long)
        if use_graph:
            try:
                r = CUDAGraphRunner()
                with torch.no_grad():
                    r.capture(self.step_mod, self.cur, self.pos, self.pos[0].clone(), self.cache, main_device=str(dev))
                self.runner, self.graph_ok = r, True
                self.pos = r.input_buffers["position_ids"]               # advanced inside the graph
                self.cur = r.input_buffers["cur_token"]
            except Exception as e:   # e.g. collectives that cannot be captured on this stack: stay eager, and SAY so
                self.graph_error = f"{type(e).__name__}: {e}"
                log(f"[bench] graph capture failed ({self.graph_error}); running eagerly")
                torch.cuda.synchronize(dev)
        self.cache_pos = self.pos[0].clone()

    def moe_handles(self):
        return [l.mlp.experts.generate_experts.handle for l in self.model.model.layers if hasattr(l.mlp, "experts")]

    def set_position(self, p):
        self.pos.fill_(p)
        if self.runner is not None:
            self.runner.input_buffers["cache_position"].fill_(p)
        else:
            self.cache_pos.fill_(p)

    @torch.no_grad()
    def step(self, i=0):
        if self.runner is not None:
            self.runner.graph.replay()
        else:
            self.step_eager()

    @torch.no_grad()
    def step_eager(self):
        """The same step as plain launches (no graph), on the same input buffers."""
        cp = self.runner.input_buffers["cache_position"] if self.runner is not None else self.cache_pos
        self.step_mod(self.cur, self.pos, self.cache, cp)

    def close(self):
        self.runner = self.step_mod = self.model = self.cache = None
        gc.collect()
        torch.cuda.empty_cache()


def timed(fn, steps, warmup, dev, dist_on):
    import torch.distributed as dist

    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize(dev)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        fn(warmup + i)
    torch.cuda.synchronize(dev)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    # a bounded hand-off inside the one-launch attention step that gave up leaves undefined results and a status word: no number then
    from ktransformers_amd import _native
    bad_dev, st = _native.attn_status_any()
    if dist_on:
        # the status word rides the same all-reduce as the time (MAX over ranks), so every rank raises TOGETHER: a rank that raised
        # alone left the others blocked in the collective until the process-group timeout
        t = torch.tensor([dt, float(st)], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, st_any = float(t[0].item()), int(t[1].item())
        if st_any != 0:
            raise RuntimeError(f"one-launch attention step: a hand-off timed out during the timed region on some rank (status {st_any:#x}"
                               + (f"; this rank: cuda:{bad_dev} {st:#x})" if st else "; not on this rank)"))
    elif st != 0:
        raise RuntimeError(f"one-launch attention step on cuda:{bad_dev}: a hand-off timed out during the timed region (status {st:#x})")
    return dt


# ---------------------------------------------------------------------------------------------------------------------
# per-kernel timing: every library launch of REAL decode steps, bracketed by HIP events on the launch stream
# ---------------------------------------------------------------------------------------------------------------------
KTX_KERNEL_NAMES = ("lin_sk_kernel", "lin_sk_gate_kernel", "lin_dec_kernel", "lin_dec_gate_kernel", "lin_qb_absorb_kernel",
                    "lin_merge_unabsorb_kernel", "lin_dequant_w4_kernel", "lin_gemm_kernel", "lin_gemm_w4n_kernel",
                    "gate_fused_kernel", "gate_logits_kernel", "gate_select_kernel", "moe_dec_gateup_kernel", "moe_dec_down_kernel",
                    "moe_dec_fp_gateup_kernel", "moe_dec_fp_down_kernel", "moe_dec_raw_gateup_kernel", "moe_dec_raw_down_kernel",
                    "moe_dec_gguf_gateup_kernel", "moe_dec_gguf_down_kernel", "moe_prep_kernel", "moe_gemm_kernel", "moe_combine_kernel",
                    "mla_decode_kernel", "mla_merge_kernel", "mla_prep_kernel", "mla_cache_append_kernel", "rmsnorm_kernel",
                    "silu_mul_kernel", "argmax_bf16_kernel", "ep_gather_kernel", "ep_reduce_kernel", "attn_decode_kernel")


def _kclass(text):
    """Kernel class of a rocprofv3 kernel name or of a launch label: the identifier in front of '<' / '(' / ' '."""
    import re
    t = re.sub(r"^void ", "", text.strip())
    t = t.replace("(anonymous namespace)::", "")
    m = re.match(r"[A-Za-z_0-9:]+", t)
    return m.group(0).split("::")[-1] if m else t[:40]


# ---------------------------------------------------------------------------------------------------------------------
# per-kernel table of the REPLAYED GRAPH: a rocprofv3 --kernel-trace child pass of the same model and graph
# ---------------------------------------------------------------------------------------------------------------------
def trace_child(args, wl, dev):
    """Child mode (run under rocprofv3 --kernel-trace): build the same model, log the launch labels (with their algorithmic
    bytes) of one eager step, capture the step, replay it; the label list goes to stdout."""
    from ktransformers_amd import _native

    n_layers = args.layers or wl["layers"]
    mr = ModelDecodeRunner(wl, n_layers, dev, args.ctx, 512, use_graph=False)
    mr.step_eager()
    torch.cuda.synchronize(dev)
    _native.timing_enable(2)
    try:
        mr.step_eager()
        torch.cuda.synchronize(dev)
        labels = [[l, b] for l, b, _ in _native.timing_collect()]
    finally:
        _native.timing_enable(0)
    mr.capture(True)
    for _ in range(args.trace_steps + 30):
        mr.step()
    torch.cuda.synchronize(dev)
    print(json.dumps({"labels": labels, "hip_graph": bool(mr.graph_ok), "layers": n_layers}), flush=True)


def graph_kernel_table(args, ms_per_step, timeout_s=420):
    """rocprofv3 --kernel-trace over a child that replays the SAME captured decode graph: per kernel class and shape, the
    average in-graph duration of the last `trace_steps` steps (a rocprofv3 duration runs from the end of the predecessor to
    the end of the kernel: the ~1.5 us boundary is inside it, and the durations of a step add up to the step).  Launch
    labels (algorithmic bytes) come from the library's own log of one eager step of the child and are lined up per kernel
    class.  Returns (rows, info)."""
    import csv
    import glob
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, {"error": "rocprofv3 not found"}
    d = tempfile.mkdtemp(prefix="ktx_trace_", dir="/tmp")
    cmd = [rocprof, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
           "--trace-child", "--workload", args.workload, "--ctx", str(args.ctx), "--layers", str(args.layers),
           "--trace-steps", str(args.trace_steps)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        meta = None
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{") and '"labels"' in line:
                meta = json.loads(line)
                break
        files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if not meta or not files:
            return None, {"error": f"trace child failed rc={r.returncode}: {(r.stderr or r.stdout).strip()[-300:]}"}
        rows = list(csv.DictReader(open(files[0])))
    except Exception as e:
        return None, {"error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    for x in rows:
        x["s"], x["e"] = int(x["Start_Timestamp"]), int(x["End_Timestamp"])
    rows.sort(key=lambda x: x["s"])
    cuts = [i for i, x in enumerate(rows) if "argmax_bf16_kernel" in x["Kernel_Name"]]
    steps = [rows[a + 1:b + 1] for a, b in zip(cuts, cuts[1:])]
    if not steps:
        return None, {"error": "no decode steps in the kernel trace"}
    lens = [len(st) for st in steps]
    L = max(set(lens), key=lens.count)
    steps = [st for st in steps if len(st) == L][-args.trace_steps:]
    # labels per kernel class, in launch order
    lab_by_class = {}
    for lab, nb in meta["labels"]:
        lab_by_class.setdefault(_kclass(lab), []).append((lab, nb))
    agg = {}
    layer_us = []
    for st in steps:
        seen = {}
        t_layers, cur = [], None
        for x in st:
            c = _kclass(x["Kernel_Name"])
            us = (x["e"] - x["s"]) / 1e3
            key, nb = f"{c} grid {x.get('Grid_Size_X', '?')}", 0
            if c in lab_by_class:
                i = seen.get(c, 0)
                seen[c] = i + 1
                n_disp = sum(1 for y in st if _kclass(y["Kernel_Name"]) == c) if i == 0 else None
                if i == 0:
                    seen["#" + c] = n_disp
                if seen["#" + c] == len(lab_by_class[c]):
                    key, nb = lab_by_class[c][i]
            elif c not in KTX_KERNEL_NAMES:
                key = "torch: " + c[:60]
            a = agg.setdefault(key, [0, 0.0, nb])
            a[0] += 1
            a[1] += us
            # one MoE layer = from one q_a|kv_a GEMV / attention-input launch to the next: cut at the launch that carries mla_prep
            if key.endswith("+mla_prep") or key.startswith("mla_prep_kernel"):
                if cur is not None:
                    t_layers.append(cur)
                cur = [0.0, False]
            if cur is not None:
                cur[0] += us
                cur[1] = cur[1] or c.startswith("moe_dec_")
        layer_us += [t for t, moe in t_layers if moe]
    n = len(steps)
    out = []
    for key, (cnt, tot, nb) in agg.items():
        us = tot / cnt
        per = cnt / n
        out.append({"kernel": key, "launches_per_step": round(per, 2), "avg_launch_us": round(us, 3), "us_per_step": round(us * per, 1),
                    "share_of_step": round(us * per / (ms_per_step * 1e3), 4), "algorithmic_bytes_per_launch": int(nb),
                    "GBs": round(nb / (us * 1e-6) / 1e9, 1) if nb and us > 0 else None,
                    "frac_of_hbm_peak": round(nb / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if nb and us > 0 else None})
    out.sort(key=lambda r: -r["us_per_step"])
    span = sum(st[-1]["e"] - st[0]["s"] for st in steps) / n / 1e3
    info = {"steps": n, "dispatches_per_step": L, "step_span_us": round(span, 1), "hip_graph": meta.get("hip_graph"),
            "moe_layer_us": round(sum(layer_us) / len(layer_us), 2) if layer_us else None,
            "source": "rocprofv3 --kernel-trace child pass replaying the captured decode graph of the same model; "
                      "durations include the inter-kernel boundary"}
    return out, info


def eager_step_log(mr, dev, flush, mode, reps):
    """Run `reps` decode steps eagerly with the library's per-launch log on (mode 1: HIP events around every kernel, mode 2:
    labels only) and return one log per step.  The GPU is kept busy with L3-flushing traffic while the host enqueues the
    whole step, so the kernels then run back to back exactly as in the captured graph, on cold caches."""
    from ktransformers_amd import _native

    logs, n_block = [], 40
    _native.timing_enable(mode)
    try:
        for r in range(reps + 1):
            for _ in range(n_block):
                flush.add_(1)                   # 2 x 512 MB of traffic each: ~0.3 ms of GPU time, and nothing stays in the L3
            t0 = time.perf_counter()
            mr.step_eager()
            host_ms = (time.perf_counter() - t0) * 1e3
            torch.cuda.synchronize(dev)
            log = _native.timing_collect()
            if r:
                logs.append(log)
            n_block = max(8, min(400, int(host_ms * 1.5 / 0.25) + 1))   # rep 0 sizes the blocker from the measured enqueue time
    finally:
        _native.timing_enable(0)
    return logs


def kernel_table(mr, dev, ms_per_step, reps=4):
    flush = torch.zeros(512 << 20, dtype=torch.int8, device=dev)
    logs = eager_step_log(mr, dev, flush, 1, reps)
    del flush
    agg: dict = {}
    for log in logs:
        for label, nbytes, us in log:
            a = agg.setdefault(label, [0, 0.0, nbytes])
            a[0] += 1
            a[1] += us or 0.0
    rows = []
    for label, (cnt, tot, nbytes) in agg.items():
        us = tot / cnt
        n = cnt / len(logs)
        rows.append({"kernel": label, "launches_per_step": round(n, 2), "avg_launch_us": round(us, 3),
                     "us_per_step": round(us * n, 1), "share_of_step": round(us * n / (ms_per_step * 1e3), 4),
                     "algorithmic_bytes_per_launch": int(nbytes),
                     "GBs": round(nbytes / (us * 1e-6) / 1e9, 1) if us > 0 else None,
                     "frac_of_hbm_peak": round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if us > 0 else None})
    rows.sort(key=lambda r: -r["us_per_step"])
    # kernel time of ONE MoE layer: cut a step's log at every mla_prep (one per layer, fused into the q-absorb launch in decode; a segment then holds one layer's
    # launches, its first GEMVs borrowed from the next layer, whose shapes are the same); average the MoE segments
    segs = []
    for log in logs:
        cuts = [i for i, (label, _, _) in enumerate(log) if label.startswith("mla_prep_kernel") or label.endswith("+mla_prep")]
        for a, b in zip(cuts, cuts[1:]):
            seg = log[a:b]
            if any(l.startswith("moe_dec_") for l, _, _ in seg):
                segs.append(sum(us or 0.0 for _, _, us in seg))
    return rows, (sum(segs) / len(segs) if segs else None)


# ---------------------------------------------------------------------------------------------------------------------
# PMC traffic of the dominant kernel: rocprofv3 child passes (counters cannot be read from inside this process)
# ---------------------------------------------------------------------------------------------------------------------
def pmc_child(args, wl, dev):
    """Child mode (run under rocprofv3 --pmc): a short model of the same dimensions decodes 3 tokens eagerly with the
    library's launch log in labels-only mode; the label sequence goes to stdout.  The parent lines it up with the LAST
    len(labels) dispatches of library kernels in the counter CSV."""
    n_layers = wl["dense"] + 2
    mr = ModelDecodeRunner(wl, n_layers, dev, args.ctx, 64, use_graph=False)
    flush = torch.zeros(512 << 20, dtype=torch.int8, device=dev)
    logs = eager_step_log(mr, dev, flush, 2, 3)
    print(json.dumps({"labels": [l for log in logs for l, _, _ in log]}), flush=True)


def pmc_traffic(args, label, timeout_s=300):
    """HBM bytes per launch of the kernel class `label`: separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE
    (MI355X_MICROARCH.md §HBM: KiB units; FETCH_SIZE x2 for wide coalesced reads on gfx950; WRITE_SIZE as reported)."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    import csv
    import glob
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix=f"ktx_pmc_{counter}_", dir="/tmp")
        cmd = [rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.abspath(__file__), "--pmc-child", "--workload", args.workload, "--ctx", str(args.ctx)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp",
                               env=dict(os.environ, TMPDIR="/tmp"))
            labels = None
            for line in reversed(r.stdout.strip().splitlines()):
                if line.startswith("{") and '"labels"' in line:
                    labels = json.loads(line)["labels"]
                    break
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not labels or not files:
                return None, f"pmc child failed rc={r.returncode}: {(r.stderr or r.stdout).strip()[-200:]}"
            rows = [x for x in csv.DictReader(open(files[0]))
                    if x["Counter_Name"] == counter and any(k in x["Kernel_Name"] for k in KTX_KERNEL_NAMES)]
            rows.sort(key=lambda x: int(x["Dispatch_Id"]))
            rows = rows[-len(labels):]
            if len(rows) < len(labels) or any(lab.split("<")[0].split(" ")[0].split("+")[0] not in row["Kernel_Name"]
                                              for lab, row in zip(labels, rows)):
                return None, "pmc: the dispatch sequence does not line up with the library's launch log"
            vals = [float(row["Counter_Value"]) for lab, row in zip(labels, rows) if lab == label]
            if not vals:      # the child's shorter cache can change a launch-time choice the label carries (the KV split count)
                stem = label.split(" nsplit=")[0]
                vals = [float(row["Counter_Value"]) for lab, row in zip(labels, rows) if lab.split(" nsplit=")[0] == stem]
            if not vals:
                return None, f"pmc: no dispatch of {label!r} in the child"
            res[counter] = sum(vals) / len(vals)
        except Exception as e:
            return None, f"pmc error: {type(e).__name__}: {e}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    read_b = res["FETCH_SIZE"] * 1024 * 2
    write_b = res["WRITE_SIZE"] * 1024
    return int(read_b + write_b), (f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this run, same model dimensions "
                                   f"(KiB -> bytes; FETCH_SIZE x2 gfx950 wide-read correction): read {int(read_b)} + write {int(write_b)} B per launch")


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline: the reference's own kernels on the host cores
# ---------------------------------------------------------------------------------------------------------------------
def host_numa_nodes():
    """NUMA nodes of the host that hold CPUs (sysfs); 1 if it cannot be read."""
    n = 0
    try:
        for d in os.listdir("/sys/devices/system/node"):
            if d.startswith("node") and d[4:].isdigit():
                try:
                    if open(f"/sys/devices/system/node/{d}/cpulist").read().strip():
                        n += 1
                except OSError:
                    pass
    except OSError:
        pass
    return max(n, 1)


def host_cpu_info():
    """What the CPU leg ran on: logical CPUs this process may use, sockets, hardware threads per core, model name."""
    info = {"logical_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)}
    try:                                      # the container's CFS allowance: "quota period" in us, or "max"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        info["cpu_quota"] = None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        info["cpu_quota"] = None
    try:
        sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
        n = 0
        for part in sib.split(","):
            a, _, b = part.partition("-")
            n += (int(b) - int(a) + 1) if b else 1
        info["threads_per_core"] = max(n, 1)
    except (OSError, ValueError):
        info["threads_per_core"] = 1
    try:
        pk = set()
        for d in os.listdir("/sys/devices/system/cpu"):
            if d.startswith("cpu") and d[3:].isdigit():
                try:
                    pk.add(open(f"/sys/devices/system/cpu/{d}/topology/physical_package_id").read().strip())
                except OSError:
                    pass
        info["sockets"] = max(len(pk), 1)
    except OSError:
        info["sockets"] = 1
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                info["model"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return info


def cpu_baseline(wl, budget_s=16.0, prefill_budget_s=8.0, threads=None):
    """The reference's own AVX512 MoE kernels (oracle/_ref) on this box's host cores, same expert shape: a bs=1 decode leg and a
    prompt leg.  Threads: ONE per physical core the process may use, pinned by the reference's own worker pool (its hwloc calls are
    served from sysfs by oracle/shim/hwloc.h), one sub-pool per NUMA node — the placement kt-kernel/README.md asks for.
    Bounded sample: 3 distinct layers' weights (>> the host's L3), forwards rotating over them.  The expert COUNT is cut to 32 per
    layer to bound host memory and set-up time: a bs=1 forward touches k experts whatever E is, and the prompt leg keeps the
    per-expert GEMM shape of the real chunk (see `prefill.sample`)."""
    import numpy as np

    try:
        from oracle.oracle import FMT_AMXINT4, Reference, f32_to_bf16, reference_available
    except Exception as e:  # pragma: no cover
        return {"value": None, "unit": "tok/s", "cores": 0, "kind": "reference", "sample": f"unavailable: {e}"}
    if not reference_available():
        return {"value": None, "unit": "tok/s", "cores": 0, "kind": "reference",
                "sample": "oracle/_ref/libkt_ref.so not present or host lacks AVX512-VNNI/BF16"}
    H, I, k = wl["H"], wl["I"], wl["k"]
    E = min(wl["E"], 32)
    Lm = wl["full_layers"] - wl["dense"]
    host = host_cpu_info()
    numa_nodes = host_numa_nodes()
    phys = max(1, host["logical_cpus"] // host["threads_per_core"])          # physical cores (reference guidance: one worker each)
    subpools = numa_nodes if (numa_nodes > 1 and I % (numa_nodes * 32) == 0) else 1
    rng = np.random.default_rng(0)
    base = f32_to_bf16((rng.standard_normal((E, I, H), dtype=np.float32) / 10))
    x = f32_to_bf16((rng.standard_normal((1, H), dtype=np.float32) / 100))
    sets = [(np.stack([rng.permutation(E)[:k]]).astype(np.int64), rng.random((1, k), dtype=np.float32)) for _ in range(64)]

    def fit(n):
        n = max(subpools, min(int(n), phys))
        return n - n % subpools

    # The reference's workers busy-wait, so a container with a CFS quota below its CPU count (measured on the pool's boxes: 16 CPUs of
    # quota on 256 logical CPUs, profiles/r04_i_cpu_leg_probe.txt) throttles a one-worker-per-core pool to a fraction of what fewer
    # workers reach (128 -> 3.4, 64 -> 6.8, 32 -> 8.8 tok/s).  Placement sweep: the candidates are timed for ~1.5 s each on one layer
    # and the best one runs the legs.
    sweep = None
    if threads:
        threads = fit(threads)
    else:
        quota = host.get("cpu_quota")
        cands = sorted({fit(phys)} | ({fit(quota), fit(2 * quota), fit(4 * quota)} if quota and quota < phys else set()))
        if len(cands) == 1:
            threads = cands[0]
        else:
            sweep = {}
            for c in cands:
                r = Reference(threads=c // subpools, subpools=subpools)
                m0 = r.make_moe(FMT_AMXINT4, base, np.roll(base, 1, axis=0), np.roll(base, 2, axis=0).reshape(E, H, I), k=k, max_len=64)
                for i in range(10):
                    r.moe_forward(m0, sets[i][0], sets[i][1], x)
                n0, t0 = 0, time.perf_counter()
                while time.perf_counter() - t0 < 1.5:
                    r.moe_forward(m0, sets[n0 % 64][0], sets[n0 % 64][1], x)
                    n0 += 1
                sweep[c] = round((time.perf_counter() - t0) / n0 * 1e6, 1)
                r.free_moe(m0)
                r.close()
            threads = min(sweep, key=sweep.get)
    ref = Reference(threads=threads // subpools, subpools=subpools)
    rng = np.random.default_rng(0)
    nlayers = 3
    T_PRE = 256
    moes = []
    t_load = time.perf_counter()
    # one block of randn/10 bf16 values (`base`), re-used with cheap permutations so that every matrix of every layer is
    # distinct in memory (what matters for a bandwidth-bound baseline) without minutes of single-threaded numpy RNG
    for li in range(nlayers):
        gate = np.roll(base, li + 1, axis=0)
        up = np.ascontiguousarray(base[::-1]) if li % 2 == 0 else np.roll(base, -(li + 2), axis=0)
        down = np.roll(base, li + 3, axis=0).reshape(E, H, I)
        moes.append(ref.make_moe(FMT_AMXINT4, gate, up, down, k=k, max_len=T_PRE))
    t_load = time.perf_counter() - t_load
    for i in range(30):
        ref.moe_forward(moes[i % nlayers], sets[i % 64][0], sets[i % 64][1], x)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s and n < 40000:
        for _ in range(25):
            ref.moe_forward(moes[n % nlayers], sets[n % 64][0], sets[n % 64][1], x)
            n += 1
    dt = time.perf_counter() - t0
    t_layer = dt / n
    layer_bytes = k * (3 * H * I * 0.5 + (2 * I + H) * 4)                     # int4 weights + per-row fp32 scales of the k experts
    out = {"value": round(1.0 / (Lm * t_layer), 3), "unit": "tok/s", "cores": threads, "kind": "reference", "numa_nodes": numa_nodes,
           "subpools": subpools, "host": host, "physical_cores": phys, "thread_sweep_us_per_layer": sweep,
           "pinned": "workers bound to distinct physical cores by the reference's worker pool (sysfs-backed hwloc shim); thread count = the best of "
                     "the placement sweep when the container's CFS quota is below its core count, else one per physical core",
           "us_per_layer": round(t_layer * 1e6, 1), "GBs": round(layer_bytes / t_layer / 1e9, 1),
           "covers": "routed experts only (the part the reference runs on the CPU)",
           "sample": f"{n} bs=1 forwards of TP_MOE<AMX_MOE_TP<GemmKernel224Int4>> (AVX512-VNNI path, no AMX on this host), "
                     f"H={H} I={I} k={k}, rotating over {nlayers} distinct layers of {E} experts, {threads} threads in {subpools} sub-pool(s) "
                     f"(host NUMA nodes with CPUs: {numa_nodes}; sub-pools placed by the reference's libnuma calls, workers pinned to cores); "
                     f"tok/s = 1/({Lm} MoE layers x t_layer) = the routed experts of the full-depth model alone; "
                     f"weight quant took {t_load:.1f}s (untimed)"}
    # ---- prompt leg (kt-kernel/bench/bench_moe_amx.py's loop at qlen > 1; forward_prefill, operators/amx/moe_base.hpp:208-436):
    # T_PRE tokens over the sample's 32 experts = T_PRE * k / 32 rows per expert, the per-expert GEMM of a 2048-token chunk over
    # the model's 256 experts; the chunk's layer time is the sample's times (E_model / 32).
    try:
        rows_per_expert = T_PRE * k // E
        chunk = rows_per_expert * wl["E"] // k                                     # tokens of the chunk this sample stands for
        xp = f32_to_bf16((rng.standard_normal((T_PRE, H), dtype=np.float32) / 100))
        idp = np.stack([rng.permutation(E)[:k] for _ in range(T_PRE)]).astype(np.int64)
        wp = rng.random((T_PRE, k), dtype=np.float32)
        for i in range(2):
            ref.moe_forward(moes[i % nlayers], idp, wp, xp)
        m, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < prefill_budget_s and m < 2000:
            ref.moe_forward(moes[m % nlayers], idp, wp, xp)
            m += 1
        tp = (time.perf_counter() - t0) / m
        t_chunk_layer = tp * (wl["E"] / E)
        flop = 2.0 * 3 * H * I * T_PRE * k
        out["prefill"] = {"value": round(chunk / (Lm * t_chunk_layer), 2), "unit": "tok/s", "chunk_tokens": chunk,
                          "ms_per_layer_chunk": round(t_chunk_layer * 1e3, 2), "int8_TOPs": round(flop / tp / 1e12, 2),
                          "covers": "routed experts only",
                          "sample": f"{m} forwards of {T_PRE} tokens over {E} experts ({rows_per_expert} rows per expert = a {chunk}-token chunk over "
                                    f"{wl['E']} experts; layer time = sample x {wl['E'] // E}); tok/s = {chunk} / ({Lm} MoE layers x layer time)"}
    except Exception as e:   # the decode leg stands without it
        out["prefill"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:200]}
    return out


def cpu_baseline_subprocess(workload, timeout_s=300):
    """Run the CPU leg in a child process: the reference kernels abort() on assertion failures and hold GBs of host memory;
    neither may take the bench line down with it."""
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", workload],
                           capture_output=True, text=True, timeout=timeout_s)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "tok/s", "cores": 0, "kind": "reference",
                "sample": f"child failed rc={r.returncode}: {r.stderr.strip()[-300:]}"}
    except Exception as e:
        return {"value": None, "unit": "tok/s", "cores": 0, "kind": "reference", "sample": f"child error: {e}"}


# ---------------------------------------------------------------------------------------------------------------------
def expert_bpw(wl):
    """Stored bytes per routed-expert weight: (gate|up, down)."""
    if wl["method"] == "GGUF":
        ty = wl["ggml"]
        return (GGML_BLOCK[ty[0]] + GGML_BLOCK[ty[1]]) / 2 / 256, GGML_BLOCK[ty[2]] / 256
    return EXPERT_BPW[wl["method"]], EXPERT_BPW[wl["method"]]


def step_bytes(cfg, n_layers, ctx, wl=None):
    """Algorithmic HBM bytes of one decode token through the model as built (weights as stored + KV + embedding row)."""
    wl = wl or WORKLOADS["v3-int4"]
    H, I, Im, E, k = cfg.hidden_size, cfg.intermediate_size, cfg.moe_intermediate_size, cfg.n_routed_experts, cfg.num_experts_per_tok
    Hq, nope, rope, v, lora = cfg.num_attention_heads, cfg.qk_nope_head_dim, cfg.qk_rope_head_dim, cfg.v_head_dim, cfg.kv_lora_rank
    wlin = LINEAR_BPW[wl.get("linear", "W4")]
    w4 = LINEAR_BPW["W4"]                                      # lm_head stays W4 in every rule file used here
    if cfg.q_lora_rank:
        q_params = H * cfg.q_lora_rank + cfg.q_lora_rank * Hq * (nope + rope)
    else:
        q_params = H * Hq * (nope + rope)
    attn = (q_params + H * (lora + rope) + Hq * v * H) * wlin + Hq * (nope + v) * lora * 2 + (ctx + 1) * (lora + rope) * 2
    dense_mlp = 3 * H * I * wlin
    n_dense = min(cfg.first_k_dense_replace, n_layers)
    n_moe = n_layers - n_dense
    gu, dn = expert_bpw(wl)
    sc = k * (2 * Im + H) * 4 if wl["method"] in ("AMXINT4", "AMXINT8") else 0          # per-row fp32 scales
    moe = k * (2 * H * Im * gu + H * Im * dn) + sc + E * H * 2 + (cfg.n_shared_experts or 0) * 3 * H * Im * wlin
    head = H * cfg.vocab_size * w4 + H * 2
    return int(n_layers * attn + n_dense * dense_mlp + n_moe * moe + head), int(attn + moe)


def box_info():
    """State of the GPU this run landed on (clocks, power cap, partition modes, firmware): boxes of the pool differ by up to
    1.5x on the latency-bound part of the step at identical clocks, so the line says where it was measured."""
    out = {}
    try:
        r = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showuniqueid", "--showperflevel", "--showclocks", "--showmaxpower", "--showpower",
                            "--showmemorypartition", "--showcomputepartition", "--showvbios", "--showdriverversion", "--showfwinfo"],
                           capture_output=True, text=True, timeout=20)
        keys = {"Unique ID": "unique_id", "Performance Level": "perf_level", "sclk clock level": "sclk", "mclk clock level": "mclk",
                "fclk clock level": "fclk", "Max Graphics Package Power (W)": "power_cap_w",
                "Current Socket Graphics Package Power (W)": "idle_power_w", "Compute Partition": "compute_partition",
                "Memory Partition": "memory_partition", "VBIOS version": "vbios", "Driver version": "driver",
                "SMC firmware version": "smc_fw", "MEC firmware version": "mec_fw"}
        for line in r.stdout.splitlines():
            for k, name in keys.items():
                if k + ":" in line and name not in out:
                    out[name] = line.split(k + ":", 1)[1].strip()
    except Exception as e:   # never let the fingerprint take the bench line down
        out["error"] = f"{type(e).__name__}: {e}"[:200]
    return out


MFMA_PEAK_BF16, MFMA_PEAK_I8 = 2.5e15, 5.0e15      # dense peaks, MI355X_MICROARCH.md (never the 2:1-sparsity figures)


def prefill_roofline(cfg, n_layers, T, wl, dt):
    """Which roof bounds ONE T-token prompt chunk through the model as built, and how close the chunk came (the
    grouped expert GEMM is HBM-bound below ~150 rows per expert — every touched expert's weights are read once — and MFMA-bound
    beyond).  flops: every linear 2 T N K, the routed experts 2 * 3 H I k T per MoE layer, causal attention over qk 192 / v 128;
    bytes: every weight of the resident layers once (all experts once T k >= 4 E) + the embedding rows."""
    H, I, Im, E, k = cfg.hidden_size, cfg.intermediate_size, cfg.moe_intermediate_size, cfg.n_routed_experts, cfg.num_experts_per_tok
    Hq, nope, rope, v, lora = cfg.num_attention_heads, cfg.qk_nope_head_dim, cfg.qk_rope_head_dim, cfg.v_head_dim, cfg.kv_lora_rank
    wlin = LINEAR_BPW[wl.get("linear", "W4")]
    q_params = (H * cfg.q_lora_rank + cfg.q_lora_rank * Hq * (nope + rope)) if cfg.q_lora_rank else H * Hq * (nope + rope)
    attn_params = q_params + H * (lora + rope) + Hq * v * H + Hq * (nope + v) * lora            # + kv_b expansion of the chunk
    n_dense = min(cfg.first_k_dense_replace, n_layers)
    n_moe = n_layers - n_dense
    shared_params = (cfg.n_shared_experts or 0) * 3 * H * Im
    lin_flop = 2.0 * T * (n_layers * attn_params + n_dense * 3 * H * I + n_moe * (shared_params + E * H))
    attn_flop = n_layers * 2.0 * Hq * (nope + rope + v) * T * (T + 1) / 2
    exp_flop = n_moe * 2.0 * 3 * H * Im * k * T
    gu, dn = expert_bpw(wl)
    touched = E if T * k >= 4 * E else min(E, T * k)
    nbytes = (n_layers * (attn_params - Hq * (nope + v) * lora) * wlin + n_layers * Hq * (nope + v) * lora * 2 + n_dense * 3 * H * I * wlin
              + n_moe * (touched * (2 * H * Im * gu + H * Im * dn) + shared_params * wlin + E * H * 2) + T * H * 2)
    int_experts = wl["method"] in ("AMXINT4", "AMXINT8", "RAWINT4", "GGUF")            # int8 MFMA paths; FP8 / BF16 experts multiply in bf16
    t_mfma = (lin_flop + attn_flop) / MFMA_PEAK_BF16 + exp_flop / (MFMA_PEAK_I8 if int_experts else MFMA_PEAK_BF16)
    t_hbm = nbytes / (HBM_PEAK_GBS * 1e9)
    hbm = t_hbm >= t_mfma
    flop = lin_flop + attn_flop + exp_flop
    return {"bound": "hbm" if hbm else "mfma", "achieved": round(nbytes / dt / 1e9, 1) if hbm else round(flop / dt / 1e12, 1),
            "peak": HBM_PEAK_GBS if hbm else round((flop / t_mfma) / 1e12, 1), "unit": "GB/s" if hbm else "TFLOP/s",
            "frac": round(max(t_hbm, t_mfma) / dt, 4), "algorithmic_bytes": int(nbytes), "flop": int(flop),
            "rows_per_expert": round(T * k / max(E, 1), 1) if n_moe else None,
            "ms_at_hbm_roof": round(t_hbm * 1e3, 3), "ms_at_mfma_roof": round(t_mfma * 1e3, 3),
            "note": "frac = the larger of the two roof times / the measured chunk time; the MFMA peak is the flop-weighted mix of the "
                    "dense bf16 (2.5 PFLOP/s) and int8 (5 POP/s) peaks of the chunk's GEMMs"}


def whole_model_prefill(mr, T, dev, reps=3, per_kernel_pass=True):
    """One T-token prompt chunk through the resident model's prefill path (every operator's T>1 kernels)."""
    from ktransformers_amd.util.generate import set_inference_mode
    from ktransformers_amd.util.utils import InferenceState

    set_inference_mode(mr.model, InferenceState.PREFILL)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    ids = torch.randint(0, mr.cfg.vocab_size, (1, T), generator=g, device=dev)
    pos = torch.arange(T, device=dev).unsqueeze(0)
    times = []
    with torch.no_grad():
        for r in range(reps + 1):
            mr.cache.past_tokens = [0] * mr.cfg.num_hidden_layers
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            mr.model(ids, pos, mr.cache, pos[0], last_token_only=True)
            torch.cuda.synchronize(dev)
            if r:
                times.append(time.perf_counter() - t0)
    # one more pass with the library's per-launch log on: where a prompt chunk spends its time
    from ktransformers_amd import _native
    per_kernel, lib_ms = [], None
    try:
        if not per_kernel_pass:
            raise StopIteration
        _native.timing_enable(1)
        with torch.no_grad():
            mr.cache.past_tokens = [0] * mr.cfg.num_hidden_layers
            mr.model(ids, pos, mr.cache, pos[0], last_token_only=True)
        agg = {}
        for label, nbytes, us in _native.timing_collect():
            key = label.split(" T=")[0] if label.startswith(("lin_gemm", "lin_dec")) else label
            a = agg.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += us or 0.0
        tot = sum(v[1] for v in agg.values())
        lib_ms = tot / 1e3
        per_kernel = [{"kernel": k2, "launches": v[0], "total_ms": round(v[1] / 1e3, 3), "share": round(v[1] / tot, 4)}
                      for k2, v in sorted(agg.items(), key=lambda kv: -kv[1][1])][:12]
    except StopIteration:
        pass
    finally:
        _native.timing_enable(0)
    set_inference_mode(mr.model, InferenceState.GENERATE)
    dt = sum(times) / len(times)
    cfg = mr.cfg
    n_moe = cfg.num_hidden_layers - min(cfg.first_k_dense_replace, cfg.num_hidden_layers)
    moe_flop = 2 * 3 * cfg.hidden_size * cfg.moe_intermediate_size * cfg.num_experts_per_tok * T * n_moe
    return {"value": round(T / dt, 1), "unit": "tok/s", "tokens": T, "ms_per_chunk": round(dt * 1e3, 3),
            "layers": cfg.num_hidden_layers, "roofline": prefill_roofline(cfg, cfg.num_hidden_layers, T, mr.wl, dt),
            "routed_expert_TOPs_share": round(moe_flop / dt / 1e12, 1), "per_kernel": per_kernel,
            # every launch of libktx_hip.so is event-bracketed in that extra pass; the rest of the chunk is torch glue
            # (elementwise adds, the GLU row permutation, copies, index ops) and launch gaps — no vendor GEMM is called
            "library_kernel_ms": None if lib_ms is None else round(lib_ms, 3),
            "outside_library_share": None if lib_ms is None else round(max(0.0, 1.0 - lib_ms / (dt * 1e3)), 4),
            "vendor_gemm": bool(os.environ.get("KTX_VENDOR_GEMM")),
            "what": "whole resident model, one prompt chunk from an empty cache (non-absorbed MLA prompt attention, grouped int8-MFMA "
                    "expert GEMMs, W4 linears expanded once per call + the library's own BF16 MFMA GEMM), last-token logits"}


def run_model_decode(name, args, dev, steps, warmup, dist_on=False, world=1, rank=0, n_layers=None, ctx=None, windows=0):
    """Build the named workload's model, capture one decode step, time it.  Returns (result dict, runner)."""
    wl = WORKLOADS[name]
    n_layers = n_layers or wl["layers"]
    ctx = ctx or args.ctx
    t0 = time.perf_counter()
    # (the cache also has to hold the reference's own 8192-token prompt chunk, local_chat.py:86, timed after the decode run)
    max_new = max(steps * (windows + 1) + warmup + 1024, (0 if args.no_prefill else max(args.prefill_tokens, PREFILL_LONG) + 64) - ctx)
    mr = ModelDecodeRunner(wl, n_layers, dev, ctx, max_new, seed=rank, use_graph=not args.no_graph)
    cfg = mr.cfg
    n_dense = min(cfg.first_k_dense_replace, n_layers)
    gib = sum(h.weight_bytes for h in mr.moe_handles()) / 2 ** 30
    if rank == 0:
        log(f"[bench] {name}: {n_layers} layers ({n_dense} dense + {n_layers - n_dense} MoE, {gib:.1f} GiB of packed experts on this rank) "
            f"injected and loaded in {time.perf_counter() - t0:.1f}s; HIP graph: {mr.graph_ok}")
    # bring the GPU to its sustained clocks before the W warm-up + K timed steps (a fresh process that has only loaded weights
    # runs its first replays at idle clocks).  N > 1: every step issues collectives, so all ranks run the SAME fixed number of
    # steps.  Declared in the JSON as `prewarm_steps`.
    n_pre, t_pre = 0, time.perf_counter()
    n_pre_dist = 20 if getattr(args, "dist_backend", "nccl") == "gloo" else 100     # (gloo = ranks time-slicing one GPU: a smoke, not a measurement)
    while n_pre < n_pre_dist if dist_on else (time.perf_counter() - t_pre < 1.0 and n_pre < 300):
        for _ in range(10):
            mr.step()
            n_pre += 1
        torch.cuda.synchronize(dev)
    mr.set_position(ctx)               # the timed run starts at the configured context length again
    dt = timed(mr.step, steps, warmup, dev, dist_on)
    ms = dt / steps * 1e3
    win = []
    for _ in range(windows):           # the same K steps again, several times: how much a box moves between windows
        win.append(timed(mr.step, steps, 0, dev, dist_on) / steps * 1e3)
    tot_bytes, layer_bytes = step_bytes(cfg, n_layers, ctx, wl)
    res = {"value": round(world * steps / dt, 2), "unit": "tok/s", "ms_per_step": round(ms, 4), "layers": n_layers,
           "dense_layers": n_dense, "moe_layers": n_layers - n_dense, "ctx": ctx, "hip_graph": bool(mr.graph_ok),
           "graph_error": mr.graph_error, "prewarm_steps": n_pre, "expert_GiB_resident": round(gib, 1),
           "whole_step": {"algorithmic_bytes": tot_bytes, "GBs": round(tot_bytes / (ms * 1e-3) / 1e9, 1),
                          "frac_of_hbm_peak": round(tot_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "moe_layer_bytes": layer_bytes}}
    if win:
        allw = sorted([ms] + win)
        res["windows_ms_per_step"] = [round(w, 4) for w in [ms] + win]
        res["median_ms_per_step"] = round(allw[len(allw) // 2], 4)
        res["median_tok_s"] = round(world * 1e3 / allw[len(allw) // 2], 2)
    return res, mr


def batched_decode(mr, B, ctx, steps, warmup, dev):
    """B requests, ONE new token each per step — the decode batch of the reference's balance_serve engine seam: the flattened-batch
    attention operator (`flashinfer_attn`, archive/ktransformers/operators/balance_serve_attention.py:66-118) over a paged latent
    cache (`KDeepSeekV3Cache`, permuted page table) and the MoE / MLP blocks on the [B, hidden] rows, greedy next tokens, one HIP
    graph per step.  What it shows that bs = 1 cannot: the attention / shared / lm_head weights are streamed once for B tokens, so
    the step approaches the HBM roof where the bs = 1 step is bound by its chain of dependent launches.  Positions are held at
    `ctx` (every step rewrites the same cache slot): the step's work does not depend on it."""
    from ktransformers_amd._native import MLAWrapper, argmax_bf16
    from ktransformers_amd.models.custom_cache import KDeepSeekV3Cache
    from ktransformers_amd.operators.balance_serve_attention import flashinfer_attn

    model, cfg = mr.model, mr.cfg
    page = 64
    ppr = (ctx + 1 + page - 1) // page + 1                       # pages per request
    kv = KDeepSeekV3Cache(cfg, page_size=page, device=str(dev))
    kv.allocate(B * ppr)
    for kc in kv.k_caches:
        kc.normal_()
    g = torch.Generator(device=dev)
    g.manual_seed(23)
    i32 = dict(dtype=torch.int32, device=dev)
    q_indptr = torch.arange(B + 1, **i32)
    kv_indptr = torch.arange(B + 1, **i32) * ppr
    kv_indices = torch.randperm(B * ppr, generator=g, device=dev).to(torch.int32)          # a scattered page table
    kv_len = torch.full((B,), ctx + 1, **i32)
    pos = torch.full((B,), ctx, dtype=torch.int64, device=dev)
    bsz = torch.tensor([B], **i32)
    page_idx, page_off = kv.get_page_table(pos, q_indptr, kv_indptr, kv_indices, bsz)
    page_idx, page_off = page_idx.to(torch.int32), page_off.to(torch.int32)
    attn0 = model.model.layers[0].self_attn
    Hp = (attn0.num_heads + 15) // 16 * 16
    wrapper = MLAWrapper(B, B * ppr, use_cuda_graph=True, device=dev, max_q_tokens=B)
    wrapper.plan(q_indptr, kv_indptr, kv_indices, kv_len, bsz, Hp, attn0.kv_lora_rank, attn0.qk_rope_head_dim, page, attn0.softmax_scale,
                 torch.bfloat16, torch.bfloat16, max_kv_len=ctx + page)
    tokens = torch.randint(0, cfg.vocab_size, (B,), generator=g, device=dev)

    def step():
        # the layer loop of the reference's serving model (custom_modeling_deepseek_v3.py:93-129): a zero residual, every norm is
        # flashinfer's fused_add_rmsnorm (residual += h; h = norm(residual)) — one launch where `h + f(norm(h))` is three
        h = model.model.embed_tokens(tokens).contiguous()                                # [B, hidden]
        res = torch.zeros_like(h)
        for layer in model.model.layers:
            h, res = layer.input_layernorm(h, bsz, res)
            h = flashinfer_attn.forward(layer.self_attn, h, kv, pos, wrapper, bsz, page_idx, page_off).contiguous()
            h, res = layer.post_attention_layernorm(h, bsz, res)
            h = layer.mlp(h.unsqueeze(0)).squeeze(0).contiguous()
        h, res = model.model.norm(h, bsz, res)
        logits = model.lm_head(h)
        tokens.copy_(argmax_bf16(logits) if logits.dtype == torch.bfloat16 else logits.float().argmax(dim=-1))

    with torch.no_grad():
        for _ in range(3):
            step()
        torch.cuda.synchronize(dev)
        graph, graph_ok = torch.cuda.CUDAGraph(), True
        try:
            with torch.cuda.graph(graph):
                step()
        except Exception as e:     # stay eager, and say so
            graph_ok = False
            log(f"[bench] bs{B}: graph capture failed ({type(e).__name__}: {e}); running eagerly")
            torch.cuda.synchronize(dev)
        run = graph.replay if graph_ok else step
        for _ in range(warmup):
            run()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
    ms = dt / steps * 1e3
    # algorithmic bytes of the step: every non-expert weight once, the expected number of DISTINCT routed experts of B tokens
    # (E (1 - (1 - k/E)^B)), B requests' latent rows
    n_layers = cfg.num_hidden_layers
    tot1, _ = step_bytes(cfg, n_layers, ctx, mr.wl)
    E, k, H, Im = cfg.n_routed_experts, cfg.num_experts_per_tok, cfg.hidden_size, cfg.moe_intermediate_size
    n_moe = n_layers - min(cfg.first_k_dense_replace, n_layers)
    gu, dn = expert_bpw(mr.wl)
    per_expert = 2 * H * Im * gu + H * Im * dn
    distinct = E * (1.0 - (1.0 - k / E) ** B)
    nbytes = tot1 + n_moe * (distinct - k) * per_expert + (B - 1) * n_layers * (ctx + 1) * (cfg.kv_lora_rank + cfg.qk_rope_head_dim) * 2
    del graph, wrapper, kv
    gc.collect()
    torch.cuda.empty_cache()
    return {"value": round(B * 1e3 / ms, 2), "unit": "tok/s (aggregate over the batch)", "batch": B, "ms_per_step": round(ms, 4), "ctx": ctx,
            "hip_graph": graph_ok, "expected_distinct_experts_per_layer": round(distinct, 1),
            "whole_step": {"algorithmic_bytes": int(nbytes), "GBs": round(nbytes / (ms * 1e-3) / 1e9, 1),
                           "frac_of_hbm_peak": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "path": "flashinfer_attn (five launches per layer: the one-launch attention step covers one token) + MoE decode kernels on B rows"}


def run_experts_decode(name, args, dev, steps):
    """`kind: experts` workloads (BASELINE.json configs[0]: kt-kernel/bench/bench_moe.py's scope): the routed experts of every
    layer alone, one token through all of them per step, one HIP graph."""
    wl = WORKLOADS[name]
    layers = build_layers(wl, dev, max_len=16)
    r = DecodeRunner(wl, layers, 1, dev)
    r.capture()
    for i in range(30):
        r.step(i)
    dt = timed(r.step, steps, 10, dev, False)
    ms = dt / steps * 1e3
    gu, dn = expert_bpw(wl)
    nbytes = wl["L"] * wl["k"] * (2 * wl["H"] * wl["I"] * gu + wl["H"] * wl["I"] * dn)
    res = {"value": round(steps / dt, 2), "unit": "tok/s (routed experts of all layers only)", "ms_per_step": round(ms, 4),
           "layers": wl["L"], "hip_graph": True, "us_per_layer": round(ms * 1e3 / wl["L"], 2),
           "whole_step": {"algorithmic_bytes": int(nbytes), "GBs": round(nbytes / (ms * 1e-3) / 1e9, 1),
                          "frac_of_hbm_peak": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
    for h in layers:
        h.close()
    del r, layers
    gc.collect()
    torch.cuda.empty_cache()
    return res


def run_experts_prefill(name, dev, T=2048, reps=2):
    """`kind: experts` workloads: one T-token chunk through the routed experts of every layer (grouped prompt kernels), uniform routing."""
    wl = WORKLOADS[name]
    layers = build_layers(wl, dev, max_len=T)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    x = (torch.randn((T, wl["H"]), generator=g, device=dev) / 10).to(torch.bfloat16)
    ids = torch.stack([torch.randperm(wl["E"], generator=g, device=dev)[:wl["k"]] for _ in range(T)]).to(torch.int64)
    w = torch.rand((T, wl["k"]), generator=g, device=dev)
    y = torch.empty_like(x)
    times = []
    for r in range(reps + 1):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for h in layers:
            h.forward(x, ids, w, out=y)
        torch.cuda.synchronize(dev)
        if r:
            times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    gu, dn = expert_bpw(wl)
    nbytes = wl["L"] * wl["E"] * (2 * wl["H"] * wl["I"] * gu + wl["H"] * wl["I"] * dn)
    flop = wl["L"] * 2.0 * 3 * wl["H"] * wl["I"] * wl["k"] * T
    t_hbm, t_mfma = nbytes / (HBM_PEAK_GBS * 1e9), flop / MFMA_PEAK_I8
    hbm = t_hbm >= t_mfma
    for h in layers:
        h.close()
    del layers
    gc.collect()
    torch.cuda.empty_cache()
    return {"value": round(T / dt, 1), "unit": "tok/s (routed experts of all layers only)", "tokens": T, "ms_per_chunk": round(dt * 1e3, 3),
            "layers": wl["L"],
            "roofline": {"bound": "hbm" if hbm else "mfma", "achieved": round(nbytes / dt / 1e9, 1) if hbm else round(flop / dt / 1e12, 1),
                         "peak": HBM_PEAK_GBS if hbm else MFMA_PEAK_I8 / 1e12, "unit": "GB/s" if hbm else "TOP/s",
                         "frac": round(max(t_hbm, t_mfma) / dt, 4), "rows_per_expert": round(T * wl["k"] / wl["E"], 1)}}


def llamafile_cpu_leg(wl, budget_s=10.0):
    """CPU leg of the q4_k_m workload: the reference's own iqk kernels (third_party/llamafile/iqk_mul_mat.inc compiled unmodified
    into oracle/_ref/libiqk_ref_*.so) over one layer's top-k experts, all host threads.  None when the library is not there."""
    try:
        from oracle.gguf_ref import iqk_forward_bench            # noqa: F401
    except Exception:
        return None
    try:
        return iqk_forward_bench(wl["H"], wl["I"], wl["k"], wl["ggml"], wl["L"], budget_s)
    except Exception as e:
        return {"value": None, "error": f"{type(e).__name__}: {e}"[:200]}


class RunClock:
    """Wall clock of one bench run: seconds per section (-> `timing_s` of the JSON line) and the point after which no further
    OPTIONAL section is started (--time-budget), so that the single JSON line — printed at the end — is not lost to a caller's
    time limit."""

    def __init__(self, budget_s: float, now=time.perf_counter):
        self.now, self.budget_s = now, float(budget_s)
        self.t_start = now()
        self.timing = {}

    def lap(self, name: str, t0: float) -> None:
        self.timing[name] = round(self.now() - t0, 1)

    def elapsed(self) -> float:
        return self.now() - self.t_start

    def over_budget(self) -> bool:
        return self.elapsed() > self.budget_s


def truncated_line(out: dict, clock: RunClock, signum: int) -> dict:
    """The line as far as the run got when a SIGTERM arrives after the headline measurement: marked `truncated`, with the
    section timings, and with the whole-step rate as `roofline` if the per-kernel pass had not been reached (never overwriting
    a measured one)."""
    line = dict(out)
    line["truncated"] = f"signal {signum} after {clock.elapsed():.0f}s: the sections not yet run are missing"
    line["timing_s"] = dict(clock.timing)
    if "roofline" not in line:
        ws = line["whole_step"]
        line["roofline"] = {"bound": "hbm", "kernel": "whole decode step (per-kernel pass not reached)", "achieved": ws["GBs"],
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ws["frac_of_hbm_peak"], "traffic": None}
    return line


LINE_LIMIT = 4096      # bytes: the driver parses the LAST stdout line; round 4's 25 KB line was recorded as `parsed: null`


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(out: dict) -> dict:
    """The ONE stdout line of a run: the driver contract's fields + `roofline` + `cpu_baseline` + one (decode, prefill) number
    pair per secondary workload — numbers only, under LINE_LIMIT bytes.  Everything else (per-kernel tables, prose, windows,
    box, sample descriptions) is the detail record written next to it (`write_detail`)."""
    cfg = out.get("config") or {}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    line["metric"] = str(line["metric"])[:160]
    line["config"] = dict(_pick(cfg, ("hidden", "intermediate", "experts", "top_k", "heads", "layers", "dense_layers", "moe_layers",
                                      "vocab", "ctx", "batch_per_gpu", "parallelism", "rccl_ranks", "dist_backend", "hip_graph", "ep_transport_status")),
                          workload=str(cfg.get("workload", ""))[:200])
    if cfg.get("ep_transport"):
        line["config"]["ep_transport"] = str(cfg["ep_transport"])[:40]
    for k in ("error", "truncated"):
        if out.get(k):
            line[k] = str(out[k])[:160]
    rf = out.get("roofline") or {}
    line["roofline"] = dict({k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")},
                            **_pick(rf, ("algorithmic_bytes_per_launch", "avg_launch_us", "launches_per_step", "share_of_step")),
                            kernel=str(rf.get("kernel", ""))[:64])
    ws = out.get("whole_step") or {}
    line["whole_step"] = _pick(ws, ("GBs", "frac_of_hbm_peak", "moe_layer_kernel_us", "moe_layer_frac_of_hbm_peak"))
    if out.get("median_tok_s") is not None:
        line["median_tok_s"] = out["median_tok_s"]
    fd = out.get("full_depth_extrapolation") or {}
    if fd.get("tok_s") is not None:
        line["full_depth_extrapolation"] = _pick(fd, ("layers", "tok_s", "ms_per_step"))
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("value", "unit", "cores", "kind", "us_per_layer", "GBs", "physical_cores", "error"))
        c["sample"] = str(cb.get("sample", ""))[:150]
        quota = (cb.get("host") or {}).get("cpu_quota")
        if quota is not None:
            c["cpu_quota"] = quota
        if isinstance(cb.get("prefill"), dict):
            c["prefill"] = _pick(cb["prefill"], ("value", "unit", "chunk_tokens", "error"))
        line["cpu_baseline"] = c

    def _prefill(p):
        if not isinstance(p, dict):
            return None
        r = _pick(p, ("value", "unit", "tokens", "ms_per_chunk", "error"))
        prf = p.get("roofline")
        if isinstance(prf, dict):
            r["roofline"] = _pick(prf, ("bound", "achieved", "peak", "unit", "frac"))
        if "error" in r:
            r["error"] = str(r["error"])[:100]
        return r

    if isinstance(out.get("bs8"), dict):
        b8 = out["bs8"]
        line["bs8"] = dict(_pick(b8, ("value", "ms_per_step", "batch", "error")),
                           **({"frac_of_hbm_peak": b8["whole_step"]["frac_of_hbm_peak"]} if isinstance(b8.get("whole_step"), dict) else {}))
        if "error" in line["bs8"]:
            line["bs8"]["error"] = str(line["bs8"]["error"])[:100]
    if out.get("prefill") is not None:
        line["prefill"] = _prefill(out["prefill"])
    if out.get("prefill_8192") is not None:
        line["prefill_8192"] = _prefill(out["prefill_8192"])
    sec = {}
    for name in SECONDARY:
        key = name.replace("-", "_")
        r2 = out.get(key)
        if not isinstance(r2, dict):
            continue
        s = _pick(r2, ("value", "ms_per_step", "layers"))
        if r2.get("value") is None:
            s["value"] = None
            s["why"] = str(r2.get("error") or r2.get("skipped") or "")[:100]
        if isinstance(r2.get("whole_step"), dict):
            s["frac_of_hbm_peak"] = r2["whole_step"].get("frac_of_hbm_peak")
        pf = r2.get("prefill")
        if isinstance(pf, dict):
            s["prefill"] = pf.get("value")
            if isinstance(pf.get("roofline"), dict):
                s["prefill_frac"] = pf["roofline"].get("frac")
                s["prefill_bound"] = pf["roofline"].get("bound")
        pl = r2.get("prefill_8192")
        if isinstance(pl, dict) and pl.get("value") is not None:
            s["prefill_8192"] = pl.get("value")
            if isinstance(pl.get("roofline"), dict):
                s["prefill_8192_frac"] = pl["roofline"].get("frac")
        if isinstance(r2.get("ctx_131072"), dict):
            s["ctx_131072"] = r2["ctx_131072"].get("value")
        if "exact" in r2:
            s["exact"] = r2["exact"]
        if isinstance(r2.get("bs8"), dict) and r2["bs8"].get("value") is not None:
            s["bs8"] = r2["bs8"]["value"]
            s["bs8_frac"] = r2["bs8"]["whole_step"]["frac_of_hbm_peak"]
        if isinstance(r2.get("cpu_llamafile"), dict):
            s["cpu_llamafile"] = _pick(r2["cpu_llamafile"], ("value", "cores"))
        sec[key] = s
    if sec:
        line["secondary"] = sec
    if out.get("timing_s"):
        line["timing_s"] = {"total": out["timing_s"].get("total")}
    line["detail"] = DETAIL_FILE
    # last resort (a pathological error string, a future field): drop optional blocks until the line fits
    for k in ("timing_s", "full_depth_extrapolation", "median_tok_s", "whole_step", "secondary", "bs8", "prefill_8192", "prefill"):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        line.pop(k, None)
    return line


DETAIL_FILE = "bench_detail.json"


def write_detail(out: dict) -> None:
    """The full record of the run (what round 1-4 printed as one line), beside the compact line: `bench_detail.json` in the
    working directory and, when it exists, under gpurun_out/ (the directory gpurun merges back)."""
    text = json.dumps(out)
    for d in (".", "gpurun_out"):
        try:
            if d == "." or os.path.isdir(d):
                with open(os.path.join(d, DETAIL_FILE), "w") as f:
                    f.write(text + "\n")
        except OSError as e:
            log(f"[bench] could not write {d}/{DETAIL_FILE}: {e}")


def emit(out: dict) -> None:
    """Detail to the side file, then the compact line as the LAST (and only) stdout line."""
    write_detail(out)
    sys.stdout.flush()
    print(json.dumps(compact_line(out)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="v3-int4", choices=sorted(WORKLOADS))
    ap.add_argument("--layers", type=int, default=0, help="decoder layers of the resident layer subset (0 = workload default)")
    ap.add_argument("--prefill-tokens", type=int, default=2048)
    ap.add_argument("--ctx", type=int, default=4096, help="cached tokens the MLA decode attends over")
    ap.add_argument("--windows", type=int, default=4, help="extra timed windows of --steps steps after the contract's one (median reported)")
    ap.add_argument("--trace-steps", type=int, default=20, help="decode steps the in-graph kernel table averages over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-prefill", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the batch-of-8 decode step (serving seam)")
    ap.add_argument("--no-prefill-long", action="store_true", help="skip the 8192-token prompt chunk (the reference's default chunk_size)")
    ap.add_argument("--no-kernels", action="store_true", help="skip the per-kernel table / roofline")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (roofline.traffic = null)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other BASELINE.json configurations")
    ap.add_argument("--secondary", default=",".join(SECONDARY), help="comma-separated secondary workloads")
    ap.add_argument("--time-budget", type=float, default=1500.0,
                    help="seconds of wall clock after which no further OPTIONAL section (PMC passes, secondary workloads) is started: the "
                         "JSON line is printed once, at the end, and must not be lost to a caller's time limit")
    ap.add_argument("--strong", action="store_true", help="N > 1: ONE token stream (every rank decodes the same token; routed experts "
                                                            "E/N per rank) instead of one stream per rank")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="CPU leg: total worker threads (0 = one per physical core)")
    ap.add_argument("--cpu-budget", type=float, default=16.0, help="CPU leg: seconds of bs=1 forwards")
    ap.add_argument("--force-dist", action="store_true",
                    help="dev / test aid: take the N > 1 code path (process group, expert parallelism, exchange transport) "
                         "even with WORLD_SIZE=1, so that path can be run on a one-GPU box")
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                    help="process-group backend of the N > 1 path.  nccl (= RCCL over xGMI) is the product path; gloo exists for the "
                         "one-GPU smoke of the N-process code path (ranks share cuda:0, the decode exchange still runs through the "
                         "peer-write transport over inter-process handles; RCCL refuses several ranks on one device)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--trace-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(wl, budget_s=args.cpu_budget, prefill_budget_s=args.cpu_budget / 2, threads=args.cpu_threads or None)), flush=True)
        return

    clock = RunClock(args.time_budget)
    t_start, timing, lap, over_budget = clock.t_start, clock.timing, clock.lap, clock.over_budget

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or args.force_dist
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    shared_gpu = dist_on and args.dist_backend == "gloo" and local_rank >= torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % torch.cuda.device_count() if shared_gpu else local_rank)
    torch.cuda.set_device(dev)
    if args.pmc_child:
        pmc_child(args, wl, dev)
        return
    if args.trace_child:
        trace_child(args, wl, dev)
        return
    ep_transport, ep_exchange = None, None
    if dist_on:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
        assert wl["E"] % world == 0
        from ktransformers_amd.parallel import enable_expert_parallel, enable_peer_exchange
        enable_expert_parallel()      # experts sharded over the ranks, attention / dense parts replicated
        if args.strong:
            from ktransformers_amd.parallel import set_replicated_input
            set_replicated_input(True)   # one stream: the ranks' rows are identical, only the fp32 partials travel
        # decode exchange: direct peer writes over xGMI (two launches per MoE layer; checked on this node's fabric while it
        # is set up), else the two collectives.  KTX_EP_TRANSPORT=collectives forces the latter for an A/B.
        ep_transport = "collectives: all-gather + reduce-scatter per MoE layer (RCCL)"
        if os.environ.get("KTX_EP_TRANSPORT", "peer") != "collectives":
            try:
                ep_exchange = enable_peer_exchange(wl["H"], wl["k"], 16, dev)
                # ranks capture their graphs at their own pace; a dead peer still ends the wait (KTX_EP_SPIN_SECONDS: the bound)
                ep_exchange.set_spin_seconds(float(os.environ.get("KTX_EP_SPIN_SECONDS", "120")))
                ep_transport = ("peer writes: tagged 8-byte granules into the peers' buffers over xGMI, two launches per MoE "
                                "layer (ktx_ep_gather / ktx_ep_reduce), partials added in rank order")
            except Exception as e:      # raised on every rank alike: all ranks fall back together, and the line says so
                ep_transport += f" [peer-write transport refused: {type(e).__name__}: {e}]"[:400]
                log(f"[bench] {ep_transport}")

    from ktransformers_amd import _native  # noqa: F401  (raises if the HIP library is missing: no CPU fallback)

    box = box_info() if rank == 0 else None     # before any work: `idle_power_w` is the idle socket power of this box
    if wl.get("kind") == "experts":
        out = {"metric": f"decode tokens/s ({wl['name']})", **run_experts_decode(args.workload, args, dev, args.steps),
               **({} if args.no_prefill else {"prefill": run_experts_prefill(args.workload, dev, args.prefill_tokens)}),
               "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "int8 x q4_k/q6_k -> int32 per block (fp32 out)", "data": "synthetic", "config": {"workload": wl["desc"]},
               "box": box}
        emit(out)
        return

    # N >= 2: the experts are sharded E/N per rank, so the WHOLE model fits from two GPUs on (327 GB of int4 experts / N)
    n_layers = args.layers or (wl["full_layers"] if (world >= 2 and wl["model"] == "v3") else wl["layers"])
    args.layers = n_layers
    if args.strong and dist_on:
        rank_seed = 0                                   # every rank starts from the same token: one stream
    else:
        rank_seed = rank
    t_sec = time.perf_counter()
    res, mr = run_model_decode(args.workload, args, dev, args.steps, args.warmup, dist_on, 1 if (args.strong and dist_on) else world,
                               rank_seed, n_layers, windows=args.windows)
    lap("decode", t_sec)
    cfg = mr.cfg
    ms_per_step = res["ms_per_step"]
    n_dense = res["dense_layers"]
    H, I, E, k = wl["H"], wl["I"], wl["E"], wl["k"]
    subset = n_layers < wl["full_layers"]
    out = {
        "metric": f"decode tokens/s ({wl['name']}: {wl['method']} routed experts + {wl['linear']} linears + MLA resident in HBM, "
                  f"whole-model greedy decode" + (f", {n_layers}-of-{wl['full_layers']}-layer subset" if subset else "") + ")",
        "value": res["value"], "unit": "tok/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if (args.strong and dist_on) else "weak",
        "vs_baseline": None,
        "dtype": {"AMXINT4": "int8xint4->int32 (bf16 io)", "GGUF": "q8_k x iq1_s/q4_k/q6_k -> int32 per block (fp32 acc)",
                  "FP8": "bf16 x e4m3 -> fp32 (bf16 io)", "RAWINT4": "int8xint4->int32 per 32-group (bf16 io)"}.get(wl["method"], wl["method"]),
        "data": "synthetic",
        "config": {"workload": f"{wl['desc']}: {n_dense} dense + {n_layers - n_dense} MoE layers"
                               + (f" of the model's {wl['dense']} + {wl['full_layers'] - wl['dense']}" if subset else "")
                               + f", decode bs=1 per GPU at ctx {args.ctx}",
                   "hidden": H, "intermediate": I, "experts": E, "top_k": k, "heads": wl["heads"], "layers": n_layers,
                   "dense_layers": n_dense, "moe_layers": n_layers - n_dense, "vocab": cfg.vocab_size, "ctx": args.ctx,
                   "batch_per_gpu": 1, "parallelism": f"ep{world}" if dist_on else "single",
                   "rccl_ranks": world if (dist_on and args.dist_backend == "nccl") else 0, "dist_backend": args.dist_backend if dist_on else None,
                   "ep_transport": ep_transport if dist_on else None,
                   "hip_graph": res["hip_graph"], "graph_error": res["graph_error"], "prewarm_steps": res["prewarm_steps"],
                   "step": "one greedy token through the YAML-injected model: embedding, per layer [RMSNorm, MLA attention operator "
                           "(q_a|kv_a, q_b, o projections, YaRN RoPE, absorb, paged MQA over the cached context, cache append), "
                           "RMSNorm, dense MLP | router + top-k routed experts + shared expert], RMSNorm, W4 lm_head, argmax; "
                           "token and position fed back inside the HIP graph; random weights"
                           + ("; routed experts sharded expert-parallel over %d ranks (every rank's token row gathered, local "
                              "experts, fp32 partials reduced at the token's home rank; transport in ep_transport), "
                              "attention / dense / router replicated, %s" % (world, "ONE token stream (all ranks decode the same token)"
                                                                            if args.strong else "one token stream per rank") if dist_on else "")},
        "whole_step": res["whole_step"],
        "windows_ms_per_step": res.get("windows_ms_per_step"), "median_ms_per_step": res.get("median_ms_per_step"),
        "median_tok_s": res.get("median_tok_s"),
        "box": box,
    }
    # the headline measurement exists from here on: a caller's SIGTERM (time limit) prints what there is instead of nothing
    import signal

    def _salvage(signum, frame):
        if rank == 0:
            emit(truncated_line(out, clock, signum))
        os._exit(0)

    try:
        signal.signal(signal.SIGTERM, _salvage)
    except (ValueError, OSError):
        pass
    if subset:
        # NOT `value`: what the measured per-layer time implies for the full depth (the extra layers are MoE layers)
        out["full_depth_extrapolation"] = {
            "note": f"extrapolated, not measured: the {wl['full_layers'] - n_layers} missing layers are MoE layers; their time is taken as "
                    "the in-graph kernel time of one MoE layer when the per-kernel table is present, else step time / layers",
            "layers": wl["full_layers"]}

    bs8 = None
    if not dist_on and not args.no_batched:
        t_sec = time.perf_counter()
        try:
            bs8 = batched_decode(mr, 8, args.ctx, 50, 10, dev)
        except Exception as e:
            bs8 = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.synchronize(dev)
        lap("bs8", t_sec)
    prefill = prefill_long = None
    if not dist_on and not args.no_prefill:
        # ---------------- prefill: one prompt chunk through the same resident model -----------------------------------------
        t_sec = time.perf_counter()
        try:
            prefill = whole_model_prefill(mr, args.prefill_tokens, dev)
        except Exception as e:
            prefill = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.synchronize(dev)
        lap("prefill", t_sec)
        if args.prefill_tokens != PREFILL_LONG and not args.no_prefill_long:
            t_sec = time.perf_counter()
            try:    # the reference's own chunk size: 256 rows per V3 expert, the MFMA-bound regime of SURVEY.md §8(d)
                prefill_long = whole_model_prefill(mr, PREFILL_LONG, dev, reps=2)
            except Exception as e:
                prefill_long = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
                torch.cuda.synchronize(dev)
            lap("prefill_8192", t_sec)
    rows_eager = None
    if dist_on and not args.no_kernels:
        # N > 1: every rank steps together (collectives), so the per-launch events run on all ranks; rank 0 reports
        rows_eager, lus_eager = kernel_table(mr, dev, ms_per_step)
        mr.set_position(args.ctx)
    mr.close()
    del mr
    gc.collect()
    torch.cuda.empty_cache()
    if bs8 is not None:
        out["bs8"] = bs8
    if prefill is not None:
        out["prefill"] = prefill
    if prefill_long is not None:
        out["prefill_8192"] = prefill_long

    lus = None
    if not args.no_kernels:
        t_sec = time.perf_counter()
        if not dist_on:
            # ---------------- per-kernel table of the replayed graph + roofline of its dominant kernel ----------------------
            rows, info = graph_kernel_table(args, ms_per_step)
            lap("per_kernel_trace_child", t_sec)
        else:
            rows, info = rows_eager, {"source": "HIP events around every library launch of eager decode steps on rank 0 (all ranks step "
                                                "together); N = 1 runs take this table from a rocprofv3 pass of the replayed graph",
                                      "moe_layer_us": lus_eager}
        if rows:
            out["per_kernel"] = rows[:24]
            out["per_kernel_info"] = info
            lus = info.get("moe_layer_us")
            top = next((r for r in rows if r["algorithmic_bytes_per_launch"]), rows[0])
            t_sec = time.perf_counter()
            traffic, src = (None, "skipped") if (args.no_pmc or dist_on) else \
                ((None, f"skipped: --time-budget {args.time_budget:.0f}s reached") if over_budget() else pmc_traffic(args, top["kernel"]))
            lap("pmc_child_passes", t_sec)
            out["roofline"] = {"bound": "hbm", "kernel": top["kernel"], "achieved": top["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round((top["GBs"] or 0.0) / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": src,
                               "algorithmic_bytes_per_launch": top["algorithmic_bytes_per_launch"],
                               "avg_launch_us": top["avg_launch_us"], "launches_per_step": top["launches_per_step"],
                               "share_of_step": top["share_of_step"],
                               "selection": "the kernel class with the largest total time per step in the per_kernel table "
                                            "(in-graph rocprofv3 durations at N = 1)"}
            ksum = sum(r["us_per_step"] for r in rows)
            out["whole_step"]["sum_of_kernels_us"] = round(ksum, 1)
            if lus:
                lb = out["whole_step"]["moe_layer_bytes"]
                out["whole_step"]["moe_layer_kernel_us"] = round(lus, 1)
                out["whole_step"]["moe_layer_frac_of_hbm_peak"] = round(lb / (lus * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        else:
            out["per_kernel_info"] = info
    if subset:
        ms_full = ms_per_step + (wl["full_layers"] - n_layers) * lus * 1e-3 if lus else ms_per_step * wl["full_layers"] / n_layers
        out["full_depth_extrapolation"].update(ms_per_step=round(ms_full, 3), tok_s=round(1e3 / ms_full, 2))
    if "roofline" not in out:   # no per-kernel pass: whole-step rate as the only (honest) roofline figure
        ws = out["whole_step"]
        out["roofline"] = {"bound": "hbm", "kernel": "whole decode step (per-kernel pass skipped or failed)", "achieved": ws["GBs"],
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ws["frac_of_hbm_peak"], "traffic": None}

    if not dist_on:
        # the reference's CPU path first (short, and part of the contract's line), then the optional configurations
        if not args.no_cpu_baseline:
            t_sec = time.perf_counter()
            out["cpu_baseline"] = cpu_baseline_subprocess(args.workload)
            lap("cpu_baseline", t_sec)
            if isinstance(out.get("prefill"), dict) and isinstance(out["cpu_baseline"].get("prefill"), dict):
                out["prefill"]["cpu_baseline"] = dict(out["cpu_baseline"]["prefill"], cores=out["cpu_baseline"].get("cores"),
                                                      kind=out["cpu_baseline"].get("kind"))
        # ---------------- the other BASELINE.json configurations, as secondary fields -------------------------------------------
        if args.workload == "v3-int4" and not args.no_secondary:
            for name in [x for x in args.secondary.split(",") if x]:
                w2 = WORKLOADS[name]
                key = name.replace("-", "_")
                if over_budget():
                    out[key] = {"value": None, "skipped": f"--time-budget {args.time_budget:.0f}s reached after "
                                                          f"{time.perf_counter() - t_start:.0f}s; run bench.py --workload {name}"}
                    continue
                t_sec = time.perf_counter()
                try:
                    n2 = max(30, min(args.steps, 100))
                    if w2.get("kind") == "experts":
                        r2 = run_experts_decode(name, args, dev, n2)
                        if not args.no_prefill:
                            try:
                                r2["prefill"] = run_experts_prefill(name, dev, args.prefill_tokens)
                                if args.prefill_tokens != PREFILL_LONG and not args.no_prefill_long:
                                    r2["prefill_8192"] = run_experts_prefill(name, dev, PREFILL_LONG, reps=1)
                            except Exception as e:
                                r2.setdefault("prefill", {"value": None, "error": f"{type(e).__name__}: {e}"[:300]})
                                torch.cuda.synchronize(dev)
                        if not args.no_cpu_baseline:
                            r2["cpu_llamafile"] = llamafile_cpu_leg(w2)
                    else:
                        r2, m2 = run_model_decode(name, args, dev, n2, 10)
                        if not args.no_batched:
                            try:
                                r2["bs8"] = batched_decode(m2, 8, args.ctx, 30, 5, dev)
                            except Exception as e:
                                r2["bs8"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
                                torch.cuda.synchronize(dev)
                        if not args.no_prefill:
                            try:   # the same resident model, one prompt chunk (no per-launch pass: the headline workload carries that table)
                                r2["prefill"] = whole_model_prefill(m2, args.prefill_tokens, dev, reps=1, per_kernel_pass=True)
                                if args.prefill_tokens != PREFILL_LONG and not args.no_prefill_long:
                                    r2["prefill_8192"] = whole_model_prefill(m2, PREFILL_LONG, dev, reps=1, per_kernel_pass=False)
                            except Exception as e:
                                r2.setdefault("prefill", {"value": None, "error": f"{type(e).__name__}: {e}"[:300]})
                                torch.cuda.synchronize(dev)
                        m2.close()
                        del m2
                        if name == "r1-iq1s":     # BASELINE.json configs[4] names a 128K context: the same model again at 131072 cached tokens
                            gc.collect()
                            torch.cuda.empty_cache()
                            try:
                                r3, m3 = run_model_decode(name, args, dev, 30, 5, ctx=131072)
                                m3.close()
                                del m3
                                r2["ctx_131072"] = {k2: r3[k2] for k2 in ("value", "ms_per_step", "hip_graph", "whole_step")}
                            except Exception as e:
                                r2["ctx_131072"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
                    r2["workload"] = w2["desc"]
                    if w2.get("method") == "RAWINT4":
                        # the prompt path timed here is the default grouped kernel, which re-associates the fp32 sum over the K groups
                        # (<= 2 bf16 ulp from the reference's order, include/ktx_moe.h); KTX_MOE_EXACT=1 selects the exact one
                        r2["exact"] = os.environ.get("KTX_MOE_EXACT", "0") not in ("", "0")
                    out[key] = r2
                except Exception as e:
                    out[key] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
                    torch.cuda.synchronize(dev)
                lap(key, t_sec)
                gc.collect()
                torch.cuda.empty_cache()
            if "v2lite_int4" in out:
                out["v2lite"] = out["v2lite_int4"]          # (the name round 1 / 2 lines used)

    if dist_on and ep_exchange is not None:
        # a poll that gave up during the run means a rank computed on rows that never arrived: the number is void, say so
        st = torch.tensor([ep_exchange.status()], device=dev, dtype=torch.int32)
        dist.all_reduce(st, op=dist.ReduceOp.MAX)
        out["config"]["ep_transport_status"] = int(st.item())
        if int(st.item()) != 0:
            out["value"], out["error"] = None, "peer-write exchange: a poll gave up waiting for a peer during the run"
    timing["total"] = round(time.perf_counter() - t_start, 1)
    out["timing_s"] = timing
    try:
        signal.signal(signal.SIGTERM, signal.SIG_DFL)      # the line is complete: no salvage print beside it
    except (ValueError, OSError):
        pass
    if rank == 0:
        emit(out)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
