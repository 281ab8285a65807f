"""Generate tests/golden/quant_model_golden.npz: GREEDY DECODE through the reference's own modules with the QUANTISED
back-ends the reference's DeepSeek-V3 rule file selects — the end-to-end pin north_star asks for ("greedy-decode token-ids
identical" against the kt-kernel cpu_backend on identical weights).

Reference side, all on CPU:
  * decoder stack = the reference's own pure-torch modules (DeepseekV3DecoderLayer / DeepseekV3Attention (eager) /
    DeepseekV3RMSNorm / MoEGate / DeepseekV3MLP, archive/ktransformers/models/modeling_deepseek_v3.py) in bf16;
  * every linear the rule file turns into KLinearMarlin (all but kv_b_proj) carries the weights Marlin computes with:
    the reference's own quantize_weights (custom_marlin/quantize/utils/quant_utils.py, imported by path) at 4 bit / group
    64, de-quantised (q - 8) * s  (KLinearMarlin.forward = x @ dequant(W), linear.py:676-714);
  * the routed experts of every MoE layer run on the reference's OWN cpu_backend kernels — TP_MOE<AMX_MOE_TP<
    GemmKernel224Int4>> compiled unmodified into oracle/_ref/libkt_ref.so — online-quantised from the same bf16 expert
    weights (the AMXInt4 backend of KExpertsCPU, experts.py:226-248), called exactly where KDeepseekV3MoE.forward calls
    them (experts.py:974-1012): gate -> routed experts -> + shared experts;
  * generation = the reference's loop (util/utils.py:483-494, do_sample=False): argmax of the last position's logits,
    token fed back; no KV cache on this side (the whole sequence is re-run every step: same function).

The model is tiny (4 layers: 1 dense + 3 MoE, hidden 256, 8 experts top-2, vocab 512) so the CPU run takes seconds.  A model
with i.i.d. random weights has no stable greedy path: its logits are Gaussian, the top-2 gap falls below bf16 noise every
few tokens, and the reference's OWN bf16 and fp32 runs then pick different tokens (measured: they diverge within ~10 tokens
for every seed tried).  A token-id comparison is only meaningful where the reference's own choice is stable, so the output
head is tied to the embedding through a fixed cyclic permutation pi of the vocabulary (lm_head[pi(i)] = embed[i]; everything
else stays i.i.d. random): the residual stream carries the current token's embedding, which makes pi(current token) the
favoured next token, while attention, the dense MLP and the three MoE layers contribute ~85 % of the final hidden state's
energy (embedding rms 3, layer outputs rms ~1, 4, 4, 4) — an error in any of them moves the logits and, if large, the
tokens.  The script still scans seeds IN ORDER
and keeps the first one for which (a) the bf16 reference and the same reference run in fp32 arithmetic pick identical
tokens at all 40 generated positions and (b) the smallest top-2 logit margin along the way exceeds MIN_MARGIN times the
logit standard deviation.  Stored: weights (bf16 bits), prompt, the 40 generated ids, per-step logits and margins,
and for every MoE layer the expert block's (input rows, expert ids, routing weights, output rows) at the last step, with
which tests/test_quant_model_gpu.py checks the HIP expert block on identical inputs.

    python tests/golden/make_quant_model_golden.py          (needs /root/reference and oracle/_ref)
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from ref_import import reference_models  # noqa: E402

from oracle.oracle import FMT_AMXINT4, Reference  # noqa: E402

v3, DeepseekV3Config = reference_models()
spec = importlib.util.spec_from_file_location(
    "ref_quant_utils", "/root/reference/archive/ktransformers/ktransformers_ext/operators/custom_marlin/quantize/utils/quant_utils.py")
qu = importlib.util.module_from_spec(spec)
spec.loader.exec_module(qu)

CFG = dict(vocab_size=512, hidden_size=256, intermediate_size=512, moe_intermediate_size=128, num_hidden_layers=4,
           num_attention_heads=2, n_shared_experts=1, n_routed_experts=8, num_experts_per_tok=2, first_k_dense_replace=1,
           moe_layer_freq=1, n_group=2, topk_group=1, topk_method="noaux_tc", scoring_func="sigmoid", norm_topk_prob=True,
           routed_scaling_factor=2.5, q_lora_rank=64, kv_lora_rank=512, qk_rope_head_dim=64, qk_nope_head_dim=128,
           v_head_dim=128, max_position_embeddings=4096, rope_theta=10000.0, rms_norm_eps=1e-6, attention_bias=False,
           rope_scaling={"type": "yarn", "factor": 40, "mscale": 1.0, "mscale_all_dim": 1.0,
                         "original_max_position_embeddings": 4096, "beta_fast": 32, "beta_slow": 1})
T_PROMPT, N_NEW, MIN_MARGIN, GROUP = 9, 40, 0.5, 64
REF = Reference(threads=4)


def marlin_weights(w_bf16: torch.Tensor) -> torch.Tensor:
    """[N, K] bf16 -> the [N, K] bf16 weights KLinearMarlin multiplies with (linear.py:645-666: quantise weight.T)."""
    q, s, _, _ = qu.quantize_weights(w_bf16.T.contiguous(), 4, GROUP, False)
    return ((q.float() - 8.0) * s.float().repeat_interleave(GROUP, dim=0)).T.contiguous().to(torch.bfloat16)


class RefExperts(torch.nn.Module):
    """KTransformersExperts(generate_op=KExpertsCPU, backend=AMXInt4).forward(x, ids, w) on the reference's own kernels."""

    def __init__(self, gate, up, down, k, record):
        super().__init__()
        self.moe = REF.make_moe(FMT_AMXINT4, gate, up, down, k=k, max_len=256)
        self.record = record

    def forward(self, x, ids, w):
        xb = x.to(torch.bfloat16).contiguous()
        y = REF.moe_forward(self.moe, ids.numpy().astype(np.int64), w.float().numpy(), xb.view(torch.uint16).numpy())
        self.record["last"] = (xb.view(torch.uint16).numpy().copy(), ids.numpy().copy(), w.float().numpy().copy(), y.copy())
        return torch.from_numpy(y.view(np.int16).copy()).view(torch.bfloat16).to(x.dtype)


def moe_forward(self, hidden_states):
    """KDeepseekV3MoE.forward (archive/ktransformers/operators/experts.py:974-1012), prefill branch."""
    identity = hidden_states
    orig_shape = hidden_states.shape
    topk_idx, topk_weight = self.gate(hidden_states)
    hidden_states = hidden_states.view(-1, hidden_states.shape[-1])
    y = self.kexperts(hidden_states, topk_idx, topk_weight).view(*orig_shape)
    return y + self.shared_experts(identity)


def build(seed, dtype):
    cfg = DeepseekV3Config(**CFG, attention_dropout=0.0, hidden_act="silu")
    cfg._attn_implementation = "eager"
    torch.set_default_dtype(dtype)
    try:
        embed = torch.nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        layers = torch.nn.ModuleList([v3.DeepseekV3DecoderLayer(cfg, i) for i in range(cfg.num_hidden_layers)])
        norm = v3.DeepseekV3RMSNorm(cfg.hidden_size, cfg.rms_norm_eps)
        head = torch.nn.Linear(cfg.hidden_size, cfg.vocab_size, bias=False)
    finally:
        torch.set_default_dtype(torch.float32)
    root = torch.nn.Module()
    root.model = torch.nn.Module()
    root.model.embed_tokens, root.model.layers, root.model.norm, root.lm_head = embed, layers, norm, head
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, p in root.named_parameters():
        if "layernorm" in name or name.endswith("norm.weight"):
            w = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        elif "e_score_correction_bias" in name:
            w = 0.1 * torch.randn(p.shape, generator=g)
        elif "embed_tokens" in name:
            w = 3.0 * torch.randn(p.shape, generator=g)          # rms 3 against ~7 for what the four layers add
        elif ".mlp.experts." in name:
            w = torch.randn(p.shape, generator=g) / 10           # randn/10 like the reference's MoE tests
        else:
            w = torch.randn(p.shape, generator=g) / (p.shape[-1] ** 0.5)
        sd[name] = w.to(torch.bfloat16)
    # head tied to the embedding through a cyclic permutation: lm_head[pi(i)] = embed[i]
    order = torch.randperm(cfg.vocab_size, generator=g)
    pi = torch.empty_like(order)
    pi[order] = order.roll(-1)
    head_w = torch.empty_like(sd["lm_head.weight"])
    head_w[pi] = sd["model.embed_tokens.weight"]
    sd["lm_head.weight"] = head_w
    eff = dict(sd)                                               # what the quantised back-ends compute with
    for name in sd:
        # every nn.Linear of the decoder layers except kv_b_proj, plus lm_head (DeepSeek-V3-Chat.yaml:10-30); the router's
        # weight is a plain Parameter of MoEGate, not a Linear, and stays as it is
        is_linear = sd[name].dim() == 2 and "embed_tokens" not in name and not name.endswith("mlp.gate.weight")
        if is_linear and ".mlp.experts." not in name and "kv_b_proj" not in name:
            eff[name] = marlin_weights(sd[name])
    root.load_state_dict({k: v.to(dtype) for k, v in eff.items()}, strict=True)
    root.eval()
    records = []
    for layer in root.model.layers:
        if hasattr(layer.mlp, "experts"):
            pre = [n for n in sd if False]
            E = cfg.n_routed_experts
            li = len(records) + cfg.first_k_dense_replace
            stack = lambda proj: np.stack([sd[f"model.layers.{li}.mlp.experts.{e}.{proj}_proj.weight"].view(torch.uint16).numpy()
                                           for e in range(E)])
            rec = {}
            records.append(rec)
            layer.mlp.kexperts = RefExperts(stack("gate"), stack("up"), stack("down"), cfg.num_experts_per_tok, rec)
            layer.mlp.forward = moe_forward.__get__(layer.mlp)
    return cfg, root, sd, records


@torch.no_grad()
def logits_of(root, ids, dtype):
    h = root.model.embed_tokens(ids)
    n = ids.shape[1]
    pos = torch.arange(n).unsqueeze(0)
    mask = torch.full((n, n), float("-inf")).triu(1)[None, None].to(dtype)
    for layer in root.model.layers:
        h = layer(h, attention_mask=mask, position_ids=pos)[0]
    return root.lm_head(root.model.norm(h)).float()[0]


@torch.no_grad()
def hidden_states(root, ids, dtype):
    """The residual stream after every decoder layer (+ the final norm) for the whole sequence `ids`."""
    h = root.model.embed_tokens(ids)
    n = ids.shape[1]
    pos = torch.arange(n).unsqueeze(0)
    mask = torch.full((n, n), float("-inf")).triu(1)[None, None].to(dtype)
    out = [h[0].float().numpy().copy()]
    for layer in root.model.layers:
        h = layer(h, attention_mask=mask, position_ids=pos)[0]
        out.append(h[0].float().numpy().copy())
    return np.stack(out)


def generate(root, prompt, dtype):
    ids, toks, lg, margins = prompt.clone(), [], [], []
    for _ in range(N_NEW):
        last = logits_of(root, ids, dtype)[-1]
        top2 = last.topk(2).values
        toks.append(int(last.argmax()))
        lg.append(last.numpy().copy())
        margins.append(float(top2[0] - top2[1]))
        ids = torch.cat([ids, torch.tensor([[toks[-1]]])], dim=1)
    return toks, np.stack(lg), np.array(margins, np.float32)


chosen = None
for seed in range(100, 140):
    cfg, root_bf16, sd, rec_bf16 = build(seed, torch.bfloat16)
    prompt = torch.randint(0, CFG["vocab_size"], (1, T_PROMPT), generator=torch.Generator().manual_seed(seed + 1))
    toks, lg, margins = generate(root_bf16, prompt, torch.bfloat16)
    _, root_f32, _, _ = build(seed, torch.float32)
    toks32, lg32, margins32 = generate(root_f32, prompt, torch.float32)
    ok = toks == toks32 and min(margins.min(), margins32.min()) > MIN_MARGIN * lg32.std()
    print(f"seed {seed}: bf16/fp32 tokens agree {toks == toks32}, min margin bf16 {margins.min():.3f} fp32 {margins32.min():.3f}, "
          f"logit std {lg32.std():.2f}, rel(bf16 vs fp32 logits) {np.linalg.norm(lg - lg32) / np.linalg.norm(lg32):.4f} -> "
          f"{'KEEP' if ok else 'skip'}", flush=True)
    if ok:
        full = torch.cat([prompt, torch.tensor([toks[:-1]])], dim=1)       # the sequence whose last position yields toks[-1]
        hid16, hid32 = hidden_states(root_bf16, full, torch.bfloat16), hidden_states(root_f32, full, torch.float32)
        chosen = (seed, cfg, sd, prompt, toks, lg, lg32, margins, margins32, rec_bf16, hid16, hid32)
        break
assert chosen is not None, "no seed in range met the stability criterion"
seed, cfg, sd, prompt, toks, lg, lg32, margins, margins32, recs, hid16, hid32 = chosen
out = {f"w.{k}": v.view(torch.uint16).numpy() for k, v in sd.items()}
out.update(seed=np.int64(seed), prompt=prompt[0].numpy(), tokens=np.array(toks, np.int64), logits_bf16=lg, logits_f32=lg32,
           margin_bf16=margins, margin_f32=margins32, min_margin=np.float32(MIN_MARGIN),
           # residual stream after the embedding and after each layer over the whole 48-token sequence (bf16 run as bf16 bits)
           hidden_f32=hid32.astype(np.float32), hidden_bf16=torch.from_numpy(hid16).to(torch.bfloat16).view(torch.uint16).numpy())
for i, rec in enumerate(recs):        # the expert block's last call (all positions of the final step) per MoE layer
    x, ids, w, y = rec["last"]
    out[f"moe{i}.x"], out[f"moe{i}.ids"], out[f"moe{i}.w"], out[f"moe{i}.y"] = x, ids, w, y
path = os.path.join(HERE, "quant_model_golden.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes; seed", seed, "tokens", toks)
