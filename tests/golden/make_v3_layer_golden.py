"""Generate tests/golden/v3_layer_golden.npz: ONE DeepSeek-V3 decoder layer at the PUBLISHED dimensions (256 routed experts of
7168 x 2048, 128 heads, q_lora 1536 — tests/v3_layer_case.py) through the reference's own code, the end-to-end pin at real size
that the 4-layer toy of make_quant_model_golden.py cannot give.

Reference side, all on the CPU of the build container:
  * DeepseekV3DecoderLayer / DeepseekV3Attention (eager) / DeepseekV3RMSNorm / MoEGate / DeepseekV3MLP from
    archive/ktransformers/models/modeling_deepseek_v3.py, once in bf16 and once in fp32 arithmetic (the yardstick for what
    a bf16 pipeline may drift);
  * every linear the reference's DeepSeek-V3-Chat.yaml turns into KLinearMarlin carries Marlin's multiplicand bf16((q-8)*s)
    from the reference's own quantize_weights at 4 bit / group 64 (linear.py:645-714); kv_b_proj stays bf16;
  * the 256 routed experts run on the reference's OWN cpu_backend: TP_MOE<AMX_MOE_TP<GemmKernel224Int4>> compiled unmodified
    into oracle/_ref/libkt_ref.so, online-quantised from the bf16 expert weights, called where KDeepseekV3MoE.forward calls
    them (experts.py:974-1012).
Input: T_PROMPT + 1 = 25 embedding rows at positions 0..24, causal.  Stored: the layer's output rows (bf16 run as bits, fp32
run), and what the MoE block saw and produced in the bf16 run — router input rows, expert ids, routing weights, routed-expert
output — so the GPU test can check the router and the expert block on identical inputs, bit for bit.

    python tests/golden/make_v3_layer_golden.py        (about 20 minutes and 45 GB of RAM: 22.5 GB of bf16 expert weights)
"""
import importlib.util
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from ref_import import reference_models  # noqa: E402
from v3_layer_case import CFG, E, H, I, L, T_PROMPT, expert_weight, small_weights, token_ids  # noqa: E402

from oracle.oracle import FMT_AMXINT4, Reference  # noqa: E402

v3, DeepseekV3Config = reference_models()
spec = importlib.util.spec_from_file_location(
    "ref_quant_utils", "/root/reference/archive/ktransformers/ktransformers_ext/operators/custom_marlin/quantize/utils/quant_utils.py")
qu = importlib.util.module_from_spec(spec)
spec.loader.exec_module(qu)
GROUP = 64
t_start = time.time()


def log(msg):
    print(f"[{time.time() - t_start:7.1f}s] {msg}", flush=True)


def marlin_weights(w_bf16):
    """[N, K] bf16 -> the [N, K] bf16 weights KLinearMarlin multiplies with (linear.py:645-666: quantise weight.T)."""
    q, s, _, _ = qu.quantize_weights(w_bf16.T.contiguous(), 4, GROUP, False)
    return ((q.float() - 8.0) * s.float().repeat_interleave(GROUP, dim=0)).T.contiguous().to(torch.bfloat16)


# ---- the reference's cpu_backend, loaded once, shared by the bf16 and the fp32 run -------------------------------------------
REF = Reference(threads=os.cpu_count() or 8)
stacks = {}
for proj in ("gate", "up", "down"):
    shape = (E, H, I) if proj == "down" else (E, I, H)
    a = np.empty(shape, np.uint16)
    for e in range(E):
        a[e] = expert_weight(e, proj).view(torch.uint16).numpy()
    stacks[proj] = a
    log(f"{proj} experts generated")
MOE = REF.make_moe(FMT_AMXINT4, stacks["gate"], stacks["up"], stacks["down"], k=CFG["num_experts_per_tok"], max_len=64)
del stacks
log("reference AMXINT4 experts quantised")


class RefExperts(torch.nn.Module):
    def __init__(self, record):
        super().__init__()
        self.record = record

    def forward(self, x, ids, w):
        xb = x.to(torch.bfloat16).contiguous()
        y = REF.moe_forward(MOE, ids.numpy().astype(np.int64), w.float().numpy(), xb.view(torch.uint16).numpy())
        self.record.update(x=xb.view(torch.uint16).numpy().copy(), ids=ids.numpy().copy(), w=w.float().numpy().copy(), y=y.copy())
        return torch.from_numpy(y.view(np.int16).copy()).view(torch.bfloat16).to(x.dtype)


def moe_forward(self, hidden_states):
    """KDeepseekV3MoE.forward (archive/ktransformers/operators/experts.py:974-1012), prefill branch."""
    identity = hidden_states
    orig_shape = hidden_states.shape
    topk_idx, topk_weight = self.gate(hidden_states)
    hidden_states = hidden_states.view(-1, hidden_states.shape[-1])
    y = self.kexperts(hidden_states, topk_idx, topk_weight).view(*orig_shape)
    return y + self.shared_experts(identity)


def run(dtype, sd):
    cfg = DeepseekV3Config(**CFG, attention_dropout=0.0, hidden_act="silu")
    cfg._attn_implementation = "eager"
    # the layer constructor would allocate and initialise 256 full-size expert MLPs: skip the initialisation (the pages are never
    # touched) and drop the experts, which the reference's KDeepseekV3MoE replaces by its CPU backend anyway
    init = torch.nn.Linear.reset_parameters
    torch.nn.Linear.reset_parameters = lambda self: None
    torch.set_default_dtype(dtype)
    try:
        layer = v3.DeepseekV3DecoderLayer(cfg, 0)
    finally:
        torch.set_default_dtype(torch.float32)
        torch.nn.Linear.reset_parameters = init
    layer.mlp.experts = torch.nn.ModuleList()
    eff = {}
    for name, w in sd.items():
        if not name.startswith(L):
            continue
        is_linear = w.dim() == 2 and not name.endswith("mlp.gate.weight")
        eff[name[len(L):]] = (marlin_weights(w) if is_linear and "kv_b_proj" not in name else w).to(dtype)
    missing, unexpected = layer.load_state_dict(eff, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    layer.eval()
    rec = {}
    layer.mlp.kexperts = RefExperts(rec)
    layer.mlp.forward = moe_forward.__get__(layer.mlp)
    ids = torch.from_numpy(token_ids())
    h = sd["model.embed_tokens.weight"][ids].to(dtype)[None]
    n = h.shape[1]
    pos = torch.arange(n).unsqueeze(0)
    mask = torch.full((n, n), float("-inf")).triu(1)[None, None].to(dtype)
    with torch.no_grad():
        out = layer(h, attention_mask=mask, position_ids=pos)[0][0]
    return out.float().numpy(), rec


sd = small_weights()
log("small weights generated")
out16, rec16 = run(torch.bfloat16, sd)
log("bf16 run done")
out32, rec32 = run(torch.float32, sd)
log("fp32 run done")
rel = np.linalg.norm(out16 - out32, axis=1) / np.linalg.norm(out32, axis=1)
log(f"reference bf16 vs fp32 layer output, per token: median {np.median(rel):.4f} max {rel.max():.4f}")
same = [set(a) == set(b) for a, b in zip(rec16["ids"].tolist(), rec32["ids"].tolist())]
log(f"tokens whose routed expert set is the same in the bf16 and the fp32 run: {sum(same)} of {len(same)}")
out = dict(out_bf16=torch.from_numpy(out16).to(torch.bfloat16).view(torch.uint16).numpy(), out_f32=out32.astype(np.float32),
           moe_x=rec16["x"], moe_ids=rec16["ids"], moe_w=rec16["w"], moe_y=rec16["y"], ids_f32=rec32["ids"],
           token_ids=token_ids(), t_prompt=np.int64(T_PROMPT))
path = os.path.join(HERE, "v3_layer_golden.npz")
np.savez_compressed(path, **out)
log(f"{path} {os.path.getsize(path)} bytes")
