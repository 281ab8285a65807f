"""GPU, end to end on BASELINE config C1's model family: a small Mixtral (8 experts, top-2, grouped-query attention) whose weights
come from a GGUF file — Q4_K gate / up and Q6_K down experts (the q4_k_m mix, llamafile arithmetic), F32 everything else — is
injected with this package's Mixtral rule file and compared with a restatement that uses the SAME quantised weights:
  * linears: Marlin's W4-g64 quantiser (oracle/linear_ref.py, bit-exact against the reference's in test_linear_gpu.py), fp64 GEMM;
  * router: softmax -> top-2 -> renormalise -> bf16 weights (archive/ktransformers/operators/experts.py:1084-1090);
  * experts: oracle/ktx_oracle_gguf.c (LLAMA_MOE_TP::forward_one's arithmetic);
  * attention core, norms, residual stream: fp32 torch.
bf16 pipeline against an fp32 one: logits norm-wise <= 3e-2; token-by-token decode reproduces the prompt pass.
The prompt is margin-selected: a top-2 router fed by a bf16 residual stream flips whenever the 2nd and 3rd logits are closer than
the stream's own rounding (the first draft's prompt had an exact bf16 tie and seven gaps below 0.02, and one flipped token moved
the norm-wise error to 7e-2), so the prompt seed is one (found by scanning seeds on the CPU restatement) whose smallest 2nd-vs-3rd log-probability gap over all (layer,
token) decisions is > 0.15 — the test re-derives the gap and asserts it, so the selection is visible, not silent."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import bf16_to_f32, write_gguf  # noqa: E402
from oracle.gguf_ref import GGML_TYPE_Q4_K, GGML_TYPE_Q6_K, QUANT, GgufOracle  # noqa: E402
from oracle.linear_ref import dequant_w4, quantize_weights_ref  # noqa: E402
from oracle.oracle import f32_to_bf16  # noqa: E402

V, H, I, L, NH, NKV, E, K = 64, 256, 512, 2, 4, 2, 8, 2
HD = H // NH
EPS, THETA = 1e-5, 10000.0


def make_weights(seed=3):
    rng = np.random.default_rng(seed)
    r = lambda *s, sc=1.0: (rng.standard_normal(s) * sc).astype(np.float32)  # noqa: E731
    w = {"token_embd": r(V, H), "output": r(V, H, sc=H ** -0.5), "output_norm": 1 + r(H, sc=0.1)}
    for l in range(L):
        w[f"blk.{l}.attn_norm"] = 1 + r(H, sc=0.1)
        w[f"blk.{l}.ffn_norm"] = 1 + r(H, sc=0.1)
        w[f"blk.{l}.attn_q"] = r(NH * HD, H, sc=H ** -0.5)
        w[f"blk.{l}.attn_k"] = r(NKV * HD, H, sc=H ** -0.5)
        w[f"blk.{l}.attn_v"] = r(NKV * HD, H, sc=H ** -0.5)
        w[f"blk.{l}.attn_output"] = r(H, NH * HD, sc=H ** -0.5)
        w[f"blk.{l}.ffn_gate_inp"] = r(E, H, sc=2 * H ** -0.5)
        w[f"blk.{l}.ffn_gate_exps"] = QUANT[GGML_TYPE_Q4_K](r(E, I, H, sc=H ** -0.5))      # raw blocks [E, I, H/256*144]
        w[f"blk.{l}.ffn_up_exps"] = QUANT[GGML_TYPE_Q4_K](r(E, I, H, sc=H ** -0.5))
        w[f"blk.{l}.ffn_down_exps"] = QUANT[GGML_TYPE_Q6_K](r(E, H, I, sc=I ** -0.5))
    return w


def write_file(path, w):
    t = {}
    for name, a in w.items():
        if name.endswith("_exps"):
            ty = GGML_TYPE_Q6_K if "down" in name else GGML_TYPE_Q4_K
            n, k = (H, I) if "down" in name else (I, H)
            t[name + ".weight"] = (ty, [k, n, E], a.tobytes())
        else:
            t[name + ".weight"] = (0, list(a.shape[::-1]), a.tobytes())
    write_gguf(path, t, {"general.architecture": "llama", "llama.expert_count": E, "llama.attention.head_count": NH,
                         "llama.attention.head_count_kv": NKV})


def hf_rows(a, n_head):
    """Row order the reference's loader gives attn_q / attn_k of a llama-architecture file (custom_loader.py:507-517)."""
    return a.reshape(n_head, a.shape[0] // n_head // 2, 2, a.shape[1]).swapaxes(1, 2).reshape(a.shape)


def q4(wf32):
    """The weights a KLinearMarlin holds: bf16 source, Marlin W4-g64 quantiser, (q - 8) * s."""
    wb = torch.from_numpy(wf32).to(torch.bfloat16)
    q, s = quantize_weights_ref(wb.T.contiguous(), 64)
    return dequant_w4(q, s, 64, False).T.double()            # [N, K]


PROMPT_SEED, PROMPT_LEN, MIN_GAP = 92, 12, 0.15


def reference_logits(w, ids, gaps=None):
    o = GgufOracle()
    bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    T = len(ids)
    h = bf(torch.from_numpy(w["token_embd"]))[ids]                                              # embedding rows as loaded (bf16)
    pos = torch.arange(T).float()
    inv = 1.0 / (THETA ** (torch.arange(0, HD, 2).float() / HD))
    fr = torch.outer(pos, inv)
    cos, sin = torch.cat([fr, fr], -1).cos()[None], torch.cat([fr, fr], -1).sin()[None]          # [1, T, HD]
    rot = lambda x: torch.cat([-x[..., HD // 2:], x[..., :HD // 2]], -1)  # noqa: E731
    norm = lambda x, g: x * torch.rsqrt((x * x).mean(-1, keepdim=True) + EPS) * bf(torch.from_numpy(g))  # noqa: E731
    lin = lambda x, name: (x.double() @ q4(w[name]).T).float()  # noqa: E731
    w = dict(w)
    for l in range(L):
        w[f"blk.{l}.attn_q"], w[f"blk.{l}.attn_k"] = hf_rows(w[f"blk.{l}.attn_q"], NH), hf_rows(w[f"blk.{l}.attn_k"], NKV)
    for l in range(L):
        x = norm(h, w[f"blk.{l}.attn_norm"])
        q = lin(x, f"blk.{l}.attn_q").view(T, NH, HD).transpose(0, 1)
        k = lin(x, f"blk.{l}.attn_k").view(T, NKV, HD).transpose(0, 1)
        v = lin(x, f"blk.{l}.attn_v").view(T, NKV, HD).transpose(0, 1)
        q, k = q * cos + rot(q) * sin, k * cos + rot(k) * sin
        k, v = k.repeat_interleave(NH // NKV, 0), v.repeat_interleave(NH // NKV, 0)
        sc = (q @ k.transpose(1, 2)) / HD ** 0.5
        sc = sc.masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool), 1)[None], float("-inf"))
        a = (sc.softmax(-1) @ v).transpose(0, 1).reshape(T, NH * HD)
        h = h + lin(a, f"blk.{l}.attn_output")
        x = norm(h, w[f"blk.{l}.ffn_norm"])
        logits = bf(lin(x, f"blk.{l}.ffn_gate_inp"))                                            # the linear returns bf16
        p, sel = torch.topk(logits.softmax(-1), K, dim=-1)
        if gaps is not None:
            top = torch.topk(logits.log_softmax(-1), K + 1, dim=-1).values
            gaps.append(float((top[:, K - 1] - top[:, K]).min()))
        p = bf(p / p.sum(-1, keepdim=True))
        y = o.moe_forward(w[f"blk.{l}.ffn_gate_exps"], w[f"blk.{l}.ffn_up_exps"], w[f"blk.{l}.ffn_down_exps"],
                          (GGML_TYPE_Q4_K, GGML_TYPE_Q4_K, GGML_TYPE_Q6_K), E, H, I, sel.numpy().astype(np.int64),
                          p.numpy().astype(np.float32), f32_to_bf16(bf(x).numpy()))
        h = h + torch.from_numpy(bf16_to_f32(y))
    return lin(norm(h, w["output_norm"]), "output")


@pytest.fixture(scope="module")
def mixtral(tmp_path_factory):
    from ktransformers_amd.models.modeling_mixtral import MixtralForCausalLM, make_mixtral_config
    from ktransformers_amd.optimize.optimize import optimize_and_load
    from ktransformers_amd.util.gguf_loader import GGUFLoader
    d = tmp_path_factory.mktemp("mixtral")
    w = make_weights()
    write_file(str(d / "toy.gguf"), w)
    cfg = make_mixtral_config(vocab_size=V, hidden_size=H, intermediate_size=I, num_hidden_layers=L, num_attention_heads=NH,
                              num_key_value_heads=NKV, num_local_experts=E, num_experts_per_tok=K, max_position_embeddings=512,
                              rope_theta=THETA, rms_norm_eps=EPS)
    rules = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ktransformers_amd", "optimize",
                         "optimize_rules", "Mixtral-8x7B.yaml")
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device("meta"):
            model = MixtralForCausalLM(cfg)
        optimize_and_load(model, rules, GGUFLoader(str(d)), cfg, default_device="cuda:0")
    finally:
        torch.set_default_dtype(torch.float32)
    return model, cfg, w


def test_prompt_and_decode_match_the_restated_model(mixtral):
    from ktransformers_amd.models.modeling_mixtral import MixtralKVCache
    from ktransformers_amd.util.generate import set_inference_mode
    from ktransformers_amd.util.utils import InferenceState
    model, cfg, w = mixtral
    ids = np.random.default_rng(PROMPT_SEED).integers(0, V, PROMPT_LEN)
    gaps = []
    ref = reference_logits(w, torch.from_numpy(ids), gaps)
    assert min(gaps) > MIN_GAP, gaps                        # every routing decision of the prompt is outside bf16 noise
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(ids).to(dev)[None]
    pos = torch.arange(len(ids), device=dev)[None]
    ex = model.model.layers[0].block_sparse_moe.experts.generate_experts
    assert ex.loaded_method == "GGUF"                       # the raw blocks went to the native k-quant kernels
    set_inference_mode(model, InferenceState.PREFILL)
    cache = MixtralKVCache(cfg, 64, dev)
    with torch.no_grad():
        logits = model(x, pos, cache, pos[0])[0].float().cpu()
    rel = float((logits - ref).norm() / ref.norm())
    per_token = ((logits - ref).norm(dim=-1) / ref.norm(dim=-1)).tolist()
    assert rel < 3e-2, (rel, per_token)
    set_inference_mode(model, InferenceState.GENERATE)
    cache = MixtralKVCache(cfg, 64, dev)
    outs = []
    with torch.no_grad():
        for t in range(len(ids)):
            p = torch.tensor([[t]], device=dev)
            outs.append(model(x[:, t:t + 1], p, cache, p[0])[0, 0].float().cpu())
    dec = torch.stack(outs)
    assert float((dec - ref).norm() / ref.norm()) < 3e-2
    assert float((dec - logits).norm() / logits.norm()) < 2e-2
