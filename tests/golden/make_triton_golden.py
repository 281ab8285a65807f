"""Generate tests/golden/triton_golden.npz by RUNNING the reference's own Triton kernels on the CPU (TRITON_INTERPRET=1: Triton's
numpy interpreter executes the @triton.jit bodies unchanged — no GPU, nothing of the reference travels):

  act_quant_kernel, fp8_gemm_kernel   /root/reference/archive/ktransformers/ktransformers_ext/triton/fp8gemm.py:10-55,117-193
  decode_attention_fwd_grouped        /root/reference/archive/ktransformers/operators/triton_attention.py:358-385
                                      (the MLA decode KDeepseekV2Attention.forward_linux_triton calls, attention.py:285-290)

These pin oracle/linear_ref.py::act_quant_ref / linear_fp8_ref and oracle/mla_ref.py, which were restatements with no
reference-produced vector behind them ("parity unpinned" for those two legs).  What the interpreter does NOT do like
the hardware, measured here (scripts below print it) and worked around:
  * its fp32 -> fp8e4nv cast is wrong where the mantissa rounds up into the next binade (127.27 -> 64 instead of 128) and rounds
    exact ties upward (168 -> 176, not the even 160); its fp32 -> bf16 cast truncates (the GPU rounds to nearest even).  So: act_quant's SCALES are taken from the reference kernel, its
    CODES are stored too but only compared where the interpreter is right; fp8_gemm is fed the codes of torch's own RNE cast
    and writes an fp32 C (no bf16 cast of the accumulator); the MLA kernels get fp32 / fp16 operands (bf16-representable values)
    and write fp32 outputs — tl.dot on bf16 operands is garbage in the interpreter;
  * fp8_gemm_kernel is wrapped in triton.autotune, which needs a GPU to time configurations: the jitted body (`.fn`) is launched
    with the first entry of the reference's own fp8_gemm_configs (BLOCK_SIZE_M 16, N 32, K 128) and the grid fp8_gemm() computes.

Run in the build container:   python tests/golden/make_triton_golden.py
"""
import importlib.util
import os
import sys
import types

os.environ["TRITON_INTERPRET"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402
import triton  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference/archive/ktransformers"


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


out = {}
torch.set_default_dtype(torch.bfloat16)          # local_chat.py:105 — fp8_gemm / weight_dequant allocate in the default dtype
fp8 = load("ref_fp8gemm", f"{REF}/ktransformers_ext/triton/fp8gemm.py")
g = torch.Generator().manual_seed(20260930)
for tag, (M, K, N) in {"a": (5, 512, 384), "b": (3, 1536, 200), "c": (17, 256, 130)}.items():
    x = (torch.randn(M, K, generator=g) * 0.7).to(torch.bfloat16)
    x[0, :128] = x[0, :128] * 40                 # a block with a large scale
    w = (torch.randn(N, K, generator=g) * 0.05).float()
    Np = (N + 127) // 128 * 128
    wf = torch.zeros(Np, K, dtype=torch.float32)
    wf[:N] = w
    blk = wf.view(Np // 128, 128, K // 128, 128)
    sc = (blk.abs().amax(dim=(1, 3)).clamp_min(1e-12) / 448.0).float().contiguous()      # DeepSeek block-fp8 checkpoint format
    wq = (blk / sc[:, None, :, None]).reshape(Np, K)[:N].to(torch.float8_e4m3fn).contiguous()
    y_ref, s_ref = fp8.act_quant(x)              # the reference's launcher + kernel
    y_rne = (x.float().view(M, K // 128, 128) / s_ref[..., None]).reshape(M, K).to(torch.float8_e4m3fn).contiguous()   # same division, torch's RNE cast
    c = torch.empty(M, N, dtype=torch.float32)
    grid = (triton.cdiv(M, 16), triton.cdiv(N, 32))
    fp8.fp8_gemm_kernel.fn[grid](y_rne, wq, c, s_ref.contiguous(), sc, M, N, K, BLOCK_SIZE_M=16, BLOCK_SIZE_N=32, BLOCK_SIZE_K=128)
    bad = (y_ref.view(torch.uint8) != y_rne.view(torch.uint8))
    print(f"fp8 {tag}: act_quant codes differ from RNE at {int(bad.sum())} of {bad.numel()} positions (binade carries and exact ties)")
    out[f"fp8_{tag}_x"] = x.view(torch.uint16).numpy()
    out[f"fp8_{tag}_wq"] = wq.view(torch.uint8).numpy()
    out[f"fp8_{tag}_wscale"] = sc.numpy()
    out[f"fp8_{tag}_act_scale"] = s_ref.numpy()
    out[f"fp8_{tag}_act_codes_interp"] = y_ref.view(torch.uint8).numpy()
    out[f"fp8_{tag}_c_f32"] = c.numpy()

# ---- MLA decode (grouped, split-KV + merge): triton_attention.py imports ktransformers.util.vendors for one AMD block-size switch
torch.set_default_dtype(torch.float32)
vend = types.ModuleType("ktransformers.util.vendors")


class _GPUVendor:
    AMD, NVIDIA = "amd", "nvidia"


vend.GPUVendor = _GPUVendor
vend.device_manager = types.SimpleNamespace(gpu_vendor=_GPUVendor.AMD)       # the BLOCK = 16 branch an MI355X takes
vend.get_device = vend.to_device = lambda *a, **k: None                       # imported by the module, not used by the decode path
for name in ("ktransformers", "ktransformers.util"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["ktransformers.util.vendors"] = vend
ta = load("ref_triton_attention", f"{REF}/operators/triton_attention.py")
# (tl.dot on bf16 operands returns garbage in the interpreter — numpy has no bf16; the kernels are dtype-generic, so the vectors
# are taken in fp32, where p.to(v.dtype) is the identity, and in fp16, numpy's own round-to-nearest-even type)
for tag, (H, n_tok, page, splits, dt) in {"a": (16, 150, 16, 4, torch.float32), "b": (32, 77, 64, 4, torch.float16),
                                          "c": (16, 513, 64, 4, torch.float32)}.items():
    n_pages = (n_tok + page - 1) // page + 2
    q = (torch.randn(1, H, 576, generator=g) * 0.3).to(torch.bfloat16).to(dt)          # bf16-representable values
    kv = (torch.randn(n_pages, page, 1, 576, generator=g) * 0.5).to(torch.bfloat16).to(dt)
    table = torch.randperm(n_pages, generator=g).to(torch.int32).view(1, n_pages)
    seq = torch.tensor([n_tok], dtype=torch.int32)
    o = torch.zeros(1, H, 512, dtype=torch.float32)
    logits = torch.empty(1, H, splits, 513, dtype=torch.float32)
    sm = 0.1147
    ta.decode_attention_fwd_grouped(q, kv, kv[..., :512], o, table, seq, logits, splits, sm, page)
    out[f"mla_{tag}_q"] = q.to(torch.bfloat16).view(torch.uint16).numpy()
    out[f"mla_{tag}_kv"] = kv.to(torch.bfloat16).view(torch.uint16).numpy()
    out[f"mla_{tag}_fp16"] = np.array([dt == torch.float16])
    out[f"mla_{tag}_table"] = table.numpy()
    out[f"mla_{tag}_meta"] = np.array([H, n_tok, page, splits], dtype=np.int64)
    out[f"mla_{tag}_sm"] = np.array([sm], dtype=np.float64)
    out[f"mla_{tag}_o_f32"] = o.numpy()
    print(f"mla {tag}: |o| max {float(o.abs().max()):.4f}")
np.savez_compressed(os.path.join(HERE, "triton_golden.npz"), **out)
print("wrote", os.path.join(HERE, "triton_golden.npz"), {k: v.shape for k, v in out.items() if k.endswith(("c_f32", "o_f32"))})
