"""CPU: oracle/linear_ref.py (act_quant_ref, linear_fp8_ref) and oracle/mla_ref.py against vectors PRODUCED BY THE REFERENCE'S OWN
TRITON KERNELS (fp8gemm.py:10-55,117-193; triton_attention.py:358-385), executed on the CPU by Triton's interpreter in the build
container — tests/golden/make_triton_golden.py, which also says what the interpreter does differently from the GPU and how the
vectors avoid it.  Before round 6 these two oracle legs were restatements with nothing of the reference behind them."""
import os

import numpy as np
import torch

from oracle.linear_ref import act_quant_ref, linear_fp8_ref
from oracle.mla_ref import mla_paged_ref

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "triton_golden.npz"))


def _bf16(a):
    return torch.from_numpy(a.astype(np.int16)).view(torch.bfloat16)


def test_act_quant_scales_and_codes_match_the_reference_kernel():
    for tag in "abc":
        x = _bf16(G[f"fp8_{tag}_x"])
        codes, scales = act_quant_ref(x)
        assert np.array_equal(scales.numpy(), G[f"fp8_{tag}_act_scale"]), f"{tag}: block scales differ from act_quant_kernel's"
        ours, ref = codes.float().numpy(), torch.from_numpy(G[f"fp8_{tag}_act_codes_interp"]).view(torch.float8_e4m3fn).float().numpy()
        bad = ours != ref
        assert bad.mean() < 0.04, f"{tag}: {bad.mean():.3f} of the codes differ"
        # every difference is one of the interpreter's two cast defects (make_triton_golden.py): the mantissa carry into the next
        # binade lost (its code is half of the round-to-nearest one), or an exact tie rounded up instead of to even
        v = (x.float().view(x.shape[0], -1, 128) / scales[..., None]).reshape(x.shape).numpy()
        o, r, vv = ours[bad], ref[bad], v[bad]
        binade = o == 2 * r
        tie = np.abs(vv - o) == np.abs(vv - r)
        assert bool((binade | tie).all()), f"{tag}: {int((~(binade | tie)).sum())} code differences are neither"


def test_fp8_gemm_accumulator_matches_the_reference_kernel():
    for tag in "abc":
        x = _bf16(G[f"fp8_{tag}_x"])
        wq = torch.from_numpy(G[f"fp8_{tag}_wq"]).view(torch.float8_e4m3fn)
        sc = torch.from_numpy(G[f"fp8_{tag}_wscale"])
        acc = linear_fp8_ref(x, wq, sc, _raw=True).numpy()
        ref = G[f"fp8_{tag}_c_f32"]
        # the same block dots x a_s x b_s summed over the K blocks in fp32; the reference's in-block sum runs in fp32 (tl.dot), the
        # oracle's exactly: a couple of fp32 ulps
        assert np.abs(acc - ref).max() <= 2.0 ** -21 * np.abs(ref).max(), (tag, float(np.abs(acc - ref).max()))
        assert (acc == ref).mean() > 0.9


def test_mla_decode_matches_the_reference_split_kv_kernels():
    for tag in "abc":
        H, n_tok, page, splits = (int(v) for v in G[f"mla_{tag}_meta"])
        fp16 = bool(G[f"mla_{tag}_fp16"][0])
        q = _bf16(G[f"mla_{tag}_q"])                       # [1, H, 576]
        kv = _bf16(G[f"mla_{tag}_kv"])                     # [pages, page, 1, 576]
        table = torch.from_numpy(G[f"mla_{tag}_table"]).reshape(-1)
        used = (n_tok + page - 1) // page
        out, _ = mla_paged_ref(q[:, :, :512], q[:, :, 512:], kv[:, :, 0, :], torch.tensor([0, 1]), torch.tensor([0, used]), table[:used],
                               torch.tensor([n_tok]), float(G[f"mla_{tag}_sm"][0]))
        ref = torch.from_numpy(G[f"mla_{tag}_o_f32"])
        err = float((out - ref).abs().max())
        # fp32 operands: only the summation order differs (4 KV splits merged by their log-sum-exps vs one softmax); fp16 operands:
        # the kernel rounds P to fp16 before the PV product — the reference's own bound for this operator is 5e-3
        assert err <= (2e-3 if fp16 else 2e-6) * max(1.0, float(ref.abs().max())), (tag, err)
