"""GPU: the small fused ops (include/ktx_ops.h) against their torch definitions in the reference.
RMSNorm/RoPE/SiLU are bf16-in/bf16-out elementwise-plus-reduction ops: results must equal the torch expression evaluated
with the same roundings, up to 1 bf16 ulp where the fp32 reduction order or the libm cos/sin differ in the last bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.attention_ref import apply_rope, rmsnorm_ref, rope_tables  # noqa: E402
from attn_helpers import make_cfg  # noqa: E402


def ulp_close(y, ref, frac=0.002):
    y, ref = y.float().cpu(), ref.float()
    bad = (y - ref).abs() > 2.0 ** -7 * ref.abs() + 1e-6
    assert not bad.any(), f"{int(bad.sum())} elements differ by more than one bf16 ulp"
    assert (y != ref).float().mean() <= frac, float((y != ref).float().mean())


@pytest.mark.parametrize("dim", [512, 1536, 2048, 7168])
@pytest.mark.parametrize("T", [1, 5])
def test_rmsnorm_native_and_fused(dim, T):
    from ktransformers_amd import _native as n
    torch.manual_seed(dim + T)
    x = torch.randn(T, dim).to(torch.bfloat16)
    w = (1 + 0.1 * torch.randn(dim)).to(torch.bfloat16)
    y = n.rmsnorm(x.cuda(), w.cuda(), 1e-6, native_rounding=True)
    ulp_close(y, rmsnorm_ref(x, w, 1e-6))
    y1 = n.rmsnorm(x.cuda(), w.cuda(), 1e-6, native_rounding=False)
    xf = x.float()
    ref1 = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float()).to(torch.bfloat16)
    ulp_close(y1, ref1)
    res = torch.randn(T, dim).to(torch.bfloat16)
    xg, rg = x.cuda().clone(), res.cuda().clone()
    bsz = torch.tensor([T], dtype=torch.int32, device="cuda")
    n.fused_add_rmsnorm(xg, rg, w.cuda(), 1e-6, bsz)
    s = x.float() + res.float()
    assert torch.equal(rg.cpu(), s.to(torch.bfloat16))
    ulp_close(xg, (s * torch.rsqrt(s.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float()).to(torch.bfloat16))


def test_rmsnorm_respects_bsz_tensor():
    from ktransformers_amd import _native as n
    x = torch.randn(4, 512).to(torch.bfloat16).cuda()
    w = torch.ones(512, dtype=torch.bfloat16, device="cuda")
    out = torch.full_like(x, 3.0)
    n.rmsnorm(x, w, 1e-6, False, torch.tensor([2], dtype=torch.int32, device="cuda"), out)
    assert torch.all(out[2:] == 3.0) and not torch.any(out[:2] == 3.0)


@pytest.mark.parametrize("T,H", [(1, 16), (7, 3), (33, 128)])
def test_mla_prep_matches_torch(T, H):
    from ktransformers_amd import _native as n
    torch.manual_seed(T * H)
    cfg = make_cfg(256, H, None)
    q = torch.randn(T, H * 192).to(torch.bfloat16)
    kv = torch.randn(T, 576).to(torch.bfloat16)
    nw = (1 + 0.1 * torch.randn(512)).to(torch.bfloat16)
    pos = torch.randint(0, 100000, (T,))
    cos, sin, inv_freq, mscale = rope_tables(cfg, pos, torch.bfloat16)
    q_pe, ckv, kpe = n.mla_prep(q.cuda(), kv.cuda(), nw.cuda(), 1e-6, pos.cuda(), inv_freq.cuda(), mscale, H, 128, 64, 512)
    ulp_close(ckv, rmsnorm_ref(kv[:, :512], nw, 1e-6))
    # cos/sin of large arguments: the device libm and torch's CPU libm may differ in the last fp32 bit -> rare bf16 flips
    ulp_close(kpe, apply_rope(kv[:, 512:].reshape(T, 1, 64), cos, sin).reshape(T, 64), frac=0.02)
    ulp_close(q_pe, apply_rope(q.view(T, H, 192)[:, :, 128:].contiguous(), cos, sin), frac=0.02)


def test_silu_mul():
    from ktransformers_amd import _native as n
    torch.manual_seed(3)
    gu = torch.randn(6, 2 * 1408).to(torch.bfloat16)
    y = n.silu_mul(gu.cuda())
    ref = torch.nn.functional.silu(gu[:, :1408]) * gu[:, 1408:]
    ulp_close(y, ref)


@pytest.mark.parametrize("rows,n", [(1, 129280), (1, 102400), (3, 4096), (1, 100), (2, 24)])
def test_argmax_bf16_first_maximum(rows, n):
    """ktx_argmax_bf16 (greedy sampling in one launch) == torch.argmax on the fp32 copy of the logits, ties included (the first
    maximum wins), replay-safe (the arrival counters return to zero)."""
    torch.manual_seed(rows * 1000 + n)
    x = torch.randn((rows, n), device="cuda").to(torch.bfloat16)
    for rep in range(3):
        got = n_native().argmax_bf16(x)
        assert torch.equal(got, x.float().argmax(dim=-1)), rep
    t = torch.zeros((rows, n), device="cuda", dtype=torch.bfloat16)          # many exact ties: the lowest index among the maxima
    for r in range(rows):
        t[r, torch.tensor([n - 1, n // 2, min(n - 1, 7 + r)], device="cuda")] = 3.0
    assert torch.equal(n_native().argmax_bf16(t), t.float().argmax(dim=-1))
    assert n_native().argmax_bf16(x[0]).shape == ()


def n_native():
    from ktransformers_amd import _native
    return _native
