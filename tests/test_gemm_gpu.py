"""GPU: the library's own prompt-sized BF16 GEMM (csrc/ktx_gemm.hip, include/ktx_gemm.h) against fp64 math on the same bf16
operands.  Bound per element: one bf16 rounding of the result (2^-8 relative, torch's round-to-nearest-even) plus fp32
accumulation noise 2^-20 * sum_k |a||b| — the contract of the tensor-core GEMMs it replaces (gptq_marlin_gemm on
bf16((q-8)*s), torch.matmul of the kv_b expansion, F.linear of the router logits).  Shapes cover tile edges (M, N not
multiples of 128, N = 8), strided rows, shared operands of the batched form, both LDS-stage variants and the fp32 output."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def ref64(a, b, bias=None):
    y = a.double() @ b.double().transpose(-1, -2)
    s = a.double().abs() @ b.double().abs().transpose(-1, -2)
    if bias is not None:
        y = y + bias.double()
        s = s + bias.double().abs()
    return y, s


def check(y, a, b, bias=None, f32=False):
    ref, s = ref64(a, b, bias)
    err = (y.double() - ref).abs()
    tol = (0 if f32 else 2.0 ** -8) * ref.abs() + 2.0 ** -20 * s + 1e-30
    bad = err > tol
    assert not bool(bad.any()), (int(bad.sum()), float((err / tol).max()))


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("M,N,K", [(1, 8, 64), (130, 2112, 1536), (300, 256, 7168), (128, 128, 64), (257, 1032, 512), (2048, 1152, 2048)])
def test_gemm_matches_fp64_math(M, N, K, variant):
    from ktransformers_amd._native import gemm_bf16_nt
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((M, K), generator=g).to(torch.bfloat16).to(DEV)
    b = (torch.randn((N, K), generator=g) * K ** -0.5).to(torch.bfloat16).to(DEV)
    y = gemm_bf16_nt(a, b, variant=variant)
    assert y.dtype == torch.bfloat16 and tuple(y.shape) == (M, N)
    check(y, a, b)


def test_variants_and_repeated_calls_are_bit_identical():
    from ktransformers_amd._native import gemm_bf16_nt
    g = torch.Generator(device="cpu").manual_seed(5)
    a = torch.randn((515, 4096), generator=g).to(torch.bfloat16).to(DEV)
    b = (torch.randn((1160, 4096), generator=g) / 64).to(torch.bfloat16).to(DEV)
    y1 = gemm_bf16_nt(a, b, variant=1)
    for _ in range(3):
        assert torch.equal(gemm_bf16_nt(a, b, variant=1), y1)          # no race between the DMA stages and the fragment reads
        for v in (2, 3, 4, 5):
            assert torch.equal(gemm_bf16_nt(a, b, variant=v), y1)      # same k order, same sums in every tile configuration


@pytest.mark.parametrize("M,N,K", [(4096, 4096, 4096), (2048, 7168, 16384), (8192, 2112, 7168), (1000, 520, 192)])
def test_ping_pong_tile_race_screen(M, N, K):
    """Round 6, variant 5 (256 x 256 x 64, half-tile ring with counted waits, the two wavefronts of a SIMD half a phase apart): at
    prompt-chunk sizes, repeated, with bias and with fp32 output, every call must give the bits of the two-barrier 128 x 128 kernel
    (the same k order) — a fragment read that overtakes its DMA, or a DMA that overwrites a slot still being read, shows up here as
    a wrong tile that comes and goes between repetitions."""
    from ktransformers_amd._native import gemm_bf16_nt
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = torch.randn((M, K), generator=g).to(torch.bfloat16).to(DEV)
    b = (torch.randn((N, K), generator=g) * K ** -0.5).to(torch.bfloat16).to(DEV)
    bias = torch.randn(N, generator=g).to(torch.bfloat16).to(DEV)
    y_ref, yb_ref, yf_ref = gemm_bf16_nt(a, b, variant=2), gemm_bf16_nt(a, b, bias=bias, variant=2), gemm_bf16_nt(a, b, out_f32=True, variant=2)
    for rep in range(4):
        assert torch.equal(gemm_bf16_nt(a, b, variant=5), y_ref), rep
        assert torch.equal(gemm_bf16_nt(a, b, bias=bias, variant=5), yb_ref), rep
        assert torch.equal(gemm_bf16_nt(a, b, out_f32=True, variant=5), yf_ref), rep
    assert torch.equal(gemm_bf16_nt(a, b), y_ref)                        # whatever the automatic choice is


def test_bias_strided_rows_and_fp32_output():
    from ktransformers_amd._native import gemm_bf16_nt
    g = torch.Generator(device="cpu").manual_seed(9)
    big = torch.randn((200, 1024 + 64), generator=g).to(torch.bfloat16).to(DEV)
    a = big[:, 64:64 + 1024]                                             # row stride 1088 elements, first row 128 B into the buffer
    b = (torch.randn((520, 1024), generator=g) / 32).to(torch.bfloat16).to(DEV)
    bias = torch.randn(520, generator=g).to(torch.bfloat16).to(DEV)
    y = gemm_bf16_nt(a, b, bias=bias)
    check(y, a, b, bias)
    yf = gemm_bf16_nt(a, b, out_f32=True)
    assert yf.dtype == torch.float32
    check(yf, a, b, f32=True)
    out = torch.zeros((200, 1040), dtype=torch.bfloat16, device=DEV)     # write into a wider buffer: columns 520.. stay untouched
    gemm_bf16_nt(a, b, out=out[:, :520])
    check(out[:, :520], a, b)
    assert not bool(out[:, 520:].any())


def test_batched_with_a_shared_operand_is_the_kv_b_expansion():
    """attention.py:77-194: K_nope[h] = latent @ W_UK[h]^T (A shared) and V^T[h] = W_UV[h] @ latent^T (B shared)."""
    from ktransformers_amd._native import gemm_bf16_nt
    g = torch.Generator(device="cpu").manual_seed(11)
    Hh, kv, lora, d = 6, 320, 512, 128
    lat = torch.randn((kv, lora), generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn((Hh, d, lora), generator=g) / 22).to(torch.bfloat16).to(DEV)
    k_nope = gemm_bf16_nt(lat, w)                                         # [H, kv, d]
    assert tuple(k_nope.shape) == (Hh, kv, d)
    check(k_nope, lat.unsqueeze(0).expand(Hh, -1, -1), w)
    v_t = gemm_bf16_nt(w, lat)                                            # [H, d, kv]
    assert tuple(v_t.shape) == (Hh, d, kv)
    check(v_t, w, lat.unsqueeze(0).expand(Hh, -1, -1))
    wt = w.transpose(0, 1)                                                # [d, H, lora] view: batch stride < row stride
    y = gemm_bf16_nt(lat, wt.transpose(0, 1))
    assert torch.equal(y, k_nope)


def test_three_bf16_planes_reproduce_an_fp32_weight_exactly_and_give_fp32_grade_logits():
    """modeling_deepseek_v3.py:434-437 F.linear(x.float(), weight.float()): hi + mid + lo == w bit for bit, and the three-plane
    GEMM's logits are as close to fp64 math as an fp32 GEMM's (2^-20 * sum |x||w|, no bf16 rounding anywhere)."""
    from ktransformers_amd._native import gemm_bf16_nt, split_f32_bf16x3
    g = torch.Generator(device="cpu").manual_seed(13)
    E, K, T = 256, 7168, 600
    w = (torch.randn((E, K), generator=g) * 0.02).to(DEV)
    w[0, :4] = torch.tensor([0.0, -0.0, 1e-30, -3.0e30])
    planes = split_f32_bf16x3(w)
    assert tuple(planes.shape) == (3 * E, K)
    back = planes[:E].float() + planes[E:2 * E].float() + planes[2 * E:].float()
    assert torch.equal(back, w)
    x = torch.randn((T, K), generator=g).to(torch.bfloat16).to(DEV)
    l3 = gemm_bf16_nt(x, planes, out_f32=True)
    logits = (l3[:, 2 * E:] + l3[:, E:2 * E]) + l3[:, :E]
    ref = x.double() @ w.double().T
    s = x.double().abs() @ w.double().abs().T
    assert bool(((logits.double() - ref).abs() <= 2.0 ** -20 * s + 1e-30).all())


def test_argument_checks_do_not_launch():
    from ktransformers_amd._native import KtxError, gemm_bf16_nt
    a = torch.zeros((4, 96), dtype=torch.bfloat16, device=DEV)
    b = torch.zeros((8, 96), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(KtxError, match="multiple of 64"):
        gemm_bf16_nt(a, b)
    a = torch.zeros((4, 128), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(KtxError, match="16 bytes"):
        gemm_bf16_nt(a, torch.zeros((12, 128), dtype=torch.bfloat16, device=DEV))
    with pytest.raises(KtxError):
        gemm_bf16_nt(a.float(), b)
