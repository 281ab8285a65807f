"""TEST INFRASTRUCTURE ONLY — CPU restatement (torch, fp32/fp64) of the reference's quantised linears.

  quantize_weights_ref : archive/ktransformers/ktransformers_ext/operators/custom_marlin/quantize/utils/quant_utils.py:36-98
                         (the quantiser behind marlin_quantize, marlin_utils.py:79-114, called by KLinearMarlin.load,
                         archive/ktransformers/operators/linear.py:664-666).  PINNED: tests/golden/linear_w4_golden.npz was
                         produced by importing the reference's own quant_utils.py (tests/golden/make_linear_golden.py).
  linear_w4_ref        : what gptq_marlin_gemm computes (linear.py:690-702): x @ dequant(q, s), fp32 accumulation.  The CUDA
                         kernel itself is un-vendored (KTransformersOps) => GEMM parity is fp-tolerance against this math;
                         `round_weights=True` applies Marlin's in-register dequant rounding bf16((q-8)*s).
  act_quant_ref / linear_fp8_ref : ktransformers_ext/triton/fp8gemm.py:10-55, 117-193:
                         s = amax/448 per 128 inputs, y = (x/s)->e4m3; acc += dot(a_blk, b_blk) * a_s * b_s per 128-K block.
                         PINNED (round 6): tests/golden/triton_golden.npz holds the outputs of the reference's own act_quant_kernel
                         and fp8_gemm_kernel, run on the CPU by Triton's interpreter (tests/golden/make_triton_golden.py);
                         tests/test_triton_pin_cpu.py: scales bit-equal, codes equal up to the interpreter's two documented cast
                         defects, fp32 accumulator within 2 ulp.
  linear_bf16_ref      : KLinearTorch.forward, linear.py:174-183.
"""
import torch


def quantize_weights_ref(w: torch.Tensor, group_size: int, num_bits: int = 4):
    """w: [K, N] floating (the reference passes bf16).  Returns (q_w int32 [K,N] in 0..15, s [K/g, N] in w.dtype)."""
    size_k, size_n = w.shape
    if group_size == -1:
        group_size = size_k
    max_q_val = 2 ** num_bits - 1
    half_q_val = (max_q_val + 1) // 2
    w = w.clone()
    if group_size < size_k:                                   # quant_utils.py:55-58
        w = w.view((-1, group_size, size_n)).permute(1, 0, 2).reshape((group_size, -1))
    s = torch.max(torch.abs(w), 0, keepdim=True)[0]            # :61
    s *= 2 / max_q_val                                         # :62 (in the tensor's dtype)
    q_w = torch.round(w / s).int()                             # :65
    q_w += half_q_val
    q_w = torch.clamp(q_w, 0, max_q_val)
    if group_size < size_k:                                    # :70-78
        q_w = q_w.reshape((group_size, -1, size_n)).permute(1, 0, 2).reshape((size_k, size_n)).contiguous()
    s = s.reshape((-1, size_n)).contiguous()
    return q_w, s


def dequant_w4(q: torch.Tensor, s: torch.Tensor, group_size: int, round_weights: bool) -> torch.Tensor:
    """[K,N] fp32 weights: (q-8)*s, optionally rounded to bf16 like Marlin's in-register dequant."""
    sf = s.float().repeat_interleave(group_size, dim=0)
    w = (q.float() - 8.0) * sf
    return w.to(torch.bfloat16).float() if round_weights else w


def _finish(y: torch.Tensor, bias):
    y = y.to(torch.bfloat16)
    if bias is not None:
        y = (y.float() + bias.float()).to(torch.bfloat16)      # x = x + self.bias in bf16 (linear.py:709)
    return y


def linear_w4_ref(x: torch.Tensor, q: torch.Tensor, s: torch.Tensor, group_size: int, bias=None, round_weights=False):
    w = dequant_w4(q, s, group_size, round_weights).double()
    return _finish((x.double() @ w).float(), bias)


def linear_bf16_ref(x: torch.Tensor, weight: torch.Tensor, bias=None):
    return _finish((x.double() @ weight.double().T).float(), bias)


def act_quant_ref(x: torch.Tensor, block_size: int = 128):
    """fp8gemm.py:10-55.  Returns (e4m3 tensor, fp32 scales [..., K/128]).  Zero blocks give NaN in the reference; 0 here."""
    xf = x.float().reshape(*x.shape[:-1], -1, block_size)
    s = xf.abs().amax(dim=-1) / 448.0
    y = torch.where(s[..., None] > 0, xf / s[..., None], torch.zeros_like(xf))
    return y.reshape(x.shape).to(torch.float8_e4m3fn), s


def linear_fp8_ref(x: torch.Tensor, w_fp8: torch.Tensor, scale_inv: torch.Tensor, bias=None, block: int = 128, _raw: bool = False):
    """x bf16 [T,K]; w_fp8 float8_e4m3fn [N,K]; scale_inv fp32 [ceil(N/128), K/128].  fp8gemm.py:117-159."""
    xq, a_s = act_quant_ref(x, block)
    T, K = x.shape
    N = w_fp8.shape[0]
    a = xq.float().reshape(T, K // block, block)
    b = w_fp8.float().reshape(N, K // block, block)
    dots = torch.einsum("tkb,nkb->tnk", a.double(), b.double()).float()         # exact products, fp32-rounded block dots
    b_s = scale_inv.repeat_interleave(block, dim=0)[:N]                           # [N, K/128]
    acc = torch.zeros(T, N, dtype=torch.float32)
    for kb in range(K // block):
        acc = acc + dots[:, :, kb] * a_s[:, kb, None] * b_s[None, :, kb]
    if _raw:                                                                      # the fp32 accumulator (tests/golden pin)
        return acc
    return _finish(acc, bias)
