"""Injected linear operators — mirror of archive/ktransformers/operators/linear.py.

Same class names, constructor kwargs, `load / unload / forward(x, bsz_tensor)` lifecycle and `LINEAR_MAP` registry as the
reference, so its YAML rules (`class: ktransformers.operators.linear.KTransformersLinear`, kwargs `generate_op:
"KLinearMarlin"`, `prefill_op: "KLinearTorch"`) resolve unchanged.  The compute is the HIP library (include/ktx_linear.h):

  KLinearMarlin (linear.py:595-720)  -> W4 g=64 handle; weights quantised on the GPU with quantize_weights' arithmetic
  KLinearFP8    (linear.py:388-436)  -> block-fp8 handle (e4m3 weight + weight_scale_inv; e4m3 activations per 128)
  KLinearTorch  (linear.py:158-216)  -> dense bf16 handle
  VLinearMarlin / KLinearQ8 / KLinearCPUInfer / KLinearIPEXLLM are vendor back-ends of the same contract; their names map
  to the closest of the three above so a rule file written for another box still loads.

One deliberate difference: the reference keeps a *prefill* and a *generate* copy (bf16 for KLinearTorch prefill, Marlin
for decode) and swaps them on `set_inference_mode` because the Marlin kernel is slow for long prompts; here one
quantised copy serves every T (the same handle runs a GEMV kernel for T <= 4 and an MFMA-tiled GEMM above), so when
both ops name a quantised format only one handle is built.
"""
from __future__ import annotations

import torch
from torch import nn

from ktransformers_amd.operators.base_operator import BaseInjectedModule
from ktransformers_amd.util.utils import InferenceState


class KLinearBase(nn.Module):
    """linear.py:57-155 — shape discovery and weight fetch through the loader."""
    FMT = "BF16"

    def __init__(self, key: str, gguf_loader, config, orig_module: nn.Module = None, device: str = "cuda", **kwargs):
        nn.Module.__init__(self)
        self.key, self.gguf_loader, self.device, self.config = key, gguf_loader, device, config
        self.has_bias = False
        self.dtype = torch.get_default_dtype()
        if orig_module is not None:
            self.in_features, self.out_features = orig_module.in_features, orig_module.out_features
        else:
            shape = gguf_loader.load_tensor(key + ".weight").shape      # [out, in]
            self.out_features, self.in_features = int(shape[0]), int(shape[1])
        self.loaded = False
        self.max_len = int(kwargs.get("max_len", getattr(config, "max_position_embeddings", 4096) or 4096))
        self._h = None
        self.weight = None

    def load_weight(self, override_key: str | None = None, device: str | None = None):
        """linear.py:92-141 for the safetensors branch: weight (+ weight_scale_inv) (+ bias)."""
        key = override_key or self.key
        ld = self.gguf_loader
        if not ld.has_tensor(key + ".weight"):
            raise FileNotFoundError(f"Weight file not found for key {key}")
        w = ld.load_tensor(key + ".weight", device=device or "cpu")
        bias = ld.load_tensor(key + ".bias", device=device or "cpu") if ld.has_tensor(key + ".bias") else None
        if ld.has_tensor(key + ".weight_scale_inv"):
            return nn.Parameter(w, requires_grad=False), nn.Parameter(
                ld.load_tensor(key + ".weight_scale_inv", device=device or "cpu"), requires_grad=False)
        if bias is not None:
            return nn.Parameter(w, requires_grad=False), nn.Parameter(bias, requires_grad=False)
        return nn.Parameter(w, requires_grad=False)

    def _dev(self) -> torch.device:
        d = torch.device(self.device)
        return torch.device("cuda", torch.cuda.current_device()) if d.type == "cuda" and d.index is None else d

    def _make_handle(self, group_size: int = 0):
        from ktransformers_amd._native import LinearHandle
        self.device = str(self._dev())
        return LinearHandle(self.in_features, self.out_features, self.FMT, group_size, self.max_len, torch.device(self.device))

    def forward(self, x: torch.Tensor, bsz_tensor: torch.Tensor = None, **fusion) -> torch.Tensor:
        """`fusion`: norm=(weight, eps), add1=, add2= — see LinearHandle.forward (an extension of the reference signature;
        plain calls behave exactly like KLinear*.forward(x, bsz_tensor))."""
        if self._h is None:
            raise RuntimeError(f"{type(self).__name__}({self.key}): forward before load()")
        dtype, dev = x.dtype, x.device
        y = self._h.forward(x.to(device=self.device, dtype=torch.bfloat16), bsz_tensor, **fusion)
        return y.to(dtype=dtype, device=dev)

    def unload(self):
        if self._h is not None:
            self._h.close()
        self._h = None
        self.weight = None
        self.loaded = False


def _split(w):
    """(weight, second) from the load() argument forms of linear.py:185-205."""
    if isinstance(w, nn.Parameter) or isinstance(w, torch.Tensor):
        return w, None
    if isinstance(w, tuple):
        return w[0], w[1]
    raise ValueError("Invalid weight type")


class KLinearTorch(KLinearBase):
    """Dense bf16: x @ W^T (+ bias) — linear.py:158-216."""
    FMT = "BF16"

    def load(self, w=None, device: str | None = None):
        if self.loaded:
            return
        if device is not None:
            self.device = device
        if w is None:
            w = self.load_weight(device=self.device)
        weight, bias = _split(w)
        weight = weight.data.to(self.device, torch.bfloat16).view(self.out_features, self.in_features).contiguous()
        self.has_bias = bias is not None
        self._h = self._make_handle()
        self._h.load_bf16(weight, bias.data.to(self.device, torch.bfloat16) if bias is not None else None)
        self.weight = weight.T                                     # modeling code may read linear.weight (linear.py:196)
        self.loaded = True


def marlin_quantize(weight: torch.Tensor, num_bits: int, group_size: int):
    """[N, K] bf16 weights -> (q [N, K/g, g] float, signed: quantised value minus 2^(bits-1); s [N, K/g, 1] bf16) = quantize_weights on
    weight.T (custom_marlin/quantize/utils/quant_utils.py:36-98 — s = max|w| * 2/(2^bits - 1) per group and output, q = clamp(round(w / s)
    + 2^(bits-1), 0, 2^bits - 1), every step in the tensor's own bf16 arithmetic).  torch ops on the weight's device."""
    n, k = weight.shape
    g = k if group_size == -1 else group_size
    if k % g:
        raise ValueError(f"in_features {k} is not a multiple of group_size {g}")
    qmax = 2 ** num_bits - 1
    half = (qmax + 1) // 2
    w = weight.to(torch.bfloat16).view(n, k // g, g)
    sc = w.abs().amax(dim=2, keepdim=True)
    sc = sc * (2 / qmax)                                            # bf16 tensor * python float: fp32 product, bf16 result
    q = torch.round(w / sc)
    q = torch.where(sc == 0, torch.full_like(q, -half), q)          # int(NaN) + half clamps to 0 in the reference
    q = torch.clamp(q.float() + half, 0, qmax) - half
    return q, sc


def marlin_multiplicand(weight: torch.Tensor, num_bits: int, group_size: int) -> torch.Tensor:
    """[N, K] bf16 weights -> the [N, K] bf16 matrix gptq_marlin_gemm multiplies activations with: marlin_quantize, then Marlin's
    in-register de-quantisation bf16((q - 2^(bits-1)) * s) (kt-kernel/cuda/gptq_marlin/gptq_marlin.cu: dequant + scale)."""
    q, sc = marlin_quantize(weight, num_bits, group_size)
    return (q * sc.float()).to(torch.bfloat16).view(weight.shape)


class KLinearMarlin(KLinearBase):
    """W4A16 / W8A16 weight-only linear — linear.py:595-720.  `num_bits`, `group_size`, `act_order`, `is_k_full` as in the
    reference.
      * num_bits = 4 (every rule file of the reference): the native W4 format — packed nibbles + bf16 group scales, the W4
        decode GEMVs and prompt GEMM of csrc/ktx_linear.hip.
      * num_bits = 8: the native W8 format (round 4) — one byte per weight + bf16 group scales in HBM, Marlin's own multiplicand
        bf16((q - 128) * s) formed in registers in front of the bf16 MFMA: the same products and fp32 accumulation as
        gptq_marlin_gemm's 8-bit path, bit-identical to holding that matrix as a BF16 handle (which is what rounds 2-3 did, and what
        still happens for group sizes the W8 tiles do not carry: anything but 32, 64 or a multiple of 128).
      * act_order: the reference "simulates" it by a random permutation of the K rows of the STORED q_w with g_idx / sort_indices
        undoing it inside the kernel (quant_utils.py:15-33,82-92) — scales and quantised values are computed before the permutation,
        so the product x @ dequant(w) is the same matrix product; this layout needs no such permutation and the flag is accepted."""
    FMT = "W4"

    def __init__(self, key, gguf_loader, config, orig_module=None, device: str = "cuda", num_bits: int = 4,
                 group_size: int = 64, act_order: bool = False, is_k_full=True, **kwargs):
        assert device.lower() != "cpu", "Marlin quantized linear only supports GPU device"
        super().__init__(key, gguf_loader, config, orig_module, device, **kwargs)
        if num_bits not in (4, 8):
            raise NotImplementedError(f"KLinearMarlin: num_bits must be 4 or 8 (quant_utils.py:5), got {num_bits}")
        if act_order and group_size == -1:
            raise ValueError("For act_order, groupsize must be less than size_k")          # quant_utils.py:85-88
        self.num_bits, self.group_size, self.act_order, self.is_k_full = num_bits, group_size, act_order, is_k_full
        self.k, self.n = self.in_features, self.out_features
        if num_bits == 8:
            g = self.in_features if group_size == -1 else group_size
            self.FMT = "W8" if (g in (32, 64) or g % 128 == 0) and self.in_features % g == 0 else "BF16"

    def load(self, w=None, device: str | None = None):
        if self.loaded:
            return
        if device is not None:
            self.device = device
        assert self.device.lower() != "cpu", "Marlin quantized linear only supports GPU device"
        if w is None:
            w = self.load_weight(device=self.device)
        weight, bias = _split(w)
        weight = weight.data.to(self.device, torch.bfloat16).view(self.out_features, self.in_features).contiguous()
        self.has_bias = bias is not None
        g = self.in_features if self.group_size == -1 else self.group_size
        b = bias.data.to(self.device, torch.bfloat16) if bias is not None else None
        if self.num_bits == 8 and self.FMT == "W8":
            q, sc = marlin_quantize(weight, 8, g)                                   # [N, K/g, g] signed, [N, K/g, 1]
            qu = (q + 128).to(torch.uint8).view(self.out_features, self.in_features).T.contiguous()          # [K, N]
            hg = min(g, 128)                                                         # groups of a multiple of 128 inputs: one scale per 128
            st = sc.view(self.out_features, -1).repeat_interleave(g // hg, dim=1).T.contiguous()            # [K/hg, N]
            self._h = self._make_handle(hg)
            self._h.load_w8(qu, st, b)
        elif self.num_bits == 8:
            self._h = self._make_handle()
            self._h.load_bf16(marlin_multiplicand(weight, 8, g).contiguous(), b)
        else:
            self._h = self._make_handle(g)
            self._h.load_bf16(weight, b)
        # the reference exposes marlin_q_w here and consumers only use it for shape/device: no bf16 copy is kept
        self.weight = torch.empty((self.in_features, self.out_features), dtype=torch.bfloat16, device="meta")
        self.loaded = True


class KLinearFP8(KLinearBase):
    """DeepSeek block-fp8 — linear.py:388-436.  Needs (weight e4m3, weight_scale_inv)."""
    FMT = "FP8"

    def __init__(self, key, gguf_loader, config, orig_module=None, device: str = "cuda", block_size: int = 128, **kwargs):
        super().__init__(key, gguf_loader, config, orig_module, device, **kwargs)
        self.block_size = block_size

    def load(self, w=None, device: str | None = None):
        if self.loaded:
            return
        if device is not None:
            self.device = device
        if w is None:
            w = self.load_weight(device=self.device)
        if not isinstance(w, tuple):
            raise ValueError("Invalid weight type")              # linear.py:426
        weight, scale_inv = w[0].data, w[1].data
        self._h = self._make_handle(self.block_size)
        self._h.load_fp8(weight.to(self.device).contiguous(), scale_inv.to(self.device, torch.float32).contiguous())
        self.weight, self.weight_scale_inv = weight, scale_inv
        self.loaded = True


def build_merged_linear(template: "KLinearBase", keys: list, loader, device: str, interleave8: bool = False):
    """One operator over the row-concatenation of several linears that share their input (q_proj | kv_a_proj_with_mqa;
    gate_proj | up_proj): per-(group, output-row) quantisation is independent of the other rows, so the merged GEMV gives
    exactly the rows the separate operators would, in one launch.  `template` supplies the operator class and its options."""
    ws = [loader.load_tensor(k + ".weight", device=device) for k in keys]
    if any(loader.has_tensor(k + ".bias") for k in keys):
        return None                                   # biases: keep the separate operators
    if any(loader.has_tensor(k + ".weight_scale_inv") for k in keys):
        # block-fp8 (round 4): the 128-row scale blocks of the concatenation are the matrices' own blocks as long as every matrix but
        # the last ends on a block boundary (q_a_proj's 1536 rows do: q_a | kv_a merge for DeepSeek-V3 / R1 fp8 checkpoints); a GLU
        # interleave would mix two matrices inside one block and is not done
        scs = [loader.load_tensor(k + ".weight_scale_inv", device=device) if loader.has_tensor(k + ".weight_scale_inv") else None for k in keys]
        if (interleave8 or not isinstance(template, KLinearFP8) or any(sc is None for sc in scs)
                or any(t.shape[0] % template.block_size for t in ws[:-1]) or len({t.shape[1] for t in ws}) != 1):
            return None
        w8 = torch.cat([t.view(torch.uint8) if t.dtype != torch.uint8 else t for t in ws], dim=0).contiguous()
        sc = torch.cat([t.to(torch.float32) for t in scs], dim=0).contiguous()
        holder = nn.Linear(w8.shape[1], w8.shape[0], bias=False, device="meta")
        op = type(template)("+".join(keys), loader, template.config, holder, device, max_len=template.max_len, block_size=template.block_size)
        op.load((nn.Parameter(w8.view(torch.float8_e4m3fn), requires_grad=False), nn.Parameter(sc, requires_grad=False)))
        return op, [t.shape[0] for t in ws]
    if interleave8:   # [gate | up] -> per 16-row strip 8 gate rows then 8 up rows (the `glu` epilogue of ktx_linear_forward_fused)
        g, u = (t.to(torch.bfloat16) for t in ws)
        if g.shape != u.shape or g.shape[0] % 8 != 0:
            return None
        w = torch.stack([g.view(-1, 8, g.shape[1]), u.view(-1, 8, u.shape[1])], dim=1).reshape(-1, g.shape[1]).contiguous()
    else:
        w = torch.cat([t.to(torch.bfloat16) for t in ws], dim=0).contiguous()
    holder = nn.Linear(w.shape[1], w.shape[0], bias=False, device="meta")
    kw = {}
    if isinstance(template, KLinearMarlin):
        kw = dict(num_bits=template.num_bits, group_size=template.group_size, act_order=template.act_order)
    op = type(template)("+".join(keys), loader, template.config, holder, device, max_len=template.max_len, **kw)
    op.load(nn.Parameter(w, requires_grad=False))
    return op, [t.shape[0] for t in ws]


LINEAR_MAP = {
    "KLinearMarlin": KLinearMarlin,
    "KLinearTorch": KLinearTorch,
    "KLinearFP8": KLinearFP8,
    # other vendors' back-ends of the same contract (linear.py:896-904)
    "VLinearMarlin": KLinearMarlin,
    "KLinearQ8": KLinearMarlin,
    "KLinearCPUInfer": KLinearTorch,
    "KLinearIPEXLLM": KLinearTorch,
}


class KTransformersLinear(BaseInjectedModule):
    """linear.py:906-990: picks the prefill / generate operator by inference mode."""

    def __init__(self, key: str, gguf_loader, config, orig_module: nn.Module, generate_device: str = "cuda",
                 generate_op: str | None = "KLinearMarlin", prefill_device: str = "cuda",
                 prefill_op: str | None = "KLinearTorch", **kwargs):
        BaseInjectedModule.__init__(self, key, gguf_loader, config, orig_module, prefill_device, generate_device, **kwargs)
        for op in (prefill_op, generate_op):
            assert op is None or op in LINEAR_MAP, f"linear_type {op} not supported"
        object.__setattr__(self, "in_features", orig_module.in_features)
        object.__setattr__(self, "out_features", orig_module.out_features)
        gen = LINEAR_MAP[generate_op](key, gguf_loader, config, orig_module, generate_device, **kwargs) if generate_op else None
        # one quantised copy serves prefill too unless the rule asks for a different *format* on a different device
        if prefill_op is None:
            pre = None
        elif gen is not None and (LINEAR_MAP[prefill_op] is type(gen) or prefill_device == generate_device):
            pre = gen
        else:
            pre = LINEAR_MAP[prefill_op](key, gguf_loader, config, orig_module, prefill_device, **kwargs)
        object.__setattr__(self, "generate_linear", gen)
        object.__setattr__(self, "prefill_linear", pre)
        object.__setattr__(self, "mode", InferenceState.UNLOAD)
        object.__setattr__(self, "weight", None)

    def forward(self, x, bsz_tensor=None, **fusion):
        if self.mode == InferenceState.PREFILL:
            assert self.prefill_linear is not None, "cpu linear is not initialized"
            return self.prefill_linear.forward(x, bsz_tensor, **fusion)
        assert self.generate_linear is not None, "gpu linear is not initialized"
        return self.generate_linear.forward(x, bsz_tensor, **fusion)

    def load(self, w=None, mode: InferenceState = InferenceState.GENERATE):
        if not mode:
            mode = InferenceState.GENERATE
        if mode == InferenceState.UNLOAD:
            return self.unload()
        if mode not in (InferenceState.GENERATE, InferenceState.PREFILL):
            raise ValueError("mode must be either InferenceState.GENERATE, InferenceState.PREFILL or InferenceState.UNLOAD")
        op = self.prefill_linear if mode == InferenceState.PREFILL else self.generate_linear
        other = self.generate_linear if mode == InferenceState.PREFILL else self.prefill_linear
        if other is not None and other is not op:
            other.unload()
        op.load(w=w)
        object.__setattr__(self, "device", op.device)
        object.__setattr__(self, "weight", op.weight)
        object.__setattr__(self, "mode", mode)

    def unload(self):
        for op in (self.prefill_linear, self.generate_linear):
            if op is not None:
                op.unload()
        object.__setattr__(self, "mode", InferenceState.UNLOAD)

    def set_inference_mode(self, mode: InferenceState):
        if not mode:
            mode = InferenceState.GENERATE
        if mode in (InferenceState.GENERATE, InferenceState.PREFILL):
            self.load(mode=mode)
        elif mode == InferenceState.UNLOAD:
            self.unload()
        else:
            raise ValueError("mode must be either InferenceState.GENERATE, InferenceState.PREFILL or InferenceState.UNLOAD")
