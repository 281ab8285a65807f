"""CPU: the host logic of ktransformers_amd.kt_kernel (the mirror of the reference's live `kt_kernel` Python package):
factory validation, expert masks and the deferred-expert split — the latter two against the reference's own functions
(tests/golden/kt_wrapper_golden.npz, made by tests/golden/make_kt_wrapper_golden.py)."""
import os
import types

import numpy as np
import pytest
import torch

from ktransformers_amd import kt_kernel
from ktransformers_amd.kt_kernel.experts_base import BaseMoEWrapper, KExpertsDeviceBuffer

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kt_wrapper_golden.npz"))
ARGS = dict(layer_idx=0, num_experts=8, num_experts_per_tok=2, hidden_size=256, moe_intermediate_size=256, gpu_experts_mask=None,
            cpuinfer_threads=4, threadpool_count=1, weight_path="/nonexistent", chunked_prefill_size=64)


@pytest.mark.parametrize("n", [0, 3, 17, 200])
def test_generate_gpu_experts_masks(n):
    got = kt_kernel.generate_gpu_experts_masks(torch.from_numpy(G["freq"]), n)
    assert got.dtype == torch.bool and got.device.type == "cpu"
    assert np.array_equal(got.numpy(), G[f"mask_{n}"])


@pytest.mark.parametrize("protected_k", [0, 2, 6, 9])
def test_select_deferred_experts(protected_k):
    me = types.SimpleNamespace(num_experts=16)
    imm, dfr = BaseMoEWrapper.select_deferred_experts(me, torch.from_numpy(G["ids"]), torch.from_numpy(G["scores"]), protected_k)
    assert np.array_equal(imm.numpy(), G[f"imm_{protected_k}"])
    assert np.array_equal(dfr.numpy(), G[f"def_{protected_k}"])
    ids = G["ids"]
    assert np.array_equal(np.where(imm.numpy() >= 0, imm.numpy(), dfr.numpy()), ids)  # a partition of the routed slots


def test_factory_validation():
    with pytest.raises(ValueError, match="Unknown mode"):
        kt_kernel.KTMoEWrapper(**ARGS, mode="train")
    with pytest.raises(ValueError, match="not supported for inference"):
        kt_kernel.KTMoEWrapper(**ARGS, method="INT3")
    with pytest.raises(ValueError, match="not supported for SFT"):
        kt_kernel.KTMoEWrapper(**ARGS, mode="sft", method="AMXINT4")
    with pytest.raises(NotImplementedError, match="inference only"):
        kt_kernel.KTMoEWrapper(**ARGS, mode="sft", method="AMXBF16_SFT")
    with pytest.raises(ValueError, match="swiglu_limit"):
        kt_kernel.KTMoEWrapper(**ARGS, method="AMXINT4", swiglu_limit=10.0)
    for m in sorted(kt_kernel.INFERENCE_METHODS - kt_kernel.SUPPORTED_METHODS):
        with pytest.raises(NotImplementedError, match="no HIP implementation"):
            kt_kernel.KTMoEWrapper(**ARGS, method=m)
    with pytest.raises(FileNotFoundError):
        kt_kernel.KTMoEWrapper(**ARGS, method="LLAMAFILE")
    with pytest.raises(FileNotFoundError):
        kt_kernel.KTMoEWrapper(**ARGS, method="FP8")  # the loader is created eagerly, like NativeMoEWrapper.__init__
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU path"):
            kt_kernel.KTMoEWrapper(**ARGS, method="AMXINT4")


def test_capture_batch_size_policy():
    kt_kernel.KTMoEWrapper.clear_buffer_cache()
    kt_kernel.KTMoEWrapper.set_capture_batch_sizes([1, 4])
    assert kt_kernel.KTMoEWrapper.get_capture_batch_sizes() == [1, 4]
    a = KExpertsDeviceBuffer.get_buffer(torch.zeros(4, 32, dtype=torch.bfloat16), 2)
    b = KExpertsDeviceBuffer.get_buffer(torch.zeros(3, 32, dtype=torch.bfloat16), 2)
    assert len(a) == 2 and a[0].shape == (4, 32) and a[0].dtype == torch.bfloat16
    assert KExpertsDeviceBuffer.get_buffer(torch.zeros(4, 32, dtype=torch.bfloat16), 2) is a      # captured size: kept
    assert KExpertsDeviceBuffer.get_buffer(torch.zeros(3, 32, dtype=torch.bfloat16), 2) is b      # last other size: cached
    KExpertsDeviceBuffer.get_buffer(torch.zeros(5, 32, dtype=torch.bfloat16), 2)
    assert KExpertsDeviceBuffer.get_buffer(torch.zeros(3, 32, dtype=torch.bfloat16), 2) is not b  # ...one entry only
    kt_kernel.KTMoEWrapper.clear_buffer_cache()
    kt_kernel.KTMoEWrapper.set_capture_batch_sizes([])
    assert KExpertsDeviceBuffer.get_buffer(torch.zeros(4, 32, dtype=torch.bfloat16), 2) is not a


def test_logical_order():
    assert BaseMoEWrapper._logical_order(None, 3) == [0, 1, 2]
    assert BaseMoEWrapper._logical_order(torch.tensor([2, 0, 1]), 3) == [2, 0, 1]
    with pytest.raises(ValueError):
        BaseMoEWrapper._logical_order(torch.tensor([0, 1]), 3)
    with pytest.raises(ValueError):
        BaseMoEWrapper._logical_order(torch.tensor([0, 1, 3]), 3)


# ---- load paths, with a recording stand-in for the HIP handle (the real one is exercised in test_kt_wrapper_gpu.py) --------
class RecordingHandle:
    """Accepts what _native.MoEHandle accepts (same shape / dtype rules) and keeps the tensors for inspection."""

    def __init__(self, E, k, H, I, max_len, method, device, group_size=0):
        self.E, self.k, self.H, self.I, self.method, self.group_size, self.max_len = E, k, H, I, method, group_size, max_len
        self.calls, self.quant, self.mask = [], {}, None

    def set_expert_mask(self, m):
        self.mask = np.array(m)

    def _chk(self, ts, shapes, dtype):
        for t, s in zip(ts, shapes):
            assert tuple(t.shape) == s and t.is_contiguous() and (dtype is None or t.dtype == dtype), (t.shape, s, t.dtype)

    def load_bf16(self, g, u, d):
        self._chk((g, u, d), [(self.E, self.I, self.H)] * 2 + [(self.E, self.H, self.I)], torch.bfloat16)
        self.calls.append(("bf16", g, u, d))

    def load_fp8(self, g, u, d, gs, us, ds):
        self._chk((g, u, d), [(self.E, self.I, self.H)] * 2 + [(self.E, self.H, self.I)], torch.uint8)
        self._chk((gs, us, ds), [(self.E, self.I // 128, self.H // 128)] * 2 + [(self.E, self.H // 128, self.I // 128)], torch.float32)
        self.calls.append(("fp8", g, u, d, gs, us, ds))

    def load_rawint4(self, g, u, d, gs, us, ds):
        self._chk((g, u, d), [(self.E, self.I, self.H // 2)] * 2 + [(self.E, self.H, self.I // 2)], torch.uint8)
        self._chk((gs, us, ds), [(self.E, self.I, self.H // 32)] * 2 + [(self.E, self.H, self.I // 32)], torch.bfloat16)
        self.calls.append(("rawint4", g, u, d, gs, us, ds))

    def load_gguf(self, g, u, d, gt, ut, dt):
        self.calls.append(("gguf", g, u, d, gt, ut, dt))

    def load_quantized(self, e, which, q, s):
        self.quant[(e, which)] = (np.array(q), np.array(s))


@pytest.fixture
def recording(monkeypatch):
    from ktransformers_amd.kt_kernel import backends, experts_base
    monkeypatch.setattr(experts_base._native, "MoEHandle", RecordingHandle)
    for cls in (backends.NativeMoEWrapper, backends.AMXMoEWrapper, backends.LlamafileMoEWrapper):
        for attr in ("_native_loader_instance", "_safetensor_loader_instance", "_gguf_loader_instance"):
            if hasattr(cls, attr):
                monkeypatch.setattr(cls, attr, None)
    yield


def _wrapper(method, path, L, **kw):
    import kt_ckpt_builders as B
    return kt_kernel.KTMoEWrapper(layer_idx=L, num_experts=B.E, num_experts_per_tok=2, hidden_size=B.H, moe_intermediate_size=B.I,
                                  gpu_experts_mask=kw.pop("mask", None), cpuinfer_threads=1, threadpool_count=1, weight_path=path,
                                  chunked_prefill_size=32, method=method, device=torch.device("cpu"), **kw)


def test_native_load_paths(recording, tmp_path, capsys):
    import kt_ckpt_builders as B
    from ktransformers_amd.kt_kernel.utils import loader as L
    perm = torch.tensor([2, 0, 1])
    for method, builder, layer, kind in (("FP8", B.fp8_block, 2, "fp8"), ("BF16", B.bf16_per_expert, 3, "bf16"),
                                          ("RAWINT4", B.compressed_int4, 5, "rawint4")):
        d = str(tmp_path / method)
        builder(d)
        mask = torch.tensor([False, True, False])
        w = _wrapper(method, d, layer, mask=mask)
        w.load_weights(perm)
        call = w.moe.calls[0]
        assert call[0] == kind and np.array_equal(w.moe.mask, [0, 1, 0]) and w.num_gpu_experts == 1
        ref = {"FP8": L.FP8SafeTensorLoader, "BF16": L.BF16SafeTensorLoader, "RAWINT4": L.CompressedSafeTensorLoader}[method](d)
        src = ref.load_experts(f"model.layers.{layer}")
        for slot, logical in enumerate(perm.tolist()):  # physical slot i holds logical expert map[i]
            assert torch.equal(call[1][slot].view(torch.uint8), src["gate"][logical].view(torch.uint8))
            assert torch.equal(call[3][slot].view(torch.uint8), src["down"][logical].view(torch.uint8))
            if method != "BF16":
                assert torch.equal(call[5][slot], src["up_scale"][logical])
        if method == "RAWINT4":
            assert w.moe.group_size == 32
        if method == "FP8":
            assert w.moe.group_size == 128
        with pytest.raises(NotImplementedError):
            w.load_weights_from_tensors(None, None, None, None)
        from ktransformers_amd.kt_kernel.backends import NativeMoEWrapper
        assert NativeMoEWrapper._native_loader_instance is None  # released after the layer, like the reference
    with pytest.raises(ValueError, match="No experts found"):
        d = str(tmp_path / "FP8")
        _wrapper("FP8", d, 7).load_weights(None)


def test_amx_from_tensors_and_packed(recording, tmp_path):
    import kt_ckpt_builders as B
    from safetensors.numpy import save_file
    from oracle import oracle as O
    g = torch.randn(B.E, B.I, B.H).to(torch.float16)
    w = _wrapper("AMXINT4", str(tmp_path), 0)
    w.load_weights_from_tensors(g, g, torch.randn(B.E, B.H, B.I), torch.arange(B.E))
    assert w.moe.calls[0][0] == "bf16" and w.moe.calls[0][1].dtype == torch.bfloat16 and w.moe.method == "AMXINT4"
    with pytest.raises(FileNotFoundError):
        w.load_weights(None)
    if not O.reference_available():
        pytest.skip("oracle/_ref not available: packed-checkpoint leg needs the reference's packer")
    from test_amx_packed_cpu import pack_with_reference, quantise_numpy
    rng = np.random.default_rng(0)
    mats, tensors = {}, {}
    for fam, (n, k) in (("gate", (B.I, B.H)), ("up", (B.I, B.H)), ("down", (B.H, B.I))):
        for e in range(B.E):
            wf = O.bf16_to_f32(O.f32_to_bf16((rng.standard_normal((n, k)) * 0.05).astype(np.float32)))
            mats[(fam, e)] = wf
            packed, scale = pack_with_reference(1, O.f32_to_bf16(wf), n, k)
            tensors[f"blk.4.ffn_{fam}_exps.{e}.numa.0.weight"] = packed.view(np.int8)
            tensors[f"blk.4.ffn_{fam}_exps.{e}.numa.0.scale"] = scale
    d = tmp_path / "packed"
    d.mkdir()
    save_file(tensors, str(d / "w.safetensors"))
    w8 = _wrapper("AMXINT8", str(d), 4)
    w8.load_weights(torch.tensor([1, 2, 0]))
    for slot, logical in enumerate([1, 2, 0]):
        for which, fam in enumerate(("gate", "up", "down")):
            q, s = w8.moe.quant[(slot, which)]
            want_q, want_d = quantise_numpy(mats[(fam, logical)], 8)
            assert np.array_equal(q, want_q) and np.array_equal(s, want_d)


def test_llamafile_load_path(recording, tmp_path):
    from helpers import write_gguf
    E, H, I = 2, 256, 256
    rng = np.random.default_rng(0)
    raw = {f: rng.integers(0, 256, n, dtype=np.uint8) for f, n in (("gate", E * I * 144), ("up", E * I * 144), ("down", E * H * 210))}
    write_gguf(str(tmp_path / "m.gguf"), {"blk.1.ffn_gate_exps.weight": (12, [H, I, E], raw["gate"].tobytes()),
                                         "blk.1.ffn_up_exps.weight": (12, [H, I, E], raw["up"].tobytes()),
                                         "blk.1.ffn_down_exps.weight": (14, [I, H, E], raw["down"].tobytes())})
    w = kt_kernel.KTMoEWrapper(layer_idx=1, num_experts=E, num_experts_per_tok=1, hidden_size=H, moe_intermediate_size=I,
                               gpu_experts_mask=None, cpuinfer_threads=1, threadpool_count=1, weight_path=str(tmp_path),
                               chunked_prefill_size=8, method="LLAMAFILE", device=torch.device("cpu"))
    w.load_weights(torch.tensor([1, 0]))
    kind, g, u, d, gt, ut, dt = w.moe.calls[0]
    assert (kind, gt, ut, dt) == ("gguf", 12, 12, 14) and g.shape == (E, I, 144) and d.shape == (E, H, 210)
    assert np.array_equal(g[0].numpy().reshape(-1), raw["gate"].reshape(E, -1)[1])
    assert np.array_equal(d[1].numpy().reshape(-1), raw["down"].reshape(E, -1)[0])
    with pytest.raises(NotImplementedError):
        w.load_weights_from_tensors(None, None, None, None)
    with pytest.raises(ValueError, match="QK_K"):
        kt_kernel.KTMoEWrapper(layer_idx=1, num_experts=E, num_experts_per_tok=1, hidden_size=H, moe_intermediate_size=1408,
                               gpu_experts_mask=None, cpuinfer_threads=1, threadpool_count=1, weight_path=str(tmp_path),
                               chunked_prefill_size=8, method="LLAMAFILE", device=torch.device("cpu"))
