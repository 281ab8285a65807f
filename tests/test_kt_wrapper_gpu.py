"""GPU: ktransformers_amd.kt_kernel.KTMoEWrapper — the mirror of the reference's live SGLang-facing API — end to end on the
HIP experts: online-quantised AMXINT4 against the oracle (bit-exact), the deferred-expert protocol across two layers,
gpu_experts_mask, physical->logical maps, and every checkpoint format against a directly loaded handle."""
import numpy as np
import pytest
import torch

import kt_ckpt_builders as B
from helpers import make_case, numpy_u16, torch_bf16
from oracle.oracle import FMT_AMXINT4

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


@pytest.fixture(autouse=True)
def fresh_state():
    from ktransformers_amd.kt_kernel import backends
    from ktransformers_amd.kt_kernel.experts_base import BaseMoEWrapper
    BaseMoEWrapper.clear_buffer_cache()
    BaseMoEWrapper._layer_has_pending_deferred.clear()
    backends.NativeMoEWrapper._native_loader_instance = None
    backends.AMXMoEWrapper._safetensor_loader_instance = None
    backends.LlamafileMoEWrapper._gguf_loader_instance = None
    yield


def make(method, layer, E, k, H, I, path="/tmp", **kw):
    from ktransformers_amd.kt_kernel import KTMoEWrapper
    return KTMoEWrapper(layer_idx=layer, num_experts=E, num_experts_per_tok=k, hidden_size=H, moe_intermediate_size=I,
                        gpu_experts_mask=kw.pop("mask", None), cpuinfer_threads=8, threadpool_count=1, weight_path=path,
                        chunked_prefill_size=64, method=method, **kw)


def stream():
    return torch.cuda.current_stream().cuda_stream


def test_amxint4_online_quant_mask_and_submit_sync(oracle, dev):
    E, k, H, I, T = 8, 2, 512, 256, 5
    c = make_case(11, E, k, H, I, T)
    mask = torch.zeros(E, dtype=torch.bool)
    mask[[1, 6]] = True
    w = make("AMXINT4", 0, E, k, H, I, mask=mask)
    w.load_weights_from_tensors(torch_bf16(c["gate"], "cpu"), torch_bf16(c["up"], "cpu"), torch_bf16(c["down"], "cpu"), torch.arange(E))
    x, ids, wt = torch_bf16(c["x"], dev), torch.from_numpy(c["ids"]).to(dev).to(torch.int32), torch.from_numpy(c["w"]).to(dev)
    w.submit_forward(x, ids, wt, stream())
    y = w.sync_forward(x, stream())
    torch.cuda.synchronize()
    ids_masked = np.where(np.isin(c["ids"], [1, 6]), -1, c["ids"])
    mo = oracle.make_moe(FMT_AMXINT4, c["gate"], c["up"], c["down"])
    assert np.array_equal(numpy_u16(y), oracle.moe_forward(mo, ids_masked, c["w"], c["x"]))
    assert w.num_gpu_experts == 2 and y.shape == (T, H) and y.dtype == torch.bfloat16


def test_deferred_experts_fold_into_the_next_layer(oracle, dev):
    E, k, H, I, T = 32, 4, 256, 256, 3   # few tokens, many experts: most 4th-ranked experts are protected by no token
    cs = [make_case(20 + l, E, k, H, I, T) for l in range(2)]
    ws = []
    for l, c in enumerate(cs):
        w = make("AMXINT4", l, E, k, H, I, max_deferred_experts_per_token=1)
        w.load_weights_from_tensors(torch_bf16(c["gate"], dev), torch_bf16(c["up"], dev), torch_bf16(c["down"], dev), torch.arange(E))
        ws.append(w)
    outs, split = [], []
    for w, c in zip(ws, cs):
        x, ids, wt = torch_bf16(c["x"], dev), torch.from_numpy(c["ids"]).to(dev), torch.from_numpy(c["w"]).to(dev)
        imm, dfr = w.select_deferred_experts(ids, wt, k - 1)
        split.append((imm.cpu().numpy(), dfr.cpu().numpy()))
        outs.append(w.forward(x, ids, wt, stream()).clone())
    torch.cuda.synchronize()
    mos = [oracle.make_moe(FMT_AMXINT4, c["gate"], c["up"], c["down"]) for c in cs]
    want0 = oracle.moe_forward(mos[0], split[0][0], cs[0]["w"], cs[0]["x"])
    deferred0 = oracle.moe_forward(mos[0], split[0][1], cs[0]["w"], cs[0]["x"])
    want1 = oracle.moe_forward(mos[1], split[1][0], cs[1]["w"], cs[1]["x"], y_prev=deferred0)
    assert (split[0][1] >= 0).any() and (split[0][0] >= 0).any()
    assert np.array_equal(numpy_u16(outs[0]), want0)
    assert np.array_equal(numpy_u16(outs[1]), want1)


def _fp8_per_channel(folder, dims=None):
    return B.fp8_block(folder, scale="weight_scale", per_channel=True, dims=dims)


@pytest.mark.parametrize("method,builder,layer", [("FP8", B.fp8_block, 2), ("BF16", B.bf16_per_expert, 3), ("RAWINT4", B.compressed_int4, 5),
                                                  ("FP8_PERCHANNEL", _fp8_per_channel, 2)])
def test_checkpoint_formats_equal_a_directly_loaded_handle(dev, tmp_path, method, builder, layer):
    from ktransformers_amd._native import MoEHandle
    from ktransformers_amd.kt_kernel.utils import loader as L
    E, H, I, k, T = 3, 512, 512, 2, 7   # RAWINT4 needs multiples of 512
    builder(str(tmp_path), dims=(E, H, I))
    perm = [2, 0, 1]
    w = make(method, layer, E, k, H, I, path=str(tmp_path))
    w.load_weights(torch.tensor(perm))
    src = {"FP8": L.FP8SafeTensorLoader, "BF16": L.BF16SafeTensorLoader, "RAWINT4": L.CompressedSafeTensorLoader,
           "FP8_PERCHANNEL": lambda p: L.FP8SafeTensorLoader(p, scale_suffix="weight_scale")}[method](str(tmp_path))
    e = src.load_experts(f"model.layers.{layer}")
    st = lambda name: torch.stack([e[name][i] for i in perm]).to(dev).contiguous()
    h = MoEHandle(E, k, H, I, max_len=64, method=method, device=dev, group_size={"FP8": 128, "RAWINT4": 32}.get(method, 0))
    if method == "BF16":
        h.load_bf16(st("gate"), st("up"), st("down"))
    elif method == "FP8":
        h.load_fp8(st("gate").view(torch.uint8), st("up").view(torch.uint8), st("down").view(torch.uint8), st("gate_scale"),
                   st("up_scale"), st("down_scale"))
    elif method == "FP8_PERCHANNEL":
        h.load_fp8_perchannel(st("gate").view(torch.uint8), st("up").view(torch.uint8), st("down").view(torch.uint8),
                              st("gate_scale"), st("up_scale"), st("down_scale"))
    else:
        h.load_rawint4(st("gate"), st("up"), st("down"), st("gate_scale"), st("up_scale"), st("down_scale"))
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(T, H, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    ids = torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(T)]).to(dev)
    wt = torch.rand(T, k, generator=g).to(dev)
    got = w.forward(x, ids, wt, stream()).clone()
    want = h.forward(x, ids, wt)
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all() and got.float().abs().max() > 0
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


def test_amx_packed_checkpoint_equals_online_quantisation(dev, tmp_path):
    from safetensors.numpy import save_file
    from oracle import oracle as O
    if not O.reference_available():
        pytest.skip("oracle/_ref not available on this host: the packed checkpoint is written with the reference's own packer")
    from test_amx_packed_cpu import pack_with_reference
    E, k, H, I, T = 4, 2, 256, 256, 9
    c = make_case(31, E, k, H, I, T)
    tensors = {}
    for fam, (n, kk) in (("gate", (I, H)), ("up", (I, H)), ("down", (H, I))):
        for e in range(E):
            packed, scale = pack_with_reference(0, np.ascontiguousarray(c[fam][e]), n, kk)
            tensors[f"blk.3.ffn_{fam}_exps.{e}.numa.0.weight"] = packed.view(np.int8)
            tensors[f"blk.3.ffn_{fam}_exps.{e}.numa.0.scale"] = scale
    save_file(tensors, str(tmp_path / "packed.safetensors"))
    a = make("AMXINT4", 3, E, k, H, I, path=str(tmp_path))
    a.load_weights(torch.arange(E))
    b = make("AMXINT4", 5, E, k, H, I)
    b.load_weights_from_tensors(torch_bf16(c["gate"], dev), torch_bf16(c["up"], dev), torch_bf16(c["down"], dev), torch.arange(E))
    x, ids, wt = torch_bf16(c["x"], dev), torch.from_numpy(c["ids"]).to(dev), torch.from_numpy(c["w"]).to(dev)
    ya = a.forward(x, ids, wt, stream()).clone()
    yb = b.forward(x, ids, wt, stream()).clone()
    torch.cuda.synchronize()
    assert torch.equal(ya.view(torch.int16), yb.view(torch.int16))


def test_numa_sharded_checkpoint_reproduces_the_reference_tp_moe(dev, tmp_path):
    """A checkpoint converted with threadpool_count = 2 (gate / up split over rows, down over K with one scale per (row, part);
    kt-kernel/python/utils/loader.py:179-290): the wrapper runs one handle per part and adds their fp32 outputs in part order —
    bit-identical to what the reference's TP_MOE (tp_count = 2, oracle/_ref) computes from the same weights, plain and
    incremental (merge_results, operators/amx/moe_base.hpp:749-791).  tests/golden/make_kt_tp_golden.py; 2263 of the 2304
    outputs differ from the one-part arithmetic, so the check tells the two apart."""
    import os
    from safetensors.numpy import save_file
    from ktransformers_amd import _native
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kt_tp_golden.npz"))
    E, k, H, I, P = (int(g[n]) for n in ("E", "k", "H", "I", "P"))
    assert int(g["differs_from_one_part"]) > 2000
    save_file({n: g[n] for n in g.files if n.startswith("blk.")}, str(tmp_path / "packed.safetensors"))
    a = make("AMXINT4", 3, E, k, H, I, path=str(tmp_path))
    a.load_weights(torch.arange(E))
    assert len(a.tp_parts) == P and a.tp_parts[0].I == I // P
    x, ids, wt = torch_bf16(g["x"], dev), torch.from_numpy(g["ids"]).to(dev), torch.from_numpy(g["w"]).to(dev)
    y = a.forward(x, ids, wt, stream()).clone()
    torch.cuda.synchronize()
    assert np.array_equal(numpy_u16(y), g["y"]), f"{int((numpy_u16(y) != g['y']).sum())} outputs differ from the reference's tp_count=2 run"
    # incremental merge: ((part0 + y_prev) + part1), one rounding
    parts = torch.stack([h.forward_partial(x, ids, wt) for h in a.tp_parts])
    out = torch_bf16(g["y_prev"], dev).clone()
    _native.moe_merge_partials(parts, out, incremental=True)
    torch.cuda.synchronize()
    assert np.array_equal(numpy_u16(out), g["y_inc"])
    # under a HIP graph, twice
    static_x = x.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        yg = a.forward(static_x, ids, wt, stream())
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(numpy_u16(yg), g["y"])


def test_llamafile_gguf_equals_a_directly_loaded_handle(dev, tmp_path):
    from helpers import write_gguf
    from ktransformers_amd._native import MoEHandle
    from oracle.gguf_ref import GGML_TYPE_Q4_K, GGML_TYPE_Q6_K, QUANT
    E, k, H, I, T = 2, 1, 256, 256, 4
    rng = np.random.default_rng(3)
    gate = QUANT[GGML_TYPE_Q4_K]((rng.standard_normal((E, I, H)) / 10).astype(np.float32))
    up = QUANT[GGML_TYPE_Q4_K]((rng.standard_normal((E, I, H)) / 10).astype(np.float32))
    down = QUANT[GGML_TYPE_Q6_K]((rng.standard_normal((E, H, I)) / 10).astype(np.float32))
    write_gguf(str(tmp_path / "m.gguf"), {"blk.0.ffn_gate_exps.weight": (12, [H, I, E], gate.tobytes()),
                                         "blk.0.ffn_up_exps.weight": (12, [H, I, E], up.tobytes()),
                                         "blk.0.ffn_down_exps.weight": (14, [I, H, E], down.tobytes())})
    w = make("LLAMAFILE", 0, E, k, H, I, path=str(tmp_path))
    w.load_weights()
    h = MoEHandle(E, k, H, I, max_len=64, method="GGUF", device=dev)
    h.load_gguf(*(torch.from_numpy(np.ascontiguousarray(a)).reshape(E, a.shape[1], -1).to(dev) for a in (gate, up, down)), 12, 12, 14)
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(T, H, generator=g)).to(torch.bfloat16).to(dev)
    ids = torch.randint(0, E, (T, k), generator=g).to(dev)
    wt = torch.rand(T, k, generator=g).to(dev)
    got = w.forward(x, ids, wt, stream()).clone()
    want = h.forward(x, ids, wt)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)) and got.float().abs().max() > 0
