"""CPU: the GGUF reader (util/gguf_loader.py) on a file written by tests/helpers.write_gguf — header parsing, offsets,
name translation, de-quantisation and raw expert blocks.  tests/golden/gguf_loader_golden.json records what the
REFERENCE's own GGUFLoader (archive/ktransformers/util/custom_loader.py) reports for the same file."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from helpers import write_gguf
from oracle.gguf_ref import dequantize_q4_k, dequantize_q6_k, quantize_q4_k, quantize_q6_k

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gguf_loader_golden.json")


def build_file(path):
    rng = np.random.default_rng(7)
    E, H, I = 4, 256, 512
    gate = quantize_q4_k((rng.standard_normal((E, I, H)) / 10).astype(np.float32))
    up = quantize_q4_k((rng.standard_normal((E, I, H)) / 10).astype(np.float32))
    down = quantize_q6_k((rng.standard_normal((E, H, I)) / 10).astype(np.float32))
    norm = rng.standard_normal(H).astype(np.float32)
    attn = (rng.standard_normal((64, H)) / 10).astype(np.float16)
    tensors = {
        "blk.1.ffn_gate_exps.weight": (12, [H, I, E], gate.tobytes()),
        "blk.1.ffn_up_exps.weight": (12, [H, I, E], up.tobytes()),
        "blk.1.ffn_down_exps.weight": (14, [I, H, E], down.tobytes()),
        "blk.1.attn_norm.weight": (0, [H], norm.tobytes()),
        "blk.1.attn_kv_a_mqa.weight": (1, [H, 64], attn.tobytes()),
    }
    write_gguf(path, tensors, {"deepseek2.expert_count": E})
    return dict(gate=gate, up=up, down=down, norm=norm, attn=attn, E=E, H=H, I=I)


def build_llama_file(path, n_head=4, n_kv=2, hd=8, H=16):
    """attn_q / attn_k / attn_v of a llama-architecture file; column 0 of stored row r holds r."""
    rng = np.random.default_rng(11)
    t = {}
    src = {}
    for name, rows in (("attn_q", n_head * hd), ("attn_k", n_kv * hd), ("attn_v", n_kv * hd)):
        a = rng.standard_normal((rows, H)).astype(np.float32)
        a[:, 0] = np.arange(rows)
        src[name] = a
        t[f"blk.0.{name}.weight"] = (0, [H, rows], a.tobytes())
    write_gguf(path, t, {"general.architecture": "llama", "llama.attention.head_count": n_head,
                         "llama.attention.head_count_kv": n_kv})
    return src


def test_llama_files_get_the_reference_q_k_row_order(tmp_path):
    """custom_loader.py:507-517: q / k rows come back de-interleaved per head (HF rotate-half order), v untouched; the golden is
    what the reference's own loader returned for the same file (tests/golden/make_gguf_llama_golden.py)."""
    from ktransformers_amd.util.gguf_loader import GGUFLoader
    src = build_llama_file(str(tmp_path / "toy.gguf"))
    gold = json.load(open(os.path.join(os.path.dirname(GOLD), "gguf_llama_golden.json")))
    ld = GGUFLoader(str(tmp_path))
    for name, g in gold.items():
        v = ld.load_gguf_tensor(name, target_dtype=torch.float32)
        assert list(v.shape) == g["shape"]
        assert [int(x) for x in v[:, 0].tolist()] == g["row_ids"], name
        assert torch.equal(v, torch.from_numpy(src[name.split(".")[2]])[g["row_ids"]])
    assert gold["blk.0.attn_v.weight"]["row_ids"] == list(range(16)) and gold["blk.0.attn_q.weight"]["row_ids"][:4] == [0, 2, 4, 6]


def test_reader_matches_reference_loader_and_roundtrips(tmp_path):
    from ktransformers_amd.util.gguf_loader import GGUFLoader, translate_name_to_gguf
    path = str(tmp_path / "toy.gguf")
    src = build_file(path)
    gold = json.load(open(GOLD))
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == gold["sha256"], "writer output changed; regenerate the golden"
    ld = GGUFLoader(str(tmp_path))
    for name, g in gold["tensors"].items():
        t = ld.tensor_info[name]
        assert (t["ggml_type"], t["shape"], t["offset"]) == (g["ggml_type"], g["shape"], g["offset"]), name
    assert ld.gguf_file_meta["deepseek2.expert_count"] == gold["meta"]["deepseek2.expert_count"]
    # HF names resolve like the reference's translate_name_to_gguf
    for hf, gg in gold["names"].items():
        assert translate_name_to_gguf(hf) == gg
    assert ld.has_tensor("model.layers.1.input_layernorm.weight") and not ld.has_tensor("model.layers.2.input_layernorm.weight")
    assert torch.equal(ld.load_gguf_tensor("model.layers.1.input_layernorm.weight", target_dtype=torch.float32),
                       torch.from_numpy(src["norm"]))
    a = ld.load_gguf_tensor("model.layers.1.self_attn.kv_a_proj_with_mqa.weight", target_dtype=torch.float32)
    assert a.shape == (64, src["H"]) and torch.equal(a, torch.from_numpy(src["attn"].astype(np.float32)))
    ex = ld.load_experts("model.layers.1.mlp.experts")
    assert (ex["gate_type"], ex["up_type"], ex["down_type"]) == (12, 12, 14)
    assert np.array_equal(ex["gate"].numpy(), src["gate"].reshape(-1)) and np.array_equal(ex["down"].numpy(), src["down"].reshape(-1))
    assert ld.get_expert_count("model.layers.1.mlp.experts") == src["E"]
    deq = ld.load_gguf_tensor("blk.1.ffn_down_exps.weight", target_dtype=torch.float32)
    assert deq.shape == (src["E"], src["H"], src["I"])
    assert np.array_equal(deq.numpy(), dequantize_q6_k(src["down"]).reshape(src["E"], src["H"], src["I"]))
    deq4 = ld.load_gguf_tensor("blk.1.ffn_gate_exps.weight", target_dtype=torch.float32)
    assert np.array_equal(deq4.numpy(), dequantize_q4_k(src["gate"]).reshape(src["E"], src["I"], src["H"]))


@pytest.mark.parametrize("t", [2, 6, 8, 10, 11, 12, 13, 14, 23])
def test_every_type_the_reference_loader_reads_dequantises_bit_for_bit(t):
    """util/gguf_loader._dequant against the REFERENCE's own numpy dequantisers on committed random blocks
    (tests/golden/make_gguf_golden.py; custom_gguf.py:225-572): Q4_0, Q5_0, Q8_0, Q2_K..Q6_K, IQ4_XS — same values, same
    float32 association (d*scale, then *q, then - dmin*min)."""
    from ktransformers_amd.util.gguf_loader import GGML_QUANT_SIZES, _dequant
    g = np.load(os.path.join(os.path.dirname(GOLD), "gguf_blocks_golden.npz"))
    blocks, want = {12: ("q4k_blocks", "q4k_values"), 14: ("q6k_blocks", "q6k_values")}.get(t, (f"t{t}_blocks", f"t{t}_values"))
    blocks, want = g[blocks], g[want]
    assert blocks.shape[1] == GGML_QUANT_SIZES[t][1] and want.shape[1] == GGML_QUANT_SIZES[t][0]
    got = _dequant(t, blocks.reshape(-1)).reshape(want.shape)
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_native_expert_types_use_the_loaders_block_sizes():
    """The ggml types the expert kernels read natively — Q2_K..Q6_K, IQ1_S, IQ4_XS (csrc/ktx_moe_gguf.inc: 256-weight blocks, Q8_K
    activations) and, round 5, Q4_0 / Q5_0 / Q8_0 (csrc/ktx_moe_legacy.inc: 32-weight blocks, Q8_0 activations) — and the loader's table
    of block sizes (custom_gguf.py:72-100) name the same (weights, bytes) per block."""
    from ktransformers_amd import _native
    from ktransformers_amd.util.gguf_loader import GGML_QUANT_SIZES, GGML_TYPES
    assert set(_native.GGML_BLOCK_BYTES) == ({GGML_TYPES[n] for n in ("Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K", "IQ4_XS")} | {19}
                                             | {GGML_TYPES[n] for n in ("Q4_0", "Q5_0", "Q8_0")})
    assert set(_native.GGML_LEGACY_TYPES) == {GGML_TYPES[n] for n in ("Q4_0", "Q5_0", "Q8_0")}
    for t, nbytes in _native.GGML_BLOCK_BYTES.items():
        assert GGML_QUANT_SIZES[t] == (_native.ggml_block_elems(t), nbytes)
