"""CPU: the FP8 + GGUF HYBRID checkpoint format (output of the reference's archive/merge_tensors/merge_safetensor_gguf.py,
read by its SafeTensorLoader — BASELINE config C5's weights) through ktransformers_amd.util.loader.SafeTensorLoader.
tests/golden/hybrid_loader_golden.json was produced by the REFERENCE's own merge script and loader on the toy sources of
tests/hybrid_ckpt_builders.py: (1) `write_hybrid` reproduces the script's files tensor for tensor, (2) our loader answers
has_tensor / load_tensor / load_experts / load_gate exactly as the reference's does."""
import json
import os

import pytest
import torch
from safetensors import safe_open

import hybrid_ckpt_builders as B

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hybrid_loader_golden.json")))


@pytest.fixture()
def hybrid_dir(tmp_path):
    d = str(tmp_path / "hybrid")
    B.write_hybrid(d)
    return d


def test_builder_reproduces_the_reference_scripts_files(hybrid_dir):
    inv = {}
    for f in sorted(os.listdir(hybrid_dir)):
        with safe_open(os.path.join(hybrid_dir, f), framework="pt") as h:
            for k in h.keys():
                inv[k] = B.digest(h.get_tensor(k))
    assert inv == GOLD["inventory"]


def test_loader_answers_like_the_reference_loader(hybrid_dir):
    from ktransformers_amd.util.loader import SafeTensorLoader
    got = B.probe(SafeTensorLoader(hybrid_dir))
    assert json.loads(json.dumps(got)) == GOLD["probe"]


def test_hybrid_experts_feed_the_gguf_backend_shapes(hybrid_dir):
    """The raw blocks come back in the shape KExpertsHIP's GGUF path consumes: E x N rows of K/256 blocks."""
    from ktransformers_amd._native import GGML_BLOCK_BYTES
    from ktransformers_amd.util.loader import SafeTensorLoader
    ex = SafeTensorLoader(hybrid_dir).load_experts("model.layers.1.mlp.experts")
    for proj, n, k in (("gate", B.I, B.H), ("up", B.I, B.H), ("down", B.H, B.I)):
        blk = GGML_BLOCK_BYTES[ex[proj + "_type"]]
        assert ex[proj].dtype == torch.uint8 and ex[proj].numel() == B.E * n * (k // 256) * blk
        assert ex[proj].view(torch.uint8).reshape(B.E, n, -1).shape[-1] == (k // 256) * blk


def test_plain_checkpoints_still_take_the_per_expert_branch(tmp_path):
    from ktransformers_amd.util.loader import SafeTensorLoader
    st, gg = str(tmp_path / "st"), str(tmp_path / "gg")
    B.write_sources(st, gg)
    ld = SafeTensorLoader(st)
    assert not ld.is_hybrid_experts("model.layers.1.mlp.experts") and ld.get_expert_count("model.layers.1.mlp.experts") == B.E
    ex = ld.load_experts("model.layers.1.mlp.experts")
    assert ex["gate"].shape == (B.E, B.I, B.H) and ex["gate_scale"].shape == (B.E, B.I // 128, B.H // 128)
    with pytest.raises(KeyError):
        ld.load_tensor("model.layers.9.nope.weight")
    with pytest.raises(FileNotFoundError):
        SafeTensorLoader(str(tmp_path / "absent"))
