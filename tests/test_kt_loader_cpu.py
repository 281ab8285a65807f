"""CPU: ktransformers_amd.kt_kernel.utils.loader against the reference's own kt_kernel loaders reading the same files
(tests/golden/kt_loader_golden.json, made by tests/golden/make_kt_loader_golden.py): every returned tensor (shape, dtype,
bytes), the detected naming / scale format attributes, and the error type for absent layers."""
import contextlib
import io
import json
import os

import pytest
import torch

import kt_ckpt_builders as B
from ktransformers_amd.kt_kernel.utils import loader as ours

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kt_loader_golden.json")))


@pytest.mark.parametrize("name", sorted(B.CASES))
def test_loader_matches_reference(name, tmp_path):
    with contextlib.redirect_stdout(io.StringIO()):
        got = B.run_case(ours, name, str(tmp_path))
    assert json.loads(json.dumps(got)) == GOLD[name]


def test_missing_path_and_empty_folder(tmp_path):
    with pytest.raises(FileNotFoundError):
        ours.SafeTensorLoader(str(tmp_path / "nope"))
    with pytest.raises(FileNotFoundError):
        ours.SafeTensorLoader(str(tmp_path))
    B.bf16_per_expert(str(tmp_path))
    ld = ours.BF16SafeTensorLoader(str(tmp_path / "model.safetensors"))  # a file path means "its folder"
    with pytest.raises(KeyError):
        ld.load_tensor("absent")
    assert ld.load_tensor("model.layers.3.mlp.experts.0.gate_proj.weight").dtype == torch.bfloat16


def test_rawint4_normalisation_errors():
    f = ours.CompressedSafeTensorLoader._normalize_rawint4_weight
    w = torch.zeros(4, 8, dtype=torch.uint8)
    s = torch.zeros(4, 2, dtype=torch.bfloat16)
    with pytest.raises(TypeError):
        f(w.to(torch.int16), s)
    with pytest.raises(ValueError):
        f(w, s, torch.tensor([4, 16, 1]))
    with pytest.raises(ValueError):
        f(w, torch.zeros(3, 2, dtype=torch.bfloat16), torch.tensor([4, 16]))
    with pytest.raises(ValueError):
        f(w, torch.zeros(4, 3, dtype=torch.bfloat16), torch.tensor([4, 16]))
    assert f(w.view(torch.int32), s, torch.tensor([4, 16])).shape == (4, 8)
    assert f(w, s, torch.tensor([5, 16])) is not None  # inconsistent record: passed through, as the reference does
