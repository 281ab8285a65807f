"""The C-ABI library loads here (no GPU) and exports every symbol include/*.h declares; argument validation that does
not need a device behaves like the reference (error code + message instead of an exception)."""
import ctypes as C
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ktransformers_amd import build
    return C.CDLL(build.build())


def declared_symbols():
    syms = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        syms |= set(re.findall(r"\b(ktx_[a-z0-9_]+)\s*\(", txt))
    return sorted(syms)


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for must in ("ktx_moe_create", "ktx_moe_forward", "ktx_moe_forward_ex", "ktx_moe_load_bf16", "ktx_moe_load_quantized",
                 "ktx_moe_destroy", "ktx_last_error", "ktx_gate_logits", "ktx_gate_select"):
        assert must in syms


@pytest.mark.parametrize("sym", declared_symbols())
def test_library_exports(lib, sym):
    assert hasattr(lib, sym), f"{sym} declared in include/ but not exported by libktx_hip.so"


def test_create_rejects_bad_config_without_touching_the_device(lib):
    from ktransformers_amd._native import _MoeConfig
    lib.ktx_last_error.restype = C.c_char_p
    h = C.c_void_p()
    bad_fmt = _MoeConfig(8, 2, 256, 256, 16, 99, 0, 0, 0, 8)
    assert lib.ktx_moe_create(C.byref(bad_fmt), C.byref(h)) != 0
    assert b"format" in lib.ktx_last_error()
    bad_k = _MoeConfig(8, 2, 200, 256, 16, 0, 0, 0, 0, 8)
    assert lib.ktx_moe_create(C.byref(bad_k), C.byref(h)) != 0
    assert b"multiples of 128" in lib.ktx_last_error()
    assert lib.ktx_moe_create(None, C.byref(h)) != 0


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ktransformers_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "from oracle" not in txt and "import oracle" not in txt and "ktx_oracle" not in txt.replace(
                    "oracle/ktx_oracle.c states the same rule", ""), f"{f} references the oracle"


def test_moe_handle_refuses_cpu_device():
    from ktransformers_amd._native import KtxError, MoEHandle
    with pytest.raises(KtxError):
        MoEHandle(8, 2, 256, 256, 16, device="cpu")
