"""CPU: the ctypes mirrors in ktransformers_amd/_native.py against the structs of include/*.h, as a C compiler lays them out —
the public headers compiled by gcc as plain C (the drop-in boundary is a C ABI: no C++-only constructs in the headers), one
sizeof / offsetof line per field, compared with ctypes' own layout.  A field added on one side only (round 5 added
ktx_linear_fusion.glu_in into what was tail padding) fails here, not as a mis-read argument on the GPU."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = [("ktx_moe_config", "_MoeConfig", "ktx_moe.h"), ("ktx_linear_config", "_LinearConfig", "ktx_linear.h"),
         ("ktx_linear_fusion", "_LinearFusion", "ktx_linear.h"), ("ktx_gate_config", "_GateConfig", "ktx_gate.h"),
         ("ktx_mla_config", "_MlaConfig", "ktx_mla.h"), ("ktx_gemm_args", "_GemmArgs", "ktx_gemm.h"),
         ("ktx_attn_decode_args", "_AttnDecodeArgs", "ktx_attn.h")]


@pytest.fixture(scope="module")
def native():
    try:
        from ktransformers_amd import _native
    except ImportError as e:          # the library is built by __graft_entry__.build(); without it there is nothing to mirror
        pytest.skip(f"libktx_hip.so not built: {e}")
    return _native


def test_ctypes_mirrors_match_the_headers(native, tmp_path):
    lines = ["#include <stddef.h>", "#include <stdio.h>"]
    lines += [f'#include "{h}"' for h in sorted({h for _, _, h in PAIRS})]
    lines.append("int main(void) {")
    for cname, pyname, _ in PAIRS:
        cls = getattr(native, pyname)
        lines.append(f'  printf("{cname} sizeof %zu\\n", sizeof({cname}));')
        for fname, _ftype in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "abi_probe.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "abi_probe"
    cc = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr      # a header that is not plain C, or a mirrored field the header does not have
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    got = {}
    for ln in out.splitlines():
        s, f, v = ln.split()
        got[(s, f)] = int(v)
    for cname, pyname, _ in PAIRS:
        cls = getattr(native, pyname)
        assert C.sizeof(cls) == got[(cname, "sizeof")], f"{pyname}: sizeof {C.sizeof(cls)} != {cname} {got[(cname, 'sizeof')]}"
        for fname, _ftype in cls._fields_:
            assert getattr(cls, fname).offset == got[(cname, fname)], f"{pyname}.{fname}: offset differs from {cname}"
    # and nothing the header declares is missing from the mirror — a member that fits into tail padding moves neither a size nor an
    # offset, so the members are counted in the header text itself
    import re
    for cname, pyname, header in PAIRS:
        text = open(os.path.join(ROOT, "include", header)).read()
        body = re.search(r"typedef struct " + cname + r" \{(.*?)\} " + cname + ";", text, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        body = re.sub(r"//[^\n]*", "", body)
        members = sum(decl.count(",") + 1 for decl in body.split(";") if decl.strip())
        assert members == len(getattr(native, pyname)._fields_), f"{cname}: {members} members in {header}, {pyname} mirrors {len(getattr(native, pyname)._fields_)}"
