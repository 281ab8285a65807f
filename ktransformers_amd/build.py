"""Build the gfx950 shared library (hipcc cross-compiles without a GPU).

    python -m ktransformers_amd.build [--force]

Every csrc/*.hip is compiled to its own object (in parallel, only when it or a header it may include changed) and the
objects are linked into ktransformers_amd/lib/libktx_hip.so (git-ignored, travels with the gpurun snapshot).
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libktx_hip.so")

# -ffp-contract=off: the parity contract needs every fp32 op rounded exactly as the reference's AVX512 code does;
# FMAs appear only where the reference issues one (explicit fmaf()).
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value"]


def sources() -> list[str]:
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers() -> list[str]:
    return (glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc"))
            + glob.glob(os.path.join(HERE, "..", "include", "*.h")))


def _obj(src: str) -> str:
    return os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build() -> bool:
    return _stale(LIB, sources() + _headers())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = _headers()
    todo = [s for s in sources() if force or _stale(_obj(s), [s] + hdrs)]

    def compile_one(src: str) -> None:
        cmd = [hipcc, *CFLAGS, "-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 4))) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[_obj(s) for s in sources()], "-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
