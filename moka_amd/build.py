"""In-tree build of libmoka_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m moka_amd.build [--force] [--diag]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "moka_kernels.hip")
OUT = os.path.join(HERE, "libmoka_hip.so")
OUT_DIAG = os.path.join(HERE, "libmoka_hip_diag.so")       # -DMOKA_DIAGNOSTICS: moka_tune() launch-heuristic overrides (tools/ only)
INC = os.path.join(ROOT, "include")


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def needs_build(out: str = OUT) -> bool:
    if not os.path.exists(out):
        return True
    newest = max(os.path.getmtime(p) for p in (SRC, os.path.join(INC, "moka_hip.h")))
    return os.path.getmtime(out) < newest


def build(force: bool = False, verbose: bool = True, diag: bool = False) -> str:
    """diag: the diagnostics variant (same ABI + working moka_tune), loaded by the tools through MOKA_HIP_LIB."""
    out = OUT_DIAG if diag else OUT
    if not force and not needs_build(out):
        return out
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-value", "-munsafe-fp-atomics", "-I", INC, SRC, "-o", out + ".tmp"] + (["-DMOKA_DIAGNOSTICS"] if diag else [])
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed building libmoka_hip.so")
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, diag="--diag" in sys.argv))
