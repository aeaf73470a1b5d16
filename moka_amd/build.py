"""In-tree build of libmoka_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m moka_amd.build [--force] [--diag] [--unity]

The library is six translation units (moka_amd/csrc/k_*.hip: one per kernel family, moka_api.hip: the entry points and launch rules)
compiled in parallel and linked; a change to the launch rules rebuilds in seconds, a change to one kernel family rebuilds that family.
``--unity`` compiles csrc/moka_kernels.hip (all of them in one) instead.  The .so is git-ignored but travels to the GPU box with the
gpurun snapshot; the object files under csrc/_obj/ are a local cache.
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
UNITS = ["k_cross", "k_expand", "k_wgrad", "k_reduce", "k_misc", "moka_api"]
HEADERS = [os.path.join(CSRC, "moka_device.h"), os.path.join(CSRC, "moka_host.h"), os.path.join(ROOT, "include", "moka_hip.h")]
SRC = os.path.join(CSRC, "moka_kernels.hip")               # the unity form (tools/microbench/passlab.hip, tools/kernel_resources.sh)
OUT = os.path.join(HERE, "libmoka_hip.so")
OUT_DIAG = os.path.join(HERE, "libmoka_hip_diag.so")       # -DMOKA_DIAGNOSTICS: moka_tune() launch-heuristic overrides (tools/ only)
INC = os.path.join(ROOT, "include")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-function", "-munsafe-fp-atomics", "-I", INC, "-I", CSRC]


def sources():
    """Every file the library is built from (what a PMC traffic summary is stamped with: bench.kernel_source_sha256)."""
    return [os.path.join(CSRC, u + ".hip") for u in UNITS] + HEADERS


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def needs_build(out: str = OUT) -> bool:
    if not os.path.exists(out):
        return True
    return os.path.getmtime(out) < max(os.path.getmtime(p) for p in sources())


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed building libmoka_hip.so")


def build(force: bool = False, verbose: bool = True, diag: bool = False, unity: bool = False) -> str:
    """diag: the diagnostics variant (same ABI + working moka_tune), loaded by the tools through MOKA_HIP_LIB."""
    out = OUT_DIAG if diag else OUT
    if not force and not needs_build(out):
        return out
    cc, defs = _hipcc(), (["-DMOKA_DIAGNOSTICS"] if diag else [])
    if unity:
        _run([cc] + FLAGS + defs + ["-shared", SRC, "-o", out + ".tmp"], verbose)
        os.replace(out + ".tmp", out)
        return out
    objdir = os.path.join(CSRC, "_obj", "diag" if diag else "prod")
    os.makedirs(objdir, exist_ok=True)
    newest_hdr = max(os.path.getmtime(h) for h in HEADERS)
    jobs = []
    for u in UNITS:
        src, obj = os.path.join(CSRC, u + ".hip"), os.path.join(objdir, u + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_hdr):
            jobs.append([cc] + FLAGS + defs + ["-c", src, "-o", obj])
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(len(jobs) or 1, os.cpu_count() or 1))) as ex:
        list(ex.map(lambda c: _run(c, verbose), jobs))
    _run([cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(objdir, u + ".o") for u in UNITS] + ["-o", out + ".tmp"], verbose)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, diag="--diag" in sys.argv, unity="--unity" in sys.argv))
