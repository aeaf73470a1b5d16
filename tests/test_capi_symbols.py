"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/moka_hip.h
declares; host-side argument validation returns error codes without touching a device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from moka_amd import build, _lib
    build.build(verbose=False)          # hipcc cross-compiles without a GPU; no-op when up to date
    return _lib.load()


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "moka_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(moka_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from moka_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 11
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in moka_hip.h but not exported"
        assert name in _lib.SYMBOLS, f"{name} has no ctypes prototype"
    assert sorted(_lib.SYMBOLS) == declared


def test_version_and_rank_pad(lib):
    assert lib.moka_version() == 700
    assert [lib.moka_tok_pad(t) for t in (1, 32, 33)] == [32, 32, 64]
    assert [lib.moka_rank_pad(r) for r in (1, 4, 8, 16, 17, 32, 33, 64)] == [16, 16, 16, 16, 32, 32, 64, 64]
    assert lib.moka_rank_pad(0) < 0 and lib.moka_rank_pad(65) < 0


def test_ksplit_covers_width(lib):
    for T, C, r in [(8192, 4096, 16), (8192, 11008, 16), (2048, 4096, 16), (96, 64, 4), (4096, 5120, 64), (10, 64, 8)]:
        ks = lib.moka_ksplit(T, C, r)
        assert 1 <= ks <= 32
    # slices per projection: one per 512 columns; rank pads 32 (forward) / 64: whole 256-column chunks, as few as still give every CU three
    # (forward) / one (backward) workgroups of 128 tokens (no device here: 256 CUs assumed)
    assert lib.moka_ksplit(8192, 5120, 16) == 10 and lib.moka_ksplit_group(8192, 5120, 16, 1) == 10 and lib.moka_ksplit_group(8192, 5120, 16, 3) == 10
    assert lib.moka_ksplit(8190, 5120, 16) == 10 and lib.moka_ksplit(8192, 11008, 16) == 22 and lib.moka_ksplit_group(8192, 11008, 16, 2) == 22
    assert lib.moka_ksplit_group(8192, 4096, 16, 0) < 0 and lib.moka_ksplit_group(8192, 4096, 16, 4) < 0
    assert lib.moka_ksplit(8192, 5120, 64) == 10 and lib.moka_ksplit(8192, 13824, 64) == 11
    assert lib.moka_ksplit(128, 5120, 64) == 20 and lib.moka_ksplit(65536, 5120, 64) == 2
    assert lib.moka_ksplit_bwd(8192, 5120, 16) == 10 and lib.moka_ksplit_bwd(8192, 5120, 64) == 4 and lib.moka_ksplit_bwd(8192, 13824, 64) == 4
    assert lib.moka_ksplit_bwd(128, 5120, 64) == 20
    assert lib.moka_ksplit(8192, 4096, 32) == 8 and lib.moka_ksplit(8192, 11008, 32) == 11 and lib.moka_ksplit_bwd(8192, 11008, 32) == 22
    # ... and at least two chunks per slice while that still gives every CU a workgroup (round 6: 4096-token launches wrote 16 slices per 4096 columns)
    assert lib.moka_ksplit(4096, 4096, 32) == 8 and lib.moka_ksplit_group(4096, 4096, 32, 3) == 8 and lib.moka_ksplit(4096, 11008, 32) == 15
    assert lib.moka_ksplit(2048, 4096, 32) == 16 and lib.moka_ksplit(4096, 5120, 64) == 10 and lib.moka_ksplit(4096, 13824, 64) == 18
    # passes over gy of moka_up_bwd: one up to rank 32 in bf16 storage, dB on its own beyond and in fp32 storage
    assert [lib.moka_up_bwd_passes(r, 0) for r in (4, 16, 32, 33, 64)] == [1, 1, 1, 2, 2] and lib.moka_up_bwd_passes(16, 1) == 2
    assert lib.moka_up_bwd_passes(65, 0) < 0 and lib.moka_up_bwd_passes(16, 7) < 0
    assert lib.moka_ksplit(8192, 16, 16) < 0          # width below one MFMA K step
    assert lib.moka_ksplit(8192, 4100, 16) < 0        # not a multiple of 32
    assert lib.moka_ksplit(8192, 4096, 65) < 0


def test_argument_validation_sets_error_message(lib):
    dummy = ctypes.c_void_p(64)
    arr = (ctypes.c_void_p * 1)(64)
    # unsupported dtype
    rc = lib.moka_down_fwd(dummy, arr, dummy, dummy, 16, 64, 4, 1, 1.0, 0.0, 0, 7, None, None)     # 0 = MOKA_BF16, 1 = MOKA_F32
    assert rc == -2 and b"MOKA_BF16" in lib.moka_last_error()
    # width not a multiple of 32
    rc = lib.moka_down_fwd(dummy, arr, dummy, dummy, 16, 72, 4, 1, 1.0, 0.0, 0, 0, None, None)
    assert rc == -1 and b"multiple of 32" in lib.moka_last_error()
    # rank out of range
    rc = lib.moka_up_fwd(dummy, dummy, dummy, dummy, 16, 65, 64, 0, None)
    assert rc == -1 and b"rank" in lib.moka_last_error()
    # dropout probability out of range
    rc = lib.moka_down_fwd(dummy, arr, dummy, dummy, 16, 64, 4, 1, 1.0, 1.5, 7, 0, None, None)
    assert rc == -1 and b"dropout" in lib.moka_last_error()
    assert abs(lib.moka_dropout_scale(0.05) - 32768.0 / (32768 - 1638)) < 1e-6
    # null pointer
    rc = lib.moka_down_fwd(None, arr, dummy, dummy, 16, 64, 4, 1, 1.0, 0.0, 0, 0, None, None)
    assert rc == -1 and b"null" in lib.moka_last_error()


def test_the_product_library_keeps_no_mutable_state(lib):
    """VERDICT r02 item 8 / SURVEY 8(b): launch-heuristic overrides exist only in the diagnostics build, the deterministic-mode
    workspace travels with the call (moka_opts) and is validated before anything is launched."""
    from moka_amd import _lib, build
    assert lib.moka_diagnostics() == 0
    assert lib.moka_tune(b"xa_ng", 4) == -1 and b"diagnostics build" in lib.moka_last_error()
    assert not hasattr(lib, "moka_deterministic")
    diag = ctypes.CDLL(build.build(verbose=False, diag=True))
    diag.moka_tune.argtypes = [ctypes.c_char_p, ctypes.c_int]
    assert diag.moka_diagnostics() == 1 and diag.moka_tune(b"xa_ng", 4) == 0 and diag.moka_tune(b"xa_ng", 0) == 0
    assert diag.moka_tune(b"nope", 1) == -1
    # a deterministic call with a workspace that is too small / misaligned fails on the host, before any launch
    dummy = ctypes.c_void_p(64)
    arr = (ctypes.c_void_p * 3)(64, 64, 64)
    need = lib.moka_deterministic_ws_bytes(256, 64, 4, 1, 3)
    assert need == 2 * 3 * 64 * 4 * 4
    small = _lib.MokaOpts(4096, need - 16)
    rc = lib.moka_down_bwd(dummy, dummy, dummy, dummy, dummy, arr, None, 256, 64, 4, 3, 0.0, 0, 0, ctypes.byref(small), None)
    assert rc == -1 and b"too small" in lib.moka_last_error()
    odd = _lib.MokaOpts(4100, need)
    so = (ctypes.c_float * 3)(1.0, 1.0, 1.0)
    rc = lib.moka_up_bwd(dummy, dummy, dummy, dummy, so, None, dummy, 256, 4, 64, 3, 0, ctypes.byref(odd), None)
    assert rc == -1 and b"aligned" in lib.moka_last_error()
    # moka_opts carries its own size: a struct too short to hold `company` is refused, a caller built against a SHORTER
    # header (no seed_dev) is served with that field at its default, and a misaligned seed_dev fails before any launch
    assert _lib.MokaOpts().struct_size == ctypes.sizeof(_lib.MokaOpts) == 40
    short = _lib.MokaOpts(4096, need)
    short.struct_size = 8
    rc = lib.moka_up_bwd(dummy, dummy, dummy, dummy, so, None, dummy, 256, 4, 64, 3, 0, ctypes.byref(short), None)
    assert rc == -1 and b"struct_size" in lib.moka_last_error()
    bad_seed = _lib.MokaOpts(seed_dev=4100)
    rc = lib.moka_down_fwd(dummy, arr, dummy, dummy, 256, 64, 4, 3, 1.0, 0.1, 7, 0, ctypes.byref(bad_seed), None)
    assert rc == -1 and b"seed_dev" in lib.moka_last_error()


def test_no_gpu_means_loud_failure():
    import torch
    from moka_amd import _lib
    from moka_amd.functional import AdapterSpec, moka_linear
    if torch.cuda.is_available():
        pytest.skip("runs on the CPU-only container")
    x = torch.zeros(1, 4, 64, dtype=torch.bfloat16)
    with pytest.raises(_lib.MokaError):
        moka_linear(x, torch.zeros(64, 64, dtype=torch.bfloat16), None, torch.zeros(64, 4, dtype=torch.bfloat16),
                    [torch.zeros(4, 64, dtype=torch.bfloat16)], None, AdapterSpec(4, 1.0, [1.0], 0.0, 0.5))
