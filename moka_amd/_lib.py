"""ctypes binding of libmoka_hip.so (the C ABI declared in include/moka_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``python -m moka_amd.build``.
There is NO fallback: if the shared object is missing, or the current device is not a
gfx950, every compute entry point raises -- a silent eager path would void the parity and
performance claims of this package.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# MOKA_HIP_LIB: alternative build of the same ABI (A/B measurements of kernel variants on one box)
LIB_PATH = os.environ.get("MOKA_HIP_LIB") or os.path.join(_HERE, "libmoka_hip.so")

MOKA_BF16 = 0
MOKA_MAX_SHADOW_BATCH = 16
MOKA_MAX_BATCH = 8        # problems of one moka_down_bwd_da_batch launch (include/moka_hip.h)
MOKA_F32 = 1
MOKA_MOD_NONE = 255
MOKA_MAX_MOD = 3


class MokaRoutingStruct(Structure):
    _fields_ = [("tok_mod", c_void_p), ("ktok", c_void_p), ("klen", c_void_p), ("kslot", c_void_p),
                ("B", c_int32), ("S", c_int32), ("Lk_max", c_int32), ("M", c_int32)]


class MokaOpts(ctypes.Structure):
    """moka_opts of include/moka_hip.h: per-call options (deterministic-mode workspace, how many launch chains run side by side, the
    device-resident part of the dropout seed).  ``struct_size`` is filled in: the library reads no field beyond it."""
    _fields_ = [("struct_size", ctypes.c_size_t), ("det_ws", c_void_p), ("det_bytes", ctypes.c_size_t), ("company", c_int), ("seed_dev", c_void_p)]

    def __init__(self, det_ws=None, det_bytes=0, company=0, seed_dev=None):
        super().__init__()
        self.struct_size = ctypes.sizeof(MokaOpts)
        self.det_ws, self.det_bytes, self.company, self.seed_dev = det_ws, int(det_bytes), int(company), seed_dev


class MokaError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes); must list EVERY symbol include/moka_hip.h declares
SYMBOLS = {
    "moka_version": (c_int, []),
    "moka_last_error": (c_char_p, []),
    "moka_device_check": (c_int, []),
    "moka_tune": (c_int, [c_char_p, c_int]),
    "moka_diagnostics": (c_int, []),
    "moka_rank_pad": (c_int, [c_int]),
    "moka_tok_pad": (c_int, [c_int]),
    "moka_ksplit": (c_int, [c_int, c_int, c_int]),
    "moka_ksplit_group": (c_int, [c_int, c_int, c_int, c_int]),
    "moka_ksplit_bwd": (c_int, [c_int, c_int, c_int]),
    "moka_up_bwd_passes": (c_int, [c_int, c_int]),
    # x, A[], tok_mod, part, T, d_in, r, M, s_in, dropout_p, seed, dtype, opts, stream
    "moka_down_fwd": (c_int, [c_void_p, POINTER(c_void_p), c_void_p, c_void_p,
                              c_int, c_int, c_int, c_int, c_float, c_float, ctypes.c_ulonglong, c_int, POINTER(MokaOpts), c_void_p]),
    # part, ks, rt, s_out[], Bw, d_out, A[], d_in, h, hp, hp_tok, hp_kmj, BwT, AT, r, w, c, stream
    "moka_cross_fwd": (c_int, [c_void_p, c_int, POINTER(MokaRoutingStruct), POINTER(c_float), c_void_p, c_int,
                               POINTER(c_void_p), c_int,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p]),
    # hp_tok, Bw, tok_mod, y, T, r, d_out, dtype, stream
    "moka_up_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "moka_up_fwd_fused_ok": (c_int, [c_int, c_int]),
    "moka_up_fwd_fused_pays": (c_int, [c_int, c_int, POINTER(c_int), c_int, c_int, c_int]),
    # part, ks, rt, s_out, Bw, y_inout, d_out, h, hp_kmj, r, w, inv_sqrt_dk, dtype, stream
    "moka_up_fwd_fused": (c_int, [c_void_p, c_int, POINTER(MokaRoutingStruct), POINTER(c_float), c_void_p, c_void_p,
                                  c_int, c_void_p, c_void_p, c_int, c_float, c_float, c_int, c_void_p]),
    # part[], ks, rt, s_out, Bw[], y_inout[], d_out[], h[], hp_kmj[], G, r, w, inv_sqrt_dk, dtype, stream
    "moka_up_fwd_fused_group": (c_int, [POINTER(c_void_p), c_int, POINTER(MokaRoutingStruct), POINTER(c_float), POINTER(c_void_p),
                                        POINTER(c_void_p), POINTER(c_int), POINTER(c_void_p), POINTER(c_void_p),
                                        c_int, c_int, c_float, c_float, c_int, c_void_p]),
    # Bw, d_out, A[], d_in, BwT, AT, r, M, stream
    "moka_weight_shadows": (c_int, [c_void_p, c_int, POINTER(c_void_p), c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    # Bw[], d_out[], A[], d_in, BwT[], AT[], G, r, M, stream
    "moka_weight_shadows_group": (c_int, [POINTER(c_void_p), POINTER(c_int), POINTER(c_void_p), c_int, POINTER(c_void_p), POINTER(c_void_p),
                                          c_int, c_int, c_int, c_void_p]),
    # gy, hp_kmj, BwT, tok_mod, s_out[], g_part, dB_acc, T, r, d_out, M, dtype, opts, stream
    "moka_up_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_float), c_void_p, c_void_p,
                            c_int, c_int, c_int, c_int, c_int, POINTER(MokaOpts), c_void_p]),
    # g_part, ks, h, rt, s_in, dh, dh_tok, dh_kmj, ws, r, w, c, stream
    "moka_cross_bwd": (c_int, [c_void_p, c_int, c_void_p, POINTER(MokaRoutingStruct), c_float,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p]),
    "moka_cross_ws_bytes": (ctypes.c_size_t, [c_int, c_int, c_int, c_int]),
    # dh_tok, dh_kmj, x, AT, tok_mod, dA_acc[], dx, T, d_in, r, M, dropout_p, seed, dtype, opts, stream
    "moka_down_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_void_p), c_void_p,
                              c_int, c_int, c_int, c_int, c_float, ctypes.c_ulonglong, c_int, POINTER(MokaOpts), c_void_p]),
    # ---- grouped entry points (host arrays of G pointers)
    # x, A[G*M], tok_mod, part[G], T, d_in, r, M, G, s_in, dropout_p, seeds[G], dtype, opts, stream
    "moka_down_fwd_group": (c_int, [c_void_p, POINTER(c_void_p), c_void_p, POINTER(c_void_p),
                                    c_int, c_int, c_int, c_int, c_int, c_float, c_float, POINTER(ctypes.c_ulonglong), c_int, POINTER(MokaOpts), c_void_p]),
    # part[G], ks, rt, s_out[], Bw[G], d_out[G], A[G*M], d_in, h[G], hp[G], hp_tok[G], hp_kmj[G], BwT[G], AT[G], G, r, w, c, stream
    "moka_cross_fwd_group": (c_int, [POINTER(c_void_p), c_int, POINTER(MokaRoutingStruct), POINTER(c_float),
                                     POINTER(c_void_p), POINTER(c_int), POINTER(c_void_p), c_int,
                                     POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                     POINTER(c_void_p), POINTER(c_void_p), c_int, c_int, c_float, c_float, c_void_p]),
    # hp_tok[G], Bw[G], tok_mod, y[G], T, r, d_out[G], G, dtype, stream
    "moka_up_fwd_group": (c_int, [POINTER(c_void_p), POINTER(c_void_p), c_void_p, POINTER(c_void_p),
                                  c_int, c_int, POINTER(c_int), c_int, c_int, c_void_p]),
    # gy[G], hp_kmj[G], BwT[G], tok_mod, s_out[], g_part[G], dB_acc[G], T, r, d_out[G], M, G, dtype, opts, stream
    "moka_up_bwd_group": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_void_p, POINTER(c_float),
                                  POINTER(c_void_p), POINTER(c_void_p), c_int, c_int, POINTER(c_int), c_int, c_int, c_int,
                                  POINTER(MokaOpts), c_void_p]),
    # g_part[G], ks, h[G], rt, s_in, dh[G], dh_tok[G], dh_kmj[G], ws[G], G, r, w, c, stream
    "moka_cross_bwd_group": (c_int, [POINTER(c_void_p), c_int, POINTER(c_void_p), POINTER(MokaRoutingStruct), c_float,
                                     POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                     c_int, c_int, c_float, c_float, c_void_p]),
    # dh_tok[G], dh_kmj[G], x, AT[G], tok_mod, dA_acc[G*M], dx, T, d_in, r, M, G, dropout_p, seeds[G], dtype, opts, stream
    "moka_down_bwd_group": (c_int, [POINTER(c_void_p), POINTER(c_void_p), c_void_p, POINTER(c_void_p), c_void_p,
                                    POINTER(c_void_p), c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                    POINTER(ctypes.c_ulonglong), c_int, POINTER(MokaOpts), c_void_p]),
    # dh_kmj[n], x[n], d_in[n], tok_mod, dA_acc[n*M], n, T, r, M, dropout_p, seeds[n], dtype, opts, stream
    "moka_down_bwd_da_batch": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int), c_void_p, POINTER(c_void_p), c_int, c_int, c_int,
                                       c_int, c_float, POINTER(ctypes.c_ulonglong), c_int, POINTER(MokaOpts), c_void_p]),
    # Bw[n], d_out[n], A[n*M], d_in[n], BwT[n], AT[n], n, r, M, stream
    "moka_weight_shadows_batch": (c_int, [POINTER(c_void_p), POINTER(c_int), POINTER(c_void_p), POINTER(c_int), POINTER(c_void_p), POINTER(c_void_p),
                                          c_int, c_int, c_int, c_void_p]),
    # gy[n], hp_kmj[n], d_out[n], tok_mod, dB_acc[n], n, T, r, M, dtype, opts, stream
    "moka_up_bwd_db_batch": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int), c_void_p, POINTER(c_void_p), c_int, c_int, c_int,
                                     c_int, c_int, POINTER(MokaOpts), c_void_p]),
    # dropout_p, seed, T, d_in, keep_out, stream
    "moka_dropout_mask": (c_int, [c_float, ctypes.c_ulonglong, c_int, c_int, c_void_p, c_void_p]),
    "moka_dropout_scale": (c_float, [c_float]),
    "moka_deterministic_ws_bytes": (ctypes.c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    # master, work_bf16, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, zero_grad, stream
    "moka_adamw_flat": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_size_t,
                                c_float, c_float, c_float, c_float, c_float, c_int, c_float, c_int, c_void_p]),
    "moka_adamw_coef": (None, [c_float, c_float, c_float, c_float, c_int, ctypes.POINTER(c_float)]),
    "moka_adamw_begin_dev": (c_int, [c_void_p, c_float, c_float, c_float, c_float, c_int, c_void_p]),
    "moka_adamw_flat_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_size_t,
                                    c_float, c_float, c_float, c_void_p, c_float, c_int, c_void_p]),
}


def load():
    """Load the shared library (once).  Raises MokaError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MokaError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). moka_amd has no CPU / eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    # MOKA_TUNE="key=value,key=value": launch-heuristic overrides (moka_tune) applied at load time -- lets bench.py and
    # the tests run unchanged under a different setting (diagnostics only; results never depend on it)
    for item in filter(None, os.environ.get("MOKA_TUNE", "").split(",")):
        key, _, val = item.partition("=")
        if lib.moka_tune(key.strip().encode(), int(val)) != 0:
            raise MokaError("MOKA_TUNE: " + lib.moka_last_error().decode("utf-8", "replace"))
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().moka_last_error().decode("utf-8", "replace")
        raise MokaError(f"{what or 'moka'} failed with code {rc}: {msg}")


def rank_pad(r: int) -> int:
    rp = load().moka_rank_pad(int(r))
    if rp < 0:
        raise ValueError(f"`r` should be an integer in 1..64 for the HIP path but the value passed is {r}")
    return rp


def ksplit(T: int, C: int, r: int, G: int = 1) -> int:
    """Split-K slices moka_down_fwd[_group] writes per projection (G = projections that share the input: a single projection at r <= 16
    gets 1024-column slices, groups 512-column ones)."""
    ks = load().moka_ksplit_group(int(T), int(C), int(r), int(G))
    if ks < 0:
        raise ValueError(f"unsupported shape for the HIP path: T={T} width={C} r={r} (width must be a multiple of 32)")
    return ks


def ksplit_bwd(T: int, C: int, r: int) -> int:
    ks = load().moka_ksplit_bwd(int(T), int(C), int(r))
    if ks < 0:
        raise ValueError(f"unsupported shape for the HIP path: T={T} width={C} r={r} (width must be a multiple of 32)")
    return ks


_PASSES = {}


def up_bwd_passes(r: int, dtype: int = 0) -> int:
    """1: moka_up_bwd takes g and dB out of one pass over gy; 2: dB is a pass of its own (it may then leave the dependency chain)."""
    key = (int(r), int(dtype))
    if key not in _PASSES:
        n = load().moka_up_bwd_passes(*key)
        if n < 0:
            raise ValueError(f"unsupported rank / storage type for the HIP path: r={r} dtype={dtype}")
        _PASSES[key] = n
    return _PASSES[key]


_FUSED = {}


def up_fwd_fused_ok(r: int, dtype: int = 0) -> bool:
    """True when moka_up_fwd_fused (the interaction inside the up-projection) exists for this rank / storage type."""
    key = (int(r), int(dtype))
    if key not in _FUSED:
        _FUSED[key] = load().moka_up_fwd_fused_ok(*key) == 1
    return _FUSED[key]


def up_fwd_fused_pays(T: int, ks: int, d_outs, r: int, dtype: int = 0) -> bool:
    """The library's advice: does the fused launch beat cross_fwd + up_fwd for this shape? (same bits either way)"""
    arr = (c_int * len(d_outs))(*[int(d) for d in d_outs])
    return load().moka_up_fwd_fused_pays(int(T), int(ks), arr, len(d_outs), int(r), int(dtype)) == 1


def tok_pad(T: int) -> int:
    return (int(T) + 31) // 32 * 32
