"""Host glue between torch tensors and the C ABI (include/moka_hip.h).

``ops`` are thin 1:1 wrappers of the exported entry points (pointer extraction, workspace
allocation, current-stream lookup); ``MokaLinearFn`` is the autograd node of one adapted
projection: frozen base GEMM on stock PyTorch-ROCm + the MokA adapter on the HIP kernels,
with the explicit backward the reference leaves to autograd.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, byref, c_float, c_void_p
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from .routing import MokaRouting


def _launch_stream(device) -> "torch.cuda.Stream":
    return torch.cuda.current_stream(device)


def _stream_ptr(device) -> c_void_p:
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_device(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise _lib.MokaError(
            f"moka_amd: `{name}` lives on {t.device}; the MokA adapter path only exists as HIP kernels for "
            "MI355X (gfx950) -- there is no CPU fallback (use oracle/ for CPU checking).")


def _require_bf16(t: torch.Tensor, name: str):
    if t.dtype != torch.bfloat16:
        raise _lib.MokaError(f"moka_amd: `{name}` is {t.dtype}; the HIP path stores activations and weights in bf16")


def _storage(tensors, names) -> int:
    """Storage dtype code of one adapted projection: all operands bf16 (the tuned path) or all fp32 (MOKA_F32: exact-fp32 FMA
    kernels, the reference's fp32 configuration -- adapters follow the base dtype, layer.py:124-132)."""
    dts = {t.dtype for t in tensors}
    if dts == {torch.bfloat16}:
        return _lib.MOKA_BF16
    if dts == {torch.float32}:
        return _lib.MOKA_F32
    desc = ", ".join(f"{n}: {t.dtype}" for n, t in zip(names, tensors))
    raise _lib.MokaError(f"moka_amd: operands must be all bf16 or all fp32 on the HIP path ({desc})")


def _ptrs(tensors: Sequence[torch.Tensor]):
    arr = (c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def _floats(vals: Sequence[float]):
    return (c_float * len(vals))(*[float(v) for v in vals])


# --------------------------------------------------------------------------------------
# 1:1 wrappers of the C entry points
# --------------------------------------------------------------------------------------
def down_fwd(x2: torch.Tensor, A: Sequence[torch.Tensor], rt: MokaRouting, r: int, s_in: float,
             dropout_p: float = 0.0, seed: int = 0, dtype: int = 0, seed_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x2 [T,d_in] bf16 -> part [ks,T,RP] fp32 (split-K partials of s_in * drop(x) A_mod^T)."""
    lib = _lib.load()
    T, d_in = x2.shape
    RP = _lib.rank_pad(r)
    ks = _lib.ksplit(T, d_in, r)
    part = torch.empty((ks, T, RP), dtype=torch.float32, device=x2.device)
    _lib.check(lib.moka_down_fwd(x2.data_ptr(), _ptrs(A), rt.tok_mod.data_ptr(), part.data_ptr(),
                                 T, d_in, r, len(A), float(s_in), float(dropout_p), int(seed), dtype,
                                 _det_opts(x2.device, T, d_in, r, 1, len(A), seed_dev, want_det=False), _stream_ptr(x2.device)), "moka_down_fwd")
    return part


class FwdState:
    """Rank-space results of the forward that the up-projection and the backward consume."""
    __slots__ = ("h", "hp", "hp_tok", "hp_kmj", "BwT", "AT")


def cross_fwd(part: torch.Tensor, rt: MokaRouting, r: int, s_out: Sequence[float], w: float, inv_sqrt_dk: float,
              Bw: Optional[torch.Tensor] = None, want_hp: bool = False, A: Optional[Sequence[torch.Tensor]] = None,
              want_tok: bool = True) -> FwdState:
    """want_tok=False: the token-major pack is not written (the up-projection is `up_fwd_fused_`, which computes it itself)."""
    lib = _lib.load()
    ks, T, RP = part.shape
    dev = part.device
    Tp = _lib.tok_pad(T)
    st = FwdState()
    st.h = torch.empty((T, RP), dtype=torch.float32, device=dev)
    st.hp = torch.empty((T, RP), dtype=torch.float32, device=dev) if want_hp else None
    st.hp_tok = torch.empty((Tp, 2 * RP), dtype=torch.bfloat16, device=dev) if want_tok else None
    st.hp_kmj = torch.empty((2, RP, Tp), dtype=torch.bfloat16, device=dev)
    st.BwT = torch.empty((RP, Bw.shape[0]), dtype=torch.bfloat16, device=dev) if Bw is not None else None
    st.AT = torch.empty((len(A), A[0].shape[1], RP), dtype=torch.bfloat16, device=dev) if A is not None else None
    _lib.check(lib.moka_cross_fwd(part.data_ptr(), ks, byref(rt.struct), _floats(s_out),
                                  None if Bw is None else Bw.data_ptr(), 0 if Bw is None else Bw.shape[0],
                                  None if A is None else _ptrs(A), 0 if A is None else A[0].shape[1],
                                  st.h.data_ptr(), None if st.hp is None else st.hp.data_ptr(),
                                  None if st.hp_tok is None else st.hp_tok.data_ptr(), st.hp_kmj.data_ptr(),
                                  None if st.BwT is None else st.BwT.data_ptr(),
                                  None if st.AT is None else st.AT.data_ptr(),
                                  r, float(w), float(inv_sqrt_dk), _stream_ptr(dev)), "moka_cross_fwd")
    return st


def up_fwd_fused_(y2: torch.Tensor, part: torch.Tensor, Bw: torch.Tensor, rt: MokaRouting, r: int, s_out: Sequence[float],
                  w: float, inv_sqrt_dk: float, want_state: bool = False) -> Optional[FwdState]:
    """In place: y2 [T,d_out] bf16 += (s_out[mod] hp) Bw^T with hp computed from the split-K slices `part` inside the kernel
    (the bits of cross_fwd + up_fwd_; bf16 storage).  want_state: the launch also writes what the backward reads from
    the rank space (h, hp_kmj); returned as a FwdState whose BwT / AT are filled by `weight_shadows`."""
    lib = _lib.load()
    ks, T, RP = part.shape
    st = None
    if want_state:
        st = FwdState()
        st.h = torch.empty((T, RP), dtype=torch.float32, device=part.device)
        st.hp_kmj = torch.empty((2, RP, _lib.tok_pad(T)), dtype=torch.bfloat16, device=part.device)
        st.hp = st.hp_tok = st.BwT = st.AT = None
    _lib.check(lib.moka_up_fwd_fused(part.data_ptr(), ks, byref(rt.struct), _floats(s_out), Bw.data_ptr(), y2.data_ptr(),
                                     y2.shape[1], None if st is None else st.h.data_ptr(), None if st is None else st.hp_kmj.data_ptr(),
                                     r, float(w), float(inv_sqrt_dk), _lib.MOKA_BF16, _stream_ptr(y2.device)),
               "moka_up_fwd_fused")
    return st


def weight_shadows(Bw: Optional[torch.Tensor], A: Optional[Sequence[torch.Tensor]], r: int) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """(BwT [RP, d_out], AT [M, d_in, RP]) bf16: the transposed, zero-padded weight copies the backward kernels read -- functions of
    the weights alone (moka_weight_shadows), so they need not sit on the forward's dependency chain."""
    lib = _lib.load()
    RP = _lib.rank_pad(r)
    ref = Bw if Bw is not None else A[0]
    BwT = torch.empty((RP, Bw.shape[0]), dtype=torch.bfloat16, device=ref.device) if Bw is not None else None
    AT = torch.empty((len(A), A[0].shape[1], RP), dtype=torch.bfloat16, device=ref.device) if A is not None else None
    _lib.check(lib.moka_weight_shadows(None if Bw is None else Bw.data_ptr(), 0 if Bw is None else Bw.shape[0],
                                       None if A is None else _ptrs(A), 0 if A is None else A[0].shape[1],
                                       None if BwT is None else BwT.data_ptr(), None if AT is None else AT.data_ptr(),
                                       r, len(A) if A is not None else 1, _stream_ptr(ref.device)), "moka_weight_shadows")
    return BwT, AT


def up_fwd_(y2: torch.Tensor, hp_tok: torch.Tensor, Bw: torch.Tensor, rt: MokaRouting, r: int, dtype: int = 0):
    """In place: y2 [T,d_out] bf16 += (s_out[mod] hp) Bw^T (the scale is already inside hp_tok)."""
    lib = _lib.load()
    T, d_out = y2.shape
    _lib.check(lib.moka_up_fwd(hp_tok.data_ptr(), Bw.data_ptr(), rt.tok_mod.data_ptr(), y2.data_ptr(),
                               T, r, d_out, dtype, _stream_ptr(y2.device)), "moka_up_fwd")
    return y2


def up_bwd(gy2: torch.Tensor, hp_kmj: Optional[torch.Tensor], BwT: Optional[torch.Tensor], rt: MokaRouting, r: int,
           s_out: Sequence[float], dB_acc: Optional[torch.Tensor], dtype: int = 0, want_g: bool = True) -> Optional[torch.Tensor]:
    """gy2 [T,d_out] bf16 -> g_part [ks,T,RP] (want_g=False: skip); dB_acc [d_out,r] fp32 += (may be None: skip)."""
    lib = _lib.load()
    T, d_out = gy2.shape
    RP = _lib.rank_pad(r)
    ks = _lib.ksplit_bwd(T, d_out, r)
    g_part = torch.empty((ks, T, RP), dtype=torch.float32, device=gy2.device) if want_g else None
    _lib.check(lib.moka_up_bwd(gy2.data_ptr(), None if hp_kmj is None else hp_kmj.data_ptr(), None if BwT is None else BwT.data_ptr(),
                               rt.tok_mod.data_ptr(), _floats(s_out), None if g_part is None else g_part.data_ptr(),
                               None if dB_acc is None else dB_acc.data_ptr(),
                               T, r, d_out, len(s_out), dtype, _det_opts(gy2.device, T, d_out, r, 1, len(s_out)) if dB_acc is not None else None,
                               _stream_ptr(gy2.device)), "moka_up_bwd")
    return g_part


class BwdState:
    __slots__ = ("dh", "dh_tok", "dh_kmj")


def cross_bwd(g_part: torch.Tensor, h: torch.Tensor, rt: MokaRouting, r: int, s_in: float, w: float, inv_sqrt_dk: float,
              want_dh: bool = False) -> BwdState:
    lib = _lib.load()
    ks, T, RP = g_part.shape
    dev = g_part.device
    Tp = _lib.tok_pad(T)
    st = BwdState()
    st.dh = torch.empty((T, RP), dtype=torch.float32, device=dev) if want_dh else None
    st.dh_tok = torch.empty((Tp, 2 * RP), dtype=torch.bfloat16, device=dev)
    st.dh_kmj = torch.empty((rt.M, 2, RP, Tp), dtype=torch.bfloat16, device=dev)
    _lib.check(lib.moka_cross_bwd(g_part.data_ptr(), ks, h.data_ptr(), byref(rt.struct), float(s_in),
                                  None if st.dh is None else st.dh.data_ptr(), st.dh_tok.data_ptr(), st.dh_kmj.data_ptr(),
                                  rt.cross_ws(r).data_ptr(), r, float(w), float(inv_sqrt_dk), _stream_ptr(dev)), "moka_cross_bwd")
    return st


def down_bwd_(bst: BwdState, x2: torch.Tensor, AT: Optional[torch.Tensor], rt: MokaRouting, r: int,
              dA_acc: Optional[Sequence[torch.Tensor]], dx2: Optional[torch.Tensor],
              dropout_p: float = 0.0, seed: int = 0, dtype: int = 0, seed_dev: Optional[torch.Tensor] = None):
    """dA_acc[m] [r,d_in] fp32 += ; dx2 [T,d_in] bf16 += (either may be None)."""
    lib = _lib.load()
    T, d_in = x2.shape
    _lib.check(lib.moka_down_bwd(bst.dh_tok.data_ptr(), None if bst.dh_kmj is None else bst.dh_kmj.data_ptr(), x2.data_ptr(),
                                 None if AT is None else AT.data_ptr(), rt.tok_mod.data_ptr(),
                                 None if dA_acc is None else _ptrs(dA_acc), None if dx2 is None else dx2.data_ptr(),
                                 T, d_in, r, rt.M, float(dropout_p), int(seed), dtype,
                                 _det_opts(x2.device, T, d_in, r, 1, rt.M, seed_dev, want_det=dA_acc is not None),
                                 _stream_ptr(x2.device)), "moka_down_bwd")


# --------------------------------------------------------------------------------------
# grouped entry points: G projections fed by the same x (q/k/v, gate/up)
# --------------------------------------------------------------------------------------
def _ints(vals: Sequence[int]):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def _u64s(vals: Sequence[int]):
    return (ctypes.c_ulonglong * len(vals))(*[int(v) for v in vals])


def weight_shadows_batch_(Bws: Sequence[torch.Tensor], As: Sequence[Sequence[torch.Tensor]], r: int,
                          BwTs: Sequence[torch.Tensor], ATs: Sequence[torch.Tensor]) -> None:
    """Rewrite the shadows (BwT [RP, d_out], AT [M, d_in, RP]) of n projections of ANY widths in place, one launch per MOKA_MAX_SHADOW_BATCH
    of them: what a trainer that keeps persistent shadows runs behind an optimizer step (moka_weight_shadows_batch)."""
    lib = _lib.load()
    M = len(As[0])
    for i in range(0, len(Bws), _lib.MOKA_MAX_SHADOW_BATCH):
        j = min(len(Bws), i + _lib.MOKA_MAX_SHADOW_BATCH)
        _lib.check(lib.moka_weight_shadows_batch(_ptrs(Bws[i:j]), _ints([b.shape[0] for b in Bws[i:j]]), _ptrs([a for Ag in As[i:j] for a in Ag]),
                                                 _ints([Ag[0].shape[1] for Ag in As[i:j]]), _ptrs(BwTs[i:j]), _ptrs(ATs[i:j]), j - i, r, M,
                                                 _stream_ptr(Bws[0].device)), "moka_weight_shadows_batch")


def _optptrs(tensors: Sequence[Optional[torch.Tensor]]):
    return (c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def down_fwd_group(x2: torch.Tensor, A: Sequence[Sequence[torch.Tensor]], rt: MokaRouting, r: int, s_in: float,
                   dropout_p: float = 0.0, seeds: Optional[Sequence[int]] = None, seed_dev: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """One read of x2 [T,d_in] for the G down-projections A[g][m]; returns the G split-K partial buffers."""
    lib = _lib.load()
    T, d_in = x2.shape
    G, M = len(A), len(A[0])
    RP = _lib.rank_pad(r)
    ks = _lib.ksplit(T, d_in, r, G)
    parts = [torch.empty((ks, T, RP), dtype=torch.float32, device=x2.device) for _ in range(G)]
    flat = [a for Ag in A for a in Ag]
    _lib.check(lib.moka_down_fwd_group(x2.data_ptr(), _ptrs(flat), rt.tok_mod.data_ptr(), _ptrs(parts),
                                       T, d_in, r, M, G, float(s_in), float(dropout_p),
                                       _u64s(seeds if seeds is not None else [0] * G), _lib.MOKA_BF16,
                                       _det_opts(x2.device, T, d_in, r, G, M, seed_dev, want_det=False), _stream_ptr(x2.device)), "moka_down_fwd_group")
    return parts


def cross_fwd_group(parts: Sequence[torch.Tensor], rt: MokaRouting, r: int, s_out: Sequence[float], w: float, inv_sqrt_dk: float,
                    Bw: Sequence[torch.Tensor], A: Optional[Sequence[Sequence[torch.Tensor]]] = None, want_tok: bool = True) -> List[FwdState]:
    lib = _lib.load()
    G = len(parts)
    ks, T, RP = parts[0].shape
    dev = parts[0].device
    Tp = _lib.tok_pad(T)
    sts = []
    for g in range(G):
        st = FwdState()
        st.h = torch.empty((T, RP), dtype=torch.float32, device=dev)
        st.hp = None
        st.hp_tok = torch.empty((Tp, 2 * RP), dtype=torch.bfloat16, device=dev) if want_tok else None
        st.hp_kmj = torch.empty((2, RP, Tp), dtype=torch.bfloat16, device=dev)
        st.BwT = torch.empty((RP, Bw[g].shape[0]), dtype=torch.bfloat16, device=dev)
        st.AT = torch.empty((len(A[g]), A[g][0].shape[1], RP), dtype=torch.bfloat16, device=dev) if A is not None else None
        sts.append(st)
    flatA = None if A is None else _ptrs([a for Ag in A for a in Ag])
    _lib.check(lib.moka_cross_fwd_group(_ptrs(parts), ks, byref(rt.struct), _floats(s_out),
                                        _ptrs(Bw), _ints([b.shape[0] for b in Bw]), flatA, 0 if A is None else A[0][0].shape[1],
                                        _ptrs([st.h for st in sts]), None, _ptrs([st.hp_tok for st in sts]) if want_tok else None,
                                        _ptrs([st.hp_kmj for st in sts]), _ptrs([st.BwT for st in sts]),
                                        None if A is None else _ptrs([st.AT for st in sts]),
                                        G, r, float(w), float(inv_sqrt_dk), _stream_ptr(dev)), "moka_cross_fwd_group")
    return sts


def up_fwd_group_(ys: Sequence[torch.Tensor], hp_toks: Sequence[torch.Tensor], Bw: Sequence[torch.Tensor], rt: MokaRouting, r: int):
    lib = _lib.load()
    T = ys[0].shape[0]
    _lib.check(lib.moka_up_fwd_group(_ptrs(hp_toks), _ptrs(Bw), rt.tok_mod.data_ptr(), _ptrs(ys), T, r,
                                     _ints([y.shape[1] for y in ys]), len(ys), _lib.MOKA_BF16, _stream_ptr(ys[0].device)),
               "moka_up_fwd_group")


def up_fwd_fused_group_(ys: Sequence[torch.Tensor], parts: Sequence[torch.Tensor], Bw: Sequence[torch.Tensor], rt: MokaRouting, r: int,
                        s_out: Sequence[float], w: float, inv_sqrt_dk: float, want_state: bool = False) -> Optional[List[FwdState]]:
    lib = _lib.load()
    ks, T, RP = parts[0].shape
    sts = None
    if want_state:
        sts = []
        for _ in ys:
            st = FwdState()
            st.h = torch.empty((T, RP), dtype=torch.float32, device=parts[0].device)
            st.hp_kmj = torch.empty((2, RP, _lib.tok_pad(T)), dtype=torch.bfloat16, device=parts[0].device)
            st.hp = st.hp_tok = st.BwT = st.AT = None
            sts.append(st)
    _lib.check(lib.moka_up_fwd_fused_group(_ptrs(parts), ks, byref(rt.struct), _floats(s_out), _ptrs(Bw), _ptrs(ys),
                                           _ints([y.shape[1] for y in ys]),
                                           None if sts is None else _ptrs([st.h for st in sts]),
                                           None if sts is None else _ptrs([st.hp_kmj for st in sts]),
                                           len(ys), r, float(w), float(inv_sqrt_dk),
                                           _lib.MOKA_BF16, _stream_ptr(ys[0].device)), "moka_up_fwd_fused_group")
    return sts


def weight_shadows_group(Bws: Sequence[torch.Tensor], As: Optional[Sequence[Sequence[torch.Tensor]]], r: int):
    """Per projection (BwT, AT) of a group in one launch (AT only when `As` is given)."""
    lib = _lib.load()
    RP = _lib.rank_pad(r)
    dev = Bws[0].device
    G = len(Bws)
    BwTs = [torch.empty((RP, b.shape[0]), dtype=torch.bfloat16, device=dev) for b in Bws]
    ATs = [torch.empty((len(Ag), Ag[0].shape[1], RP), dtype=torch.bfloat16, device=dev) for Ag in As] if As is not None else None
    M = len(As[0]) if As is not None else 1
    _lib.check(lib.moka_weight_shadows_group(_ptrs(Bws), _ints([b.shape[0] for b in Bws]),
                                             None if As is None else _ptrs([a for Ag in As for a in Ag]), 0 if As is None else As[0][0].shape[1],
                                             _ptrs(BwTs), None if ATs is None else _ptrs(ATs), G, r, M, _stream_ptr(dev)),
               "moka_weight_shadows_group")
    return BwTs, ATs


def up_bwd_group(gys: Sequence[torch.Tensor], hp_kmjs: Sequence[torch.Tensor], BwTs: Sequence[torch.Tensor], rt: MokaRouting, r: int,
                 s_out: Sequence[float], dB_accs: Optional[Sequence[torch.Tensor]], want_g: bool = True) -> Optional[List[torch.Tensor]]:
    lib = _lib.load()
    G = len(gys)
    T = gys[0].shape[0]
    RP = _lib.rank_pad(r)
    d_outs = [g_.shape[1] for g_ in gys]
    ks = _lib.ksplit_bwd(T, max(d_outs), r)
    g_parts = [torch.empty((ks, T, RP), dtype=torch.float32, device=gys[0].device) for _ in range(G)] if want_g else None
    _lib.check(lib.moka_up_bwd_group(_ptrs(gys), _ptrs(hp_kmjs), _ptrs(BwTs), rt.tok_mod.data_ptr(), _floats(s_out),
                                     None if g_parts is None else _ptrs(g_parts), None if dB_accs is None else _ptrs(dB_accs),
                                     T, r, _ints(d_outs), len(s_out), G, _lib.MOKA_BF16,
                                     _det_opts(gys[0].device, T, max(d_outs), r, G, len(s_out)) if dB_accs is not None else None,
                                     _stream_ptr(gys[0].device)),
               "moka_up_bwd_group")
    return g_parts


def cross_bwd_group(g_parts: Sequence[torch.Tensor], hs: Sequence[torch.Tensor], rt: MokaRouting, r: int, s_in: float,
                    w: float, inv_sqrt_dk: float) -> List[BwdState]:
    lib = _lib.load()
    G = len(g_parts)
    ks, T, RP = g_parts[0].shape
    dev = g_parts[0].device
    Tp = _lib.tok_pad(T)
    sts = []
    for g in range(G):
        st = BwdState()
        st.dh = None
        st.dh_tok = torch.empty((Tp, 2 * RP), dtype=torch.bfloat16, device=dev)
        st.dh_kmj = torch.empty((rt.M, 2, RP, Tp), dtype=torch.bfloat16, device=dev)
        sts.append(st)
    _lib.check(lib.moka_cross_bwd_group(_ptrs(g_parts), ks, _ptrs(hs), byref(rt.struct), float(s_in), None,
                                        _ptrs([st.dh_tok for st in sts]), _ptrs([st.dh_kmj for st in sts]),
                                        _ptrs([rt.cross_ws(r, g) for g in range(G)]), G, r, float(w), float(inv_sqrt_dk),
                                        _stream_ptr(dev)), "moka_cross_bwd_group")
    return sts


def down_bwd_group_(bsts: Sequence[BwdState], x2: torch.Tensor, ATs: Optional[Sequence[torch.Tensor]], rt: MokaRouting, r: int,
                    dA_accs: Optional[Sequence[Sequence[torch.Tensor]]], dx2: Optional[torch.Tensor],
                    dropout_p: float = 0.0, seeds: Optional[Sequence[int]] = None, seed_dev: Optional[torch.Tensor] = None):
    """dA_accs[g][m] += (one read of x2 for all G); dx2 += sum_g dh_g A_g (one read-modify-write)."""
    lib = _lib.load()
    T, d_in = x2.shape
    G = len(bsts)
    _lib.check(lib.moka_down_bwd_group(_ptrs([b.dh_tok for b in bsts]), _ptrs([b.dh_kmj for b in bsts]), x2.data_ptr(),
                                       None if ATs is None else _ptrs(ATs), rt.tok_mod.data_ptr(),
                                       None if dA_accs is None else _ptrs([a for Ag in dA_accs for a in Ag]),
                                       None if dx2 is None else dx2.data_ptr(), T, d_in, r, rt.M, G, float(dropout_p),
                                       _u64s(seeds if seeds is not None else [0] * G), _lib.MOKA_BF16,
                                       _det_opts(x2.device, T, d_in, r, G, rt.M, seed_dev, want_det=dA_accs is not None), _stream_ptr(x2.device)),
               "moka_down_bwd_group")


def down_bwd_da_batch_(dh_kmjs: Sequence[torch.Tensor], xs: Sequence[torch.Tensor], rt: MokaRouting, r: int,
                       dA_accs: Sequence[Sequence[torch.Tensor]], dropout_p: float = 0.0, seeds: Optional[Sequence[int]] = None,
                       seed_dev: Optional[torch.Tensor] = None):
    """dA_accs[i][m] += for n independent projections of one token set (their own x [T, d_in_i] and operand pack) in ONE launch: what a
    trainer defers per decoder layer (moka_down_bwd_da_batch; with the deterministic mode on: one launch per projection)."""
    lib = _lib.load()
    n, T = len(xs), xs[0].shape[0]
    d_ins = [int(x.shape[1]) for x in xs]
    _lib.check(lib.moka_down_bwd_da_batch(_ptrs(dh_kmjs), _ptrs(xs), (ctypes.c_int * n)(*d_ins), rt.tok_mod.data_ptr(),
                                          _ptrs([a for Ai in dA_accs for a in Ai]), n, T, r, rt.M, float(dropout_p),
                                          _u64s(seeds if seeds is not None else [0] * n), _lib.MOKA_BF16,
                                          _det_opts(xs[0].device, T, max(d_ins), r, 1, rt.M, seed_dev), _stream_ptr(xs[0].device)),
               "moka_down_bwd_da_batch")


def up_bwd_db_batch_(gys: Sequence[torch.Tensor], hp_kmjs: Sequence[torch.Tensor], rt: MokaRouting, r: int, dB_accs: Sequence[torch.Tensor]):
    """dB_accs[i] [d_out_i, r] += for n independent projections of one token set in ONE launch (moka_up_bwd_db_batch): the deferred dB of a
    decoder layer at the ranks where dB is a pass of its own."""
    lib = _lib.load()
    n, T = len(gys), gys[0].shape[0]
    d_outs = [int(g.shape[1]) for g in gys]
    _lib.check(lib.moka_up_bwd_db_batch(_ptrs(gys), _ptrs(hp_kmjs), (ctypes.c_int * n)(*d_outs), rt.tok_mod.data_ptr(), _ptrs(dB_accs), n, T, r, rt.M,
                                        _lib.MOKA_BF16, _det_opts(gys[0].device, T, max(d_outs), r, 1, rt.M), _stream_ptr(gys[0].device)),
               "moka_up_bwd_db_batch")


def dropout_mask(dropout_p: float, seed: int, T: int, d_in: int, device) -> torch.Tensor:
    """The keep mask (uint8 [T,d_in]) the kernels derive from (dropout_p, seed) -- for checkers."""
    lib = _lib.load()
    out = torch.empty((T, d_in), dtype=torch.uint8, device=device)
    _lib.check(lib.moka_dropout_mask(float(dropout_p), int(seed), T, d_in, out.data_ptr(), _stream_ptr(device)), "moka_dropout_mask")
    return out


def _token_scale(rt: MokaRouting, s_out: Sequence[float], device) -> torch.Tensor:
    """s_out[mod(t)] per token (0 for tokens of no modality), fp32 [T] -- the scale the bf16 packs carry built in."""
    lut = torch.zeros(256, dtype=torch.float32, device=device)
    lut[:len(s_out)] = torch.tensor([float(v) for v in s_out], dtype=torch.float32, device=device)
    return lut[rt.tok_mod[:rt.T].long()]


# Deterministic mode is HOST-side state of this Python module: the library itself keeps none (the workspace travels with every
# backward call as `moka_opts`).  One workspace per (device, stream): calls that run concurrently on two streams never share one.
_DET_ON = {}            # device -> True
_DET_WS = {}            # (device, stream handle) -> uint8 workspace tensor


def set_deterministic(enabled: bool, T: int = 0, C_max: int = 0, r: int = 16, G: int = 3, M: int = 3, device=None) -> None:
    """Bitwise-reproducible weight gradients: the dA_m / dB kernels write one partial tile per token run into a workspace and a
    second launch adds the runs in order, instead of fp32 atomics.  The setting is per device and lives here, not in the library;
    the workspace of a call is sized for THAT call (``moka_deterministic_ws_bytes``), allocated on first use per stream, kept
    alive and handed to the backward entry points through ``moka_opts``.  The size arguments are optional pre-allocation hints
    for the current stream."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    if not enabled:
        _DET_ON.pop(dev, None)
        for k in [k for k in _DET_WS if k[0] == dev]:
            del _DET_WS[k]
        return
    _DET_ON[dev] = True
    if T and C_max:
        if _det_opts(dev, int(T), int(C_max), int(r), int(G), int(M)) is None:
            raise ValueError("set_deterministic: bad T / C_max / r / G / M")


def _det_opts(device, T: int, C_max: int, r: int, G: int, M: int, seed_dev: Optional[torch.Tensor] = None, want_det: bool = True):
    """moka_opts of one call: the deterministic-mode workspace (set_deterministic on for the device and the call produces weight gradients)
    and / or the device-resident part of the dropout seed (``seed_dev``: an int64 tensor of one element); None when neither applies."""
    sd = None if seed_dev is None else seed_dev.data_ptr()
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    if not want_det or not _DET_ON.get(dev):
        return None if sd is None else ctypes.byref(_lib.MokaOpts(seed_dev=sd))
    n = int(_lib.load().moka_deterministic_ws_bytes(int(T), int(C_max), int(r), int(G), int(M)))
    if n == 0:
        return None if sd is None else ctypes.byref(_lib.MokaOpts(seed_dev=sd))
    key = (dev, _launch_stream(dev).cuda_stream)
    ws = _DET_WS.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(n, dtype=torch.uint8, device=dev)
        _DET_WS[key] = ws
    return ctypes.byref(_lib.MokaOpts(ws.data_ptr(), ws.numel(), seed_dev=sd))


def effective_seed(seed: int, epoch: int) -> int:
    """The seed whose mask a call with (seed, *seed_dev == epoch) uses: 32-bit halves, low ^ low and high + high (include/moka_hip.h,
    moka_opts.seed_dev) -- what a checker hands to ``dropout_mask`` to replay a captured step's masks."""
    seed, epoch = int(seed) & (2 ** 64 - 1), int(epoch) & (2 ** 64 - 1)
    lo = (seed ^ epoch) & 0xffffffff
    hi = ((seed >> 32) + (epoch >> 32)) & 0xffffffff
    return (hi << 32) | lo


def draw_seed() -> int:
    """Per-call dropout seed from torch's CPU generator (torch.manual_seed controls it; activation
    checkpointing restores that generator before the re-forward, so the mask replays)."""
    a, b = torch.randint(0, 2 ** 31 - 1, (2,)).tolist()
    return (a << 31) | b


def _grad_accumulators(shapes: Sequence[Tuple[int, int]], device) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """One zeroed fp32 buffer carved into the [rows, cols] accumulators the weight-gradient kernels add into
    (one memset and, afterwards, one cast for all of them instead of one tiny kernel per tensor)."""
    sizes = [a * b for a, b in shapes]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=device)
    views, off = [], 0
    for (a, b), n in zip(shapes, sizes):
        views.append(flat[off:off + n].view(a, b))
        off += n
    return flat, views


def _check_sinks(dB_acc, dA_acc, d_out: int, r: int, d_in: int, M: int, device) -> None:
    ts = ([dB_acc] if dB_acc is not None else []) + (list(dA_acc) if dA_acc is not None else [])
    if dA_acc is not None and len(dA_acc) != M:
        raise ValueError(f"moka_amd: {len(dA_acc)} gradient sinks for {M} lora_A matrices")
    for t in ts:
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != device:
            raise TypeError("moka_amd: gradient sinks must be contiguous fp32 tensors on the device of the activations")
    if dB_acc is not None and tuple(dB_acc.shape) != (d_out, r):
        raise ValueError(f"moka_amd: lora_B gradient sink is {tuple(dB_acc.shape)}, expected {(d_out, r)}")
    if dA_acc is not None:
        for t in dA_acc:
            if tuple(t.shape) != (r, d_in):
                raise ValueError(f"moka_amd: lora_A gradient sink is {tuple(t.shape)}, expected {(r, d_in)}")


def _split_like(flat: torch.Tensor, shapes: Sequence[Tuple[int, int]]) -> List[torch.Tensor]:
    out, off = [], 0
    for a, b in shapes:
        out.append(flat[off:off + a * b].view(a, b))
        off += a * b
    return out


# The forward of the autograd nodes: True (default) = moka_down_fwd -> moka_up_fwd_fused (+ moka_weight_shadows) where the fused launch
# exists (bf16 storage) and pays for the shape (moka_up_fwd_fused_pays: not at rank pad 64); False = the three-launch path moka_down_fwd -> moka_cross_fwd -> moka_up_fwd.  Same bits either way
# (tests/test_gpu_fused.py); a module-level switch for A/B runs, not a tuning knob.
FUSE_FORWARD = True

# --------------------------------------------------------------------------------------
# autograd node of one adapted projection
# --------------------------------------------------------------------------------------
class AdapterSpec:
    """Static description of one adapted projection (what varies between AVT and VT)."""

    __slots__ = ("r", "s_in", "s_out", "w", "inv_sqrt_dk", "dropout_p", "seed", "sinks", "defer", "shadows", "seed_dev")

    def __init__(self, r: int, s_in: float, s_out: Sequence[float], w: float, inv_sqrt_dk: float,
                 dropout_p: float = 0.0, seed: Optional[int] = None, sinks=None, defer=None, shadows=None, seed_dev=None):
        self.r, self.s_in, self.s_out, self.w, self.inv_sqrt_dk = int(r), float(s_in), [float(s) for s in s_out], float(w), float(inv_sqrt_dk)
        self.dropout_p = float(dropout_p)
        self.seed = (draw_seed() if seed is None else int(seed)) if self.dropout_p > 0.0 else 0
        # sinks = (dB_acc [d_out, r] fp32, [dA_acc_m [r, d_in] fp32 ...]): views of a flat gradient buffer
        # (moka_amd.parallel.attach) the weight-gradient kernels accumulate into DIRECTLY; the autograd node then returns no
        # gradient for lora_B / lora_A (the data-parallel step works on the flat buffer).  None: ordinary autograd gradients.
        self.sinks = sinks
        # defer: callable(fn) installed with the sinks (moka_amd.parallel.attach(defer_dA=True)).  dA_m is needed by the optimizer
        # only: the backward then runs the dx half of moka_down_bwd on the dependency chain and hands the dA_m half to `defer`,
        # which launches it later on a side stream (beside the next layer's chain), before the gradients are used.
        self.defer = defer
        # shadows = (BwT [RP, d_out], AT [M, d_in, RP]) bf16: PERSISTENT weight shadows kept by whoever owns the weights
        # (moka_amd.parallel.attach rewrites them behind every optimizer update, one batched launch per gradient bucket).  The forward
        # then launches no moka_weight_shadows and the backward reads these; None: the node computes them per call.
        self.shadows = shadows
        # seed_dev: int64 device tensor of one element the dropout kernels fold into `seed` when they start (moka_opts.seed_dev).  0 in
        # live training (the seed drawn per call is the whole seed); a captured step (schedule.GraphedTrainStep) rewrites it before every
        # replay, so replays of the frozen launch arguments still draw fresh masks.  None: no device part.
        self.seed_dev = seed_dev if self.dropout_p > 0.0 else None


class MokaLinearFn(torch.autograd.Function):
    """y = x W^T (+ bias) + adapter(x).  Inputs: x, W, bias|None, Bw, A_0..A_{M-1}.

    Forward: base GEMM (hipBLASLt through torch), then three launches -- down-projection,
    cross-modal interaction, up-projection with the residual add done in place on the base
    output.  Backward: base input-gradient GEMM (frozen W: no dW), then one pass over gy
    (dL/dhp and dB), the rank-space backward, and one pass over x / dx (dA_m, dx += dh A_m).
    Saves x, h and two small operand packs; the softmax is recomputed in rank space.
    """

    @staticmethod
    def forward(ctx, x, W, bias, Bw, rt: MokaRouting, spec: AdapterSpec, *A):
        _require_device(x, "x")
        dt = _storage([x, Bw, *A] + ([W] if W is not None else []), ["x", "lora_B"] + ["lora_A"] * len(A) + ["base weight"])
        if len(spec.s_out) != rt.M or len(A) != rt.M:
            raise ValueError(f"routing describes {rt.M} modalities but {len(A)} adapters / {len(spec.s_out)} scales were given")
        d_in = x.shape[-1]
        x2 = x.reshape(-1, d_in)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        if x2.shape[0] != rt.T:
            raise ValueError(f"x has {x2.shape[0]} tokens but the masks describe {rt.T}")
        A = [a if a.is_contiguous() else a.contiguous() for a in A]
        Bw_c = Bw if Bw.is_contiguous() else Bw.contiguous()
        fused = (dt == _lib.MOKA_BF16 and FUSE_FORWARD and
                 _lib.up_fwd_fused_pays(x2.shape[0], _lib.ksplit(x2.shape[0], d_in, spec.r), [Bw.shape[0]], spec.r, dt))
        kept = spec.shadows if dt == _lib.MOKA_BF16 else None          # persistent (BwT, AT) of the weights' owner
        if W is not None:
            y = torch.nn.functional.linear(x2, W, bias)               # frozen base, stock PyTorch-ROCm
        else:                                                         # adapter term alone (per-sample adapter_names: several adapters add to one base output)
            y = torch.zeros((x2.shape[0], Bw.shape[0]), dtype=x2.dtype, device=x2.device)
        part = down_fwd(x2, A, rt, spec.r, spec.s_in, spec.dropout_p, spec.seed, dtype=dt, seed_dev=spec.seed_dev)
        if dt == _lib.MOKA_F32:
            # fp32 storage: the rank-space rows themselves are the operands (no bf16 packs, no weight shadows)
            st = cross_fwd(part, rt, spec.r, spec.s_out, spec.w, spec.inv_sqrt_dk, want_hp=True)
            hps = st.hp * _token_scale(rt, spec.s_out, x2.device)[:, None]
            up_fwd_(y, hps, Bw_c, rt, spec.r, dtype=dt)
            ctx.save_for_backward(x2, W, Bw_c, st.h, hps, Bw_c, None, *A)
        elif fused:
            # two launches on the dependency chain: the up-projection computes the interaction itself from the slices and writes what
            # the backward reads from the rank space (h, hp_kmj); the weight shadows are functions of the weights alone
            st = up_fwd_fused_(y, part, Bw_c, rt, spec.r, spec.s_out, spec.w, spec.inv_sqrt_dk, want_state=True)
            st.BwT, st.AT = kept if kept is not None else weight_shadows(Bw_c, A if ctx.needs_input_grad[0] else None, spec.r)
            ctx.save_for_backward(x2, W, Bw_c, st.h, st.hp_kmj, st.BwT, st.AT, *A)
        else:
            if kept is not None:
                st = cross_fwd(part, rt, spec.r, spec.s_out, spec.w, spec.inv_sqrt_dk)
                st.BwT, st.AT = kept
            else:
                st = cross_fwd(part, rt, spec.r, spec.s_out, spec.w, spec.inv_sqrt_dk, Bw=Bw_c, A=A if ctx.needs_input_grad[0] else None)
            up_fwd_(y, st.hp_tok, Bw_c, rt, spec.r)
            ctx.save_for_backward(x2, W, Bw_c, st.h, st.hp_kmj, st.BwT, st.AT, *A)
        ctx.rt, ctx.spec, ctx.x_shape, ctx.has_bias, ctx.dt = rt, spec, x.shape, bias is not None, dt
        return y.reshape(*x.shape[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, gy):
        x2, W, Bw, h, hp_kmj, BwT, AT, *A = ctx.saved_tensors
        rt, spec = ctx.rt, ctx.spec
        r = spec.r
        gy2 = gy.reshape(-1, gy.shape[-1])
        if not gy2.is_contiguous():
            gy2 = gy2.contiguous()
        need_x = ctx.needs_input_grad[0]
        need_B = ctx.needs_input_grad[3]
        need_A = any(ctx.needs_input_grad[6:])
        if W is not None and ctx.needs_input_grad[1]:
            raise _lib.MokaError("moka_amd: the base weight is frozen in MokA; requires_grad on it is not supported")
        if spec.sinks is not None:
            # the flat data-parallel gradient buffer is the accumulator (no temporaries, no cast, nothing returned to autograd)
            dB_acc = spec.sinks[0] if need_B else None
            dA_acc = list(spec.sinks[1]) if need_A else None
            _check_sinks(dB_acc, dA_acc, Bw.shape[0], r, x2.shape[1], len(A), gy2.device)
            flat, shapes = None, []
        else:
            shapes = ([(Bw.shape[0], r)] if need_B else []) + ([(r, x2.shape[1])] * len(A) if need_A else [])
            flat, acc = _grad_accumulators(shapes, gy2.device) if shapes else (None, [])
            dB_acc = acc[0] if need_B else None
            dA_acc = acc[(1 if need_B else 0):] if need_A else None
        dt = ctx.dt
        # dB is needed by the optimizer only: where it is a pass of its own over gy anyway (r > 32), it leaves the dependency chain like dA_m
        split_dB = spec.sinks is not None and spec.defer is not None and dB_acc is not None and _lib.up_bwd_passes(r, dt) == 2
        bst = None
        g_part = up_bwd(gy2, hp_kmj, BwT, rt, r, spec.s_out, None if split_dB else dB_acc, dtype=dt)
        if split_dB:
            fn_db = lambda: up_bwd(gy2, hp_kmj, BwT, rt, r, spec.s_out, dB_acc, dtype=dt, want_g=False)   # noqa: E731
            if dt == _lib.MOKA_BF16 and getattr(spec.defer, "accepts_da", False):
                spec.defer(fn_db, [gy2, hp_kmj], db=((rt, r), [(gy2, hp_kmj, dB_acc)]))
            else:
                spec.defer(fn_db, [gy2, hp_kmj])
        dx2 = None
        if need_x:                                                   # frozen base: dx only, never dW
            dx2 = torch.matmul(gy2, W) if W is not None else torch.zeros_like(x2)
        if need_A or need_x:
            if dt == _lib.MOKA_F32:
                bst = cross_bwd(g_part, h, rt, r, spec.s_in, spec.w, spec.inv_sqrt_dk, want_dh=True)
                bst.dh_tok, bst.dh_kmj = bst.dh * spec.s_in, None    # the fp32 rows, scaled, stand in for the packs
                AT = torch.stack(list(A)).contiguous()
            else:
                bst = cross_bwd(g_part, h, rt, r, spec.s_in, spec.w, spec.inv_sqrt_dk)
            if spec.sinks is not None and spec.defer is not None and dA_acc is not None:
                if dx2 is not None:
                    down_bwd_(bst, x2, AT, rt, r, None, dx2, spec.dropout_p, spec.seed, dtype=dt, seed_dev=spec.seed_dev)
                fn = lambda bst=bst, x2=x2, AT=AT, dA_acc=dA_acc: down_bwd_(bst, x2, AT, rt, r, dA_acc, None, spec.dropout_p, spec.seed, dtype=dt, seed_dev=spec.seed_dev)   # noqa: E731
                if dt == _lib.MOKA_BF16 and getattr(spec.defer, "accepts_da", False):
                    # (described as well: attach() sends a decoder layer's dA_m halves out as ONE launch, moka_down_bwd_da_batch)
                    spec.defer(fn, [x2, bst.dh_tok, bst.dh_kmj, AT], da=((rt, r, float(spec.dropout_p), spec.seed_dev), [(bst.dh_kmj, x2, list(dA_acc), spec.seed or 0)]))
                else:
                    spec.defer(fn, [x2, bst.dh_tok, bst.dh_kmj, AT])
            else:
                down_bwd_(bst, x2, AT, rt, r, dA_acc, dx2, spec.dropout_p, spec.seed, dtype=dt, seed_dev=spec.seed_dev)
        gB, gA = None, [None] * len(A)
        if flat is not None:
            cast = _split_like(flat.to(Bw.dtype), shapes)             # one cast kernel for all weight gradients
            if need_B:
                gB = cast[0]
            if need_A:
                ca = cast[(1 if need_B else 0):]
                gA = [ca[m] if ctx.needs_input_grad[6 + m] else None for m in range(len(A))]
        gbias = gy2.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return (None if dx2 is None else dx2.reshape(ctx.x_shape), None, gbias, gB, None, None, *gA)


def moka_linear(x, W, bias, Bw, A: Sequence[torch.Tensor], rt: MokaRouting, spec: AdapterSpec):
    if getattr(rt, "dup_src", None) is not None:
        # tokens of several modalities (MokaRouting.from_avt_masks): the adapter term over the real + virtual tokens, the virtual rows folded
        # back onto the tokens they stand for (autograd carries their input gradients the same way); the frozen base sees the real tokens only
        ya = rt.fold(MokaLinearFn.apply(rt.extend(x), None, None, Bw, rt, spec, *A)).reshape(*x.shape[:-1], Bw.shape[0])
        return ya if W is None else torch.nn.functional.linear(x, W, bias) + ya
    return MokaLinearFn.apply(x, W, bias, Bw, rt, spec, *A)


# --------------------------------------------------------------------------------------
# autograd node of a GROUP of adapted projections fed by the same x
# --------------------------------------------------------------------------------------
class MokaLinearGroupFn(torch.autograd.Function):
    """(y_0 .. y_{G-1}) = (x W_g^T (+ b_g) + adapter_g(x)) for the G projections that read the same x:
    q/k/v of the attention block, gate/up of the MLP.  Same arithmetic as G MokaLinearFn nodes; x is
    read once by the G down-projections and once by the G dA kernels, and the G input gradients are
    added to dx in one read-modify-write pass.

    Inputs: x, rt, specs (list of G AdapterSpec: r / s_in / s_out / w / d_k / dropout_p must agree, seeds
    differ), G, M, then per projection W_g, bias_g|None, Bw_g, A_g0..A_g{M-1}.
    """

    @staticmethod
    def forward(ctx, x, rt: MokaRouting, specs, G: int, M: int, *flat):
        per = 3 + M
        Ws = [flat[g * per] for g in range(G)]
        biases = [flat[g * per + 1] for g in range(G)]
        Bws = [flat[g * per + 2] for g in range(G)]
        As = [list(flat[g * per + 3:g * per + 3 + M]) for g in range(G)]
        _require_device(x, "x")
        _require_bf16(x, "x")
        for g in range(G):
            _require_bf16(Ws[g], "base weight")
            _require_bf16(Bws[g], "lora_B")
            for a in As[g]:
                _require_bf16(a, "lora_A")
        sp = specs[0]
        for o in specs[1:]:
            if (o.r, o.s_in, o.s_out, o.w, o.inv_sqrt_dk, o.dropout_p) != (sp.r, sp.s_in, sp.s_out, sp.w, sp.inv_sqrt_dk, sp.dropout_p) or o.seed_dev is not sp.seed_dev:
                raise ValueError("moka_amd: the projections of one group must share r, scaling, blc/attn weight and dropout")
        if len(sp.s_out) != rt.M or M != rt.M:
            raise ValueError(f"routing describes {rt.M} modalities but {M} adapters / {len(sp.s_out)} scales were given")
        d_in = x.shape[-1]
        x2 = x.reshape(-1, d_in)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        if x2.shape[0] != rt.T:
            raise ValueError(f"x has {x2.shape[0]} tokens but the masks describe {rt.T}")
        As = [[a if a.is_contiguous() else a.contiguous() for a in Ag] for Ag in As]
        Bws = [b if b.is_contiguous() else b.contiguous() for b in Bws]
        seeds = [s_.seed for s_ in specs]
        need_x = ctx.needs_input_grad[0]
        fused = FUSE_FORWARD and _lib.up_fwd_fused_pays(x2.shape[0], _lib.ksplit(x2.shape[0], d_in, sp.r, len(Bws)), [b.shape[0] for b in Bws], sp.r)
        kept = [s_.shadows for s_ in specs] if all(s_.shadows is not None for s_ in specs) else None     # persistent (BwT, AT) per projection
        ys = [torch.nn.functional.linear(x2, Ws[g], biases[g]) for g in range(G)]      # frozen base, stock PyTorch-ROCm
        parts = down_fwd_group(x2, As, rt, sp.r, sp.s_in, sp.dropout_p, seeds, seed_dev=sp.seed_dev)
        if fused:
            sts = up_fwd_fused_group_(ys, parts, Bws, rt, sp.r, sp.s_out, sp.w, sp.inv_sqrt_dk, want_state=True)     # (see MokaLinearFn.forward)
            if kept is not None:
                BwTs, ATs = [k[0] for k in kept], [k[1] for k in kept]
            else:
                BwTs, ATs = weight_shadows_group(Bws, As if need_x else None, sp.r)
            for g in range(G):
                sts[g].BwT, sts[g].AT = BwTs[g], (ATs[g] if ATs is not None else None)
        else:
            sts = cross_fwd_group(parts, rt, sp.r, sp.s_out, sp.w, sp.inv_sqrt_dk, Bws, As if need_x else None)
            if kept is not None:
                for g in range(G):
                    sts[g].BwT, sts[g].AT = kept[g]
            up_fwd_group_(ys, [st.hp_tok for st in sts], Bws, rt, sp.r)
        saved = [x2]
        for g in range(G):
            saved += [Ws[g], Bws[g], sts[g].h, sts[g].hp_kmj, sts[g].BwT, sts[g].AT if need_x else None, *As[g]]
        ctx.save_for_backward(*saved)
        ctx.rt, ctx.specs, ctx.G, ctx.M, ctx.x_shape = rt, specs, G, M, x.shape
        ctx.has_bias = [b is not None for b in biases]
        return tuple(y.reshape(*x.shape[:-1], y.shape[-1]) for y in ys)

    @staticmethod
    def backward(ctx, *gys):
        G, M, rt, specs = ctx.G, ctx.M, ctx.rt, ctx.specs
        sp = specs[0]
        r = sp.r
        saved = ctx.saved_tensors
        x2 = saved[0]
        per_s = 6 + M
        Ws = [saved[1 + g * per_s] for g in range(G)]
        Bws = [saved[2 + g * per_s] for g in range(G)]
        hs = [saved[3 + g * per_s] for g in range(G)]
        hp_kmjs = [saved[4 + g * per_s] for g in range(G)]
        BwTs = [saved[5 + g * per_s] for g in range(G)]
        ATs = [saved[6 + g * per_s] for g in range(G)]
        As = [list(saved[7 + g * per_s:7 + g * per_s + M]) for g in range(G)]
        per = 3 + M
        nig = ctx.needs_input_grad
        base = 5                                                     # x, rt, specs, G, M
        if any(nig[base + g * per] for g in range(G)):
            raise _lib.MokaError("moka_amd: the base weight is frozen in MokA; requires_grad on it is not supported")
        dev = x2.device
        gy2 = []
        for g in range(G):
            t_ = gys[g]
            if t_ is None:                                           # an output that did not reach the loss
                t_ = torch.zeros((x2.shape[0], Ws[g].shape[0]), dtype=x2.dtype, device=dev)
            t_ = t_.reshape(-1, t_.shape[-1])
            gy2.append(t_ if t_.is_contiguous() else t_.contiguous())
        need_x = nig[0]
        need_B = any(nig[base + g * per + 2] for g in range(G))
        need_A = any(nig[base + g * per + 3 + m] for g in range(G) for m in range(M))
        use_sinks = all(s_.sinks is not None for s_ in specs)
        if use_sinks:
            for g in range(G):
                _check_sinks(specs[g].sinks[0] if need_B else None, list(specs[g].sinks[1]) if need_A else None,
                             Bws[g].shape[0], r, x2.shape[1], M, dev)
            shapes, flat = [], None
            acc = ([specs[g].sinks[0] for g in range(G)] if need_B else []) + \
                  ([a_ for g in range(G) for a_ in specs[g].sinks[1]] if need_A else [])
        else:
            shapes = ([(Bws[g].shape[0], r) for g in range(G)] if need_B else []) + ([(r, x2.shape[1])] * (G * M) if need_A else [])
            flat, acc = _grad_accumulators(shapes, dev) if shapes else (None, [])
        dB_accs = acc[:G] if need_B else None
        split_dB = use_sinks and sp.defer is not None and dB_accs is not None and _lib.up_bwd_passes(r, _lib.MOKA_BF16) == 2
        bsts = None
        g_parts = up_bwd_group(gy2, hp_kmjs, BwTs, rt, r, sp.s_out, None if split_dB else dB_accs)
        if split_dB:                                                 # (see MokaLinearFn.backward)
            fn_db = lambda: up_bwd_group(gy2, hp_kmjs, BwTs, rt, r, sp.s_out, dB_accs, want_g=False)   # noqa: E731
            if getattr(sp.defer, "accepts_da", False):
                sp.defer(fn_db, list(gy2) + list(hp_kmjs), db=((rt, r), [(gy2[g], hp_kmjs[g], dB_accs[g]) for g in range(G)]))
            else:
                sp.defer(fn_db, list(gy2) + list(hp_kmjs))
        dx2 = None
        if need_x:
            dx2 = torch.matmul(gy2[0], Ws[0])                        # frozen base: dx only, never dW
            for g in range(1, G):
                dx2.addmm_(gy2[g], Ws[g])
        dA_accs = None
        if need_A or need_x:
            if bsts is None:
                bsts = cross_bwd_group(g_parts, hs, rt, r, sp.s_in, sp.w, sp.inv_sqrt_dk)
            if need_A:
                a0 = G if need_B else 0
                dA_accs = [acc[a0 + g * M:a0 + (g + 1) * M] for g in range(G)]
            seeds_ = [s_.seed for s_ in specs]
            if use_sinks and sp.defer is not None and dA_accs is not None:
                if dx2 is not None:
                    down_bwd_group_(bsts, x2, ATs, rt, r, None, dx2, sp.dropout_p, seeds_, seed_dev=sp.seed_dev)
                fn = lambda: down_bwd_group_(bsts, x2, None, rt, r, dA_accs, None, sp.dropout_p, seeds_, seed_dev=sp.seed_dev)   # noqa: E731
                keep_ = [x2] + [b.dh_tok for b in bsts] + [b.dh_kmj for b in bsts]
                if getattr(sp.defer, "accepts_da", False):
                    sp.defer(fn, keep_, da=((rt, r, float(sp.dropout_p), sp.seed_dev), [(bsts[g].dh_kmj, x2, list(dA_accs[g]), seeds_[g] or 0) for g in range(G)]))
                else:
                    sp.defer(fn, keep_)
            else:
                down_bwd_group_(bsts, x2, ATs if need_x else None, rt, r, dA_accs, dx2, sp.dropout_p, seeds_, seed_dev=sp.seed_dev)
        cast = _split_like(flat.to(x2.dtype), shapes) if flat is not None else []      # one cast kernel for all weight gradients
        grads = []
        for g in range(G):
            gbias = gy2[g].sum(0) if (ctx.has_bias[g] and nig[base + g * per + 1]) else None
            gB = cast[g] if (cast and need_B and nig[base + g * per + 2]) else None
            a0 = G if need_B else 0
            gA = [cast[a0 + g * M + m] if (cast and need_A and nig[base + g * per + 3 + m]) else None for m in range(M)]
            grads += [None, gbias, gB, *gA]
        return (None if dx2 is None else dx2.reshape(ctx.x_shape), None, None, None, None, *grads)


def moka_linear_group(x, projections, rt: MokaRouting, specs: Sequence[AdapterSpec]):
    """projections: list of (W, bias|None, Bw, [A_0..A_{M-1}]) fed by the same x.  Returns the list of outputs."""
    if x.dtype == torch.float32 or getattr(rt, "dup_src", None) is not None:          # fp32 storage / virtual tokens: correctness paths, one projection at a time
        return [moka_linear(x, W, b, Bw, A, rt, sp) for (W, b, Bw, A), sp in zip(projections, specs)]
    G = len(projections)
    M = len(projections[0][3])
    flat = []
    for W, b, Bw, A in projections:
        flat += [W, b, Bw, *A]
    return list(MokaLinearGroupFn.apply(x, rt, list(specs), G, M, *flat))
