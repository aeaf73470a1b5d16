#!/usr/bin/env python3
"""Probe: weight-gradient kernel (dA path) under different routings / pack sizes, cold input."""
import math, os, sys
from ctypes import byref, c_float, c_void_p
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from moka_amd import _lib
from moka_amd.routing import MokaRouting
from oracle import cases as C
lib = _lib.load(); dev = torch.device("cuda:0")
B, S, r = 4, 2048, 16; T = B * S
tok, q = C.build_layout(C.synthetic_sequence_layout(S), S)
masks = [(tok == m).to(torch.int32).reshape(1, S, 1).repeat(B, 1, 1).to(dev) for m in range(3)] + [q.to(torch.int32).reshape(1, S, 1).repeat(B, 1, 1).to(dev)]
bf, f32 = torch.bfloat16, torch.float32
RP, Tp = 16, _lib.tok_pad(T)
d = 4096
xs = [torch.randn(T, d, device=dev, dtype=bf) for _ in range(6)]
def timeit(fn, iters=24):
    for i in range(4): assert fn(i) == 0, lib.moka_last_error()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fn(i)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
for name, rt, M in (("synthetic AVT routing M=3", MokaRouting.from_avt_masks(masks), 3), ("all text, M=3", MokaRouting.plain(B, S, dev, 3), 3),
                    ("all text, M=1", MokaRouting.plain(B, S, dev, 1), 1)):
    A = [torch.randn(r, d, device=dev, dtype=bf) * 0.01 for _ in range(M)]
    dh_tok = torch.zeros(Tp, 2 * RP, dtype=bf, device=dev); dh_kmj = torch.randn(M, 2, RP, Tp, device=dev).to(bf)
    dA = [torch.zeros(r, d, dtype=f32, device=dev) for _ in range(M)]
    Ap = (c_void_p * M)(*[a.data_ptr() for a in A]); dAp = (c_void_p * M)(*[a.data_ptr() for a in dA])
    tm = rt.tok_mod.data_ptr(); sp = lambda: c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda i: lib.moka_down_bwd(dh_tok.data_ptr(), dh_kmj.data_ptr(), xs[i % 6].data_ptr(), Ap, tm, dAp, None, T, d, r, M, 0.0, 0, 0, sp())
    fwarm = lambda i: lib.moka_down_bwd(dh_tok.data_ptr(), dh_kmj.data_ptr(), xs[0].data_ptr(), Ap, tm, dAp, None, T, d, r, M, 0.0, 0, 0, sp())
    print(f"{name:30s} cold {timeit(f):7.1f} us   same-buffer {timeit(fwarm):7.1f} us")
