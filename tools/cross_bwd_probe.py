#!/usr/bin/env python3
"""How much of moka_cross_bwd is the attention?  Times the entry point (both launches) on 4 x 2048 tokens, r = 16, ks = 8, for a VT
routing with and without question tokens (no question -> klen = 0: every block takes the plain path) -> floor of a one-launch form."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moka_amd import functional as F
from moka_amd.routing import MokaRouting

dev = torch.device("cuda:0")
B, S, r, ks = 4, 2048, 16, 8
T = B * S
tok = torch.zeros(B, S, dtype=torch.long)
tok[:, 16:272] = 1
tok[:, 288:416] = 1
q = torch.zeros(B, S, dtype=torch.bool)
q[:, 416:480] = True
for name, qq in (("with 64 question tokens", q), ("no question tokens (plain path only)", torch.zeros_like(q))):
    rt = MokaRouting.from_vt_masks((tok == 0).to(dev), (tok == 1).to(dev), qq.to(dev))
    g_part = torch.randn(ks, T, 16, device=dev)
    h = torch.randn(T, 16, device=dev)
    for _ in range(20):
        F.cross_bwd(g_part, h, rt, r, 1.0, 0.05, 0.25)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 300
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            F.cross_bwd(g_part, h, rt, r, 1.0, 0.05, 0.25)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=st):
        for _ in range(n):
            F.cross_bwd(g_part, h, rt, r, 1.0, 0.05, 0.25)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / n * 1e3:.2f} us per moka_cross_bwd (two launches, back to back in a hipGraph)")
