#!/usr/bin/env python3
"""Time the non-adapter parts of a bench step (zero_, fused AdamW, bf16 working copy) at the 7B r=16 M=3 size."""
import torch
n = 76_414_976
dev = torch.device("cuda:0")
master = torch.randn(n, device=dev)
g = torch.randn(n, device=dev)
work = torch.empty(n, device=dev, dtype=torch.bfloat16)
p = torch.nn.Parameter(master); p.grad = g
opt = torch.optim.AdamW([p], lr=1e-4, fused=True)
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
print("zero_  %.3f ms" % t(lambda: g.zero_()))
print("adamw  %.3f ms" % t(lambda: opt.step()))
print("copy   %.3f ms" % t(lambda: work.copy_(master)))
