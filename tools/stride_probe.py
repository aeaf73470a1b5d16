import math, os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from moka_amd import _lib, functional as F
from moka_amd.routing import MokaRouting
dev = torch.device("cuda:0")
B, S, r, M = 4, 2048, 16, 3
T = B * S
tok, q = bench.synthetic_layout(S)
masks = [(tok == m).to(torch.int32).reshape(1, S, 1).repeat(B, 1, 1).to(dev) for m in range(3)]
masks.append(q.to(torch.int32).reshape(1, S, 1).repeat(B, 1, 1).to(dev))
rt = MokaRouting.from_avt_masks(masks)
bf = torch.bfloat16
def timeit(fn, iters=24):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
lib = _lib.load()
for d_in in (4096, 11008):
    xs = [torch.randn(T, d_in, device=dev, dtype=bf) for _ in range(6)]
    for G in (1, 3):
        As = [[torch.randn(r, d_in, device=dev, dtype=bf) * 0.01 for _ in range(M)] for _ in range(G)]
        t = timeit(lambda i: F.down_fwd_group(xs[i % 6], As, rt, r, 1.0, 0.0, None))
        print(f"d_in {d_in:6d} G={G}: {t:7.1f} us   {2*T*d_in/(t*1e-6)/1e9:7.0f} GB/s unique")
    del xs
