"""Host-side cost of one adapted projection through the autograd nodes (tiny tensors, so the GPU time is negligible): Python / ctypes
overhead per forward + backward of moka_linear and moka_linear_group.  Run on the GPU box from the repo root."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from moka_amd.functional import AdapterSpec, moka_linear, moka_linear_group
from moka_amd.routing import MokaRouting
dev = torch.device("cuda:0")
B, S, r, M = 1, 256, 16, 3
tok, q = bench.synthetic_layout(S)
masks = [(tok == m).to(torch.int32).reshape(1, S, 1).repeat(B, 1, 1).to(dev) for m in range(3)]
masks.append(q.to(torch.int32).reshape(1, S, 1).repeat(B, 1, 1).to(dev))
rt = MokaRouting.from_avt_masks(masks)
bf = torch.bfloat16
d = 4096
W = torch.randn(d, d, device=dev, dtype=bf) * 0.02
A = [(torch.randn(r, d, device=dev, dtype=bf) * 0.01).requires_grad_(True) for _ in range(M)]
Bw = (torch.randn(d, r, device=dev, dtype=bf) * 0.02).requires_grad_(True)
x = torch.randn(B, S, d, device=dev, dtype=bf, requires_grad=True)
spec = AdapterSpec(r, 1.0, [1.0] * 3, 1.0, 0.25, 0.05)
gy = torch.randn(B, S, d, device=dev, dtype=bf)
def one():
    y = moka_linear(x, W, None, Bw, A, rt, spec)
    y.backward(gy)
for _ in range(5): one()
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 50
for _ in range(N): one()
t1 = time.perf_counter()   # host time to enqueue
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"single projection fwd+bwd: host enqueue {(t1-t0)/N*1e6:.0f} us per call, wall {(t2-t0)/N*1e6:.0f} us")
projs = [(W, None, Bw, A)] * 3
specs = [AdapterSpec(r, 1.0, [1.0] * 3, 1.0, 0.25, 0.05) for _ in range(3)]
def grp():
    ys = moka_linear_group(x, projs, rt, specs)
    torch.autograd.backward(ys, [gy, gy, gy])
for _ in range(5): grp()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): grp()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"group of 3 fwd+bwd: host enqueue {(t1-t0)/N*1e6:.0f} us per call, wall {(t2-t0)/N*1e6:.0f} us")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(100): one()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
