"""Design probe: the batched y-expand (moka_up_fwd_group) and gy (moka_up_bwd_group) launches for members of different width
(grouped-query k / v beside q: d_out 8192 / 1024 / 1024) against the same shapes launched alone, per expand_bpc setting.
Run on the GPU box from the repo root."""
import os as _os
_os.environ.setdefault("MOKA_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "moka_amd", "libmoka_hip_diag.so"))   # moka_tune: diagnostics build only
import math, os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from moka_amd import _lib
from moka_amd import functional as F
from moka_amd.routing import MokaRouting
lib=_lib.load(); dev=torch.device("cuda:0")
B,S,r,M=4,2048,16,3; T=B*S
tok,q=bench.synthetic_layout(S)
masks=[(tok==m).to(torch.int32).reshape(1,S,1).repeat(B,1,1).to(dev) for m in range(3)]
masks.append(q.to(torch.int32).reshape(1,S,1).repeat(B,1,1).to(dev))
rt=MokaRouting.from_avt_masks(masks)
bf=torch.bfloat16
NB=4
def timeit(fn, iters=20):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
RP=16; Tp=_lib.tok_pad(T)
def case(d_outs):
    ys=[[torch.randn(T,d,device=dev,dtype=bf) for d in d_outs] for _ in range(NB)]
    hp=[torch.randn(Tp,2*RP,device=dev,dtype=bf) for _ in d_outs]
    Bw=[torch.randn(d,r,device=dev,dtype=bf)*0.02 for d in d_outs]
    gys=[[torch.randn(T,d,device=dev,dtype=bf) for d in d_outs] for _ in range(NB)]
    kmj=[torch.randn(2,RP,Tp,device=dev,dtype=bf) for _ in d_outs]
    BwT=[torch.randn(RP,d,device=dev,dtype=bf) for d in d_outs]
    dB=[torch.zeros(d,r,device=dev) for d in d_outs]
    t_up=timeit(lambda i: F.up_fwd_group_(ys[i%NB], hp, Bw, rt, r))
    t_gy=timeit(lambda i: F.up_bwd_group(gys[i%NB], kmj, BwT, rt, r, [1.0]*M, dB))
    mb=sum(d_outs)*T*2/1e6
    return t_up, t_gy, mb
for tune in ("", "expand_bpc=2", "expand_bpc=4", "expand_bpc=16"):
    if tune:
        k,v=tune.split("="); lib.moka_tune(k.encode(), int(v))
    for d_outs in ((8192,),(8192,1024,1024),(1024,1024),(1024,),(8192,8192,8192)):
        tu,tg,mb=case(d_outs)
        print(f"{tune or 'default':14s} {str(d_outs):22s} y-expand {tu:7.1f} us ({2*mb/tu/1e0*1e-0/1e3:5.2f} TB/s rmw)   gy {tg:7.1f} us ({mb/tg/1e3:5.2f} TB/s)")
    lib.moka_tune(b"expand_bpc", 0)
