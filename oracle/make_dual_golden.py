#!/usr/bin/env python3
"""tests/golden/avt_dual_modality.npz: what the REAL AVT layer does with a token that sits in two modality masks.

Build container only (needs /root/reference).  Checks ``oracle/dense_avt.py`` against the reference layer (forward + autograd
gradients, fp64, <= 1e-10) on a tiny case whose text and video masks overlap on a few tokens, checks that on disjoint masks the
dense form equals the routed oracle, and stores inputs and reference outputs (numbers only)."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

from oracle import moka_oracle as O                      # noqa: E402
from oracle.dense_avt import avt_dense_forward           # noqa: E402
from oracle.make_goldens import _import_reference, rel   # noqa: E402


def main():
    AvtLinear, _ = _import_reference()
    g = torch.Generator().manual_seed(20260929)
    B, L, d_in, d_out, r, alpha, w = 2, 24, 32, 64, 4, 16.0, 1.0          # (widths: multiples of 32, what the HIP path takes)
    dt = torch.float64
    x = torch.randn(B, L, d_in, generator=g, dtype=dt)
    W = torch.randn(d_out, d_in, generator=g, dtype=dt) * 0.05
    A = [torch.randn(r, d_in, generator=g, dtype=dt) * 0.2 for _ in range(3)]
    Bw = torch.randn(d_out, r, generator=g, dtype=dt) * 0.1
    gy = torch.randn(B, L, d_out, generator=g, dtype=dt)
    tok = torch.zeros(B, L, dtype=torch.int64)           # 0 text, 1 video, 2 audio
    tok[:, 3:9] = 1
    tok[:, 11:15] = 2
    q = torch.zeros(B, L, dtype=torch.int64)
    q[:, 17:21] = 1

    def masks(overlap):
        mt, mv, ma = [(tok == m).to(torch.int32) for m in range(3)]
        if overlap:
            mt = mt.clone()
            mt[0, 5:8] = 1                               # three video tokens of sample 0 are ALSO text tokens
            mt[1, 12] = 1                                # one audio token of sample 1 too
        return [m.reshape(B, L, 1) for m in (mt, mv, ma, q.to(torch.int32))]

    def reference(ms):
        lin = AvtLinear(d_in, d_out, r=444, lora_alpha=alpha, lora_nums=3, blc_alpha=1, blc_weight=w, lora_dropout=0.0,
                        loramethod="train", bias=False).to(dt)
        with torch.no_grad():
            lin.weight.copy_(W)
            for i in range(3):
                getattr(lin, f"lora_A{i}").weight.copy_(A[i])
            lin.lora_B0.weight.copy_(Bw)
        for p in lin.parameters():
            p.requires_grad_(True)
        xr = x.clone().requires_grad_(True)
        y = lin(xr, [m.clone() for m in ms])
        (y * gy).sum().backward()
        return dict(y=y.detach(), dx=xr.grad, dA=[getattr(lin, f"lora_A{i}").weight.grad for i in range(3)], dB=lin.lora_B0.weight.grad)

    def dense(ms):
        xr = x.clone().requires_grad_(True)
        Ar = [a.clone().requires_grad_(True) for a in A]
        Br = Bw.clone().requires_grad_(True)
        y = avt_dense_forward(xr, W, Ar, Br, ms, alpha, r, w)
        (y * gy).sum().backward()
        return dict(y=y.detach(), dx=xr.grad, dA=[a.grad for a in Ar], dB=Br.grad)

    out = {}
    for overlap in (False, True):
        ms = masks(overlap)
        ref, den = reference(ms), dense(ms)
        errs = [rel(den["y"], ref["y"]), rel(den["dx"], ref["dx"]), rel(den["dB"], ref["dB"])] + [rel(a, b) for a, b in zip(den["dA"], ref["dA"])]
        assert max(errs) < 1e-10, (overlap, errs)
        if not overlap:
            yo, _ = O.avt_forward(x, W, A, Bw, ms, alpha, r, w)
            assert rel(yo, ref["y"]) < 1e-10          # disjoint masks: the routed oracle IS the dense form
        else:
            try:
                O.routing_from_avt_masks(ms)
                raise AssertionError("the routed oracle accepted overlapping masks")
            except ValueError:
                pass
            out = dict(x=x.numpy(), W=W.numpy(), A=np.stack([a.numpy() for a in A]), Bw=Bw.numpy(), gy=gy.numpy(),
                       masks=np.stack([m.numpy() for m in ms]), alpha=np.array(alpha), r=np.array(r), w=np.array(w),
                       ref_y=ref["y"].numpy(), ref_dx=ref["dx"].numpy(), ref_dA=np.stack([a.numpy() for a in ref["dA"]]), ref_dB=ref["dB"].numpy(),
                       dense_vs_ref=np.array(errs))
        print("overlap" if overlap else "disjoint", "dense oracle vs reference:", ["%.1e" % e for e in errs])
    path = os.path.join(ROOT, "tests", "golden", "avt_dual_modality.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
