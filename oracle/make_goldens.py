#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference layers.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU
box).  For every case of ``oracle/cases.py`` it

  1. builds the reference layer (AVT ``peft_hyper.tuners.lora.Linear`` or VT
     ``modified_peft.tuners.lora.layer.Linear``), loads the case's weights,
  2. runs forward + autograd backward in fp64 (exact answer), fp32 and bf16 (the
     reference's own low-precision answers, recorded for context),
  3. runs this repo's oracle restatement (``oracle/moka_oracle.py``) and asserts it
     equals the fp64 reference to <= 1e-10 relative on y, dx, dA_m, dB,
  4. writes ``tests/golden/<case>.npz``: input checksums, fp64 reference outputs (full
     tensors for small cases, strided samples + norms for the 4096/11008-wide ones) and
     the reference's fp32/bf16 errors.

The import recipes follow SURVEY.md section 8(c).  Nothing from the reference is copied:
the .npz files hold numbers only.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

from oracle import cases as C            # noqa: E402
from oracle import moka_oracle as O      # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _import_reference():
    sys.path.insert(0, os.path.join(REF, "AudioVisualText"))
    from peft_hyper.tuners.lora import Linear as AvtLinear          # noqa
    # VT: modified_peft imports `peft.*`; it is a full PEFT tree, so alias it.
    tmp = tempfile.mkdtemp(prefix="moka_ref_alias_")
    os.symlink(os.path.join(REF, "VisualText", "modified_peft"), os.path.join(tmp, "peft"))
    os.symlink(os.path.join(REF, "VisualText", "modified_peft"), os.path.join(tmp, "modified_peft"))
    sys.path.insert(0, tmp)
    from modified_peft.tuners.lora.layer import Linear as VtLinear  # noqa
    return AvtLinear, VtLinear


def build_avt(AvtLinear, cd: C.CaseData, dtype):
    c = cd.case
    method = "test" if c.masks_none else "train"
    lin = AvtLinear(c.d_in, c.d_out, r=444, lora_alpha=c.alpha, lora_nums=3, blc_alpha=1,
                    blc_weight=c.w, lora_dropout=0.0, loramethod=method, bias=False)
    # r >= 10 cannot be expressed in the digit encoding (lora.py:256-259): swap in real rank-r modules
    for i in range(3):
        setattr(lin, f"lora_A{i}", torch.nn.Linear(c.d_in, c.r, bias=False))
    lin.lora_B0 = torch.nn.Linear(c.r, c.d_out, bias=False)
    lin.r = [c.r] * 3
    lin.d_k = c.r
    lin.scaling = [c.alpha / c.r]
    lin = lin.to(dtype)
    with torch.no_grad():
        lin.weight.copy_(cd.W.to(dtype))
        for i in range(3):
            getattr(lin, f"lora_A{i}").weight.copy_(cd.A[i].to(dtype))
        lin.lora_B0.weight.copy_(cd.Bw.to(dtype))
    for p in lin.parameters():
        p.requires_grad_(True)
    return lin


def build_vt(VtLinear, cd: C.CaseData, dtype):
    c = cd.case
    base = torch.nn.Linear(c.d_in, c.d_out, bias=False).to(dtype)
    lin = VtLinear(base, "image", r=c.r, lora_alpha=c.alpha, lora_dropout=0.0, attn_weight=c.w)
    lin.update_layer("text", c.r, lora_alpha=c.alpha, lora_dropout=0.0, init_lora_weights=True, use_rslora=False)
    lin.set_adapter(["image", "text"])
    with torch.no_grad():
        lin.base_layer.weight.copy_(cd.W.to(dtype))
        lin.lora_A["text"].weight.copy_(cd.A[0].to(dtype))
        lin.lora_A["image"].weight.copy_(cd.A[1].to(dtype))
        lin.lora_B["text"].weight.copy_(cd.Bw.to(dtype))
        lin.lora_B["image"].weight.normal_(0, 0.02)      # allocated but unused by the forward
    for p in lin.parameters():
        p.requires_grad_(True)
    assert lin.lora_A["text"].weight.dtype == dtype
    return lin


def run_reference(kind, lin, cd: C.CaseData, dtype):
    c = cd.case
    x = cd.x.to(dtype).clone().requires_grad_(True)
    if kind == "avt":
        masks = None if cd.masks is None else [m.clone() for m in cd.masks]
        y = lin(x, masks)
    else:
        if cd.masks is None:
            y = lin(x, None, None, None)
        else:
            y = lin(x, *[m.clone() for m in cd.masks])
    (y * cd.gy.to(dtype)).sum().backward()
    if kind == "avt":
        dA = [getattr(lin, f"lora_A{i}").weight.grad for i in range(3)]
        dB = lin.lora_B0.weight.grad
    else:
        dA = [lin.lora_A["text"].weight.grad, lin.lora_A["image"].weight.grad]
        dB = lin.lora_B["text"].weight.grad
        assert lin.lora_B["image"].weight.grad is None
    dA = [torch.zeros_like(cd.A[0], dtype=dtype) if g is None else g for g in dA]
    return dict(y=y.detach(), dx=x.grad.detach(), dA=[g.detach() for g in dA], dB=dB.detach())


def run_oracle(cd: C.CaseData):
    c = cd.case
    dt = torch.float64
    y0 = cd.x.to(dt) @ cd.W.to(dt).t()
    s = c.alpha / c.r
    if cd.masks is None:
        # plain LoRA with the text adapter; gradients by autograd of the restatement itself
        x = cd.x.to(dt).clone().requires_grad_(True)
        A0 = cd.A[0].to(dt).clone().requires_grad_(True)
        Bw = cd.Bw.to(dt).clone().requires_grad_(True)
        y = O.plain_lora_forward(x, x @ cd.W.to(dt).t(), A0, Bw, s)
        (y * cd.gy.to(dt)).sum().backward()
        dA = [A0.grad] + [torch.zeros_like(A0) for _ in cd.A[1:]]
        return dict(y=y.detach(), dx=x.grad, dA=dA, dB=Bw.grad, h=None, hp=None, dh=None)
    if c.variant == "avt":
        y, ctx = O.avt_forward(cd.x, cd.W, cd.A, cd.Bw, cd.masks, c.alpha, c.r, c.w)
    else:
        y, ctx = O.vt_forward(cd.x, cd.W, cd.A[0], cd.A[1], cd.Bw, *cd.masks, s, s, c.w)
    dx_ad, dA, dB, dh = O.adapter_backward(cd.gy, ctx)
    dx = dx_ad + cd.gy.to(dt) @ cd.W.to(dt)
    return dict(y=y, dx=dx, dA=dA, dB=dB, h=ctx.h, hp=ctx.hp, dh=dh, dx_adapter=dx_ad, y0=y0)


def rel(a, b):
    a, b = a.double(), b.double()
    d = (a - b).norm().item()
    n = b.norm().item()
    return d / n if n > 0 else d


def sample_idx(n, k):
    """k roughly evenly spread indices in [0,n) incl. both ends (deterministic)."""
    if n <= k:
        return np.arange(n)
    return np.unique(np.round(np.linspace(0, n - 1, k)).astype(np.int64))


def main():
    torch.manual_seed(0)
    AvtLinear, VtLinear = _import_reference()
    os.makedirs(OUT, exist_ok=True)
    log = []
    for name in C.case_names():
        c = C.get_case(name)
        cd = C.make_case_data(name)
        build = build_avt if c.variant == "avt" else build_vt
        Lin = AvtLinear if c.variant == "avt" else VtLinear
        entry = {"case": name, "variant": c.variant, "shape": [c.B, c.S, c.d_in, c.d_out, c.r]}
        if c.expect == "IndexError":
            lin = build(Lin, cd, torch.float64)
            try:
                run_reference(c.variant, lin, cd, torch.float64)
                raise AssertionError("reference did not raise")
            except IndexError as e:
                entry["reference_raises"] = "IndexError: " + str(e)
            try:
                run_oracle(cd)
                raise AssertionError("oracle did not raise")
            except IndexError:
                entry["oracle_raises"] = "IndexError"
            np.savez_compressed(os.path.join(OUT, name + ".npz"), expect=np.array("IndexError"))
            log.append(entry)
            print(json.dumps(entry))
            continue

        ref64 = run_reference(c.variant, build(Lin, cd, torch.float64), cd, torch.float64)
        orc = run_oracle(cd)
        errs = {"y": rel(orc["y"], ref64["y"]), "dx": rel(orc["dx"], ref64["dx"]), "dB": rel(orc["dB"], ref64["dB"])}
        for m in range(len(cd.A)):
            errs[f"dA{m}"] = rel(orc["dA"][m], ref64["dA"][m])
        entry["oracle_vs_ref_fp64"] = errs
        assert max(errs.values()) < 1e-10, (name, errs)

        # the reference's own low-precision answers vs its fp64 answer (context for tolerances)
        lowp = {}
        for dt, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
            try:
                rl = run_reference(c.variant, build(Lin, cd, dt), cd, dt)
                lowp[tag] = {"y": rel(rl["y"], ref64["y"]), "dx": rel(rl["dx"], ref64["dx"]),
                             "dB": rel(rl["dB"], ref64["dB"]),
                             "dA": max(rel(rl["dA"][m], ref64["dA"][m]) for m in range(len(cd.A))
                                       if ref64["dA"][m].norm() > 0)}
            except Exception as e:  # pragma: no cover
                lowp[tag] = {"error": repr(e)}
        entry["reference_lowp_vs_fp64"] = lowp

        out = {
            "chk_x": np.array([cd.x.double().sum().item(), cd.x.double().pow(2).sum().item()]),
            "chk_W": np.array([cd.W.double().sum().item(), cd.W.double().pow(2).sum().item()]),
            "chk_A": np.array([[a.double().sum().item(), a.double().pow(2).sum().item()] for a in cd.A]),
            "chk_B": np.array([cd.Bw.double().sum().item(), cd.Bw.double().pow(2).sum().item()]),
            "chk_gy": np.array([cd.gy.double().sum().item(), cd.gy.double().pow(2).sum().item()]),
            "tok_mod": cd.tok_mod.numpy().astype(np.int8),
            "question": cd.question.numpy(),
        }
        tensors = {"y": ref64["y"], "dx": ref64["dx"], "dB": ref64["dB"]}
        for m in range(len(cd.A)):
            tensors[f"dA{m}"] = ref64["dA"][m]
        for k, t in tensors.items():
            t = t.double()
            out[f"norm_{k}"] = np.array([t.norm().item(), t.sum().item()])
            if c.big:
                t2 = t.reshape(-1, t.shape[-1])
                ri, ci = sample_idx(t2.shape[0], 24), sample_idx(t2.shape[1], 96)
                out[f"ri_{k}"], out[f"ci_{k}"] = ri, ci
                out[f"ref_{k}"] = t2[ri][:, ci].numpy()
            else:
                out[f"ref_{k}"] = t.numpy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        log.append(entry)
        print(json.dumps(entry))
    with open(os.path.join(OUT, "GENERATION_LOG.json"), "w") as f:
        json.dump({"torch": torch.__version__, "entries": log}, f, indent=1)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"wrote {len(log)} cases, {tot/1e6:.2f} MB total")


if __name__ == "__main__":
    main()
