"""CPU: the oracle restatement (oracle/moka_oracle.py) against the golden vectors that
oracle/make_goldens.py produced from the real reference layers (fp64).  This is the pin
that makes the oracle trustworthy on the GPU box, where /root/reference does not exist."""
import pytest
import torch

from oracle import cases as C
from oracle import moka_oracle as O
from tests.golden_util import check_inputs, golden_rel_err, load_golden

TOL = 1e-10   # fp64 restatement vs fp64 reference


def _run_oracle(cd):
    c = cd.case
    dt = torch.float64
    s = c.alpha / c.r
    if cd.masks is None:
        x = cd.x.to(dt).clone().requires_grad_(True)
        A0 = cd.A[0].to(dt).clone().requires_grad_(True)
        Bw = cd.Bw.to(dt).clone().requires_grad_(True)
        y = O.plain_lora_forward(x, x @ cd.W.to(dt).t(), A0, Bw, s)
        (y * cd.gy.to(dt)).sum().backward()
        dA = [A0.grad] + [torch.zeros_like(A0) for _ in cd.A[1:]]
        return y.detach(), x.grad, dA, Bw.grad
    if c.variant == "avt":
        y, ctx = O.avt_forward(cd.x, cd.W, cd.A, cd.Bw, cd.masks, c.alpha, c.r, c.w)
    else:
        y, ctx = O.vt_forward(cd.x, cd.W, cd.A[0], cd.A[1], cd.Bw, *cd.masks, s, s, c.w)
    dx_ad, dA, dB, _ = O.adapter_backward(cd.gy, ctx)
    return y, dx_ad + cd.gy.to(dt) @ cd.W.to(dt), dA, dB


@pytest.mark.parametrize("name", C.case_names(include_errors=False))
def test_oracle_matches_reference_golden(name):
    cd = C.make_case_data(name)
    g = load_golden(name)
    check_inputs(cd, g)
    y, dx, dA, dB = _run_oracle(cd)
    big = cd.case.big
    assert golden_rel_err(g, "y", y, big) < TOL
    assert golden_rel_err(g, "dx", dx, big) < TOL
    assert golden_rel_err(g, "dB", dB, big) < TOL
    for m in range(len(cd.A)):
        assert golden_rel_err(g, f"dA{m}", dA[m], big) < TOL


def test_avt_no_question_raises_like_reference():
    cd = C.make_case_data("avt_noquestion")
    assert str(load_golden("avt_noquestion")["expect"]) == "IndexError"
    with pytest.raises(IndexError):
        O.routing_from_avt_masks(cd.masks)


def test_oracle_backward_matches_autograd_of_itself():
    """Independent check of the hand-derived backward: autograd through the oracle forward."""
    cd = C.make_case_data("avt_tiny")
    c = cd.case
    dt = torch.float64
    x = cd.x.to(dt).clone().requires_grad_(True)
    A = [a.to(dt).clone().requires_grad_(True) for a in cd.A]
    Bw = cd.Bw.to(dt).clone().requires_grad_(True)
    rt = O.routing_from_avt_masks(cd.masks)
    y, _ = O.adapter_forward(x, torch.zeros(c.B, c.S, c.d_out, dtype=dt), A, Bw, rt,
                             s_in=c.alpha / c.r, s_out=[1.0] * 3, w=c.w, d_k=c.r)
    (y * cd.gy.to(dt)).sum().backward()
    y2, ctx = O.adapter_forward(cd.x, torch.zeros(c.B, c.S, c.d_out, dtype=dt), cd.A, cd.Bw, rt,
                                s_in=c.alpha / c.r, s_out=[1.0] * 3, w=c.w, d_k=c.r)
    dx, dA, dB, _ = O.adapter_backward(cd.gy, ctx)
    assert torch.allclose(dx, x.grad, rtol=1e-10, atol=1e-12)
    assert torch.allclose(dB, Bw.grad, rtol=1e-10, atol=1e-12)
    for m in range(3):
        assert torch.allclose(dA[m], A[m].grad, rtol=1e-10, atol=1e-12)
