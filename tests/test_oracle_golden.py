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


def test_reference_on_overlapping_masks_is_pinned_and_the_hip_routing_expresses_them_with_virtual_tokens():
    """A token in TWO modality masks (lora.py:468 runs every adapter on its masked copy of x): the real AVT layer gives it one
    rank-space row per modality stream, each stream interacts on its own, the streams are summed -- pinned by
    tests/golden/avt_dual_modality.npz (oracle/make_dual_golden.py: reference fp64 outputs + gradients) through the dense-mask
    restatement oracle/dense_avt.py.  One modality id per token cannot express that: the routed oracle raises ValueError; the host
    routing of the HIP path (round 5) gives the token one VIRTUAL token per further membership behind the sample's real ones
    (MokaRouting.dup_src / extend / fold) -- checked here on the CPU by running the routed oracle on the extended rows and folding,
    tests/test_gpu_parity.py::test_tokens_of_two_modalities_against_the_reference_golden runs the kernels on them."""
    import os
    import numpy as np
    from moka_amd.routing import MokaRouting
    from oracle.dense_avt import avt_dense_forward
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "avt_dual_modality.npz"))
    t = lambda k: torch.from_numpy(g[k])          # noqa: E731
    masks = [m for m in t("masks")]
    assert int((masks[0] + masks[1] + masks[2]).max()) == 2          # the overlap is really there
    x = t("x").clone().requires_grad_(True)
    A = [a.clone().requires_grad_(True) for a in t("A")]
    Bw = t("Bw").clone().requires_grad_(True)
    y = avt_dense_forward(x, t("W"), A, Bw, masks, float(g["alpha"]), int(g["r"]), float(g["w"]))
    (y * t("gy")).sum().backward()
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()          # noqa: E731
    assert rel(y.detach(), t("ref_y")) < 1e-12 and rel(x.grad, t("ref_dx")) < 1e-12 and rel(Bw.grad, t("ref_dB")) < 1e-12
    for m in range(3):
        assert rel(A[m].grad, t("ref_dA")[m]) < 1e-12
    # ... and it is NOT what routing every token to one adapter would give: dropping the second membership changes y
    single = [masks[0] * (1 - masks[1]) * (1 - masks[2])] + masks[1:]
    y1 = avt_dense_forward(t("x"), t("W"), [a.detach() for a in A], Bw.detach(), single, float(g["alpha"]), int(g["r"]), float(g["w"]))
    assert rel(y1, t("ref_y")) > 1e-3
    with pytest.raises(ValueError):
        O.routing_from_avt_masks(masks)
    rt = MokaRouting.from_avt_masks(masks)
    B, L = masks[0].shape[:2]
    n_extra = int(((masks[0] + masks[1] + masks[2]).clamp(min=1) - 1).sum())
    assert rt.dup_src is not None and rt.S_real == L and rt.S == L + 16 and rt.T == B * rt.S and n_extra == 4
    tok = rt.tok_mod[:rt.T].reshape(B, rt.S)
    assert int((tok[:, L:] != 255).sum()) == n_extra                              # one virtual token per further membership
    assert (tok[0, 5:8] == 0).all() and tok[1, 12] == 0                             # the token itself keeps its first (text) membership
    ext = [(tok == m).to(torch.int32).reshape(B, rt.S, 1) for m in range(3)] + [torch.cat([masks[3], torch.zeros(B, 16, 1, dtype=masks[3].dtype)], 1)]
    xe = rt.extend(t("x"))
    assert torch.equal(xe[:, :L], t("x")) and torch.equal(xe[0, L:L + 3], t("x")[0, 5:8]) and torch.equal(xe[1, L], t("x")[1, 12])
    ya, _ = O.avt_forward(xe, torch.zeros_like(t("W")), [a.detach() for a in A], Bw.detach(), ext, float(g["alpha"]), int(g["r"]), float(g["w"]))
    y2 = t("x") @ t("W").t() + rt.fold(ya)
    assert rel(y2, t("ref_y")) < 1e-12
