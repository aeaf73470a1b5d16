"""Helpers shared by the CPU and GPU parity tests: load a golden file, regenerate the
case inputs, compare a tensor against the (possibly strided-sampled) reference values."""
import os

import numpy as np
import torch

from oracle import cases as C

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def check_inputs(cd, g):
    """The regenerated inputs must be the ones the golden file was made from."""
    def chk(t):
        return np.array([t.double().sum().item(), t.double().pow(2).sum().item()])
    np.testing.assert_allclose(chk(cd.x), g["chk_x"], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(chk(cd.W), g["chk_W"], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(chk(cd.Bw), g["chk_B"], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(chk(cd.gy), g["chk_gy"], rtol=1e-12, atol=1e-9)
    for m, a in enumerate(cd.A):
        np.testing.assert_allclose(chk(a), g["chk_A"][m], rtol=1e-12, atol=1e-9)
    assert (cd.tok_mod.numpy() == g["tok_mod"]).all()
    assert (cd.question.numpy() == g["question"]).all()


def golden_rel_err(g, key, t, big):
    """Relative Frobenius error of tensor ``t`` against golden entry ``key``.
    For 'big' cases the golden holds a strided sample; the error is measured on the sample
    and normalised by the sample's own norm."""
    t = t.detach().double().cpu()
    ref = torch.from_numpy(g["ref_" + key])
    if big:
        t2 = t.reshape(-1, t.shape[-1])
        t = t2[torch.from_numpy(g["ri_" + key])][:, torch.from_numpy(g["ci_" + key])]
    else:
        t = t.reshape(ref.shape)
    n = ref.norm().item()
    d = (t - ref).norm().item()
    return d / n if n > 0 else d
