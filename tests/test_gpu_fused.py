"""GPU tests of moka_up_fwd_fused (round 4): the up-projection that computes the cross-modal interaction itself from the
split-K slices of moka_down_fwd (replaces lora.py:485-530 / layer.py:627-669 on the forward's dependency chain).

The bar is bit-identity with the two-launch path moka_cross_fwd + moka_up_fwd on the same slices -- which `test_gpu_parity.py`
pins stage by stage against the fp64 oracle -- plus one direct comparison with the oracle."""
import pytest
import torch

from oracle import cases as C
from oracle import moka_oracle as O
from tests.test_gpu_parity import (TOL_BF16, _dev, _fuzz_seeds, _group_data, _long_question_case, _random_case, _random_group_cfg,
                                   _spec_and_routing, rel, ulp_bf16_diff)

pytestmark = pytest.mark.gpu


def _operands(cd, dev):
    c = cd.case
    T = c.B * c.S
    bf = torch.bfloat16
    x2 = cd.x.reshape(T, c.d_in).to(dev, bf).contiguous()
    A = [a.to(dev, bf).contiguous() for a in cd.A]
    Bw = cd.Bw.to(dev, bf).contiguous()
    g = torch.Generator().manual_seed(c.seed + 7)
    y0 = torch.randn(c.B, c.S, c.d_out, generator=g).to(bf)
    return x2, A, Bw, y0.reshape(T, c.d_out).to(dev).contiguous()


def _fused_vs_two_launches(cd, p=0.0):
    """y of the fused launch == y of cross_fwd + up_fwd, bit for bit; the state launch without the token-major pack writes the
    same h / hp_kmj / BwT / AT."""
    from moka_amd import _lib
    from moka_amd import functional as F
    dev = _dev()
    c = cd.case
    spec, rt, _ = _spec_and_routing(cd, dev)
    x2, A, Bw, y0 = _operands(cd, dev)
    if not _lib.up_fwd_fused_ok(c.r):
        with pytest.raises(_lib.MokaError):
            part = F.down_fwd(x2, A, rt, c.r, spec.s_in)
            F.up_fwd_fused_(y0.clone(), part, Bw, rt, c.r, spec.s_out, spec.w, spec.inv_sqrt_dk)
        return
    part = F.down_fwd(x2, A, rt, c.r, spec.s_in, p, 12345)
    st = F.cross_fwd(part, rt, c.r, spec.s_out, spec.w, spec.inv_sqrt_dk, Bw=Bw, A=A)
    y_ref = F.up_fwd_(y0.clone(), st.hp_tok, Bw, rt, c.r)
    y_f = y0.clone()
    stf = F.up_fwd_fused_(y_f, part, Bw, rt, c.r, spec.s_out, spec.w, spec.inv_sqrt_dk, want_state=True)
    y_f2 = y0.clone()
    assert F.up_fwd_fused_(y_f2, part, Bw, rt, c.r, spec.s_out, spec.w, spec.inv_sqrt_dk) is None
    torch.cuda.synchronize()
    assert torch.equal(y_f, y_ref), f"fused y differs from the two-launch path: {(y_f.float() - y_ref.float()).abs().max().item()}"
    assert torch.equal(y_f2, y_ref)
    # what the backward reads: h / hp_kmj out of the fused launch, BwT / AT out of moka_weight_shadows -- the bits of moka_cross_fwd
    assert torch.equal(stf.h, st.h), "h"
    assert torch.equal(stf.hp_kmj, st.hp_kmj), "hp_kmj"
    BwT, AT = F.weight_shadows(Bw, A, c.r)
    assert torch.equal(BwT, st.BwT) and torch.equal(AT, st.AT)
    BwT_only, none = F.weight_shadows(Bw, None, c.r)
    assert none is None and torch.equal(BwT_only, st.BwT)
    st2 = F.cross_fwd(part, rt, c.r, spec.s_out, spec.w, spec.inv_sqrt_dk, Bw=Bw, A=A, want_tok=False)
    assert st2.hp_tok is None
    for name in ("h", "hp_kmj", "BwT", "AT"):
        assert torch.equal(getattr(st, name), getattr(st2, name)), name
    return y_f, y0


SMALL = [n for n in C.case_names(include_errors=False) if not C.get_case(n).big]
BIG = [n for n in C.case_names(include_errors=False) if C.get_case(n).big]


@pytest.mark.parametrize("name", SMALL + BIG)
def test_fused_equals_two_launches_on_the_golden_cases(name):
    cd = C.make_case_data(name)
    if cd.masks is None:
        pytest.skip("masks None: plain LoRA branch")
    _fused_vs_two_launches(cd)


@pytest.mark.parametrize("seed", _fuzz_seeds(range(24)))
def test_fused_equals_two_launches_on_random_shapes(seed):
    """Random spans (several samples inside one 128-token block, span boundaries inside 16-token tiles, padding, ragged T, ranks
    below their pad, rank pad 32), with dropout in the producer."""
    _fused_vs_two_launches(_random_case(seed, ranks=(4, 8, 16, 16, 16, 24, 32, 48, 64)), p=0.1 if seed % 2 else 0.0)


@pytest.mark.parametrize("variant,r,n_q", [("avt", 16, 500), ("vt", 16, 300), ("avt", 32, 450), ("avt", 16, 600), ("vt", 16, 1100), ("avt", 64, 200),
                                            ("vt", 48, 700)])
def test_fused_long_question_spans(variant, r, n_q):
    """Keys far beyond one chunk of 64: the running softmax over key chunks inside the y kernel."""
    _fused_vs_two_launches(_long_question_case(f"fused_longq_{variant}_{r}_{n_q}", variant, r, n_q))


@pytest.mark.parametrize("shape", [(4096, 4096), (11008, 4096), (4096, 11008)])
def test_fused_full_size_seq2048_against_the_oracle(shape):
    """Llama-2-7B widths, 2 x 2048 tokens (BASELINE configs[2] layout): bit-identical to the two-launch path and within the
    north-star tolerance of the fp64 oracle."""
    d_in, d_out = shape
    name = f"fused_full_{d_in}_{d_out}"
    C._CASES[name] = dict(variant="avt", B=2, S=2048, d_in=d_in, d_out=d_out, r=16, alpha=16.0, w=1.0,
                          layouts=[C.synthetic_sequence_layout(2048)] * 2, seed=77, big=True)
    cd = C.make_case_data(name)
    y_f, y0 = _fused_vs_two_launches(cd, p=0.0)
    c = cd.case
    spec, rt, ort = _spec_and_routing(cd, _dev())
    bf = torch.bfloat16
    rb = lambda t_: t_.to(bf).double()      # noqa: E731
    yo, _ = O.adapter_forward(rb(cd.x), y0.reshape(c.B, c.S, c.d_out).double().cpu(), [rb(a) for a in cd.A], rb(cd.Bw), ort,
                              spec.s_in, spec.s_out, spec.w, c.r)
    y_exact = yo.reshape(-1, c.d_out).to(bf)
    assert rel(y_f, y_exact) < TOL_BF16
    assert ulp_bf16_diff(y_f, y_exact, y0) <= 1.0 + 1e-6


def _group_fused(cfg):
    from moka_amd import _lib
    from moka_amd import functional as F
    from moka_amd.functional import AdapterSpec
    dev = _dev()
    bf = torch.bfloat16
    cds = _group_data(cfg["variant"], cfg["B"], cfg["S"], cfg["d_in"], cfg["d_outs"], cfg["r"], 777, cfg.get("layouts"))
    G = len(cds)
    spec, rt, _ = _spec_and_routing(cds[0], dev)
    if not _lib.up_fwd_fused_ok(spec.r):
        pytest.skip("no fused launch at this rank")
    T = cfg["B"] * cfg["S"]
    x2 = cds[0].x.reshape(T, -1).to(dev, bf).contiguous()
    As = [[a.to(dev, bf).contiguous() for a in cd.A] for cd in cds]
    Bws = [cd.Bw.to(dev, bf).contiguous() for cd in cds]
    parts = F.down_fwd_group(x2, As, rt, spec.r, spec.s_in, cfg["p"], [991 + 7 * g for g in range(G)])
    sts = F.cross_fwd_group(parts, rt, spec.r, spec.s_out, spec.w, spec.inv_sqrt_dk, Bws, As)
    y0 = [torch.randn(T, d, device=dev).to(bf) for d in cfg["d_outs"]]
    y_ref = [y.clone() for y in y0]
    F.up_fwd_group_(y_ref, [st.hp_tok for st in sts], Bws, rt, spec.r)
    y_f = [y.clone() for y in y0]
    stf = F.up_fwd_fused_group_(y_f, parts, Bws, rt, spec.r, spec.s_out, spec.w, spec.inv_sqrt_dk, want_state=True)
    BwTs, ATs = F.weight_shadows_group(Bws, As, spec.r)
    # ... and the single-projection fused launches
    y_1 = [y.clone() for y in y0]
    for g in range(G):
        F.up_fwd_fused_(y_1[g], parts[g], Bws[g], rt, spec.r, spec.s_out, spec.w, spec.inv_sqrt_dk)
    torch.cuda.synchronize()
    for g in range(G):
        assert torch.equal(y_f[g], y_ref[g]), f"group member {g}"
        assert torch.equal(y_1[g], y_ref[g]), f"single launch {g}"
        assert torch.equal(stf[g].h, sts[g].h) and torch.equal(stf[g].hp_kmj, sts[g].hp_kmj), f"state of member {g}"
        assert torch.equal(BwTs[g], sts[g].BwT) and torch.equal(ATs[g], sts[g].AT), f"shadows of member {g}"
    sts2 = F.cross_fwd_group(parts, rt, spec.r, spec.s_out, spec.w, spec.inv_sqrt_dk, Bws, As, want_tok=False)
    for a_, b_ in zip(sts, sts2):
        for name in ("h", "hp_kmj", "BwT", "AT"):
            assert torch.equal(getattr(a_, name), getattr(b_, name)), name


@pytest.mark.parametrize("cfg", [
    dict(variant="avt", B=2, S=2048, d_in=4096, d_outs=(4096, 4096, 4096), r=16, p=0.05),    # q/k/v, Llama-2-7B
    dict(variant="avt", B=2, S=2048, d_in=4096, d_outs=(4096, 1024, 1024), r=16, p=0.0),     # GQA-shaped: members of different width
    dict(variant="avt", B=1, S=2048, d_in=4096, d_outs=(11008, 11008), r=16, p=0.05),        # gate/up
    dict(variant="vt", B=2, S=2048, d_in=4096, d_outs=(4096, 4096, 4096), r=16, p=0.0),
    dict(variant="avt", B=3, S=700, d_in=1376, d_outs=(352, 96, 1376), r=16, p=0.1),         # ragged
    dict(variant="vt", B=2, S=333, d_in=11008, d_outs=(4096, 160), r=8, p=0.05),
    dict(variant="avt", B=3, S=700, d_in=160, d_outs=(1376, 1376, 1376), r=24, p=0.05),      # rank pad 32
    dict(variant="avt", B=1, S=4096, d_in=5120, d_outs=(5120, 5120, 5120), r=64, p=0.05),    # q/k/v, Llama-2-13B, rank pad 64
    dict(variant="vt", B=2, S=333, d_in=1056, d_outs=(160, 96), r=48, p=0.1),                # rank pad 64, r < pad, ragged
])
def test_fused_group_equals_two_launches(cfg):
    _group_fused(cfg)


@pytest.mark.parametrize("seed", list(range(8)))
def test_fused_random_groups(seed):
    _group_fused(_random_group_cfg(seed))


@pytest.mark.parametrize("r", [8, 24, 64])
def test_weight_shadows_of_many_projections_in_one_launch(r):
    """moka_weight_shadows_batch: 19 projections of different input / output widths (two launches: 16 + 3) against moka_weight_shadows
    per projection -- the same bits."""
    from moka_amd import functional as F
    dev = _dev()
    g = torch.Generator().manual_seed(9)
    widths = [(256, 96), (256, 96), (160, 1376), (704, 64), (32, 32)] * 4
    widths = widths[:19]
    Bws = [torch.randn(do, r, generator=g).to(dev, torch.bfloat16) for _, do in widths]
    As = [[torch.randn(r, di, generator=g).to(dev, torch.bfloat16) for _ in range(3)] for di, _ in widths]
    ref = [F.weight_shadows(Bws[i], As[i], r) for i in range(len(widths))]
    BwTs = [torch.full_like(b, 7.0) for b, _ in ref]
    ATs = [torch.full_like(a, 7.0) for _, a in ref]
    F.weight_shadows_batch_(Bws, As, r, BwTs, ATs)
    for i in range(len(widths)):
        assert torch.equal(BwTs[i], ref[i][0]) and torch.equal(ATs[i], ref[i][1]), i


def test_fused_refuses_what_it_was_not_built_for():
    from moka_amd import _lib
    lib = _lib.load()
    assert lib.moka_up_fwd_fused_ok(16, _lib.MOKA_BF16) == 1 and lib.moka_up_fwd_fused_ok(32, _lib.MOKA_BF16) == 1
    assert lib.moka_up_fwd_fused_ok(64, _lib.MOKA_BF16) == 1 and lib.moka_up_fwd_fused_ok(16, _lib.MOKA_F32) == 0
    assert lib.moka_up_fwd_fused_ok(65, _lib.MOKA_BF16) == 0


@pytest.mark.parametrize("name", ["avt_tiny", "avt_r16_q", "vt_r16_q"])
def test_autograd_node_is_bitwise_the_same_with_and_without_the_fused_forward(name):
    """MokaLinearFn end to end: y, dx and the weight gradients (deterministic mode) of the two forward paths are identical."""
    from moka_amd import functional as F
    from moka_amd.functional import AdapterSpec, moka_linear
    dev = _dev()
    cd = C.make_case_data(name)
    c = cd.case
    spec0, rt, _ = _spec_and_routing(cd, dev)
    bf = torch.bfloat16
    outs = []
    F.set_deterministic(True, device=dev)
    try:
        for fused in (True, False):
            F.FUSE_FORWARD = fused
            spec = AdapterSpec(spec0.r, spec0.s_in, spec0.s_out, spec0.w, spec0.inv_sqrt_dk, 0.1, seed=4711)
            x = cd.x.to(dev, bf).requires_grad_(True)
            A = [a.to(dev, bf).requires_grad_(True) for a in cd.A]
            Bw = cd.Bw.to(dev, bf).requires_grad_(True)
            y = moka_linear(x, cd.W.to(dev, bf), None, Bw, A, rt, spec)
            y.backward(cd.gy.to(dev, bf))
            outs.append([y.detach(), x.grad, Bw.grad] + [a.grad for a in A])
    finally:
        F.FUSE_FORWARD = True
        F.set_deterministic(False, device=dev)
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)
