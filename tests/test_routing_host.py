"""Host routing (moka_amd/routing.py, what the kernels consume) against the oracle's routing (oracle/moka_oracle.py, the
restatement of the reference's mask handling) on the masks of every golden case -- CPU only, no kernel involved -- plus the
cache and error behaviour of the host side."""
import pytest
import torch

from moka_amd.routing import MOD_NONE, MokaRouting, RoutingCache
from oracle import cases as C
from oracle import moka_oracle as O

MASKED = [n for n, c in C._CASES.items() if not c.get("masks_none") and not c.get("expect") and not n.startswith(("fuzz_", "full_", "longq_", "smoke"))]


def _check(rt: MokaRouting, ro, B, S):
    T = B * S
    tok = rt.tok_mod[:T].reshape(B, S).long()
    exp = torch.where(ro.tok_mod < 0, torch.full_like(ro.tok_mod, MOD_NONE), ro.tok_mod)
    assert torch.equal(tok, exp)
    assert (rt.tok_mod[T:] == MOD_NONE).all()                      # padding the kernels rely on
    for b in range(B):
        n = len(ro.kpos[b])
        assert int(rt.klen[b]) == n
        for j in range(n):
            p = int(ro.kpos[b][j])
            live = bool(ro.kvalid[b][j]) and int(ro.tok_mod[b, p]) >= 0
            assert int(rt.ktok[b, j]) == (b * S + p if live else -1)
            if live:
                assert int(rt.kslot[b * S + p]) == j
        assert (rt.ktok[b, n:] == -1).all()
        # query rows: non-text tokens of a sample that has keys
        q = (tok[b] >= 1) & (tok[b] != MOD_NONE) & (n > 0)
        assert torch.equal(q, ro.is_query[b].bool())
    assert int((rt.kslot >= 0).sum()) == int((rt.ktok >= 0).sum())


@pytest.mark.parametrize("name", MASKED)
def test_host_routing_equals_oracle_routing(name):
    cd = C.make_case_data(name)
    c = cd.case
    if c.variant == "avt":
        rt, ro = MokaRouting.from_avt_masks(cd.masks), O.routing_from_avt_masks(cd.masks)
    else:
        rt, ro = MokaRouting.from_vt_masks(*cd.masks), O.routing_from_vt_masks(*cd.masks)
    _check(rt, ro, c.B, c.S)


def test_errors_follow_the_reference():
    t = torch.ones(1, 6, 1, dtype=torch.int32)
    z = torch.zeros(1, 6, 1, dtype=torch.int32)
    with pytest.raises(IndexError):                                 # AVT sample without a question token (lora.py:489-490)
        MokaRouting.from_avt_masks([t, z, z, z])
    # a token in two modalities: one virtual token per further membership behind the sample's real ones (as the dense-mask reference
    # computes it, lora.py:468-477; tests/test_oracle_golden.py checks the arithmetic) -- round 4 refused these masks
    rt = MokaRouting.from_avt_masks([t, t, z, t])
    assert rt.dup_src is not None and rt.S_real == 6 and rt.S == 6 + 16 and rt.dup_src[0, :6].tolist() == [0, 1, 2, 3, 4, 5]
    assert (rt.tok_mod[:6] == 0).all() and (rt.tok_mod[6:12] == 1).all() and (rt.tok_mod[12:22] == 255).all()
    assert MokaRouting.from_avt_masks([t, z, z, t]).dup_src is None
    with pytest.raises(ValueError):
        MokaRouting.from_vt_masks(torch.ones(1, 4, dtype=torch.bool), torch.ones(1, 4, dtype=torch.bool), torch.zeros(1, 4, dtype=torch.bool))


def test_routing_cache_is_keyed_on_mask_identity_and_version():
    cache = RoutingCache(capacity=2)
    t = torch.tensor([[1, 1, 0, 0, 1, 1]], dtype=torch.bool)
    i = ~t
    q = torch.tensor([[0, 0, 0, 0, 1, 0]], dtype=torch.bool)
    r1 = cache.get("vt", [t, i, q])
    assert cache.get("vt", [t, i, q]) is r1                         # the 7 x n_layers calls of a forward share one routing
    q[0, 5] = True                                                   # in-place edit bumps the version: rebuilt
    r2 = cache.get("vt", [t, i, q])
    assert r2 is not r1 and int(r2.klen[0]) == 2
    assert cache.get("vt", [t.clone(), i, q]) is not r2             # another tensor object: another key
    assert len(cache._items) == 2                                    # capacity honoured (oldest entry dropped)
    p1 = cache.plain(2, 1, "cpu", 3)
    assert cache.plain(2, 1, "cpu", 3) is p1 and p1.Lk_max == 0 and int((p1.tok_mod[:2] == 0).sum()) == 2


def test_routing_cache_with_inference_tensors_views_and_data_rewrites():
    """masks made under torch.inference_mode() have no version counter (the reference works there);
    views of one buffer must not alias; a rewrite through .data is caught by the MOKA_ROUTING_VERIFY fingerprint."""
    cache = RoutingCache(capacity=4)
    with torch.inference_mode():
        t = torch.tensor([[1, 1, 0, 0, 1, 1]], dtype=torch.bool)
        i = ~t
        q = torch.tensor([[0, 0, 0, 0, 1, 0]], dtype=torch.bool)
        r = cache.get("vt", [t, i, q])                              # would raise "Inference tensors do not track version counter"
        assert int(r.klen[0]) == 1 and len(cache._items) == 0      # built on the spot, never cached
    # two views of one buffer: same data_ptr for the first, different offset / stride for the others
    buf = torch.zeros(2, 1, 6, dtype=torch.bool)
    buf[0, 0, 4] = True
    buf[1, 0, 5] = True
    t = torch.tensor([[1, 1, 0, 0, 1, 1]], dtype=torch.bool)
    ra = cache.get("vt", [t, ~t, buf[0]])
    rb = cache.get("vt", [t, ~t, buf[1]])
    assert ra is not rb
    # .data rewrite does not bump the version: stale by default (documented), caught in verify mode
    vc = RoutingCache(capacity=4)
    vc.verify = True
    q = torch.tensor([[0, 0, 0, 0, 1, 0]], dtype=torch.bool)
    i = ~t
    r1 = vc.get("vt", [t, i, q])
    assert vc.get("vt", [t, i, q]) is r1
    q.data[0, 5] = True
    r2 = vc.get("vt", [t, i, q])
    assert r2 is not r1 and int(r2.klen[0]) == 2
