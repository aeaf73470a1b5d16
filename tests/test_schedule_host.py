"""Host logic of moka_amd/schedule.py and of the pieces a captured step rests on (no GPU: CPU tensors, no launches)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _masks(B, S, q_lo, q_hi, v=(4, 20), a=(24, 36)):
    tok = torch.zeros(B, S, dtype=torch.int64)
    tok[:, v[0]:v[1]] = 1
    tok[:, a[0]:a[1]] = 2
    q = torch.zeros(B, S, dtype=torch.bool)
    q[:, q_lo:q_hi] = True
    return [(tok == m).to(torch.int32).unsqueeze(-1) for m in range(3)] + [q.to(torch.int32).unsqueeze(-1)]


def test_families_cover_every_call_of_a_unit_and_skips_is_exact():
    from moka_amd.schedule import FAMILIES, ScheduleConfig
    calls = {"moka_down_fwd", "moka_cross_fwd", "moka_up_fwd", "moka_up_fwd:fused", "moka_cross_fwd:state", "moka_weight_shadows", "moka_up_bwd", "moka_up_bwd:g",
             "moka_up_bwd:dB", "moka_cross_bwd", "moka_down_bwd:dx", "moka_down_bwd:dA"}
    covered = set().union(*FAMILIES.values())
    assert calls <= covered | {"moka_down_bwd"}            # (the combined call only exists in the in-chain schedule, which splits it when a half is ablated)
    fams_of = {c: [f for f, s in FAMILIES.items() if c in s] for c in calls}
    assert all(len(v) == 1 for v in fams_of.values()), fams_of          # every launch belongs to exactly one family
    cfg = ScheduleConfig(skip=frozenset({"dx"}))
    assert cfg.skips("moka_down_bwd:dx") and not cfg.skips("moka_down_bwd:dA") and not cfg.skips("moka_up_bwd")
    assert not ScheduleConfig().skips("moka_down_bwd:dx") and not ScheduleConfig(skip=frozenset({"none"})).skips("moka_up_fwd:fused")
    dA = ScheduleConfig(skip=frozenset({"dA"}))
    assert dA.skips("moka_up_bwd:dB") and dA.skips("moka_down_bwd:dA") and not dA.skips("moka_up_bwd:g")      # what only the optimizer needs


def test_split_batch_by_sample_keeps_containers_and_non_tensors():
    from moka_amd.schedule import _clone_part, _split_batch, _tensors
    b = {"x": torch.arange(10).reshape(5, 2), "m": torch.arange(5), "flag": "keep"}
    parts = _split_batch(b, [3, 2])
    assert [p["x"].shape[0] for p in parts] == [3, 2] and parts[1]["m"].tolist() == [3, 4] and parts[0]["flag"] == "keep"
    tp = _split_batch((torch.zeros(4, 1), torch.ones(4)), [2, 2])
    assert isinstance(tp[0], tuple) and tp[1][1].tolist() == [1.0, 1.0]
    c = _clone_part(parts[0])
    c["x"].zero_()
    assert int(parts[0]["x"].sum()) > 0 and len(_tensors(c)) == 2


def test_static_routing_holds_a_batch_routing_at_fixed_addresses():
    from moka_amd.routing import MokaRouting, StaticRouting
    S = 64
    a = MokaRouting.from_avt_masks(_masks(2, S, 40, 50))               # 10 key slots
    st = StaticRouting(a)
    assert st.Lk_max >= a.Lk_max + 64 and st.Lk_max % 64 == 0          # default capacity: the batch's span rounded up + one chunk of room
    ptrs = (st.tok_mod.data_ptr(), st.ktok.data_ptr(), st.klen.data_ptr(), st.kslot.data_ptr())
    assert st.struct.Lk_max == st.Lk_max == st.ktok.shape[1] and st.klen.tolist() == a.klen.tolist()
    assert torch.equal(st.ktok[:, :a.Lk_max], a.ktok) and bool((st.ktok[:, a.Lk_max:] == -1).all())
    b = MokaRouting.from_avt_masks(_masks(2, S, 38, 60, v=(2, 10), a=(12, 30)))     # another layout, 22 key slots
    st.load(b)
    assert (st.tok_mod.data_ptr(), st.ktok.data_ptr(), st.klen.data_ptr(), st.kslot.data_ptr()) == ptrs          # same buffers: what a captured launch points at
    assert torch.equal(st.tok_mod, b.tok_mod) and torch.equal(st.kslot, b.kslot) and st.klen.tolist() == [22, 22]
    assert torch.equal(st.ktok[:, :22], b.ktok) and bool((st.ktok[:, 22:] == -1).all())
    with pytest.raises(ValueError, match="capacity"):
        StaticRouting(a, key_capacity=12).load(b)
    with pytest.raises(ValueError, match="captured step was built"):
        st.load(MokaRouting.from_avt_masks(_masks(3, S, 40, 50)))
    dual = _masks(2, S, 40, 50)
    dual[1][:, 45:47] = 1                                               # video tokens that are also text: virtual tokens
    with pytest.raises(ValueError, match="virtual"):
        StaticRouting(MokaRouting.from_avt_masks(dual))


def test_use_routing_overrides_the_mask_compilation_of_this_thread_only():
    import threading
    from moka_amd.routing import GLOBAL_ROUTING_CACHE, MokaRouting, StaticRouting, use_routing
    m1, m2 = _masks(1, 64, 40, 50), _masks(1, 64, 30, 34)
    st = StaticRouting(MokaRouting.from_avt_masks(m1))
    plain = GLOBAL_ROUTING_CACHE.get("avt", m2)
    assert plain is not st and plain.klen.tolist() == [4]
    seen = {}
    with use_routing(st):
        assert GLOBAL_ROUTING_CACHE.get("avt", m2) is st               # whatever masks the adapters are handed
        t = threading.Thread(target=lambda: seen.setdefault("other", GLOBAL_ROUTING_CACHE.get("avt", m2)))
        t.start(); t.join()
        with use_routing(None):
            assert GLOBAL_ROUTING_CACHE.get("avt", m2) is plain
        assert GLOBAL_ROUTING_CACHE.get("avt", m2) is st
    assert GLOBAL_ROUTING_CACHE.get("avt", m2) is plain and seen["other"] is plain


def test_effective_seed_is_the_documented_combination():
    from moka_amd.functional import effective_seed
    seed, e = (0x12345678 << 32) | 0x9abcdef0, (0xfffffff0 << 32) | 0x0f0f0f0f
    lo = (0x9abcdef0 ^ 0x0f0f0f0f) & 0xffffffff
    hi = (0x12345678 + 0xfffffff0) & 0xffffffff
    assert effective_seed(seed, e) == (hi << 32) | lo and effective_seed(seed, 0) == seed
    assert effective_seed(seed, -1 & (2 ** 64 - 1)) == effective_seed(seed, 2 ** 64 - 1)
