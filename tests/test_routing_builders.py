"""routing built once per batch from what the data pipeline knows (moka_amd/routing.py:from_vt_batch,
from_segments) equals the routing the mask tensors of the reference give (from_vt_masks / from_avt_masks), and the
mask recipes themselves follow the reference (VT train.py:206-231 restated per sample in the test)."""
import pytest
import torch

from moka_amd.routing import MokaRouting


def _same(a: MokaRouting, b: MokaRouting):
    assert (a.B, a.S, a.Lk_max, a.M) == (b.B, b.S, b.Lk_max, b.M)
    for f in ("tok_mod", "ktok", "klen", "kslot"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f


def _vt_sample_masks(input_ids, labels, image_pad_id):
    """The per-sample recipe of the reference's dataset class, literally (one sample, no padding)."""
    image = input_ids == image_pad_id
    text = input_ids != image_pad_id
    pos = torch.where(image)[0]
    if len(pos) > 0:
        after = torch.arange(len(input_ids)) > pos[-1]
    else:
        after = torch.zeros_like(image)
    return text, image, (~image) & (labels == -100) & after


def test_vt_batch_recipe_matches_per_sample_masks_with_right_padding():
    g = torch.Generator().manual_seed(0)
    IMG, S = 32000, 40
    samples = []
    for n_pre, n_img, n_q, n_ans in [(3, 8, 6, 5), (0, 4, 9, 12), (5, 0, 7, 3), (2, 6, 0, 4)]:
        ids = torch.cat([torch.randint(5, 1000, (n_pre,), generator=g), torch.full((n_img,), IMG),
                         torch.randint(5, 1000, (n_q + n_ans,), generator=g)])
        lab = torch.cat([torch.full((n_pre + n_img + n_q,), -100), ids[n_pre + n_img + n_q:]])
        samples.append((ids, lab))
    B = len(samples)
    input_ids = torch.full((B, S), 2)
    labels = torch.full((B, S), -100)
    att = torch.zeros(B, S, dtype=torch.bool)
    t_m, i_m, q_m = (torch.zeros(B, S, dtype=torch.bool) for _ in range(3))
    for b, (ids, lab) in enumerate(samples):
        n = len(ids)
        input_ids[b, :n], labels[b, :n], att[b, :n] = ids, lab, True
        t, i, q = _vt_sample_masks(ids, lab, IMG)
        t_m[b, :n], i_m[b, :n], q_m[b, :n] = t, i, q            # the collator pads the masks with False
    _same(MokaRouting.from_vt_batch(input_ids, labels, IMG, att), MokaRouting.from_vt_masks(t_m, i_m, q_m))
    # sample 2 has no image token: no question, no interaction
    rt = MokaRouting.from_vt_batch(input_ids, labels, IMG, att)
    assert rt.klen.tolist() == [6, 9, 0, 0]


def _masks_from_segments(segments, S, variant, pad):
    B = len(segments)
    t, v, a, q = (torch.zeros(B, S, dtype=torch.int32) for _ in range(4))
    for b, segs in enumerate(segments):
        n = sum(l for _, l in segs)
        at = S - n if pad == "left" else 0
        for kind, l in segs:
            if kind in ("t", "q"):
                t[b, at:at + l] = 1
            if kind == "q":
                q[b, at:at + l] = 1
            if kind == "v":
                v[b, at:at + l] = 1
            if kind == "a":
                a[b, at:at + l] = 1
            at += l
    return t, v, a, q


def test_segments_match_avt_masks_left_padded():
    segs = [[("t", 5), ("v", 32), ("t", 2), ("a", 16), ("q", 9), ("t", 7)],
            [("v", 10), ("a", 10), ("t", 1), ("q", 1), ("t", 30)],
            [("t", 3), ("q", 4), ("p", 2), ("v", 8), ("t", 5)]]
    S = 80
    t, v, a, q = _masks_from_segments(segs, S, "avt", "left")
    ref = MokaRouting.from_avt_masks([m.reshape(len(segs), S, 1) for m in (t, v, a, q)])
    _same(MokaRouting.from_segments(segs, S, "cpu", "avt", "left"), ref)


def test_segments_with_a_split_question_keep_the_gap_as_zero_keys():
    """Two question runs with text between them: AVT keys are the contiguous span, the gap rows are zero keys that
    still enter the softmax (lora.py:482,489-491)."""
    segs = [[("t", 2), ("q", 3), ("t", 4), ("q", 2), ("v", 6)]]
    S = 20
    t, v, a, q = _masks_from_segments(segs, S, "avt", "left")
    ref = MokaRouting.from_avt_masks([m.reshape(1, S, 1) for m in (t, v, a, q)])
    got = MokaRouting.from_segments(segs, S, "cpu", "avt", "left")
    _same(got, ref)
    assert got.klen.tolist() == [9] and (got.ktok[0] < 0).sum().item() == 4


def test_segments_match_vt_masks_right_padded():
    segs = [[("t", 4), ("v", 8), ("q", 6), ("t", 9)], [("t", 2), ("q", 5), ("t", 3)], [("v", 4), ("t", 10)]]
    S = 32
    t, v, a, q = _masks_from_segments(segs, S, "vt", "right")
    ref = MokaRouting.from_vt_masks(t.bool(), v.bool(), q.bool())
    _same(MokaRouting.from_segments(segs, S, "cpu", "vt", "right"), ref)


def test_segments_errors_follow_the_reference():
    with pytest.raises(IndexError):
        MokaRouting.from_segments([[("t", 4), ("v", 4)]], 16, "cpu", "avt")       # no question token (lora.py:489-490)
    with pytest.raises(ValueError):
        MokaRouting.from_segments([[("t", 40)]], 16, "cpu", "avt")
    with pytest.raises(ValueError):
        MokaRouting.from_segments([[("a", 4), ("q", 2)]], 16, "cpu", "vt")
