"""Deterministic parity cases for the MokA adapter path -- TEST INFRASTRUCTURE ONLY.

Every case is a pure function of its name: shapes, seeds, token layout.  All floating
point inputs are bf16-representable, so the very same case feeds the fp64 oracle, the
reference run in fp32/bf16 and the bf16 HIP kernels without any input rounding.

Used by ``oracle/make_goldens.py`` (generation, needs /root/reference), by ``tests/``
(regeneration of the inputs; checked against checksums stored in the golden files) and
by ``bench.py`` (the synthetic workload of SURVEY.md section 8(d)).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

TEXT, VIS, AUD = 0, 1, 2


@dataclass
class Case:
    name: str
    variant: str                      # "avt" | "vt"
    B: int
    S: int
    d_in: int
    d_out: int
    r: int
    alpha: float
    w: float                          # blc_weight (avt) / attn_weight (vt)
    layouts: List[List[Tuple[str, int]]]   # per sample: [(kind, length), ...]
    seed: int = 1234
    expect: Optional[str] = None      # "IndexError" for the AVT no-question case
    masks_none: bool = False          # decode / plain-LoRA fallback
    gap_question: bool = False        # AVT: punch a hole into the question span
    big: bool = False                 # store strided samples instead of full tensors

# layout kinds:  p = padding (no modality), t = text, v = video/image, a = audio,
#                q = question text (text AND question), Q = question token that is NOT text


def _avt_tiny(**kw):
    base = dict(variant="avt", B=2, S=48, d_in=64, d_out=96, r=4, alpha=16.0, w=1.0,
                layouts=[[("p", 5), ("t", 4), ("v", 12), ("t", 2), ("a", 9), ("q", 7), ("t", 9)],
                         [("t", 3), ("v", 16), ("a", 8), ("t", 1), ("q", 11), ("t", 9)]])
    base.update(kw)
    return base


def _vt_tiny(**kw):
    base = dict(variant="vt", B=2, S=48, d_in=64, d_out=96, r=8, alpha=16.0, w=0.05,
                layouts=[[("t", 6), ("v", 16), ("q", 9), ("t", 12), ("p", 5)],
                         [("t", 2), ("v", 8), ("t", 3), ("v", 8), ("q", 14), ("t", 13)]])
    base.update(kw)
    return base


def _syn_layout(S: int, nv: int, na: int, nq: int, pre: int = 16, mid: int = 16):
    rest = S - (pre + nv + mid + na + nq)
    assert rest >= 0
    return [("t", pre), ("v", nv), ("t", mid), ("a", na), ("q", nq), ("t", rest)]


_CASES: Dict[str, dict] = {
    # ---------------- AVT (3 modalities, peft_hyper) ----------------
    "avt_tiny": _avt_tiny(),
    "avt_tiny_w0": _avt_tiny(w=0.0, seed=11),
    "avt_tiny_w005": _avt_tiny(w=0.05, seed=12),
    "avt_q1": _avt_tiny(seed=13, layouts=[[("t", 8), ("v", 12), ("a", 9), ("q", 1), ("t", 18)],
                                          [("p", 20), ("v", 10), ("a", 10), ("q", 1), ("t", 7)]]),
    "avt_gap": _avt_tiny(seed=14, gap_question=True),
    "avt_missing_modality": _avt_tiny(seed=15, layouts=[[("t", 4), ("a", 20), ("q", 8), ("t", 16)],
                                                        [("p", 10), ("t", 4), ("v", 20), ("q", 8), ("t", 6)]]),
    "avt_question_first": _avt_tiny(seed=16, layouts=[[("q", 6), ("v", 20), ("a", 10), ("t", 12)],
                                                      [("p", 3), ("q", 9), ("t", 4), ("a", 16), ("v", 16)]]),
    "avt_nontext_question": _avt_tiny(seed=17, layouts=[[("t", 4), ("v", 12), ("q", 3), ("Q", 2), ("q", 3), ("a", 8), ("t", 16)],
                                                        [("t", 6), ("v", 10), ("a", 10), ("q", 12), ("t", 10)]]),
    "avt_noquestion": _avt_tiny(seed=18, expect="IndexError",
                                layouts=[[("t", 10), ("v", 20), ("a", 10), ("t", 8)],
                                         [("t", 3), ("v", 16), ("a", 8), ("t", 1), ("q", 11), ("t", 9)]]),
    "avt_decode": dict(variant="avt", B=3, S=1, d_in=64, d_out=96, r=4, alpha=16.0, w=1.0, masks_none=True,
                       layouts=[[("t", 1)]] * 3, seed=19),
    "avt_r8": dict(variant="avt", B=2, S=64, d_in=128, d_out=64, r=8, alpha=16.0, w=1.0, seed=20,
                   layouts=[_syn_layout(64, 16, 8, 6, 4, 4), [("p", 9)] + _syn_layout(55, 12, 12, 5, 3, 2)]),
    "avt_r16_q": dict(variant="avt", B=1, S=256, d_in=4096, d_out=4096, r=16, alpha=16.0, w=1.0, seed=21, big=True,
                      layouts=[_syn_layout(256, 64, 32, 24, 8, 8)]),
    "avt_r16_down": dict(variant="avt", B=2, S=128, d_in=11008, d_out=4096, r=16, alpha=16.0, w=1.0, seed=22, big=True,
                         layouts=[_syn_layout(128, 32, 16, 12, 4, 4), [("p", 17)] + _syn_layout(111, 32, 16, 9, 4, 4)]),
    "avt_r16_up": dict(variant="avt", B=1, S=128, d_in=4096, d_out=11008, r=16, alpha=16.0, w=1.0, seed=23, big=True,
                       layouts=[_syn_layout(128, 32, 16, 12, 4, 4)]),
    "avt_r4_q": dict(variant="avt", B=1, S=256, d_in=4096, d_out=4096, r=4, alpha=16.0, w=1.0, seed=24, big=True,
                     layouts=[_syn_layout(256, 64, 32, 24, 8, 8)]),
    "avt_r64": dict(variant="avt", B=1, S=192, d_in=512, d_out=384, r=64, alpha=16.0, w=1.0, seed=25, big=True,
                    layouts=[_syn_layout(192, 48, 24, 16, 8, 8)]),
    # ---------------- VT (2 modalities, modified_peft) ----------------
    "vt_tiny": _vt_tiny(),
    "vt_tiny_w0": _vt_tiny(w=0.0, seed=31),
    "vt_tiny_w1": _vt_tiny(w=1.0, seed=32),
    "vt_noimage_sample": _vt_tiny(seed=33, layouts=[[("t", 10), ("q", 9), ("t", 24), ("p", 5)],
                                                    [("t", 2), ("v", 16), ("q", 14), ("t", 16)]]),
    "vt_noquestion_sample": _vt_tiny(seed=34, layouts=[[("t", 6), ("v", 16), ("t", 21), ("p", 5)],
                                                       [("t", 2), ("v", 16), ("q", 14), ("t", 16)]]),
    "vt_noimage_batch": _vt_tiny(seed=35, layouts=[[("t", 10), ("q", 9), ("t", 24), ("p", 5)],
                                                   [("t", 20), ("q", 14), ("t", 14)]]),
    "vt_q1": _vt_tiny(seed=36, layouts=[[("t", 6), ("v", 16), ("q", 1), ("t", 20), ("p", 5)],
                                        [("v", 32), ("q", 1), ("t", 15)]]),
    "vt_nontext_question": _vt_tiny(seed=37, layouts=[[("t", 6), ("v", 16), ("q", 4), ("Q", 3), ("q", 2), ("t", 12), ("p", 5)],
                                                      [("t", 2), ("v", 16), ("q", 14), ("t", 16)]]),
    "vt_none": dict(variant="vt", B=2, S=5, d_in=64, d_out=96, r=8, alpha=16.0, w=0.05, masks_none=True,
                    layouts=[[("t", 5)]] * 2, seed=38),
    "vt_cfg1_q": dict(variant="vt", B=1, S=256, d_in=4096, d_out=4096, r=8, alpha=16.0, w=0.05, seed=39, big=True,
                      layouts=[[("t", 12), ("v", 32), ("q", 24), ("t", 188)]]),
    "vt_r16_q": dict(variant="vt", B=2, S=256, d_in=4096, d_out=4096, r=16, alpha=16.0, w=0.05, seed=40, big=True,
                     layouts=[[("t", 12), ("v", 64), ("q", 24), ("t", 156)],
                              [("t", 5), ("v", 32), ("t", 3), ("v", 32), ("q", 17), ("t", 120), ("p", 47)]]),
    "vt_r16_down": dict(variant="vt", B=1, S=128, d_in=11008, d_out=4096, r=16, alpha=16.0, w=0.05, seed=41, big=True,
                        layouts=[[("t", 6), ("v", 32), ("q", 14), ("t", 60), ("p", 16)]]),
}


def case_names(variant: Optional[str] = None, include_errors: bool = True) -> List[str]:
    out = []
    for k, v in _CASES.items():
        if variant and v["variant"] != variant:
            continue
        if not include_errors and v.get("expect"):
            continue
        out.append(k)
    return out


def get_case(name: str) -> Case:
    return Case(name=name, **_CASES[name])


def _bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


@dataclass
class CaseData:
    case: Case
    x: torch.Tensor          # [B,S,d_in] fp32 (bf16-representable)
    W: torch.Tensor          # [d_out,d_in]
    A: List[torch.Tensor]    # M x [r,d_in]
    Bw: torch.Tensor         # [d_out,r]
    gy: torch.Tensor         # [B,S,d_out] upstream gradient
    tok_mod: torch.Tensor    # [B,S] int64, -1 = none
    question: torch.Tensor   # [B,S] bool
    masks: Optional[list]    # reference-format masks (None for the masks_none cases)


def build_layout(layout: List[Tuple[str, int]], S: int, gap_question: bool = False):
    tok_mod = torch.full((S,), -1, dtype=torch.int64)
    question = torch.zeros(S, dtype=torch.bool)
    pos = 0
    for kind, n in layout:
        sl = slice(pos, pos + n)
        if kind == "t":
            tok_mod[sl] = TEXT
        elif kind == "v":
            tok_mod[sl] = VIS
        elif kind == "a":
            tok_mod[sl] = AUD
        elif kind == "q":
            tok_mod[sl] = TEXT
            question[sl] = True
        elif kind == "Q":
            question[sl] = True       # question token that belongs to no modality
        elif kind != "p":
            raise ValueError(kind)
        pos += n
    assert pos == S, (pos, S)
    if gap_question:
        idx = torch.where(question)[0]
        if idx.numel() >= 3:
            question[idx[idx.numel() // 2]] = False
    return tok_mod, question


def make_case_data(name: str) -> CaseData:
    c = get_case(name)
    g = torch.Generator().manual_seed(c.seed)
    M = 3 if c.variant == "avt" else 2
    x = _bf16_round(torch.randn(c.B, c.S, c.d_in, generator=g))
    W = _bf16_round(torch.randn(c.d_out, c.d_in, generator=g) * 0.02)
    bound = 1.0 / math.sqrt(c.d_in)   # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    A = [_bf16_round((torch.rand(c.r, c.d_in, generator=g) * 2 - 1) * bound) for _ in range(M)]
    Bw = _bf16_round(torch.randn(c.d_out, c.r, generator=g) * 0.02)   # reference init is 0; randomised so grads are non-trivial
    gy = _bf16_round(torch.randn(c.B, c.S, c.d_out, generator=g))
    tm, qq = [], []
    for b in range(c.B):
        t, q = build_layout(c.layouts[b], c.S, c.gap_question)
        tm.append(t)
        qq.append(q)
    tok_mod = torch.stack(tm)
    question = torch.stack(qq)
    if c.masks_none:
        masks = None
    elif c.variant == "avt":
        masks = [(tok_mod == m).to(torch.int32).unsqueeze(-1) for m in range(3)] + [question.to(torch.int32).unsqueeze(-1)]
    else:
        masks = [tok_mod == TEXT, tok_mod == VIS, question]
    return CaseData(c, x, W, A, Bw, gy, tok_mod, question, masks)


def synthetic_sequence_layout(S: int = 2048):
    """[16 text][256 image/video][16 text][128 audio][64 question][rest text]."""
    scale = S / 2048.0
    return _syn_layout(S, int(256 * scale), int(128 * scale), int(64 * scale), int(16 * scale), int(16 * scale))
