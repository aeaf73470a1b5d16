"""CPU oracle for the MokA adapter hot path -- TEST INFRASTRUCTURE ONLY.

This file is a vectorised CPU restatement (torch, fp32 or fp64) of the two reference
adapter layers.  It is *not* product code: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker.  The
product path (``moka_amd``) never imports anything under ``oracle/``.

Parity pin: the reference ships no tests or golden vectors for this path (SURVEY.md
section 4), so the pin is the reference itself run in the build container:
``oracle/make_goldens.py`` imports both reference layers from /root/reference, checks
this restatement against them (forward + all gradients) and commits the resulting
vectors under ``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks the
restatement against those vectors everywhere (no /root/reference needed).

Reference semantics restated here (paths relative to /root/reference):

* AVT, 3 modalities, dense-mask form:
  ``AudioVisualText/peft_hyper/tuners/lora.py:460-532`` (train branch; the 'test'
  prefill branch :385-457 is the same math; decode branch :373-381).
* VT, 2 modalities, gather form:
  ``VisualText/modified_peft/tuners/lora/layer.py:589-671`` (masked path) and
  ``:672-678`` (masks None -> plain LoRA with the 'text' adapter).

Both are expressed through one routed formulation::

    h[t]   = s_in * x[t] @ A[mod(t)]^T                (0 for tokens of no modality)
    K_b    = rows kpos_b of h, zeroed where not kvalid (keys == values)
    h'[t]  = h[t] + w * softmax(h[t] K_b^T / sqrt(d_k)) K_b     for query rows t
    y[t]   = y0[t] + s_out[mod(t)] * h'[t] @ B^T

  AVT: s_in = alpha/r0, s_out = 1, keys = contiguous span first..last question token
       with rows zeroed unless (question AND text), queries = video|audio rows,
       IndexError when a sample has no question token (lora.py:489-490).
  VT:  s_in = 1, s_out = scaling[text|image], keys = exact question index set,
       queries = image rows, samples with no image or no question token skipped
       (layer.py:630-637).

The backward is hand derived (the reference relies on autograd) and is itself verified
against autograd of the reference layers when the goldens are generated.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch


# --------------------------------------------------------------------------------------
# routing: what the reference's masks mean, as plain index data
# --------------------------------------------------------------------------------------
@dataclass
class Routing:
    """Token routing derived from the reference's masks.

    tok_mod  [B,S] int64   modality id of each token (0 = text, 1 = image/video,
                           2 = audio), -1 for tokens in no modality (padding)
    is_query [B,S] bool    rows that receive the cross-modal update
    kpos     list of B int64 tensors: key/value positions inside the sample
    kvalid   list of B bool tensors : key row is h[kpos] (True) or a zero row (False)
    """

    tok_mod: torch.Tensor
    is_query: torch.Tensor
    kpos: List[torch.Tensor]
    kvalid: List[torch.Tensor]
    n_mod: int


def routing_from_avt_masks(modality_mask: Sequence[torch.Tensor]) -> Routing:
    """AVT masks: [text, video, audio, question], each int {0,1} of shape [B,L,1]
    (``AudioVisualText/models/unified_arch.py:159-240``; consumed at ``lora.py:462-468``).

    Keys are the contiguous span between the first and last question token
    (``lora.py:489-491``); inside the span only rows that are question AND text are
    non-zero because K = (h_text * question_mask) (``lora.py:482``).  Raises IndexError
    like the reference when a sample has no question token.
    """
    t, v, a, q = [m.reshape(m.shape[0], m.shape[1]).to(torch.int64) for m in modality_mask]
    if ((t + v + a) > 1).any():
        raise ValueError("modality masks overlap: a token belongs to more than one modality")
    tok_mod = torch.full_like(t, -1)
    tok_mod[t == 1] = 0
    tok_mod[v == 1] = 1
    tok_mod[a == 1] = 2
    is_query = (v == 1) | (a == 1)
    kpos, kvalid = [], []
    for b in range(t.shape[0]):
        idx = torch.where(q[b] == 1)[0]
        if idx.numel() == 0:
            raise IndexError("index 0 is out of bounds for dimension 0 with size 0")
        span = torch.arange(int(idx[0]), int(idx[-1]) + 1)
        kpos.append(span)
        kvalid.append((q[b, span] == 1) & (t[b, span] == 1))
    return Routing(tok_mod, is_query, kpos, kvalid, 3)


def routing_from_vt_masks(text_mask: torch.Tensor, image_mask: torch.Tensor,
                          question_mask: torch.Tensor) -> Routing:
    """VT masks: bool [B,S] (``VisualText/train/train.py:206-231``; consumed at
    ``layer.py:594-669``).  Keys are the exact question index set; a sample with no image
    or no question token is skipped (``layer.py:630-637``); the whole interaction is
    skipped when the batch has no image token (``layer.py:627``).
    """
    t = (text_mask == 1)
    i = (image_mask == 1)
    q = (question_mask == 1)
    if (t & i).any():
        raise ValueError("text and image masks overlap")
    tok_mod = torch.full(t.shape, -1, dtype=torch.int64)
    tok_mod[t] = 0
    tok_mod[i] = 1
    is_query = torch.zeros_like(t)
    kpos, kvalid = [], []
    for b in range(t.shape[0]):
        qi = torch.where(q[b])[0]
        if i[b].any() and qi.numel() > 0:
            is_query[b] = i[b]
            kpos.append(qi)
            kvalid.append(torch.ones(qi.numel(), dtype=torch.bool))
        else:
            kpos.append(torch.zeros(0, dtype=torch.int64))
            kvalid.append(torch.zeros(0, dtype=torch.bool))
    return Routing(tok_mod, is_query, kpos, kvalid, 2)


# --------------------------------------------------------------------------------------
# forward / backward of the routed formulation
# --------------------------------------------------------------------------------------
@dataclass
class Ctx:
    x: torch.Tensor
    A: List[torch.Tensor]
    Bw: torch.Tensor
    routing: Routing
    s_in: float
    s_out: List[float]
    w: float
    d_k: int
    h: torch.Tensor = None
    hp: torch.Tensor = None
    probs: List[Optional[torch.Tensor]] = field(default_factory=list)


def adapter_forward(x: torch.Tensor, y0: torch.Tensor, A: Sequence[torch.Tensor], Bw: torch.Tensor,
                    routing: Routing, s_in: float, s_out: Sequence[float], w: float, d_k: int,
                    dtype: torch.dtype = torch.float64) -> Tuple[torch.Tensor, Ctx]:
    """y = y0 + adapter(x).  x [B,S,d_in], y0 [B,S,d_out], A[m] [r,d_in], Bw [d_out,r].

    All arithmetic in ``dtype`` (fp64 by default: the oracle is the exact answer the fp32-
    accumulating kernels are compared with).
    """
    x = x.to(dtype)
    A = [a.to(dtype) for a in A]
    Bw = Bw.to(dtype)
    Bsz, S, _ = x.shape
    r = A[0].shape[0]
    h = torch.zeros(Bsz, S, r, dtype=dtype)
    for m in range(routing.n_mod):
        sel = routing.tok_mod == m
        if sel.any():
            h[sel] = s_in * (x[sel] @ A[m].t())
    hp = h.clone()
    probs: List[Optional[torch.Tensor]] = []
    c = 1.0 / math.sqrt(d_k)
    for b in range(Bsz):
        kp = routing.kpos[b]
        qrows = torch.where(routing.is_query[b])[0]
        if kp.numel() == 0 or qrows.numel() == 0:
            probs.append(None)
            continue
        K = h[b, kp] * routing.kvalid[b].to(dtype).unsqueeze(-1)
        P = torch.softmax((h[b, qrows] @ K.t()) * c, dim=-1)
        hp[b, qrows] = h[b, qrows] + w * (P @ K)
        probs.append(P)
    scale = torch.zeros(Bsz, S, 1, dtype=dtype)
    for m in range(routing.n_mod):
        scale[routing.tok_mod == m] = s_out[m]
    y = y0.to(dtype) + scale * (hp @ Bw.t())
    return y, Ctx(x, A, Bw, routing, s_in, list(s_out), w, d_k, h, hp, probs)


def adapter_backward(gy: torch.Tensor, ctx: Ctx):
    """Gradients of sum(y * gy) w.r.t. x (adapter part only), A[m], Bw.

    Returns (dx_adapter, [dA_m], dB, dh) -- dx of the frozen base (gy @ W) is not included.
    """
    dtype = ctx.x.dtype
    gy = gy.to(dtype)
    rt = ctx.routing
    Bsz, S, _ = ctx.x.shape
    scale = torch.zeros(Bsz, S, 1, dtype=dtype)
    for m in range(rt.n_mod):
        scale[rt.tok_mod == m] = ctx.s_out[m]
    gs = gy * scale                                     # [B,S,d_out]
    dB = torch.einsum("bso,bsk->ok", gs, ctx.hp)        # [d_out, r]
    ghp = gs @ ctx.Bw                                   # dL/dh'  [B,S,r]
    dh = ghp.clone()
    c = 1.0 / math.sqrt(ctx.d_k)
    for b in range(Bsz):
        P = ctx.probs[b]
        if P is None:
            continue
        kp, kv = rt.kpos[b], rt.kvalid[b].to(dtype).unsqueeze(-1)
        qrows = torch.where(rt.is_query[b])[0]
        K = ctx.h[b, kp] * kv
        q = ctx.h[b, qrows]
        do = ctx.w * ghp[b, qrows]                      # dL/d(P K)
        dP = do @ K.t()
        dS = P * (dP - (P * dP).sum(-1, keepdim=True))  # softmax backward
        dq = c * (dS @ K)
        dK = P.t() @ do + c * (dS.t() @ q)              # value role + key role
        dh[b, qrows] += dq
        dh[b].index_add_(0, kp, dK * kv)
    dA = []
    dx = torch.zeros_like(ctx.x)
    for m in range(rt.n_mod):
        sel = rt.tok_mod == m
        if sel.any():
            dA.append(ctx.s_in * (dh[sel].t() @ ctx.x[sel]))
            dx[sel] = ctx.s_in * (dh[sel] @ ctx.A[m])
        else:
            dA.append(torch.zeros_like(ctx.A[m]))
    return dx, dA, dB, dh


def plain_lora_forward(x, y0, A_text, Bw, s, dtype=torch.float64):
    """Masks None / decode step: y = y0 + s * (x A_text^T) B^T
    (AVT ``lora.py:373-381``; VT ``layer.py:672-678``)."""
    x = x.to(dtype)
    return y0.to(dtype) + s * ((x @ A_text.to(dtype).t()) @ Bw.to(dtype).t())


# --------------------------------------------------------------------------------------
# variant front-ends with the reference's own argument conventions
# --------------------------------------------------------------------------------------
def avt_forward(x, W, A, B0, modality_mask, lora_alpha, r0, blc_weight, bias=None, dtype=torch.float64):
    """AVT ``Linear.forward`` train branch: returns (y, ctx).  ``A`` = [A0, A1, A2]."""
    y0 = torch.nn.functional.linear(x.to(dtype), W.to(dtype), None if bias is None else bias.to(dtype))
    rt = routing_from_avt_masks(modality_mask)
    s = lora_alpha / r0
    return adapter_forward(x, y0, A, B0, rt, s_in=s, s_out=[1.0, 1.0, 1.0], w=blc_weight, d_k=r0, dtype=dtype)


def vt_forward(x, W, A_text, A_image, B_text, text_mask, image_mask, question_mask,
               scaling_text, scaling_image, attn_weight, bias=None, dtype=torch.float64):
    """VT ``Linear.forward`` masked path: returns (y, ctx)."""
    y0 = torch.nn.functional.linear(x.to(dtype), W.to(dtype), None if bias is None else bias.to(dtype))
    rt = routing_from_vt_masks(text_mask, image_mask, question_mask)
    r = A_text.shape[0]
    return adapter_forward(x, y0, [A_text, A_image], B_text, rt, s_in=1.0,
                           s_out=[scaling_text, scaling_image], w=attn_weight, d_k=r, dtype=dtype)
