"""Dense-mask restatement of the AVT train branch -- TEST INFRASTRUCTURE ONLY (see moka_oracle.py for the rules).

``AudioVisualText/peft_hyper/tuners/lora.py:460-532`` multiplies x by each modality mask, runs EVERY adapter on its masked copy,
lets the video and the audio stream attend to the question rows of the text stream separately, and adds the three streams before
``lora_B0``.  For the data the reference is trained on the three masks are disjoint and this equals the routed formulation of
``moka_oracle.py`` (one adapter per token).  A token that sits in TWO masks is different: it contributes one rank-space row PER
modality stream, each stream's interaction acts on its own row, and the rows are summed afterwards -- not representable by one
modality id per token.  This file states that general form so that the behaviour of the reference on overlapping masks is pinned
(``oracle/make_dual_golden.py`` -> ``tests/golden/avt_dual_modality.npz``); the HIP path and the routed oracle refuse such masks with
a ``ValueError``.
"""
from __future__ import annotations

import math
from typing import Sequence

import torch


def avt_dense_forward(x: torch.Tensor, W: torch.Tensor, A: Sequence[torch.Tensor], Bw: torch.Tensor,
                      masks: Sequence[torch.Tensor], alpha: float, r: int, w: float) -> torch.Tensor:
    """y [B,L,d_out] of lora.py:460-532 with masks = [text, video, audio, question], each {0,1} [B,L,1]; differentiable (torch ops)."""
    dt = x.dtype
    mt, mv, ma, mq = [m.to(dt) for m in masks]
    s = alpha / r
    h = [(x * m) @ a.t() * s for m, a in zip((mt, mv, ma), A)]           # one rank-space stream per modality  (:468-477)
    qrows = h[0] * mq                                                    # (:482)
    out = [h[0]]
    for stream, m in ((h[1], mv), (h[2], ma)):                           # (:485-521)
        new = torch.zeros_like(stream)
        for b in range(x.shape[0]):
            idx = torch.where(mq[b, :, 0] == 1)[0]
            kv = qrows[b, int(idx[0]):int(idx[-1]) + 1]                  # IndexError when the sample has no question token (:489)
            p = torch.softmax(stream[b] @ kv.t() / math.sqrt(r), dim=-1)
            new[b] = stream[b] + w * m[b] * (p @ kv)
        out.append(new)
    return x @ W.t() + (out[0] + out[1] + out[2]) @ Bw.t()               # (:524-530)
