"""Decoder-layer shim: the projections of a Llama block that are fed by the same
tensor run as ONE group on the HIP path.

The forked decoders of the reference call the adapted projections one after the other with the same
input and the same masks::

    AVT  models/modeling_llama.py:326-328   q_proj(x, m), k_proj(x, m), v_proj(x, m)
         models/modeling_llama.py:222-224   down_proj(act(gate_proj(x, m)) * up_proj(x, m), m)
    VT   modified_models/modeling_llama.py:251-253 / :152-159   (same, masks passed positionally)

``forward_group`` takes those modules (either mirror: ``moka_amd.peft_hyper.lora.Linear`` or
``moka_amd.modified_peft.layer.Linear``) and returns their outputs; the results are those of calling
the modules one by one (``tests/test_gpu_parity.py::test_group_*``), but x is read once by the G
down-projections and once by the G dA kernels, and the G input gradients are added to dx in one pass
(``include/moka_hip.h``: ``moka_*_group``).  ``MokaLlamaMLP`` / ``qkv_forward`` are the two call sites.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
from torch import nn

from .functional import moka_linear_group


def _compatible(plans) -> bool:
    p0 = plans[0]
    rt0, s0 = p0[4], p0[5]
    for p in plans[1:]:
        rt, s = p[4], p[5]
        if rt is not rt0 or len(p[3]) != len(p0[3]):
            return False
        if (s.r, s.s_in, s.s_out, s.w, s.inv_sqrt_dk, s.dropout_p) != (s0.r, s0.s_in, s0.s_out, s0.w, s0.inv_sqrt_dk, s0.dropout_p):
            return False
    return True


def forward_group(modules: Sequence[nn.Module], x: torch.Tensor, *mask_args, **kwargs) -> List[torch.Tensor]:
    """Outputs of ``[m(x, *mask_args) for m in modules]`` with the adapter path of the group fused.
    Falls back to the per-module calls whenever a module takes one of the reference's non-adapter
    branches (adapters disabled / merged, ``loramethod`` without train/test) or the modules disagree on
    rank, scaling, interaction weight or dropout."""
    if len(modules) < 2 or len(modules) > 3 or not all(hasattr(m, "_plan") for m in modules):
        return [m(x, *mask_args, **kwargs) for m in modules]
    plans = [m._plan(x, *mask_args, **kwargs) for m in modules]
    if any(p is None for p in plans) or not _compatible(plans):
        return [m(x, *mask_args, **kwargs) for m in modules]
    rt = plans[0][4]
    return moka_linear_group(x, [(W, b, Bw, A) for (W, b, Bw, A, _, _) in plans], rt, [p[5] for p in plans])


def qkv_forward(attn: nn.Module, hidden_states: torch.Tensor, *mask_args, **kwargs):
    """``q_proj / k_proj / v_proj`` of an attention block on the same hidden states."""
    return tuple(forward_group([attn.q_proj, attn.k_proj, attn.v_proj], hidden_states, *mask_args, **kwargs))


class MokaLlamaMLP(nn.Module):
    """SwiGLU MLP of the forked decoders with gate/up grouped: ``down(act(gate(x)) * up(x))``, every
    projection receiving the masks (AVT ``modeling_llama.py:202-226``, VT ``modified_models/modeling_llama.py:150-161``).
    Wraps the three (already adapted) projections of an existing MLP module; owns no parameters of its own."""

    def __init__(self, mlp: nn.Module):
        super().__init__()
        self.gate_proj, self.up_proj, self.down_proj = mlp.gate_proj, mlp.up_proj, mlp.down_proj
        self.act_fn = getattr(mlp, "act_fn", nn.SiLU())

    def forward(self, x: torch.Tensor, *mask_args, **kwargs) -> torch.Tensor:
        gate, up = forward_group([self.gate_proj, self.up_proj], x, *mask_args, **kwargs)
        return self.down_proj(self.act_fn(gate) * up, *mask_args, **kwargs)
