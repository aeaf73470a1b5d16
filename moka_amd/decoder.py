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

``MokaLlamaAttention`` / ``MokaLlamaDecoderLayer`` / ``MokaLlamaStack`` are the rest of the forked block: they thread
the masks to all seven projections and switch them off for cached decode steps (VT
``modified_models/modeling_llama.py:294-358``, AVT ``models/modeling_llama.py:647-654,721-743``).  Everything that
is not an adapted projection (RMSNorm, rotary embedding, causal attention, SwiGLU) is stock PyTorch-ROCm -- the
frozen backbone is not part of the hand-written path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch
from torch import nn
from torch.nn import functional as TF

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
    if len(modules) < 2 or len(modules) > 3 or not all(hasattr(m, "_plan") for m in modules) or x.numel() == 0:
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

    def forward(self, x: torch.Tensor, *mask_args, grouped: bool = True, **kwargs) -> torch.Tensor:
        if grouped:
            gate, up = forward_group([self.gate_proj, self.up_proj], x, *mask_args, **kwargs)
        else:                       # the reference's call pattern: one module call per projection
            gate, up = self.gate_proj(x, *mask_args, **kwargs), self.up_proj(x, *mask_args, **kwargs)
        return self.down_proj(self.act_fn(gate) * up, *mask_args, **kwargs)


# ------------------------------------------------------------------------------------------
# the rest of the forked decoder block
# ------------------------------------------------------------------------------------------
@dataclass
class LlamaDims:
    """Shape of one Llama block (defaults: Llama-2-7B)."""
    hidden: int = 4096
    ff: int = 11008
    n_heads: int = 32
    n_kv_heads: int = 32
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0

    @property
    def head_dim(self) -> int:
        return self.hidden // self.n_heads


def rotary_tables(S: int, head_dim: int, theta: float, device, dtype, offset: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos / sin [S, head_dim] of the rotate-half convention, positions offset .. offset + S - 1."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, device=device, dtype=torch.float32) / head_dim))
    ang = torch.arange(offset, offset + S, device=device, dtype=torch.float32)[:, None] * inv[None, :]
    ang = torch.cat([ang, ang], dim=-1)
    return ang.cos().to(dtype), ang.sin().to(dtype)


def _rotate(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return x * cos + torch.cat([-x[..., h:], x[..., :h]], dim=-1) * sin


def _project(proj: nn.Module, x: torch.Tensor, masks: tuple, merged: Optional[dict]):
    """One adapted projection; with ``merged`` (see ``MokaLlamaStack.merge_for_decode``) a decode step is a single GEMM."""
    if merged is not None:
        Wd, bias = merged[id(proj)]
        return TF.linear(x, Wd, bias)
    return proj(x, *masks)


def _live_masks(mask_args: tuple, past_len: int) -> tuple:
    """The forked decoders hand the masks to the projections for training and for the prefill, and ``None`` for
    every cached decode step (VT ``modeling_llama.py:311-329``; AVT gates the list the same way, ``:647-654``)."""
    if past_len == 0:
        return mask_args
    return tuple(None for _ in mask_args)


def decode_weight(proj: nn.Module) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """(weight, bias) of the ONE dense layer a cached decode step of an adapted projection amounts to:
    without masks only the text adapter acts and there is no interaction, ``y = x W^T + s x A_text^T B^T`` (AVT
    ``lora.py:373-381``, VT ``layer.py:672-678``), i.e. ``W + s B A_text``.  Not the reference's ``merge()``, which folds
    every active adapter (``layer.py:425-546``) and is not equivalent to the masked forward."""
    if hasattr(proj, "lora_A0"):                                   # AVT mirror (weight shared on the module itself)
        W, A, Bw, s = proj.weight, proj.lora_A0.weight, proj.lora_B0.weight, proj.scaling[0]
        if getattr(proj, "fan_in_fan_out", False):
            W = W.T
        bias = proj.bias
    elif hasattr(proj, "lora_A") and "text" in getattr(proj, "lora_A", {}):     # VT mirror (wrapped base layer)
        base = proj.get_base_layer()
        W, A, Bw, s, bias = base.weight, proj.lora_A["text"].weight, proj.lora_B["text"].weight, proj.scaling["text"], base.bias
        if getattr(proj, "fan_in_fan_out", False):
            W = W.T
    else:                                                          # a projection the adapter was not attached to
        return proj.weight, getattr(proj, "bias", None)
    Wd = (W.float() + s * (Bw.float() @ A.float())).to(W.dtype)
    return Wd, bias


class MokaLlamaAttention(nn.Module):
    """Self-attention of the forked block.  ``make_proj(d_in, d_out)`` builds one (adapted) projection; q/k/v run as
    one group on the same normalised hidden states, o_proj on the attention output, all four with the masks."""

    def __init__(self, dims: LlamaDims, make_proj: Callable[[int, int], nn.Module]):
        super().__init__()
        self.dims = dims
        hd = dims.head_dim
        self.q_proj = make_proj(dims.hidden, dims.n_heads * hd)
        self.k_proj = make_proj(dims.hidden, dims.n_kv_heads * hd)
        self.v_proj = make_proj(dims.hidden, dims.n_kv_heads * hd)
        self.o_proj = make_proj(dims.n_heads * hd, dims.hidden)

    def forward(self, x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, *mask_args,
                kv_cache: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, grouped: bool = True, merged: Optional[dict] = None):
        B, S, _ = x.shape
        d = self.dims
        if merged is not None:
            q, k, v = (_project(p_, x, mask_args, merged) for p_ in (self.q_proj, self.k_proj, self.v_proj))
        elif grouped:
            q, k, v = qkv_forward(self, x, *mask_args)
        else:
            q, k, v = self.q_proj(x, *mask_args), self.k_proj(x, *mask_args), self.v_proj(x, *mask_args)
        q = q.view(B, S, d.n_heads, d.head_dim).transpose(1, 2)
        k = k.view(B, S, d.n_kv_heads, d.head_dim).transpose(1, 2)
        v = v.view(B, S, d.n_kv_heads, d.head_dim).transpose(1, 2)
        q, k = _rotate(q, cos, sin), _rotate(k, cos, sin)
        past = 0 if kv_cache is None else kv_cache[0].shape[2]
        if past > 0:
            k, v = torch.cat([kv_cache[0], k], dim=2), torch.cat([kv_cache[1], v], dim=2)
        new_cache = (k, v)
        if d.n_kv_heads != d.n_heads:
            rep = d.n_heads // d.n_kv_heads
            k, v = k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1)
        if past > 0 and S > 1:
            # several new tokens behind a cache (chunked prefill, speculative decoding): causal inside the chunk, offset by the past
            allow = torch.ones(S, past + S, dtype=torch.bool, device=x.device).tril(diagonal=past)
            o = TF.scaled_dot_product_attention(q, k, v, attn_mask=allow)
        else:
            o = TF.scaled_dot_product_attention(q, k, v, is_causal=(S > 1))     # (an empty cache is no cache)
        o = o.transpose(1, 2).reshape(B, S, d.n_heads * d.head_dim)
        return _project(self.o_proj, o, mask_args, merged), new_cache


class MokaLlamaDecoderLayer(nn.Module):
    """pre-norm attention + SwiGLU block with all seven projections adapted and mask-threaded."""

    def __init__(self, dims: LlamaDims, make_proj: Callable[[int, int], nn.Module]):
        super().__init__()
        self.dims = dims
        self.input_layernorm = nn.RMSNorm(dims.hidden, eps=dims.rms_eps)
        self.post_attention_layernorm = nn.RMSNorm(dims.hidden, eps=dims.rms_eps)
        self.self_attn = MokaLlamaAttention(dims, make_proj)

        from types import SimpleNamespace
        self.mlp = MokaLlamaMLP(SimpleNamespace(gate_proj=make_proj(dims.hidden, dims.ff), up_proj=make_proj(dims.hidden, dims.ff),
                                                down_proj=make_proj(dims.ff, dims.hidden), act_fn=nn.SiLU()))

    def forward(self, h: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, *mask_args,
                kv_cache=None, grouped: bool = True, merged: Optional[dict] = None):
        past_len = 0 if kv_cache is None else kv_cache[0].shape[2]
        masks = _live_masks(mask_args, past_len)
        if past_len == 0:
            merged = None                                   # the merged weights stand for the mask-free decode branch only
        a, new_cache = self.self_attn(self.input_layernorm(h), cos, sin, *masks, kv_cache=kv_cache, grouped=grouped, merged=merged)
        h = h + a
        x = self.post_attention_layernorm(h)
        if merged is not None:
            mlp = self.mlp
            m = _project(mlp.down_proj, mlp.act_fn(_project(mlp.gate_proj, x, masks, merged)) * _project(mlp.up_proj, x, masks, merged), masks, merged)
        else:
            m = self.mlp(x, *masks, grouped=grouped)
        return h + m, new_cache


class MokaLlamaStack(nn.Module):
    """L decoder layers on given input embeddings (embedding table, encoders and LM head are out of scope): the harness
    ``bench.py --e2e`` and the layer tests drive.  ``forward`` returns the final hidden states (and the caches)."""

    def __init__(self, dims: LlamaDims, n_layers: int, make_proj: Callable[[int, int], nn.Module]):
        super().__init__()
        self.dims = dims
        self.layers = nn.ModuleList(MokaLlamaDecoderLayer(dims, make_proj) for _ in range(n_layers))

    def merge_for_decode(self) -> None:
        """Pre-compute ``decode_weight`` of all 7 x L projections (inference only; call again after the adapter changed).
        Decode steps (``kv_caches`` given, past > 0) then run one GEMM per projection instead of base GEMM + three adapter
        launches; prefill and training are unaffected."""
        self._merged = {}
        for layer in self.layers:
            a, m = layer.self_attn, layer.mlp
            for p_ in (a.q_proj, a.k_proj, a.v_proj, a.o_proj, m.gate_proj, m.up_proj, m.down_proj):
                with torch.no_grad():
                    self._merged[id(p_)] = decode_weight(p_)

    def unmerge(self) -> None:
        self._merged = None

    def forward(self, h: torch.Tensor, *mask_args, kv_caches: Optional[list] = None, grouped: bool = True,
                on_layer: Optional[Callable[[int], None]] = None):
        S = h.shape[1]
        past = 0 if kv_caches is None else kv_caches[0][0].shape[2]
        cos, sin = rotary_tables(S, self.dims.head_dim, self.dims.rope_theta, h.device, h.dtype, offset=past)
        caches = []
        for i, layer in enumerate(self.layers):
            h, c = layer(h, cos, sin, *mask_args, kv_cache=None if kv_caches is None else kv_caches[i], grouped=grouped,
                         merged=getattr(self, "_merged", None))
            caches.append(c)
            if on_layer is not None:
                on_layer(i)
        return h, caches
