"""VT (2-modality) MokA adapter layer on the HIP kernels.

Mirrors the surface of ``VisualText/modified_peft/tuners/lora/layer.py``: ``LoraLayer`` (:33-140,
named adapters kept in ``ModuleDict``s ``lora_A`` / ``lora_B``, ``scaling`` / ``r`` / ``lora_alpha``
dicts), ``Linear`` (:389-681) wrapping the frozen ``base_layer`` with the forward signature
``forward(x, my_text_mask, my_image_mask, question_mask, *args, **kwargs)``.  Parameter names:
``<proj>.base_layer.weight``, ``<proj>.lora_A.{image,text}.weight``, ``<proj>.lora_B.{image,text}.weight``
(``lora_B.image`` is allocated but never used by the forward, exactly like the reference).
"""
from __future__ import annotations

import math
import warnings
from typing import Any, List, Optional, Union

import torch
import torch.nn as nn

from ..functional import AdapterSpec, moka_linear
from ..routing import GLOBAL_ROUTING_CACHE, routing_for_samples
from .. import _lib


class LoraLayer:
    adapter_layer_names = ("lora_A", "lora_B", "lora_embedding_A", "lora_embedding_B")
    other_param_names = ("r", "lora_alpha", "scaling", "lora_dropout")

    def __init__(self, base_layer: nn.Module, **kwargs) -> None:
        self.base_layer = base_layer
        self.r = {}
        self.lora_alpha = {}
        self.scaling = {}
        self.lora_dropout = nn.ModuleDict({})
        self.lora_A = nn.ModuleDict({})
        self.lora_B = nn.ModuleDict({})
        self.lora_embedding_A = nn.ParameterDict({})
        self.lora_embedding_B = nn.ParameterDict({})
        self._disable_adapters = False
        self.merged_adapters: List[str] = []
        self.use_dora = {}
        self.kwargs = kwargs
        base = self.get_base_layer()
        if not isinstance(base, nn.Linear):
            raise ValueError(f"Unsupported layer type {type(base)}")
        self.in_features, self.out_features = base.in_features, base.out_features

    # -- BaseTunerLayer-style helpers ---------------------------------------------------------
    def get_base_layer(self) -> nn.Module:
        base = self
        while hasattr(base, "base_layer"):
            base = base.base_layer
        return base

    @property
    def weight(self) -> torch.Tensor:
        return self.get_base_layer().weight

    @property
    def bias(self):
        return self.get_base_layer().bias

    @property
    def merged(self) -> bool:
        return bool(self.merged_adapters)

    @property
    def disable_adapters(self) -> bool:
        return self._disable_adapters

    @property
    def active_adapter(self):
        return self._active_adapter

    @property
    def active_adapters(self) -> List[str]:
        return [self._active_adapter] if isinstance(self._active_adapter, str) else list(self._active_adapter)

    def enable_adapters(self, enabled: bool) -> None:
        if enabled:
            self.set_adapter(self.active_adapters)
            self._disable_adapters = False
        else:
            for name in self.adapter_layer_names:
                getattr(self, name).requires_grad_(False)
            self._disable_adapters = True

    def set_adapter(self, adapter_names: Union[str, List[str]]) -> None:
        if isinstance(adapter_names, str):
            adapter_names = [adapter_names]
        for layer_name in self.adapter_layer_names:
            for key, layer in getattr(self, layer_name).items():
                layer.requires_grad_(key in adapter_names)
        self._active_adapter = adapter_names

    def _check_forward_args(self, x, *args, **kwargs):
        adapter_names = kwargs.get("adapter_names", None)
        if adapter_names is None:
            return
        if len(x) != len(adapter_names):
            raise ValueError(f"Length of `adapter_names` should be the same as the number of inputs, but got "
                             f"{len(adapter_names)} and {len(x)} respectively.")
        if self.merged:
            raise ValueError("Cannot pass `adapter_names` when there are merged adapters, please call `unmerge_adapter` first.")

    # -- adapter construction (layer.py:93-155) -------------------------------------------------
    def update_layer(self, adapter_name, r, lora_alpha, lora_dropout, init_lora_weights, use_rslora, use_dora: bool = False):
        if r <= 0:
            raise ValueError(f"`r` should be a positive integer value but the value passed is {r}")
        _lib.rank_pad(r) if r <= 64 else (_ for _ in ()).throw(ValueError(f"`r`={r} exceeds the HIP path limit of 64"))
        if use_dora:
            raise ValueError("moka_amd: DoRA is not part of the MokA path")
        self.r[adapter_name] = r
        self.lora_alpha[adapter_name] = lora_alpha
        self.lora_dropout.update(nn.ModuleDict({adapter_name: nn.Dropout(p=lora_dropout) if lora_dropout > 0.0 else nn.Identity()}))
        self.lora_A[adapter_name] = nn.Linear(self.in_features, r, bias=False, dtype=torch.bfloat16)
        self.lora_B[adapter_name] = nn.Linear(r, self.out_features, bias=False, dtype=torch.bfloat16)
        self.scaling[adapter_name] = lora_alpha / math.sqrt(r) if use_rslora else lora_alpha / r
        if isinstance(init_lora_weights, str) and init_lora_weights.lower() not in ("gaussian",):
            raise ValueError(f"Unknown initialization {init_lora_weights=}")
        if init_lora_weights:
            self.reset_lora_parameters(adapter_name, init_lora_weights)
        w = self.get_base_layer().weight
        if w.dtype.is_floating_point:
            self.to(w.device, dtype=w.dtype)          # adapters follow the base weight's dtype/device
        else:
            self.to(w.device)
        self.use_dora[adapter_name] = False
        self.set_adapter(self.active_adapters)

    def reset_lora_parameters(self, adapter_name, init_lora_weights):
        if init_lora_weights is False or adapter_name not in self.lora_A:
            return
        if init_lora_weights is True:
            nn.init.kaiming_uniform_(self.lora_A[adapter_name].weight, a=math.sqrt(5))
        elif init_lora_weights.lower() == "gaussian":
            nn.init.normal_(self.lora_A[adapter_name].weight, std=1 / self.r[adapter_name])
        else:
            raise ValueError(f"Unknown initialization {init_lora_weights=}")
        nn.init.zeros_(self.lora_B[adapter_name].weight)


class Linear(nn.Module, LoraLayer):
    def __init__(self, base_layer, adapter_name: str, r: int = 0, lora_alpha: int = 1, lora_dropout: float = 0.0,
                 fan_in_fan_out: bool = False, is_target_conv_1d_layer: bool = False,
                 init_lora_weights: Union[bool, str] = True, use_rslora: bool = False, use_dora: bool = False, **kwargs) -> None:
        super().__init__()
        LoraLayer.__init__(self, base_layer, **kwargs)
        self.fan_in_fan_out = fan_in_fan_out
        self.attn_weight = kwargs.get("attn_weight", 1)
        self._active_adapter = adapter_name
        self.update_layer(adapter_name, r, lora_alpha=lora_alpha, lora_dropout=lora_dropout,
                          init_lora_weights=init_lora_weights, use_rslora=use_rslora, use_dora=use_dora)
        self.is_target_conv_1d_layer = is_target_conv_1d_layer

    # -- plain-LoRA merge API (layer.py:425-546).  NOT equivalent to the MokA forward: it ignores
    #    the token routing and the cross-modal interaction, exactly like the reference's.
    def get_delta_weight(self, adapter) -> torch.Tensor:
        wA, wB = self.lora_A[adapter].weight, self.lora_B[adapter].weight
        delta = (wB.float() @ wA.float()) * self.scaling[adapter]
        if self.fan_in_fan_out:
            delta = delta.T
        return delta.to(wA.dtype)

    def merge(self, safe_merge: bool = False, adapter_names: Optional[List[str]] = None) -> None:
        names = self.active_adapters if adapter_names is None else adapter_names
        base = self.get_base_layer()
        for name in names:
            if name in self.lora_A and name not in self.merged_adapters:
                new_w = base.weight.data + self.get_delta_weight(name).to(base.weight.dtype)
                if safe_merge and not torch.isfinite(new_w).all():
                    raise ValueError(f"NaNs detected in the merged weights. The adapter {name} seems to be broken")
                base.weight.data = new_w
                self.merged_adapters.append(name)

    def unmerge(self) -> None:
        if not self.merged:
            warnings.warn("Already unmerged. Nothing to do.")
            return
        base = self.get_base_layer()
        while self.merged_adapters:
            name = self.merged_adapters.pop()
            if name in self.lora_A:
                base.weight.data -= self.get_delta_weight(name).to(base.weight.dtype)

    # -- the hot path (layer.py:548-681) ---------------------------------------------------------
    def forward(self, x: torch.Tensor, my_text_mask: Optional[torch.Tensor], my_image_mask: Optional[torch.Tensor],
                question_mask: Optional[torch.Tensor], *args: Any, **kwargs: Any) -> torch.Tensor:
        self._check_forward_args(x, *args, **kwargs)
        adapter_names = kwargs.pop("adapter_names", None)
        if self.disable_adapters:
            if self.merged:
                self.unmerge()
            return self.base_layer(x, *args, **kwargs)
        if adapter_names is not None:
            return self._mixed_batch_forward(x, *args, adapter_names=adapter_names, **kwargs)
        if self.merged:
            return self.base_layer(x, *args, **kwargs)
        if x.numel() == 0:                      # empty batch: the adapter adds nothing to an empty base output
            return self.base_layer(x, *args, **kwargs)
        W, bias, Bw, A, rt, spec = self._plan(x, my_text_mask, my_image_mask, question_mask)
        return moka_linear(x, W, bias, Bw, A, rt, spec)

    def _mixed_batch_forward(self, x: torch.Tensor, *args: Any, adapter_names, **kwargs: Any) -> torch.Tensor:
        """Per-sample adapters in one batch (``layer.py:346-381``; PEFT's inference-time mixed-batch LoRA, not the MokA
        interaction): sample b gets plain LoRA with adapter ``adapter_names[b]`` -- ``__base__`` and unknown names get the
        base output alone.  One adapter pass per distinct name over the whole batch, with a routing in which the other
        samples' tokens belong to no modality (their tiles are skipped), added to the one base output."""
        result = self.base_layer(x, *args, **kwargs)
        if x.numel() == 0:
            return result
        xs = x if x.dim() == 3 else x.unsqueeze(1)
        B_, S_ = xs.shape[0], xs.shape[1]
        for name in sorted(set(adapter_names)):
            if name == "__base__" or name not in self.lora_A.keys():
                continue
            idx = [i for i, item in enumerate(adapter_names) if item == name]
            rt = routing_for_samples(B_, S_, idx, x.device)
            drop = self.lora_dropout[name]
            p = float(drop.p) if (isinstance(drop, nn.Dropout) and self.training) else 0.0
            r = self.r[name]
            spec = AdapterSpec(r, 1.0, [self.scaling[name]], 0.0, 1.0 / math.sqrt(r), dropout_p=p)
            A_, B_w = self.lora_A[name].weight, self.lora_B[name].weight
            if A_.dtype != x.dtype:
                A_, B_w = A_.to(x.dtype), B_w.to(x.dtype)
            result = result + moka_linear(xs, None, None, B_w, [A_], rt, spec).reshape(result.shape)
        return result

    # gradient sinks installed by moka_amd.parallel.attach(): {"B": fp32 view for lora_B['text'], "A": [views for lora_A['text'],
    # lora_A['image']]} of the flat data-parallel gradient buffer (None: ordinary autograd gradients)
    _moka_sinks = None
    _moka_defer = None          # attach(defer_dA=True): callable that takes the dA_m half of the backward off the dependency chain
    _moka_seed_dev = None       # attach(): the device-resident part of the dropout seed (0 live; a captured step rewrites it per replay)
    _moka_shadows = None        # attach(): persistent (BwT, AT) weight shadows of the masked (text + image) plan, rewritten behind every optimizer update

    def _sinks(self, n_adapters: int):
        sk = self._moka_sinks
        return None if sk is None else (sk["B"], sk["A"][:n_adapters])

    def _plan(self, x, my_text_mask, my_image_mask, question_mask, *args, **kwargs):
        """(W, bias, Bw, [A_m], routing, spec) of the adapter path of this call (``layer.py:589-678``), or None when
        the call takes one of the base-layer fallbacks.  Shared with the grouped decoder shim (``moka_amd/decoder.py``)."""
        if self.disable_adapters or self.merged or kwargs.get("adapter_names") is not None:
            return None
        base = self.get_base_layer()
        W = base.weight.T if self.fan_in_fan_out else base.weight
        A_t, A_i, B_t = self.lora_A["text"].weight, None, self.lora_B["text"].weight
        drop = self.lora_dropout["text"]
        p = float(drop.p) if (isinstance(drop, nn.Dropout) and self.training) else 0.0
        if my_text_mask is not None and "image" in self.lora_dropout:
            di = self.lora_dropout["image"]
            pi = float(di.p) if (isinstance(di, nn.Dropout) and self.training) else 0.0
            if pi != p:
                raise ValueError("moka_amd: the text and image adapters must use the same lora_dropout")
        r = self.r["text"]
        if my_text_mask is not None:
            A_i = self.lora_A["image"].weight
            rt = GLOBAL_ROUTING_CACHE.get("vt", [my_text_mask, my_image_mask, question_mask])
            spec = AdapterSpec(r, 1.0, [self.scaling["text"], self.scaling["image"]], self.attn_weight, 1.0 / math.sqrt(r), dropout_p=p,
                               sinks=self._sinks(2), defer=self._moka_defer, shadows=self._moka_shadows, seed_dev=self._moka_seed_dev)
            return (W, base.bias, B_t, [A_t, A_i], rt, spec)
        # masks None (cached decode steps): plain LoRA with the text adapter (layer.py:672-678)
        B_, S_ = (x.shape[0], x.shape[1]) if x.dim() == 3 else (1, x.shape[0])
        rt = GLOBAL_ROUTING_CACHE.plain(B_, S_, x.device, 1)
        spec = AdapterSpec(r, 1.0, [self.scaling["text"]], 0.0, 1.0 / math.sqrt(r), dropout_p=p, sinks=self._sinks(1), defer=self._moka_defer)
        return (W, base.bias, B_t, [A_t], rt, spec)

    def __repr__(self) -> str:
        return "lora." + super().__repr__()
