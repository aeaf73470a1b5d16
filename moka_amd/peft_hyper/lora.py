"""AVT (3-modality) MokA adapter layer and tuner on the HIP kernels.

Mirrors the public surface of ``AudioVisualText/peft_hyper/tuners/lora.py``:
``LoraLayer`` / ``Linear`` (:248-532), ``LoraModel`` (:93-216), ``mark_only_lora_as_trainable``
(:230-245) -- same constructor arguments, attribute and parameter names (``weight``,
``lora_A0..2.weight``, ``lora_B0.weight``), same forward signature
``forward(x, modality_mask=None)`` -- while the arithmetic runs in ``libmoka_hip.so``.
"""
from __future__ import annotations

import math
import re
from enum import Enum
from typing import List, Optional

import torch
import torch.nn as nn

from .. import _lib
from ..functional import AdapterSpec, moka_linear
from ..routing import GLOBAL_ROUTING_CACHE
from .config import LoraConfig, parse_rank


class LoraLayer:
    def __init__(self, r, lora_alpha, lora_dropout: float, merge_weights: bool, lora_nums: int = 3):
        self.r = parse_rank(r, lora_nums)
        self.lora_alpha = lora_alpha
        self.lora_dropout_p = float(lora_dropout or 0.0)
        self.lora_dropout = nn.Dropout(p=self.lora_dropout_p) if self.lora_dropout_p > 0.0 else (lambda x: x)
        self.merged = False
        self.merge_weights = merge_weights
        self.disable_adapters = False


class Linear(nn.Linear, LoraLayer):
    """Frozen dense layer + per-modality down-projections ``lora_A{i}`` + shared up-projection
    ``lora_B0`` with the rank-r cross-modal interaction in between (reference ``lora.py:277-532``)."""

    def __init__(self, in_features: int, out_features: int, r=0, lora_alpha: int = 1, lora_nums: int = 2,
                 blc_alpha: float = 0.0, blc_weight: float = 0.0, lora_dropout: float = 0.0,
                 reserved_modality="text", loramethod="uni", fan_in_fan_out: bool = False,
                 merge_weights: bool = True, **kwargs):
        nn.Linear.__init__(self, in_features, out_features, **kwargs)
        LoraLayer.__init__(self, r=r, lora_alpha=lora_alpha, lora_dropout=lora_dropout,
                           merge_weights=merge_weights, lora_nums=lora_nums)
        self.loramethod = loramethod
        self.lora_num = lora_nums
        self.blc_alpha = blc_alpha
        self.blc_weight = blc_weight
        self.reserved_modality = reserved_modality
        self.fan_in_fan_out = fan_in_fan_out
        rr = self.r
        self.d_k = rr[0]
        if rr[0] > 0:
            if any(v != rr[0] for v in rr):
                raise ValueError(f"all modality ranks must be equal (the shared lora_B0 has rank {rr[0]}), got {rr}")
            _lib.rank_pad(rr[0])           # ValueError for ranks the kernels do not cover
            for i in range(self.lora_num):
                setattr(self, f"lora_A{i}", nn.Linear(in_features, rr[i], bias=False))
            self.lora_B0 = nn.Linear(rr[0], out_features, bias=False)
            self.scaling = [self.lora_alpha / rr[0]]
            self.weight.requires_grad = False
        self.reset_parameters()
        if fan_in_fan_out:
            self.weight.data = self.weight.data.T

    def reset_parameters(self):
        nn.Linear.reset_parameters(self)
        if hasattr(self, "lora_A0"):
            for i in range(self.lora_num):
                nn.init.kaiming_uniform_(getattr(self, f"lora_A{i}").weight, a=math.sqrt(5))
            nn.init.zeros_(self.lora_B0.weight)

    # -- the hot path -------------------------------------------------------------------------
    # gradient sinks installed by moka_amd.parallel.attach(): {"B": fp32 [d_out, r] view, "A": [fp32 [r, d_in] views]} of the flat
    # data-parallel gradient buffer -- the weight-gradient kernels accumulate into them directly (None: autograd gradients)
    _moka_sinks = None
    _moka_defer = None          # attach(defer_dA=True): callable that takes the dA_m half of the backward off the dependency chain
    _moka_seed_dev = None       # attach(): the device-resident part of the dropout seed (0 live; a captured step rewrites it per replay)
    _moka_shadows = None        # attach(): persistent (BwT, AT) weight shadows of the full-M plan, rewritten behind every optimizer update

    def _sinks(self, n_adapters: int):
        sk = self._moka_sinks
        return None if sk is None else (sk["B"], sk["A"][:n_adapters])

    def _spec(self) -> AdapterSpec:
        # lora_dropout acts on x before every A_m (lora.py:477); one counter-based mask per call
        p = self.lora_dropout_p if self.training else 0.0
        return AdapterSpec(self.d_k, self.scaling[0], [1.0] * self.lora_num, self.blc_weight, 1.0 / math.sqrt(self.d_k), dropout_p=p,
                           sinks=self._sinks(self.lora_num), defer=self._moka_defer, shadows=self._moka_shadows, seed_dev=self._moka_seed_dev)

    def _adapter_weights(self, dtype):
        A = [getattr(self, f"lora_A{i}").weight for i in range(self.lora_num)]
        A = [a if a.dtype == dtype else a.to(dtype) for a in A]
        Bw = self.lora_B0.weight
        return A, (Bw if Bw.dtype == dtype else Bw.to(dtype))

    def _plan(self, x: torch.Tensor, modality_mask: Optional[List[torch.Tensor]] = None):
        """(W, bias, Bw, [A_m], routing, spec) of this call, or None when the layer produces no output
        (``lora.py:532``).  Shared by ``forward`` and by the grouped decoder shim (``moka_amd/decoder.py``)."""
        method = self.loramethod or ""
        W = self.weight.T if self.fan_in_fan_out else self.weight
        A, Bw = self._adapter_weights(x.dtype)
        spec = self._spec()
        if "test" in method and x.size(1) == 1:
            # decode step: only the text adapter, no masks (lora.py:373-381)
            rt = GLOBAL_ROUTING_CACHE.plain(x.shape[0], x.shape[1], x.device, 1)
            return (W, self.bias, Bw, A[:1], rt,
                    AdapterSpec(spec.r, spec.s_in, [1.0], 0.0, spec.inv_sqrt_dk, spec.dropout_p, spec.seed, sinks=self._sinks(1), defer=self._moka_defer))
        if "test" in method or "train" in method:
            # prefill / train: token-routed adapters + cross-modal interaction (lora.py:385-532)
            if modality_mask is None:
                # the reference subscripts the list unconditionally (lora.py:463-466): same exception type, clearer text
                raise TypeError("modality_mask is None: the AVT layer needs [text, video, audio, question] masks unless "
                                "loramethod contains 'test' and the input is a single decode position")
            rt = GLOBAL_ROUTING_CACHE.get("avt", list(modality_mask[:4]))
            return (W, self.bias, Bw, A, rt, spec)
        return None                      # the reference falls off the end of forward (lora.py:532)

    def forward(self, x: torch.Tensor, modality_mask: Optional[List[torch.Tensor]] = None):
        if x.numel() == 0 and ("test" in (self.loramethod or "") or "train" in (self.loramethod or "")):
            # empty batch: the reference's per-sample loops run zero times and the adapter adds nothing (lora.py:485,524)
            return torch.nn.functional.linear(x, self.weight.T if self.fan_in_fan_out else self.weight, self.bias)
        plan = self._plan(x, modality_mask)
        if plan is None:
            return None
        W, bias, Bw, A, rt, spec = plan
        return moka_linear(x, W, bias, Bw, A, rt, spec)


def mark_only_lora_as_trainable(model: nn.Module, bias: str = "none") -> None:
    for n, p in model.named_parameters():
        if "lora_" not in n:
            p.requires_grad = False
    if bias == "none":
        return
    if bias == "all":
        for n, p in model.named_parameters():
            if "bias" in n:
                p.requires_grad = True
    elif bias == "lora_only":
        for m in model.modules():
            if isinstance(m, LoraLayer) and getattr(m, "bias", None) is not None:
                m.bias.requires_grad = True
    else:
        raise NotImplementedError


class LoraModel(nn.Module):
    """Swaps every ``nn.Linear`` whose name ends with a target key for the MokA ``Linear``,
    sharing the frozen weight tensor (reference ``lora.py:125-188``)."""

    def __init__(self, config: LoraConfig, model: nn.Module):
        super().__init__()
        self.peft_config = config
        self.model = model
        self._find_and_replace()
        mark_only_lora_as_trainable(self.model, self.peft_config.bias)
        self.forward = self.model.forward

    def _layer_kwargs(self):
        c = self.peft_config
        return dict(r=c.r, lora_alpha=c.lora_alpha, lora_dropout=c.lora_dropout, lora_nums=c.lora_nums,
                    blc_alpha=c.blc_alpha, blc_weight=c.blc_weight, reserved_modality=c.reserved_modality,
                    loramethod=c.loramethod, fan_in_fan_out=c.fan_in_fan_out,
                    merge_weights=(c.merge_weights or c.inference_mode) and not hasattr(self.model, "hf_device_map"))

    def _find_and_replace(self):
        if getattr(self.model, "is_loaded_in_4bit", False) or getattr(self.model, "is_loaded_in_8bit", False):
            raise ImportError("To use Lora with 8-bit or 4-bit quantization, please install the `bitsandbytes` package. "
                              "You can install it with `pip install bitsandbytes`.")
        targets = self.peft_config.target_modules
        found = False
        kwargs = self._layer_kwargs()
        for key in [k for k, _ in self.model.named_modules()]:
            hit = re.fullmatch(targets, key) if isinstance(targets, str) else any(key.endswith(t) for t in targets)
            if not hit:
                continue
            found = True
            parent = self.model.get_submodule(".".join(key.split(".")[:-1]))
            old = self.model.get_submodule(key)
            if isinstance(old, nn.Linear) and self.peft_config.enable_lora is None:
                new = Linear(old.in_features, old.out_features, bias=old.bias is not None, **kwargs)
                self._replace_module(parent, key.split(".")[-1], new, old)
        if not found:
            raise ValueError(f"Target modules {targets} not found in the base model. "
                             f"Please check the target modules and try again.")

    @staticmethod
    def _replace_module(parent, child_name, new, old):
        setattr(parent, child_name, new)
        new.weight = old.weight                      # shared frozen tensor, not a copy
        if old.bias is not None:
            new.bias = old.bias
        for name, module in new.named_modules():
            if "lora_" in name:
                module.to(old.weight.device)

    def __getattr__(self, name: str):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.model, name)

    @property
    def modules_to_save(self):
        return None

    def get_peft_config_as_dict(self, inference: bool = False):
        cfg = {k: (v.value if isinstance(v, Enum) else v) for k, v in self.peft_config.to_dict().items()}
        if inference:
            cfg["inference_mode"] = True
        return cfg

    def _set_adapter_layers(self, enabled=True):
        for module in self.model.modules():
            if isinstance(module, LoraLayer):
                module.disable_adapters = not enabled

    def enable_adapter_layers(self):
        self._set_adapter_layers(True)

    def disable_adapter_layers(self):
        self._set_adapter_layers(False)
