"""Tuner + wrappers of the VT fork: ``LoraModel`` / ``MixedModel`` (module replacement,
``modified_peft/tuners/lora/model.py:173-326``, ``tuners/mixed/model.py:48-347``) and
``PeftMixedModel`` (``mixed_model.py:83-285``) as used by ``train.py:559-561``:

    model = PeftMixedModel(model, lora_config, adapter_name='image')
    model.add_adapter('text', lora_config)
    model.set_adapter(['image', 'text'])
"""
from __future__ import annotations

import re
from contextlib import contextmanager
from typing import Any, Dict, List, Union

import torch
import torch.nn as nn

from .config import LoraConfig, PeftConfig
from .layer import Linear, LoraLayer


def _target_matches(cfg: LoraConfig, key: str) -> bool:
    t = cfg.target_modules
    if isinstance(t, str):
        return re.fullmatch(t, key) is not None
    return key in t or any(key.endswith(f".{name}") for name in t)


class LoraModel(nn.Module):
    prefix: str = "lora_"

    def __init__(self, model: nn.Module, config: Union[LoraConfig, Dict[str, LoraConfig]], adapter_name: str) -> None:
        super().__init__()
        self.model = model
        self.targeted_module_names: List[str] = []
        self.peft_config: Dict[str, LoraConfig] = dict(config) if isinstance(config, dict) else {adapter_name: config}
        self.active_adapter: Union[str, List[str]] = adapter_name
        self.inject_adapter(self.model, adapter_name)
        self.model.peft_config = self.peft_config

    @property
    def active_adapters(self) -> List[str]:
        return [self.active_adapter] if isinstance(self.active_adapter, str) else list(self.active_adapter)

    # -- module replacement -----------------------------------------------------------------------
    def inject_adapter(self, model: nn.Module, adapter_name: str) -> None:
        cfg = self.peft_config[adapter_name]
        if cfg.target_modules is None:
            raise ValueError("Please specify `target_modules` in `peft_config`")
        found = False
        for key in [k for k, _ in model.named_modules()]:
            if not _target_matches(cfg, key):
                continue
            found = True
            self.targeted_module_names.append(key)
            parent = model.get_submodule(".".join(key.split(".")[:-1]))
            self._create_and_replace(cfg, adapter_name, model.get_submodule(key), key.split(".")[-1], parent, key)
        if not found:
            raise ValueError(f"Target modules {cfg.target_modules} not found in the base model. "
                             f"Please check the target modules and try again.")
        self._mark_only_adapters_as_trainable(model)
        if cfg.inference_mode:
            for n, p in model.named_parameters():
                if adapter_name in n:
                    p.requires_grad = False

    def _create_and_replace(self, cfg, adapter_name, target, target_name, parent, current_key):
        r = cfg.rank_pattern.get(target_name, cfg.r) if cfg.rank_pattern else cfg.r
        alpha = cfg.alpha_pattern.get(target_name, cfg.lora_alpha) if cfg.alpha_pattern else cfg.lora_alpha
        if isinstance(target, LoraLayer):
            # second adapter on an already wrapped projection (model.py:213-222)
            target.update_layer(adapter_name, r, lora_alpha=alpha, lora_dropout=cfg.lora_dropout,
                                init_lora_weights=cfg.init_lora_weights, use_rslora=cfg.use_rslora, use_dora=cfg.use_dora)
            return
        if not isinstance(target, nn.Linear):
            raise ValueError(f"Target module {target} is not supported. Currently, only `torch.nn.Linear` is supported "
                             f"on the MokA path.")
        new = Linear(target, adapter_name, r=r, lora_alpha=alpha, lora_dropout=cfg.lora_dropout,
                     fan_in_fan_out=cfg.fan_in_fan_out, init_lora_weights=cfg.init_lora_weights,
                     use_rslora=cfg.use_rslora, use_dora=cfg.use_dora, attn_weight=cfg.attn_weight)
        if adapter_name not in self.active_adapters:
            new.requires_grad_(False)
        setattr(parent, target_name, new)
        new.to(target.weight.device)

    def _mark_only_adapters_as_trainable(self, model: nn.Module) -> None:
        for n, p in model.named_parameters():
            if self.prefix not in n:
                p.requires_grad = False

    # -- adapter management -----------------------------------------------------------------------
    def add_adapter(self, adapter_name: str, config: LoraConfig) -> None:
        self.peft_config[adapter_name] = config
        self.inject_adapter(self.model, adapter_name)

    def set_adapter(self, adapter_name: Union[str, List[str]]) -> None:
        for module in self.model.modules():
            if isinstance(module, LoraLayer):
                if module.merged:
                    module.unmerge()
                module.set_adapter(adapter_name)
        self.active_adapter = adapter_name

    def _set_adapter_layers(self, enabled: bool) -> None:
        for module in self.model.modules():
            if isinstance(module, LoraLayer):
                module.enable_adapters(enabled)

    def enable_adapter_layers(self) -> None:
        self._set_adapter_layers(True)

    def disable_adapter_layers(self) -> None:
        self._set_adapter_layers(False)

    def forward(self, *args: Any, **kwargs: Any):
        return self.model.forward(*args, **kwargs)

    def generate(self, *args: Any, **kwargs: Any):
        return self.model.generate(*args, **kwargs)

    def __getattr__(self, name: str):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.model, name)


class MixedModel(LoraModel):
    """Only LoRA-type adapters exist on this path, so the mixed tuner is the LoRA tuner."""


class PeftMixedModel(nn.Module):
    def __init__(self, model: nn.Module, peft_config: PeftConfig, adapter_name: str = "default") -> None:
        super().__init__()
        if not isinstance(peft_config, LoraConfig):
            raise ValueError(f"The provided `peft_type` '{getattr(peft_config, 'peft_type', None)}' is not compatible "
                             f"with the `PeftMixedModel`.")
        self.modules_to_save = None
        self.base_model = MixedModel(model, {adapter_name: peft_config}, adapter_name)
        self.set_modules_to_save(peft_config, adapter_name)
        self.config = getattr(model, "config", {"model_type": "custom"})
        if hasattr(self.base_model, "config") and hasattr(self.base_model.config, "pretraining_tp"):
            self.base_model.config.pretraining_tp = 1

    @property
    def peft_config(self) -> Dict[str, PeftConfig]:
        return self.base_model.peft_config

    @property
    def active_adapter(self):
        return self.base_model.active_adapter

    @property
    def active_adapters(self) -> List[str]:
        return self.base_model.active_adapters

    def get_nb_trainable_parameters(self):
        trainable = total = 0
        for _, p in self.named_parameters():
            n = p.numel() or getattr(p, "ds_numel", 0)
            total += n
            trainable += n if p.requires_grad else 0
        return trainable, total

    def print_trainable_parameters(self):
        t, a = self.get_nb_trainable_parameters()
        print(f"trainable params: {t:,d} || all params: {a:,d} || trainable%: {100 * t / a:.4f}")

    def __getattr__(self, name: str):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.base_model, name)

    def forward(self, *args: Any, **kwargs: Any):
        return self.base_model(*args, **kwargs)

    def generate(self, *args: Any, **kwargs: Any):
        return self.base_model.generate(*args, **kwargs)

    @contextmanager
    def disable_adapter(self):
        try:
            self.base_model.disable_adapter_layers()
            yield
        finally:
            self.base_model.enable_adapter_layers()

    def add_adapter(self, adapter_name: str, peft_config: PeftConfig):
        if not isinstance(peft_config, LoraConfig):
            raise ValueError("only LoRA-type configs are compatible with `PeftMixedModel`")
        try:
            self.base_model.add_adapter(adapter_name, peft_config)
        except Exception:
            self.peft_config.pop(adapter_name, None)
            raise
        self.set_modules_to_save(peft_config, adapter_name)

    def set_modules_to_save(self, peft_config: PeftConfig, adapter_name: str) -> None:
        if getattr(peft_config, "modules_to_save", None) is None:
            return
        self.modules_to_save = set(peft_config.modules_to_save) | (self.modules_to_save or set())
        for n, p in self.named_parameters():
            if any(m in n for m in self.modules_to_save):
                p.requires_grad = True

    def set_adapter(self, adapter_name: Union[str, List[str]]) -> None:
        names = [adapter_name] if isinstance(adapter_name, str) else list(adapter_name)
        missing = set(names) - set(self.peft_config.keys())
        if missing:
            raise ValueError(f"Adapter(s) {sorted(missing)} not found, available adapters: {sorted(self.peft_config.keys())}")
        self.base_model.set_adapter(adapter_name)

    def get_base_model(self):
        return self.base_model.model


def get_peft_model(model, peft_config, adapter_name: str = "default", mixed: bool = False):
    """Both flavours wrap the model the same way on this path (keys ``base_model.model.<path>...``)."""
    peft_config.base_model_name_or_path = model.__dict__.get("name_or_path", None)
    return PeftMixedModel(model, peft_config, adapter_name=adapter_name)
