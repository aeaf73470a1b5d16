"""Model wrappers and state-dict helpers with the reference's names
(``AudioVisualText/peft_hyper/peft_model.py:48-324,497-612``, ``mapping.py:129-151``,
``utils/save_and_load.py:19-79``): ``get_peft_model(model, cfg)`` returns a
``PeftModelForCausalLM`` whose parameters are ``base_model.model.<path>.lora_A{i}.weight`` etc."""
from __future__ import annotations

import os
from contextlib import contextmanager

import torch

from .config import WEIGHTS_NAME, LoraConfig, PeftConfig, PeftType, TaskType
from .lora import LoraModel


def get_peft_model_state_dict(model, state_dict=None):
    """Adapter-only state dict: keys containing ``lora_`` (+ biases as configured)."""
    if state_dict is None:
        state_dict = model.state_dict()
    bias = model.peft_config.bias
    if bias == "none":
        out = {k: v for k, v in state_dict.items() if "lora_" in k}
    elif bias == "all":
        out = {k: v for k, v in state_dict.items() if "lora_" in k or "bias" in k}
    elif bias == "lora_only":
        out = {}
        for k, v in state_dict.items():
            if "lora_" in k:
                out[k] = v
                b = k.split("lora_")[0] + "bias"
                if b in state_dict:
                    out[b] = state_dict[b]
    else:
        raise NotImplementedError
    if getattr(model, "modules_to_save", None) is not None:
        for k, v in state_dict.items():
            if any(name in k for name in model.modules_to_save):
                out[k] = v
    return out


def set_peft_model_state_dict(model, peft_model_state_dict):
    model.load_state_dict(peft_model_state_dict, strict=False)
    return model


class PeftModel(torch.nn.Module):
    def __init__(self, model, peft_config: PeftConfig):
        super().__init__()
        self.peft_config = peft_config
        self.config = getattr(model, "config", None)
        self.modules_to_save = None
        self.base_model = LoraModel(peft_config, model)
        if getattr(peft_config, "modules_to_save", None) is not None:
            self.modules_to_save = peft_config.modules_to_save
            for name, p in self.named_parameters():
                if any(m in name for m in self.modules_to_save):
                    p.requires_grad = True
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")

    def save_pretrained(self, save_directory, **kwargs):
        if os.path.isfile(save_directory):
            raise ValueError(f"Provided path ({save_directory}) should be a directory, not a file")
        os.makedirs(save_directory, exist_ok=True)
        torch.save(get_peft_model_state_dict(self, kwargs.get("state_dict", None)),
                   os.path.join(save_directory, WEIGHTS_NAME))
        if self.peft_config.base_model_name_or_path is None:
            self.peft_config.base_model_name_or_path = self.base_model.model.__dict__.get("name_or_path", None)
        mode = self.peft_config.inference_mode
        self.peft_config.inference_mode = True
        self.peft_config.save_pretrained(save_directory)
        self.peft_config.inference_mode = mode

    @classmethod
    def from_pretrained(cls, model, model_id, **kwargs):
        cfg = LoraConfig.from_pretrained(model_id)
        wrapped = cls(model, cfg)
        weights = torch.load(os.path.join(model_id, WEIGHTS_NAME), map_location="cpu")
        return set_peft_model_state_dict(wrapped, weights)

    def print_trainable_parameters(self):
        trainable = total = 0
        for _, p in self.named_parameters():
            n = p.numel() or getattr(p, "ds_numel", 0)
            total += n
            trainable += n if p.requires_grad else 0
        print(f"trainable params: {trainable} || all params: {total} || trainable%: {100 * trainable / total}")

    def __getattr__(self, name: str):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.base_model, name)

    def forward(self, *args, **kwargs):
        return self.get_base_model()(*args, **kwargs)

    @contextmanager
    def disable_adapter(self):
        self.base_model.disable_adapter_layers()
        yield
        self.base_model.enable_adapter_layers()

    def get_base_model(self):
        return self.base_model.model


class PeftModelForCausalLM(PeftModel):
    def __init__(self, model, peft_config: PeftConfig):
        super().__init__(model, peft_config)
        self.base_model_prepare_inputs_for_generation = getattr(self.base_model, "prepare_inputs_for_generation", None)

    def forward(self, input_ids=None, attention_mask=None, inputs_embeds=None, labels=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, **kwargs):
        return self.base_model(input_ids=input_ids, attention_mask=attention_mask, inputs_embeds=inputs_embeds,
                               labels=labels, output_attentions=output_attentions,
                               output_hidden_states=output_hidden_states, return_dict=return_dict, **kwargs)

    def generate(self, **kwargs):
        return self.base_model.generate(**kwargs)

    def prepare_inputs_for_generation(self, *args, **kwargs):
        return self.base_model_prepare_inputs_for_generation(*args, **kwargs)


MODEL_TYPE_TO_PEFT_MODEL_MAPPING = {"CAUSAL_LM": PeftModelForCausalLM, TaskType.CAUSAL_LM: PeftModelForCausalLM}


def get_peft_model(model, peft_config):
    """``mapping.py:129-151``: wrap ``model`` according to ``peft_config.task_type``."""
    peft_config.base_model_name_or_path = model.__dict__.get("name_or_path", None)
    if peft_config.target_modules is None:
        raise ValueError("Please specify `target_modules` in `peft_config`")
    if peft_config.inference_mode:
        peft_config.merge_weights = True
    cls = MODEL_TYPE_TO_PEFT_MODEL_MAPPING.get(peft_config.task_type, PeftModel)
    return cls(model, peft_config)
