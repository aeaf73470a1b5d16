"""``LoraConfig`` with the field names of the VT fork (``VisualText/modified_peft/tuners/lora/
config.py:44-311``; the MokA-specific field is ``attn_weight`` :166) so ``train.py:548-556`` builds
it unchanged."""
from __future__ import annotations

import enum
import json
import os
from dataclasses import asdict, dataclass, field
from typing import List, Optional, Union

CONFIG_NAME = "adapter_config.json"


class PeftType(str, enum.Enum):
    LORA = "LORA"


class TaskType(str, enum.Enum):
    SEQ_CLS = "SEQ_CLS"
    SEQ_2_SEQ_LM = "SEQ_2_SEQ_LM"
    CAUSAL_LM = "CAUSAL_LM"
    TOKEN_CLS = "TOKEN_CLS"
    QUESTION_ANS = "QUESTION_ANS"
    FEATURE_EXTRACTION = "FEATURE_EXTRACTION"


@dataclass
class PeftConfig:
    peft_type: Optional[Union[str, PeftType]] = None
    auto_mapping: Optional[dict] = None
    base_model_name_or_path: Optional[str] = None
    revision: Optional[str] = None
    task_type: Optional[Union[str, TaskType]] = None
    inference_mode: bool = False

    def to_dict(self):
        # enums by value; sets (LoraConfig turns a list `target_modules` into a set) as sorted lists, as the reference's
        # save_pretrained does before json.dumps (modified_peft/config.py:67-70)
        def plain(v):
            if isinstance(v, enum.Enum):
                return v.value
            if isinstance(v, (set, frozenset)):
                return sorted(v)
            return v
        return {k: plain(v) for k, v in asdict(self).items()}

    def save_pretrained(self, save_directory, **kwargs):
        if os.path.isfile(save_directory):
            raise AssertionError(f"Provided path ({save_directory}) should be a directory, not a file")
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, CONFIG_NAME), "w") as f:
            f.write(json.dumps(self.to_dict(), indent=2, sort_keys=True))


@dataclass
class LoraConfig(PeftConfig):
    r: int = 8
    target_modules: Optional[Union[List[str], str]] = None
    lora_alpha: int = 8
    lora_dropout: float = 0.0
    fan_in_fan_out: bool = False
    bias: str = "none"
    use_rslora: bool = False
    modules_to_save: Optional[List[str]] = None
    attn_weight: float = 0.5
    init_lora_weights: Union[bool, str] = True
    layers_to_transform: Optional[Union[List[int], int]] = None
    layers_pattern: Optional[Union[List[str], str]] = None
    rank_pattern: Optional[dict] = field(default_factory=dict)
    alpha_pattern: Optional[dict] = field(default_factory=dict)
    megatron_config: Optional[dict] = None
    megatron_core: Optional[str] = "megatron.core"
    loftq_config: Optional[dict] = field(default_factory=dict)
    use_dora: bool = False
    layer_replication: Optional[list] = None

    def __post_init__(self):
        self.peft_type = PeftType.LORA
        if isinstance(self.target_modules, list):
            self.target_modules = set(self.target_modules)
        if isinstance(self.target_modules, str) and self.layers_to_transform is not None:
            raise ValueError("`layers_to_transform` cannot be used when `target_modules` is a str.")
        if isinstance(self.target_modules, str) and self.layers_pattern is not None:
            raise ValueError("`layers_pattern` cannot be used when `target_modules` is a str.")
        if self.use_dora:
            raise ValueError("moka_amd: DoRA is not part of the MokA path")
