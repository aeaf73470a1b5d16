"""Adapter configuration objects with the field names of the AVT reference fork
(``AudioVisualText/peft_hyper/utils/config.py`` and ``tuners/lora.py:31-90``), so that
``finetune.py:87-99`` can build the config unchanged and saved ``adapter_config.json`` files
stay interchangeable."""
from __future__ import annotations

import enum
import json
import os
from dataclasses import asdict, dataclass, field
from typing import List, Optional, Union

CONFIG_NAME = "adapter_config.json"
WEIGHTS_NAME = "adapter_model.bin"


class PeftType(str, enum.Enum):
    LORA = "LORA"


class TaskType(str, enum.Enum):
    SEQ_CLS = "SEQ_CLS"
    SEQ_2_SEQ_LM = "SEQ_2_SEQ_LM"
    CAUSAL_LM = "CAUSAL_LM"
    TOKEN_CLS = "TOKEN_CLS"


@dataclass
class PeftConfig:
    peft_type: Optional[Union[str, PeftType]] = None
    base_model_name_or_path: Optional[str] = None
    task_type: Optional[Union[str, TaskType]] = None
    inference_mode: bool = False

    def to_dict(self):
        out = {}
        for k, v in asdict(self).items():
            out[k] = v.value if isinstance(v, enum.Enum) else v
        return out

    def save_pretrained(self, save_directory, **kwargs):
        if os.path.isfile(save_directory):
            raise AssertionError(f"Provided path ({save_directory}) should be a directory, not a file")
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, CONFIG_NAME), "w") as f:
            f.write(json.dumps(self.to_dict(), indent=2, sort_keys=True))

    @classmethod
    def from_pretrained(cls, path, **kwargs):
        cfg_file = os.path.join(path, CONFIG_NAME)
        if not os.path.isfile(cfg_file):
            raise ValueError(f"Can't find config.json at '{path}'")
        with open(cfg_file) as f:
            loaded = json.load(f)
        cfg = cls(**kwargs)
        for k, v in loaded.items():
            if hasattr(cfg, k):
                setattr(cfg, k, v)
        return cfg


@dataclass
class LoraConfig(PeftConfig):
    """Same fields as the reference's ``LoraConfig`` (``tuners/lora.py:31-90``).

    ``r`` keeps the reference's digit encoding: ``r=444`` with ``lora_nums=3`` means three rank-4
    adapters (``lora.py:256-259``).  Ranks >= 10 cannot be written that way; pass a tuple/list
    (``r=(16, 16, 16)``) or an int whose digit count differs from ``lora_nums`` (``r=16,
    lora_nums=3`` -> rank 16 for every modality).
    """
    r: Union[int, List[int]] = 8
    target_modules: Optional[Union[List[str], str]] = None
    lora_alpha: Optional[float] = None
    lora_nums: Optional[int] = None
    blc_alpha: Optional[float] = None
    blc_weight: Optional[float] = None
    lora_dropout: Optional[float] = None
    reserved_modality: Optional[str] = None
    loramethod: Optional[str] = None
    merge_weights: bool = False
    fan_in_fan_out: bool = False
    enable_lora: Optional[List[bool]] = None
    bias: str = "none"
    modules_to_save: Optional[List[str]] = None

    def __post_init__(self):
        self.peft_type = PeftType.LORA


def parse_rank(r, lora_nums: int) -> List[int]:
    """Per-modality ranks from the reference's encoding (see LoraConfig)."""
    if isinstance(r, (list, tuple)):
        ranks = [int(v) for v in r]
        if len(ranks) == 1:
            ranks = ranks * lora_nums
    else:
        digits = [int(ch) for ch in str(int(r))]
        ranks = digits if len(digits) == lora_nums else [int(r)] * lora_nums
    if len(ranks) < lora_nums:
        raise IndexError("list index out of range")       # what lora.py:318 raises for a short digit string
    return ranks[:lora_nums] if len(ranks) > lora_nums else ranks
