"""Drop-in for the reference's ``peft_hyper`` package on the MokA path:
``from moka_amd.peft_hyper import LoraConfig, get_peft_model`` (``finetune.py:78``)."""
from .config import LoraConfig, PeftConfig, PeftType, TaskType  # noqa: F401
from .lora import Linear, LoraLayer, LoraModel, mark_only_lora_as_trainable  # noqa: F401
from .peft_model import (MODEL_TYPE_TO_PEFT_MODEL_MAPPING, PeftModel, PeftModelForCausalLM, get_peft_model,  # noqa: F401
                         get_peft_model_state_dict, set_peft_model_state_dict)

__version__ = "0.3.0.dev0+moka_amd"
