"""Drop-in for the reference's ``modified_peft`` package on the MokA path:
``from moka_amd.modified_peft import LoraConfig, PeftMixedModel`` (``train.py:51-52``)."""
from .config import LoraConfig, PeftConfig, PeftType, TaskType  # noqa: F401
from .layer import Linear, LoraLayer  # noqa: F401
from .model import LoraModel, MixedModel, PeftMixedModel, get_peft_model  # noqa: F401

__version__ = "0.11.1+moka_amd"
