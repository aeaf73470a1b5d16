"""moka_amd -- MI355X-native MokA adapter path (hand-written HIP kernels behind the PEFT surface).

Sub-packages mirror the two reference forks:
  moka_amd.peft_hyper     <->  AudioVisualText/peft_hyper      (LoraConfig, get_peft_model, Linear)
  moka_amd.modified_peft  <->  VisualText/modified_peft        (LoraConfig, PeftMixedModel, Linear)
"""
from . import _lib  # noqa: F401
from .routing import MokaRouting, RoutingCache  # noqa: F401

__version__ = "0.1.0"
