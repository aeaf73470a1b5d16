"""A tiny Llama-shaped module tree (2 decoder layers, the 7 projection names the reference
targets) shared by the surface fixture generator and the surface tests.  TEST INFRASTRUCTURE."""
import torch
import torch.nn as nn


class _Attn(nn.Module):
    def __init__(self, d, dt):
        super().__init__()
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            setattr(self, n, nn.Linear(d, d, bias=False, dtype=dt))


class _Mlp(nn.Module):
    def __init__(self, d, ff, dt):
        super().__init__()
        self.gate_proj = nn.Linear(d, ff, bias=False, dtype=dt)
        self.up_proj = nn.Linear(d, ff, bias=False, dtype=dt)
        self.down_proj = nn.Linear(ff, d, bias=False, dtype=dt)


class _Layer(nn.Module):
    def __init__(self, d, ff, dt):
        super().__init__()
        self.self_attn = _Attn(d, dt)
        self.mlp = _Mlp(d, ff, dt)
        self.input_layernorm = nn.LayerNorm(d, dtype=dt)


class _Cfg:
    model_type = "llama"

    def to_dict(self):
        return {"model_type": "llama"}


class Toy(nn.Module):
    def __init__(self, d=64, ff=96, n_layers=2, dt=torch.float32):
        super().__init__()
        self.embed_tokens = nn.Embedding(50, d, dtype=dt)
        self.layers = nn.ModuleList([_Layer(d, ff, dt) for _ in range(n_layers)])
        self.lm_head = nn.Linear(d, 50, bias=False, dtype=dt)
        self.config = _Cfg()

    def prepare_inputs_for_generation(self, *args, **kwargs):      # the AVT wrapper grabs this attribute
        return kwargs


def make_toy(dt=torch.float32):
    torch.manual_seed(0)
    return Toy(dt=dt)
