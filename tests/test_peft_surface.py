"""CPU: the PEFT-shaped host surface of moka_amd (peft_hyper / modified_peft mirrors) produces
exactly the parameter surface of the reference wrappers -- state-dict keys, shapes, dtypes,
requires_grad -- recorded in tests/golden/surface.json by oracle/make_surface_fixture.py from
the real reference.  Also the reference's error behaviour at construction time."""
import json
import os

import pytest
import torch
import torch.nn as nn

from oracle.toy_model import make_toy

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "surface.json")))
PROJS = "q_proj,k_proj,v_proj,o_proj,gate_proj,down_proj,up_proj".split(",")


def _surface(model):
    req = {n: p.requires_grad for n, p in model.named_parameters()}
    return {k: {"shape": list(v.shape), "dtype": str(v.dtype), "requires_grad": req.get(k)} for k, v in model.state_dict().items()}


def _avt_model(**over):
    from moka_amd.peft_hyper import LoraConfig, get_peft_model
    kw = dict(task_type="CAUSAL_LM", target_modules=PROJS, inference_mode=False, r=444, loramethod="train",
              reserved_modality=None, lora_alpha=16, lora_dropout=0.05, lora_nums=3, blc_alpha=1, blc_weight=1.0)
    kw.update(over)
    return get_peft_model(make_toy(torch.float32), LoraConfig(**kw))


def _vt_model():
    from moka_amd.modified_peft import LoraConfig, PeftMixedModel
    m = make_toy(torch.bfloat16)
    targets = [n for n, _ in m.named_modules() if "layers" in n and any(p in n for p in PROJS)]
    cfg = LoraConfig(inference_mode=False, r=4, target_modules=targets, lora_alpha=16, lora_dropout=0.05,
                     task_type="CAUSAL_LM", attn_weight=0.05)
    pm = PeftMixedModel(m, cfg, adapter_name="image")
    pm.add_adapter("text", cfg)
    pm.set_adapter(["image", "text"])
    for n, p in pm.named_parameters():            # train.py:573-579
        p.requires_grad = ("lora" in n)
    return pm


def test_avt_surface_matches_reference():
    pm = _avt_model()
    assert _surface(pm) == GOLD["avt"]["surface"]
    ex = GOLD["avt"]["extra"]
    assert type(pm).__name__ == ex["class"]
    lin = pm.base_model.model.layers[0].self_attn.q_proj
    assert lin.scaling == ex["scaling"] and lin.d_k == ex["d_k"] and lin.r == ex["r"]
    from moka_amd.peft_hyper import get_peft_model_state_dict
    assert len(get_peft_model_state_dict(pm)) == ex["n_adapter_keys"]


def test_avt_weight_is_shared_not_copied():
    from moka_amd.peft_hyper import LoraConfig, get_peft_model
    base = make_toy(torch.float32)
    w = base.layers[0].self_attn.q_proj.weight
    pm = get_peft_model(base, LoraConfig(task_type="CAUSAL_LM", target_modules=PROJS, r=444, lora_alpha=16, lora_nums=3,
                                         lora_dropout=0.0, loramethod="train", blc_weight=1.0, blc_alpha=1))
    assert pm.base_model.model.layers[0].self_attn.q_proj.weight is w
    assert float(pm.base_model.model.layers[0].self_attn.q_proj.lora_B0.weight.abs().max()) == 0.0   # B starts at zero


def test_vt_surface_matches_reference():
    pm = _vt_model()
    assert _surface(pm) == GOLD["vt"]["surface"]
    ex = GOLD["vt"]["extra"]
    lin = pm.base_model.model.layers[0].self_attn.q_proj
    assert type(pm).__name__ == ex["class"]
    assert dict(lin.scaling) == ex["scaling"] and dict(lin.r) == ex["r"]
    assert lin.attn_weight == ex["attn_weight"] and list(lin.active_adapters) == ex["active_adapters"]


def test_rank_encoding():
    from moka_amd.peft_hyper.config import parse_rank
    assert parse_rank(444, 3) == [4, 4, 4]           # the reference's digit encoding (lora.py:256-259)
    assert parse_rank(88, 2) == [8, 8]
    assert parse_rank(16, 3) == [16, 16, 16]         # not expressible in digits: plain rank
    assert parse_rank((16, 16, 16), 3) == [16, 16, 16]
    assert parse_rank(8, 1) == [8]


def test_error_behaviour_matches_reference():
    from moka_amd.modified_peft import LoraConfig as VtCfg, PeftMixedModel
    from moka_amd.modified_peft.layer import Linear as VtLinear
    from moka_amd.peft_hyper import LoraConfig as AvtCfg, get_peft_model
    with pytest.raises(ValueError, match="not found in the base model"):      # lora.py:164-168
        get_peft_model(make_toy(), AvtCfg(task_type="CAUSAL_LM", target_modules=["nope"], r=444, lora_alpha=16, lora_nums=3,
                                           lora_dropout=0.0, loramethod="train", blc_weight=1.0))
    with pytest.raises(ValueError, match="positive integer"):                 # layer.py:97-98
        VtLinear(nn.Linear(64, 64, bias=False), "image", r=0, lora_alpha=16)
    with pytest.raises(ValueError, match="not found in the base model"):
        PeftMixedModel(make_toy(torch.bfloat16), VtCfg(r=4, target_modules=["nope"], lora_alpha=16), adapter_name="image")
    pm = _vt_model()
    with pytest.raises(ValueError, match="not found"):
        pm.set_adapter(["image", "audio"])


def test_avt_unknown_loramethod_returns_none():
    """The reference falls off the end of forward when loramethod has neither 'train' nor 'test'."""
    from moka_amd.peft_hyper import Linear
    lin = Linear(64, 64, r=444, lora_alpha=16, lora_nums=3, lora_dropout=0.0, loramethod="uni", bias=False)
    assert lin(torch.zeros(1, 4, 64), None) is None


def test_vt_disabled_and_merged_paths_use_base_only():
    pm = _vt_model()
    lin = pm.base_model.model.layers[0].self_attn.q_proj
    x = torch.randn(1, 4, 64, dtype=torch.bfloat16)
    with pm.disable_adapter():
        y = lin(x, None, None, None)                 # adapters disabled -> base layer only, on CPU too
    assert torch.equal(y, lin.base_layer(x))
    with torch.no_grad():
        lin.lora_B["text"].weight.normal_(0, 0.02)
    w0 = lin.base_layer.weight.clone()
    lin.merge(adapter_names=["text"])
    assert lin.merged and not torch.equal(lin.base_layer.weight, w0)
    assert torch.equal(lin(x, None, None, None), lin.base_layer(x))
    lin.unmerge()
    assert torch.allclose(lin.base_layer.weight.float(), w0.float(), atol=1e-2)


def test_empty_batch_returns_the_empty_base_output():
    """B = 0: the reference's per-sample loops run zero times; nothing is launched, on any device."""
    from moka_amd.peft_hyper import Linear as AvtLinear
    from moka_amd.modified_peft import layer as vt_layer
    a = AvtLinear(32, 64, r=(4, 4, 4), lora_alpha=16, lora_nums=3, blc_weight=1.0, blc_alpha=1, lora_dropout=0.0,
                  loramethod="train", bias=False)
    y = a(torch.zeros(0, 5, 32), [torch.zeros(0, 5, 1, dtype=torch.int32)] * 4)
    assert y.shape == (0, 5, 64)
    v = vt_layer.Linear(torch.nn.Linear(32, 64, bias=False), "image", r=4, lora_alpha=16, lora_dropout=0.0, attn_weight=0.05)
    v.update_layer("text", 4, lora_alpha=16, lora_dropout=0.0, init_lora_weights=True, use_rslora=False)
    v.set_adapter(["image", "text"])
    m = torch.zeros(0, 5, dtype=torch.bool)
    assert v(torch.zeros(0, 5, 32, dtype=v.get_base_layer().weight.dtype), m, m, m).shape == (0, 5, 64)


def test_vt_config_with_list_target_modules_round_trips_through_save_pretrained(tmp_path):
    """`target_modules=[...]` becomes a set in __post_init__; adapter_config.json must still be written
    (the reference converts sets to lists, modified_peft/config.py:67-70)."""
    import json
    from moka_amd.modified_peft import LoraConfig
    cfg = LoraConfig(r=8, target_modules=["q_proj", "v_proj"], lora_alpha=16, lora_dropout=0.05, attn_weight=0.05)
    assert cfg.target_modules == {"q_proj", "v_proj"}
    cfg.save_pretrained(str(tmp_path))
    d = json.load(open(tmp_path / "adapter_config.json"))
    assert d["target_modules"] == ["q_proj", "v_proj"] and d["r"] == 8 and d["attn_weight"] == 0.05 and d["peft_type"] == "LORA"
