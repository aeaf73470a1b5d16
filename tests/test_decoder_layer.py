"""the forked decoder block (moka_amd/decoder.py) -- masks threaded to all seven projections, switched
off for cached decode steps, q/k/v and gate/up through the grouped entry points.

CPU: the host logic (mask gating, kv cache, causality) with projections that ignore the masks.
GPU: the adapted block against (a) the reference's call pattern (one module call per projection) and (b) an fp64
restatement of the whole block built from the oracle's adapter (autograd through oracle/moka_oracle.py)."""
import math

import pytest
import torch

from moka_amd.decoder import LlamaDims, MokaLlamaDecoderLayer, MokaLlamaStack, _live_masks, rotary_tables

DIMS = LlamaDims(hidden=128, ff=352, n_heads=4, n_kv_heads=2)


class _Plain(torch.nn.Linear):
    """A projection that takes (and ignores) the masks, like a frozen layer the adapter was not attached to."""

    def forward(self, x, *masks):
        return super().forward(x)


def _plain_stack(n_layers=2, seed=0):
    torch.manual_seed(seed)
    return MokaLlamaStack(DIMS, n_layers, lambda i, o: _Plain(i, o, bias=False)).eval()


def test_masks_are_switched_off_for_cached_decode_steps():
    m = (torch.ones(1, 4), torch.zeros(1, 4), None)
    assert _live_masks(m, 0) is m
    assert _live_masks(m, 7) == (None, None, None)
    assert _live_masks(([1, 2, 3],), 3) == (None,)


def test_prefill_plus_decode_equals_full_forward_and_is_causal():
    st = _plain_stack()
    h = torch.randn(2, 9, DIMS.hidden)
    full, _ = st(h, "mask-a", "mask-b")
    pre, caches = st(h[:, :8], "mask-a", "mask-b")
    step, caches2 = st(h[:, 8:], "mask-a", "mask-b", kv_caches=caches)
    assert torch.allclose(pre, full[:, :8], atol=1e-5)            # causal: the 9th token does not change the first 8
    assert torch.allclose(step, full[:, 8:], atol=1e-5)           # rotary offset + cache
    assert caches2[0][0].shape[2] == 9
    h2 = h.clone()
    h2[:, 5:] += 1.0
    assert torch.allclose(st(h2, None)[0][:, :5], full[:, :5], atol=1e-5)


def test_state_dict_names_follow_the_llama_block():
    keys = set(MokaLlamaDecoderLayer(DIMS, lambda i, o: _Plain(i, o, bias=False)).state_dict())
    assert keys == {"input_layernorm.weight", "post_attention_layernorm.weight"} | {
        f"self_attn.{p}_proj.weight" for p in "qkvo"} | {f"mlp.{p}_proj.weight" for p in ("gate", "up", "down")}


def test_rotary_tables_offset():
    c0, s0 = rotary_tables(6, 32, 10000.0, "cpu", torch.float32)
    c1, s1 = rotary_tables(2, 32, 10000.0, "cpu", torch.float32, offset=4)
    assert torch.allclose(c0[4:], c1) and torch.allclose(s0[4:], s1)


# ------------------------------------------------------------------------------------------ GPU
def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an AMD GPU")
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _avt_layer(dev, r=16, p=0.0, seed=3):
    from moka_amd.peft_hyper import Linear
    g = torch.Generator().manual_seed(seed)

    def make(d_in, d_out):
        m = Linear(d_in, d_out, r=(r, r, r), lora_alpha=16, lora_nums=3, blc_weight=1.0, blc_alpha=1, lora_dropout=p,
                   loramethod="train", bias=False)
        with torch.no_grad():
            m.weight.copy_(torch.randn(d_out, d_in, generator=g) / math.sqrt(d_in))
            m.lora_B0.weight.copy_(torch.randn(d_out, r, generator=g) * 0.05)
        return m

    return MokaLlamaDecoderLayer(DIMS, make).to(dev, torch.bfloat16)


def _avt_masks(B, S, dev):
    """[text, video, audio, question] int32 [B, S, 1]: 8 text, 24 video, 8 text, 16 audio, 12 question, rest text."""
    tok = torch.zeros(S, dtype=torch.long)
    tok[8:32] = 1
    tok[40:56] = 2
    q = torch.zeros(S, dtype=torch.int32)
    q[56:68] = 1
    ms = [(tok == m).to(torch.int32).reshape(1, S, 1).repeat(B, 1, 1).to(dev) for m in range(3)]
    return ms + [q.reshape(1, S, 1).repeat(B, 1, 1).to(dev)]


@pytest.mark.gpu
def test_block_grouped_equals_reference_call_pattern():
    dev = _dev()
    layer = _avt_layer(dev).eval()
    B, S = 2, 96
    masks = _avt_masks(B, S, dev)
    cos, sin = rotary_tables(S, DIMS.head_dim, DIMS.rope_theta, dev, torch.bfloat16)
    h0 = torch.randn(B, S, DIMS.hidden, generator=torch.Generator().manual_seed(1)).to(dev, torch.bfloat16)

    def run(grouped):
        for p_ in layer.parameters():
            p_.grad = None
        h = h0.clone().requires_grad_(True)
        out, _ = layer(h, cos, sin, masks, grouped=grouped)
        out.float().square().mean().backward()
        return out, h.grad, {n: p_.grad.clone() for n, p_ in layer.named_parameters() if p_.grad is not None}

    o1, dh1, g1 = run(False)
    o2, dh2, g2 = run(True)
    assert torch.equal(o1, o2)                                     # eval(): the grouped forward is bit-identical
    assert rel(dh2, dh1) < 1e-2
    assert set(g1) == set(g2) and any("lora_A2" in n for n in g1)
    for n in g1:
        if g1[n].float().norm().item() > 0:
            assert rel(g2[n], g1[n]) < 1e-2, n


@pytest.mark.gpu
def test_block_against_fp64_oracle_block():
    """The whole block in fp64 on the CPU with the oracle's adapter in place of every projection (autograd through the
    oracle), against the bf16 block on the HIP path: forward, input gradient and every adapter gradient."""
    from oracle import moka_oracle as O
    dev = _dev()
    layer = _avt_layer(dev).eval()
    B, S, r = 2, 96, 16
    masks = _avt_masks(B, S, dev)
    cos, sin = rotary_tables(S, DIMS.head_dim, DIMS.rope_theta, dev, torch.bfloat16)
    h0 = torch.randn(B, S, DIMS.hidden, generator=torch.Generator().manual_seed(2)).to(dev, torch.bfloat16)
    gout = torch.randn(B, S, DIMS.hidden, generator=torch.Generator().manual_seed(4)).to(dev, torch.bfloat16)
    h = h0.clone().requires_grad_(True)
    out, _ = layer(h, cos, sin, masks)
    out.backward(gout)

    # ---- fp64 restatement
    P = {n: p_.detach().double().cpu().requires_grad_(p_.requires_grad) for n, p_ in layer.named_parameters()}
    mcpu = [m.cpu() for m in masks]

    def proj(name, x):
        A = [P[f"{name}.lora_A{i}.weight"] for i in range(3)]
        return O.avt_forward(x, P[f"{name}.weight"], A, P[f"{name}.lora_B0.weight"], mcpu, 16, r, 1.0, dtype=torch.float64)[0]

    def rms(x, w):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + DIMS.rms_eps) * w

    def rot(x, c, s_):
        hh = x.shape[-1] // 2
        return x * c + torch.cat([-x[..., hh:], x[..., :hh]], dim=-1) * s_

    hd = h0.double().cpu().requires_grad_(True)
    c64, s64 = rotary_tables(S, DIMS.head_dim, DIMS.rope_theta, "cpu", torch.float64)
    x = rms(hd, P["input_layernorm.weight"])
    q = proj("self_attn.q_proj", x).view(B, S, DIMS.n_heads, -1).transpose(1, 2)
    k = proj("self_attn.k_proj", x).view(B, S, DIMS.n_kv_heads, -1).transpose(1, 2)
    v = proj("self_attn.v_proj", x).view(B, S, DIMS.n_kv_heads, -1).transpose(1, 2)
    q, k = rot(q, c64, s64), rot(k, c64, s64)
    rep = DIMS.n_heads // DIMS.n_kv_heads
    k, v = k.repeat_interleave(rep, 1), v.repeat_interleave(rep, 1)
    att = (q @ k.transpose(-1, -2)) / math.sqrt(DIMS.head_dim)
    att = att.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf")).softmax(-1)
    o = (att @ v).transpose(1, 2).reshape(B, S, -1)
    h1 = hd + proj("self_attn.o_proj", o)
    x2 = rms(h1, P["post_attention_layernorm.weight"])
    act = torch.nn.functional.silu(proj("mlp.gate_proj", x2)) * proj("mlp.up_proj", x2)
    ref = h1 + proj("mlp.down_proj", act)
    ref.backward(gout.double().cpu())

    assert rel(out, ref) < 1.5e-2                                   # bf16 storage of every intermediate of the block
    assert rel(h.grad, hd.grad) < 3e-2
    n_checked = 0
    for n, p_ in layer.named_parameters():
        if "lora_" in n:
            assert p_.grad is not None, n
            assert rel(p_.grad, P[n].grad) < 4e-2, n
            n_checked += 1
    assert n_checked == 7 * 4


@pytest.mark.gpu
def test_prefill_with_masks_then_cached_decode_equals_full_forward():
    """A decode step gets no masks (text adapter, no interaction); the same token inside a full masked forward is a text
    token and takes exactly that path -- so prefill + decode must reproduce the full forward."""
    dev = _dev()
    torch.manual_seed(0)
    from moka_amd.peft_hyper import Linear

    def make(d_in, d_out):
        m = Linear(d_in, d_out, r=(8, 8, 8), lora_alpha=16, lora_nums=3, blc_weight=1.0, blc_alpha=1, lora_dropout=0.0,
                   loramethod="test", bias=False)              # inference_cut.py builds the model with the test method
        torch.nn.init.normal_(m.lora_B0.weight, std=0.05)
        return m

    st = MokaLlamaStack(DIMS, 2, make).to(dev, torch.bfloat16).eval()
    B, S = 1, 81
    masks_full = _avt_masks(B, S, dev)
    masks_pre = [m[:, :S - 1] for m in masks_full]
    h = torch.randn(B, S, DIMS.hidden, device=dev).to(torch.bfloat16)
    with torch.no_grad():
        full, _ = st(h, masks_full)
        pre, caches = st(h[:, :S - 1], masks_pre)
        step, _ = st(h[:, S - 1:], masks_pre, kv_caches=caches)     # stale masks of the wrong length: must be ignored
    assert rel(pre, full[:, :S - 1]) < 2e-2
    assert rel(step, full[:, S - 1:]) < 2e-2


@pytest.mark.gpu
def test_chunk_of_several_tokens_behind_a_cache_is_causal():
    """S > 1 with a cache (chunked prefill / speculative decoding) must stay causal inside the chunk, and an
    empty-but-present cache is no cache.  The chunk's tokens are text tokens, so the mask-free branch equals the full forward."""
    dev = _dev()
    torch.manual_seed(1)
    from moka_amd.modified_peft import Linear as VtLinear

    def make(d_in, d_out):
        base = torch.nn.Linear(d_in, d_out, bias=False)
        m = VtLinear(base, "image", r=8, lora_alpha=16, lora_dropout=0.0, attn_weight=0.05)
        m.update_layer("text", 8, lora_alpha=16, lora_dropout=0.0, init_lora_weights=True, use_rslora=False)
        m.set_adapter(["image", "text"])
        for n in ("image", "text"):
            torch.nn.init.normal_(m.lora_B[n].weight, std=0.05)
        return m

    st = MokaLlamaStack(DIMS, 2, make).to(dev, torch.bfloat16).eval()
    B, S, C = 1, 80, 3
    text = torch.ones(B, S, dtype=torch.bool, device=dev)
    image = torch.zeros(B, S, dtype=torch.bool, device=dev)
    image[:, 4:36] = True
    text &= ~image
    q = torch.zeros(B, S, dtype=torch.bool, device=dev)
    q[:, 40:52] = True
    h = torch.randn(B, S, DIMS.hidden, device=dev).to(torch.bfloat16)
    with torch.no_grad():
        full, _ = st(h, text, image, q)
        pre, caches = st(h[:, :S - C], text[:, :S - C], image[:, :S - C], q[:, :S - C])
        chunk, _ = st(h[:, S - C:], None, None, None, kv_caches=caches)
        empty = [(c[0][:, :, :0], c[1][:, :, :0]) for c in caches]
        again, _ = st(h[:, :S - C], text[:, :S - C], image[:, :S - C], q[:, :S - C], kv_caches=empty)
    assert rel(chunk, full[:, S - C:]) < 2e-2
    assert rel(again, pre) < 1e-6


def test_decode_weight_equals_the_text_adapter_branch_on_cpu():
    """decode_weight(W + s B A_text) against the oracle's mask-free branch (plain LoRA with the text adapter), both mirrors'
    parameter layouts emulated with bare modules (no GPU needed: pure tensor algebra)."""
    from types import SimpleNamespace
    from moka_amd.decoder import decode_weight
    from oracle import moka_oracle as O
    g = torch.Generator().manual_seed(3)
    d_in, d_out, r, s_ = 48, 40, 8, 2.0
    W, A, Bw = torch.randn(d_out, d_in, generator=g), torch.randn(r, d_in, generator=g), torch.randn(d_out, r, generator=g)
    x = torch.randn(2, 1, d_in, generator=g)
    y_ref = O.plain_lora_forward(x, torch.nn.functional.linear(x.double(), W.double()), A, Bw, s_)
    avt = SimpleNamespace(weight=W, bias=None, lora_A0=SimpleNamespace(weight=A), lora_B0=SimpleNamespace(weight=Bw), scaling=[s_], fan_in_fan_out=False)
    Wd, b = decode_weight(avt)
    assert b is None and torch.allclose(torch.nn.functional.linear(x, Wd).double(), y_ref, atol=1e-4)
    vt = SimpleNamespace(lora_A={"text": SimpleNamespace(weight=A), "image": SimpleNamespace(weight=A * 0)},
                         lora_B={"text": SimpleNamespace(weight=Bw), "image": SimpleNamespace(weight=Bw * 0)}, scaling={"text": s_, "image": 9.0},
                         get_base_layer=lambda: SimpleNamespace(weight=W, bias=None))
    Wd2, _ = decode_weight(vt)
    assert torch.equal(Wd, Wd2)
    plain = torch.nn.Linear(d_in, d_out, bias=False)
    assert decode_weight(plain)[0] is plain.weight


@pytest.mark.gpu
def test_merged_decode_steps_match_the_adapter_path():
    dev = _dev()
    torch.manual_seed(1)
    from moka_amd.peft_hyper import Linear

    def make(d_in, d_out):
        m = Linear(d_in, d_out, r=(16, 16, 16), lora_alpha=16, lora_nums=3, blc_weight=1.0, blc_alpha=1, lora_dropout=0.0,
                   loramethod="test", bias=False)
        torch.nn.init.normal_(m.lora_B0.weight, std=0.05)
        return m

    st = MokaLlamaStack(DIMS, 2, make).to(dev, torch.bfloat16).eval()
    B, S = 2, 80
    masks = _avt_masks(B, S, dev)
    h = torch.randn(B, S + 3, DIMS.hidden, device=dev).to(torch.bfloat16)
    with torch.no_grad():
        pre, caches = st(h[:, :S], masks)
        outs = []
        c = caches
        for i in range(3):                                          # three decode steps on the HIP adapter path
            o, c = st(h[:, S + i:S + i + 1], masks, kv_caches=c)
            outs.append(o)
        st.merge_for_decode()
        pre2, caches2 = st(h[:, :S], masks)                         # prefill is untouched by the merge
        assert torch.equal(pre, pre2)
        c = caches2
        for i in range(3):                                          # the same steps as one GEMM per projection
            o, c = st(h[:, S + i:S + i + 1], masks, kv_caches=c)
            assert rel(o, outs[i]) < 1e-2
        st.unmerge()


@pytest.mark.gpu
@pytest.mark.parametrize("reentrant", [False, True])
def test_activation_checkpointing_replays_the_dropout_masks(reentrant):
    """The reference trains AVT with gradient checkpointing (ft_musicavqa.sh:41).  The per-call dropout seeds come from torch's
    CPU generator, which checkpointing restores before the re-forward: gradients with and without checkpointing must agree
    (up to the atomics' summation order), i.e. the recomputed masks are the ones the first forward used."""
    from torch.utils.checkpoint import checkpoint
    dev = _dev()
    layer = _avt_layer(dev, p=0.1).train()
    B, S = 2, 96
    masks = _avt_masks(B, S, dev)
    cos, sin = rotary_tables(S, DIMS.head_dim, DIMS.rope_theta, dev, torch.bfloat16)
    h0 = torch.randn(B, S, DIMS.hidden, generator=torch.Generator().manual_seed(1)).to(dev, torch.bfloat16)

    def run(ckpt):
        torch.manual_seed(123)
        for p_ in layer.parameters():
            p_.grad = None
        h = h0.clone().requires_grad_(True)
        if ckpt:
            out = checkpoint(lambda hh: layer(hh, cos, sin, masks)[0], h, use_reentrant=reentrant)
        else:
            out = layer(h, cos, sin, masks)[0]
        out.float().square().mean().backward()
        return out.detach(), h.grad, {n: p_.grad.clone() for n, p_ in layer.named_parameters() if p_.grad is not None}

    o1, dh1, g1 = run(False)
    o2, dh2, g2 = run(True)
    assert torch.equal(o1, o2)
    assert rel(dh2, dh1) < 1e-3
    assert set(g1) == set(g2) and sum("lora_" in n for n in g1) == 7 * 4
    for n in g1:
        assert rel(g2[n], g1[n]) < 1e-3, n
    torch.manual_seed(124)                     # another seed gives other masks (the check above is not vacuous)
    h = h0.clone()
    assert not torch.equal(layer(h, cos, sin, masks)[0], o1)
