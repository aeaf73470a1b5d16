"""Host-side behaviour of moka_amd.parallel.attach: EVERY trainable parameter rides in the flat
buffers (the reference scripts also train the Q-Former projectors, finetune.py:151-160 / train.py:573-579), decoder-layer
grouping is anchored to the decoder stack (an encoder's layers.N do not alias it), fp32-storage adapters, the accumulation
guard, state round trips, and the torch.optim.Optimizer-shaped handle.  CPU only: the optimizer kernel itself is GPU-only."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Toy(torch.nn.Module):
    """A projector in front of a decoder stack of adapted layers + a vision encoder with its own `layers.N`."""

    def __init__(self, dtype=torch.bfloat16, n=3):
        super().__init__()
        from moka_amd.decoder import LlamaDims, MokaLlamaStack
        from moka_amd.peft_hyper import Linear
        dims = LlamaDims(hidden=64, ff=96, n_heads=2, n_kv_heads=2)

        def make(d_in, d_out):
            return Linear(d_in, d_out, r=(4, 4, 4), lora_alpha=16, lora_nums=3, blc_weight=1.0, blc_alpha=1, lora_dropout=0.0,
                          loramethod="train", bias=False)
        self.vl_projector = torch.nn.Linear(32, 64)
        self.vision = torch.nn.Module()
        self.vision.layers = torch.nn.ModuleList(torch.nn.Linear(8, 8) for _ in range(2))
        self.model = MokaLlamaStack(dims, n, make)
        self.to(dtype)
        for n_, p in self.named_parameters():
            p.requires_grad = ("lora_" in n_) or n_.startswith("vl_projector") or n_ == "vision.layers.1.weight"


def test_every_trainable_parameter_is_flattened_and_grouped_by_decoder_layer():
    from moka_amd.parallel import attach
    m = _Toy()
    ref = {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}
    dp = attach(m, n_buckets=2)
    assert set(dp.names) == set(ref)
    assert set(dp.hooked) == {"vl_projector.weight", "vl_projector.bias", "vision.layers.1.weight"}
    assert len(dp.kernel_fed) == 3 * 7 * 4
    # group -1 (everything outside the decoder stack, the encoder's layers.1 included) comes first, then layers 0, 1, 2
    assert dp.bucket.n_layers == 4
    first = dp.names[:3]
    assert set(first) == set(dp.hooked)
    k = dp.names.index("model.layers.0.self_attn.q_proj.lora_A0.weight")
    assert dp.bucket.layer_end[0] == dp.offsets[k]
    for n, o, sz in zip(dp.names, dp.offsets, dp.sizes):
        p = dict(m.named_parameters())[n]
        assert p.data_ptr() == dp.work[o:].data_ptr() and torch.equal(p.detach(), ref[n])
        assert torch.equal(dp.master[o:o + sz].view(p.shape), ref[n].float())
    # the hooked parameters' autograd gradients land in the flat buffer and .grad stays empty; accumulation adds
    x = torch.randn(5, 32).to(torch.bfloat16)
    for rep in (1, 2):
        m.vl_projector(x).float().square().sum().backward()
        o = dp.offsets[dp.names.index("vl_projector.weight")]
        g = dp.bucket.flat[o:o + m.vl_projector.weight.numel()].view_as(m.vl_projector.weight)
        assert m.vl_projector.weight.grad is None and g.abs().sum() > 0
        if rep == 1:
            g1 = g.clone()
        else:
            assert torch.allclose(g, 2 * g1, rtol=1e-6)
    dp.detach()


def test_fp32_storage_parameters_are_views_of_the_master_copy():
    from moka_amd.parallel import attach
    m = _Toy(dtype=torch.float32, n=1)
    dp = attach(m)
    p = m.model.layers[0].mlp.down_proj.lora_B0.weight
    o = dp.offsets[dp.names.index("model.layers.0.mlp.down_proj.lora_B0.weight")]
    assert p.dtype == torch.float32 and p.data_ptr() == dp.master[o:].data_ptr()
    sk = m.model.layers[0].mlp.down_proj._moka_sinks
    assert sk["B"].dtype == torch.float32 and sk["B"].data_ptr() == dp.bucket.flat[o:].data_ptr()


def test_a_forward_with_buckets_in_flight_is_an_error_unless_no_sync():
    """accumulation without no_sync would race the in-place all-reduce and skip layers -- it must raise."""
    from moka_amd.parallel import attach
    m = _Toy(n=2)
    dp = attach(m, n_buckets=2)
    pre = [h for h in m.model.layers[0]._forward_pre_hooks.values()]
    assert len(pre) == 1
    x = torch.zeros(1, 4, 64, dtype=torch.bfloat16, requires_grad=True)
    pre[0](m.model.layers[0], (x,))                       # fine: nothing in flight
    dp._done.add(1)                                        # as if a backward had shipped layer 1
    with pytest.raises(RuntimeError, match="no_sync"):
        pre[0](m.model.layers[0], (x,))
    with dp.no_sync():
        pre[0](m.model.layers[0], (x,))                   # accumulation: allowed, and nothing is marked / shipped
        dp._layer_done(0)
        assert 0 not in dp._done
    assert dp.sync is True
    dp.finish()                                            # world 1: joins nothing, clears the marks
    pre[0](m.model.layers[0], (x,))


def test_state_round_trips_and_safetensors_after_attach(tmp_path):
    from safetensors.torch import load_file, save_file, save_model
    from moka_amd.parallel import MokaFlatOptimizer, attach
    m = _Toy(n=2)
    dp = attach(m, lr=3e-4, weight_decay=0.01)
    ref = {k: v.detach().clone() for k, v in m.state_dict().items()}
    # (a) torch.save of the module state and of the flat state
    torch.save(m.state_dict(), tmp_path / "m.pt")
    torch.save(dp.state_dict(), tmp_path / "dp.pt")
    # (b) safetensors: save_model refuses views of one storage; detached_state_dict() is the documented way
    with pytest.raises(RuntimeError):
        save_model(m, str(tmp_path / "raw.safetensors"))
    save_file(dp.detached_state_dict(), str(tmp_path / "m.safetensors"))
    back = load_file(str(tmp_path / "m.safetensors"))
    assert set(back) == set(ref) and all(torch.equal(back[k], ref[k]) for k in ref)
    # (c) the reference trainer's filter (AudioVisualText/trainer.py:213-218: keep what requires grad)
    keep = {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}
    assert set(keep) == set(dp.names)
    # perturb, then restore both
    with torch.no_grad():
        dp.master.add_(1.0)
        dp.work.copy_(dp.master)
        dp.optimizer.exp_avg.fill_(2.0)
        dp.optimizer.t = 7
    m.load_state_dict(torch.load(tmp_path / "m.pt"))
    assert all(torch.equal(v, ref[k]) for k, v in m.state_dict().items())
    assert m.model.layers[0].self_attn.q_proj.lora_A0.weight.data_ptr() == dp.work[dp.offsets[dp.names.index("model.layers.0.self_attn.q_proj.lora_A0.weight")]:].data_ptr()
    sd = torch.load(tmp_path / "dp.pt")
    sd["optimizer"]["step"] = 3
    dp.load_state_dict(sd)
    assert dp.optimizer.t == 3 and float(dp.optimizer.exp_avg.abs().max()) == 0.0
    o = dp.offsets[dp.names.index("vl_projector.weight")]
    assert torch.equal(dp.master[o:o + 64 * 32].view(64, 32), ref["vl_projector.weight"].float())
    # the optimizer handle: param groups for schedulers, state through the Optimizer API
    opt = MokaFlatOptimizer(dp, lr=1e-3, max_grad_norm=1.0)
    assert len(opt.param_groups[0]["params"]) == len(dp.names) and opt.param_groups[0]["lr"] == 1e-3
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda k: 0.5)
    assert opt.param_groups[0]["lr"] == 5e-4
    osd = opt.state_dict()
    assert "moka_flat" in osd and osd["moka_flat"]["names"] == dp.names
    opt.load_state_dict(osd)
    opt.zero_grad()
    del sched


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _ProjThenStack(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(4)
        self.vl_projector = torch.nn.Linear(12, 16)
        self.layers = torch.nn.ModuleList(torch.nn.Linear(16, 16) for _ in range(3))

    def forward(self, x):
        h = self.vl_projector(x)
        for l in self.layers:
            h = h + torch.tanh(l(h))
        return h


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from moka_amd.parallel import attach
    x = torch.randn(4, 12, generator=torch.Generator().manual_seed(1))
    ref = _ProjThenStack()
    ref(x).square().sum().div(4).backward()                      # mean over the 4 samples, one rank
    ref_g = {n: p.grad.clone() for n, p in ref.named_parameters()}
    m = _ProjThenStack()
    dp = attach(m, n_buckets=2, optimizer=False)                 # fp32 toy parameters, no adapted projections: all hooked
    with dp.no_sync():
        m(x[2 * rank:2 * rank + 1]).square().sum().div(2).backward()          # micro-batch 1 of my shard
    m(x[2 * rank + 1:2 * rank + 2]).square().sum().div(2).backward()          # micro-batch 2 (synchronised)
    dp.finish(average=True)
    ok = all(p.grad is None for p in m.parameters())
    for n, o, sz in zip(dp.names, dp.offsets, dp.sizes):
        ok = ok and torch.allclose(dp.bucket.flat[o:o + sz].view_as(ref_g[n]), ref_g[n], rtol=1e-5, atol=1e-6)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_projector_gradient_is_averaged_over_two_ranks_with_accumulation():
    """2 ranks x 2 micro-batches: the flat gradient (projector in front of the stack included) == the full-batch mean gradient."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert got == {0: True, 1: True}
