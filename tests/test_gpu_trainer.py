"""the data-parallel step behind a torch.optim.Optimizer-shaped handle that HF Trainer drives
(AudioVisualText/trainer.py:163-218, VisualText/train/train.py:601-617 both call Trainer.train()), and fp32-storage adapters
through attach().  World 1 on the one GPU: the machinery (sinks, hooks, fused AdamW, accumulation callback) is the N > 1 one."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


class TinyLM(torch.nn.Module):
    """projector -> 2 adapted decoder layers -> frozen head, MSE loss; takes the AVT mask list as batch columns."""

    def __init__(self, dev, dtype=torch.bfloat16, seed=13):
        super().__init__()
        from moka_amd.decoder import LlamaDims, MokaLlamaStack
        from moka_amd.peft_hyper import Linear
        self.dims = LlamaDims(hidden=128, ff=256, n_heads=4, n_kv_heads=4)
        torch.manual_seed(seed)

        def make(d_in, d_out):
            m = Linear(d_in, d_out, r=(8, 8, 8), lora_alpha=16, lora_nums=3, blc_weight=1.0, blc_alpha=1, lora_dropout=0.0,
                       loramethod="train", bias=False)
            torch.nn.init.normal_(m.weight, std=0.05)
            torch.nn.init.normal_(m.lora_B0.weight, std=0.05)
            return m
        self.vl_projector = torch.nn.Linear(40, self.dims.hidden)
        self.model = MokaLlamaStack(self.dims, 2, make)
        self.head = torch.nn.Linear(self.dims.hidden, 8)
        self.to(dev, dtype).train()
        for n, p in self.named_parameters():
            p.requires_grad = ("lora_" in n) or n.startswith("vl_projector")

    def forward(self, feats, target, m_t, m_v, m_a, m_q):
        h = self.vl_projector(feats)
        out, _ = self.model(h, [m_t, m_v, m_a, m_q])
        loss = (self.head(out).float() - target.float()).square().mean()
        return {"loss": loss}


def _data(S=64, n=2):
    g = torch.Generator().manual_seed(3)
    tok = torch.zeros(n, S, dtype=torch.int64)
    tok[:, 4:20] = 1
    tok[:, 24:36] = 2
    q = torch.zeros(n, S, dtype=torch.bool)
    q[:, 40:50] = True
    rows = []
    for b in range(n):
        rows.append({"feats": torch.randn(S, 40, generator=g).to(torch.bfloat16), "target": torch.randn(S, 8, generator=g),
                     **{k: (tok[b] == m).to(torch.int32).unsqueeze(-1) for k, m in (("m_t", 0), ("m_v", 1), ("m_a", 2))},
                     "m_q": q[b].to(torch.int32).unsqueeze(-1)})
    return rows


def _collate(rows):
    return {k: torch.stack([r[k] for r in rows]) for k in rows[0]}


@pytest.mark.parametrize("accum", [1, 2])
def test_hf_trainer_drives_the_flat_optimizer(tmp_path, accum):
    from transformers import Trainer, TrainingArguments
    from moka_amd.parallel import MokaFlatOptimizer, attach, keep_out_of_ddp, trainer_callback
    dev = torch.device("cuda:0")
    rows = _data(n=2 * accum)
    steps, lr = 4, 2e-3
    # ---- the hand-written loop: every optimizer step sees all rows (accum micro-batches of 2 rows)
    ref = TinyLM(dev)
    dp_r = attach(ref, n_buckets=2, lr=lr, weight_decay=0.0)
    ref_losses = []
    for _ in range(steps):
        tot = 0.0
        for k in range(accum):
            batch = {kk: v.to(dev) for kk, v in _collate(rows[2 * k:2 * k + 2]).items()}
            ctx = dp_r.no_sync() if k < accum - 1 else torch.enable_grad()
            with ctx:
                loss = ref(**batch)["loss"] / accum
                loss.backward()
            tot += float(loss)
        dp_r.step(max_grad_norm=1.0)
        ref_losses.append(tot)
    torch.cuda.synchronize()
    # ---- the same under transformers.Trainer
    m = TinyLM(dev)
    dp = attach(m, n_buckets=2, lr=lr, weight_decay=0.0)
    opt = MokaFlatOptimizer(dp, lr=lr, max_grad_norm=1.0)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda k: 1.0)
    args = TrainingArguments(output_dir=str(tmp_path), per_device_train_batch_size=2, gradient_accumulation_steps=accum, max_steps=steps,
                             learning_rate=lr, max_grad_norm=0.0, report_to=[], save_strategy="no", logging_steps=1,
                             remove_unused_columns=False, dataloader_pin_memory=False, seed=1, data_seed=1,
                             dataloader_drop_last=False, disable_tqdm=True)

    class Seq(Trainer):                                           # fixed row order (the comparison is against a fixed-order loop)
        def _get_train_sampler(self, *a, **k):
            return torch.utils.data.SequentialSampler(self.train_dataset)

    tr = Seq(model=m, args=args, train_dataset=rows, data_collator=_collate, optimizers=(opt, sched), callbacks=[trainer_callback(dp)])
    keep_out_of_ddp(tr, dp)                                       # (what a torchrun launch needs; harmless in one process)
    assert tr.accelerator.prepare_model(m) is m
    tr.train()
    torch.cuda.synchronize()
    losses = [e["loss"] for e in tr.state.log_history if "loss" in e]
    assert len(losses) == steps and losses[-1] < losses[0], losses
    assert ref_losses[-1] < ref_losses[0]
    assert dp.optimizer.t == steps and all(p.grad is None for p in m.parameters())
    err = ((dp.master - dp_r.master).norm() / dp_r.master.norm()).item()
    assert err <= 2e-4, err                                       # same arithmetic; atomics order and bf16 activations differ in the last bits
    o = dp.offsets[dp.names.index("vl_projector.weight")]
    w0 = TinyLM(dev).vl_projector.weight.float().reshape(-1)
    assert (dp.master[o:o + w0.numel()] - w0).abs().max().item() > 1e-4          # the projector trained too


def test_fp32_storage_adapters_through_attach_and_the_fused_step():
    """The reference's adapters follow the base dtype (layer.py:124-132): an fp32 model under attach() -- parameters are views of
    the fp32 master, the kernels add into the flat buffer, the fused AdamW updates them in place -- against torch.optim.AdamW on
    the plain autograd gradients of an identical model."""
    from moka_amd.parallel import attach
    dev = torch.device("cuda:0")
    rows = _data(n=2)
    batch = {k: v.to(dev) for k, v in _collate(rows).items()}
    batch["feats"] = batch["feats"].float()
    a, b = TinyLM(dev, torch.float32), TinyLM(dev, torch.float32)
    dp = attach(a, n_buckets=2, lr=1e-3, weight_decay=0.01)
    opt_b = torch.optim.AdamW([p for p in b.parameters() if p.requires_grad], lr=1e-3, weight_decay=0.01, eps=1e-8)
    for _ in range(3):
        a(**batch)["loss"].backward()
        dp.step()
        b(**batch)["loss"].backward()
        opt_b.step()
        opt_b.zero_grad()
    torch.cuda.synchronize()
    pb = dict(b.named_parameters())
    worst = 0.0
    for n, p in a.named_parameters():
        if p.requires_grad:
            assert p.dtype == torch.float32 and p.grad is None
            worst = max(worst, ((p.detach() - pb[n].detach()).norm() / pb[n].detach().norm().clamp_min(1e-20)).item())
    assert worst <= 2e-4, worst


def _batch2(dev, S=64, n=4, shift=0):
    """_data() with the modality / question spans moved by `shift` tokens (a second batch with another routing)."""
    rows = _data(S=S, n=n)
    if shift:
        for r_ in rows:
            for k in ("m_t", "m_v", "m_a", "m_q"):
                r_[k] = torch.roll(r_[k], shift, dims=0)
    return {k: v.to(dev) for k, v in _collate(rows).items()}


@pytest.mark.parametrize("chains", [1, 2])
def test_graphed_train_step_equals_the_live_loop(chains):
    """moka_amd.schedule.GraphedTrainStep: the whole step (forward, backward through MokaLinearFn, deferred dA_m, the bucket's AdamW slices
    and weight shadows on the hub, a hooked projector parameter) captured as ONE hub-shaped hipGraph and replayed on changing batches ==
    the live attach() loop on the same batches."""
    from moka_amd.parallel import attach
    from moka_amd.routing import MokaRouting
    from moka_amd.schedule import GraphedTrainStep
    dev = torch.device("cuda:0")
    batches = [_batch2(dev, shift=0), _batch2(dev, shift=3)]
    steps, lr = 5, 2e-3
    ref = TinyLM(dev)
    dp_r = attach(ref, n_buckets=2, lr=lr, weight_decay=0.01)
    ref_losses = []
    n_rows = batches[0]["feats"].shape[0]
    for i in range(steps):
        # the live loop on the same part-batches (the step's arithmetic: every part's mean loss weighted by its share of the samples,
        # accumulated into the flat gradient; a hooked bf16 parameter's gradient is rounded per part)
        tot = 0.0
        for c in range(chains):
            lo_, hi_ = c * n_rows // chains, (c + 1) * n_rows // chains
            part = {k: v[lo_:hi_] for k, v in batches[i % 2].items()}
            ctx = dp_r.no_sync() if c < chains - 1 else torch.enable_grad()
            with ctx:
                loss = ref(**part)["loss"] * ((hi_ - lo_) / n_rows)
                loss.backward()
            tot += float(loss.detach())
        dp_r.step()
        ref_losses.append(tot)
    m = TinyLM(dev)
    dp = attach(m, n_buckets=2, lr=lr, weight_decay=0.01)
    gs = GraphedTrainStep(dp, lambda p: m(**p)["loss"], batches[0], chains=chains,
                          routing_fn=lambda p: MokaRouting.from_avt_masks([p["m_t"], p["m_v"], p["m_a"], p["m_q"]]))
    init = attach(TinyLM(dev), n_buckets=2, lr=lr, weight_decay=0.01)
    assert torch.equal(dp.master, init.master) and dp.optimizer.t == 0 and float(dp.bucket.flat.abs().max()) == 0.0   # the capture's warm-up steps were undone
    losses = [float(gs(batches[i % 2])) for i in range(steps)]
    torch.cuda.synchronize()
    assert dp.optimizer.t == steps
    if chains == 1:
        assert losses[0] == ref_losses[0], (losses[0], ref_losses[0])          # same kernels on the same data before any update: the same bits
    for a_, b_ in zip(losses, ref_losses):
        assert abs(a_ - b_) <= 2e-3 * abs(b_) + 1e-6, (losses, ref_losses)
    assert ref_losses[-1] < ref_losses[0]
    err = ((dp.master - dp_r.master).norm() / dp_r.master.norm()).item()
    assert err <= 2e-4, err
    o = dp.offsets[dp.names.index("vl_projector.weight")]
    w0 = TinyLM(dev).vl_projector.weight.float().reshape(-1)
    assert (dp.master[o:o + w0.numel()] - w0).abs().max().item() > 1e-4          # the hooked (non-kernel-fed) projector trained inside the graph too
    # the live path still works on the same handle after the capture (the graph state was left clean)
    m(**batches[0])["loss"].backward()
    dp.step()
    torch.cuda.synchronize()


def test_persistent_weight_shadows_equal_per_call_shadows():
    """attach(persistent_shadows=True) (default): BwT / AT kept per projection and rewritten behind the optimizer == recomputed in every forward."""
    from moka_amd.parallel import attach
    dev = torch.device("cuda:0")
    batch = _batch2(dev)
    outs = []
    for keep in (True, False):
        m = TinyLM(dev)
        dp = attach(m, n_buckets=2, lr=2e-3, weight_decay=0.0, persistent_shadows=keep)
        assert bool(dp._shadowed) == keep
        for _ in range(4):
            m(**batch)["loss"].backward()
            dp.step()
        torch.cuda.synchronize()
        outs.append(dp.master.clone())
    err = ((outs[0] - outs[1]).norm() / outs[1].norm()).item()
    assert err <= 1e-5, err                                                       # (same arithmetic; fp32 atomics order differs in the last bits)


def test_graphed_train_step_draws_fresh_dropout_masks_per_replay():
    """lora_dropout under a captured step: the per-call seeds are frozen with the launch arguments, the device word every dropout kernel
    folds into its seed (moka_opts.seed_dev = dp.seed_epoch) is rewritten from torch's CPU generator in front of every replay -- replays of
    the same batch differ, replays under the same torch seed repeat bit for bit (what activation checkpointing / a resumed run need)."""
    from moka_amd.parallel import attach
    from moka_amd.routing import MokaRouting
    from moka_amd.schedule import GraphedTrainStep
    dev = torch.device("cuda:0")
    batch = _batch2(dev)
    m = TinyLM(dev)
    for mod in m.modules():
        if hasattr(mod, "lora_dropout_p"):
            mod.lora_dropout_p = 0.25
    dp = attach(m, n_buckets=2, lr=0.0, weight_decay=0.0)                          # lr 0: the weights stand still, only the masks move
    gs = GraphedTrainStep(dp, lambda p: m(**p)["loss"], batch, chains=2,
                          routing_fn=lambda p: MokaRouting.from_avt_masks([p["m_t"], p["m_v"], p["m_a"], p["m_q"]]))
    torch.manual_seed(11); a1 = float(gs(batch))
    torch.manual_seed(12); a2 = float(gs(batch))
    torch.manual_seed(11); a3 = float(gs(batch))
    assert a1 != a2 and a1 == a3, (a1, a2, a3)
    assert int(dp.seed_epoch.item()) != 0


def _hf_rank(rank, world, port, tmp, q):
    """One of `world` ranks of a torchrun-style launch (gloo: the ranks share the one GPU) driving transformers.Trainer."""
    sys.path.insert(0, ROOT)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": "0", "LOCAL_WORLD_SIZE": str(world)})      # (LOCAL_RANK 0 for both: one device)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)                   # (accelerate adopts an initialised group)
    from transformers import Trainer, TrainingArguments
    from moka_amd.parallel import MokaFlatOptimizer, attach, keep_out_of_ddp, trainer_callback
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    steps, lr = 3, 2e-3
    rows = _data(n=4 * steps)
    m = TinyLM(dev)
    dp = attach(m, n_buckets=2, lr=lr, weight_decay=0.0)
    opt = MokaFlatOptimizer(dp, lr=lr)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda k: 1.0)
    args = TrainingArguments(output_dir=os.path.join(tmp, "r%d" % rank), per_device_train_batch_size=2, gradient_accumulation_steps=1, max_steps=steps,
                             learning_rate=lr, max_grad_norm=0.0, report_to=[], save_strategy="no", logging_steps=1,
                             remove_unused_columns=False, dataloader_pin_memory=False, seed=1, data_seed=1, dataloader_drop_last=False,
                             disable_tqdm=True, ddp_backend="gloo")

    class Seq(Trainer):
        def _get_train_sampler(self, *a, **k):
            return torch.utils.data.SequentialSampler(self.train_dataset)

    tr = Seq(model=m, args=args, train_dataset=rows, data_collator=_collate, optimizers=(opt, sched), callbacks=[trainer_callback(dp)])
    keep_out_of_ddp(tr, dp)
    tr.train()
    torch.cuda.synchronize()
    wrapped = type(tr.model_wrapped).__name__
    q.put((rank, dp.master.cpu().numpy(), wrapped, dp.bucket.world, dp.optimizer.t))
    dist.barrier()
    dist.destroy_process_group()


def test_hf_trainer_under_two_ranks_stays_out_of_ddp(tmp_path):
    """transformers.Trainer under a two-rank launch (what `torchrun --nproc_per_node` sets up; gloo, the ranks share
    the one GPU) with keep_out_of_ddp: accelerate does NOT wrap the model in DistributedDataParallel (whose reducer would wait for autograd
    gradients that never come), attach()'s bucketed all-reduce averages the ranks' gradients, both ranks end with the same parameters, and
    those are the parameters of one process training on the union of the ranks' rows."""
    import multiprocessing as mp
    import socket
    from moka_amd.parallel import attach
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hf_rank, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r_, master, wrapped, world, t = q.get(timeout=600)
        got[r_] = (torch.from_numpy(master), wrapped, world, t)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] == "TinyLM" and got[0][2] == 2 and got[0][3] == 3         # unwrapped, two ranks, three optimizer steps
    assert torch.equal(got[0][0], got[1][0])
    # one process on the union of the rows (step k: rows 4k .. 4k + 3 -- rank 0 the first two, rank 1 the other two)
    dev = torch.device("cuda:0")
    rows = _data(n=12)
    ref = TinyLM(dev)
    dp_r = attach(ref, n_buckets=2, lr=2e-3, weight_decay=0.0)
    for k in range(3):
        batch = {kk: v.to(dev) for kk, v in _collate(rows[4 * k:4 * k + 4]).items()}
        ref(**batch)["loss"].backward()
        dp_r.step()
    torch.cuda.synchronize()
    err = ((got[0][0].to(dev) - dp_r.master).norm() / dp_r.master.norm()).item()
    assert err <= 2e-3, err                       # (the hooked bf16 projector gradient is rounded per rank; AdamW at step 1-3 follows its sign)
