"""CPU, world_size 2 over gloo: the bucketed flat-gradient all-reduce used for data-parallel
training (moka_amd/parallel.py) gives every rank the averaged gradient, bucket by bucket."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_layers, n_buckets, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from moka_amd.parallel import FlatGradBucket
    sizes = [37 + 11 * l for l in range(n_layers)]
    ends = []
    acc = 0
    for s in sizes:
        acc += s
        ends.append(acc)
    bucket = FlatGradBucket(acc, ends, "cpu", n_buckets=n_buckets)
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.randn(acc, generator=g)
    for step in range(2):                       # two steps: state must reset cleanly
        bucket.zero_()
        for l in range(n_layers - 1, -1, -1):   # backward order: last layer first
            bucket.layer_slice(l).add_(local[(ends[l - 1] if l else 0):ends[l]] * (step + 1))
            bucket.layer_done(l)
        bucket.finish(average=True)
        exp = sum(torch.randn(acc, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world * (step + 1)
        ok = torch.allclose(bucket.flat, exp, rtol=1e-6, atol=1e-6)
        q.put((rank, step, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_layers,n_buckets", [(8, 4), (5, 8), (7, 3)])
def test_bucketed_allreduce_world2(n_layers, n_buckets):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_layers, n_buckets, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = [q.get(timeout=5) for _ in range(4)]
    assert all(ok for _, _, ok in res), res


def test_single_process_is_a_noop():
    sys.path.insert(0, ROOT)
    from moka_amd.parallel import FlatGradBucket, bind_param_grads
    b = FlatGradBucket(10, [4, 10], "cpu", n_buckets=2)
    b.flat.fill_(3.0)
    b.layer_done(1)
    b.layer_done(0)
    b.finish()
    assert float(b.flat.sum()) == 30.0
    p = torch.nn.Parameter(torch.zeros(2, 3))
    bind_param_grads([p], b, [4])
    assert p.grad.data_ptr() == b.flat[4:].data_ptr()


def _shard_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from moka_amd.parallel import ShardedFrozenBase
    g = torch.Generator().manual_seed(7)           # same full weights on every rank before sharding
    full = [[("q.weight", torch.randn(6, 5, generator=g)), ("mlp.weight", torch.randn(7, 3, generator=g))] for _ in range(5)]
    ref = [[(n, t.clone()) for n, t in layer] for layer in full]
    store = ShardedFrozenBase(full, "cpu", dtype=torch.float32)
    total = sum(t.numel() for layer in ref for _, t in layer)
    ok = store.shard_bytes() <= (total // world + world * len(ref)) * 4           # each rank keeps ~1/N
    order = list(range(5)) + list(range(4, -1, -1))                                # forward, then backward
    for i, l in enumerate(order):
        nxt = order[i + 1] if i + 1 < len(order) else None
        got = store.layer(l, prefetch_next=nxt)
        for n, t in ref[l]:
            ok = ok and torch.equal(got[n], t)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_frozen_base_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = [q.get(timeout=5) for _ in range(2)]
    assert all(ok for _, ok in res), res


def test_sharded_frozen_base_single_process():
    sys.path.insert(0, ROOT)
    from moka_amd.parallel import ShardedFrozenBase
    layers = [[("w", torch.arange(12.0).reshape(3, 4) + 100 * l)] for l in range(3)]
    ref = [layer[0][1].clone() for layer in layers]
    store = ShardedFrozenBase(layers, "cpu", dtype=torch.float32)
    for l in (0, 1, 2, 1, 0):
        assert torch.equal(store.layer(l, prefetch_next=None)["w"], ref[l])


def test_attach_flattens_the_adapter_layer_by_layer_on_cpu():
    """attach(): parameters re-seated as views of ONE bf16 buffer in decoder-layer order, fp32 master copy, fp32 gradient sinks on
    both mirrors, one bucket entry per layer; the optimizer kernel itself is GPU-only (loud error on CPU tensors)."""
    sys.path.insert(0, ROOT)
    from moka_amd import _lib
    from moka_amd.decoder import LlamaDims, MokaLlamaStack
    from moka_amd.parallel import attach
    from moka_amd.peft_hyper import Linear
    dims = LlamaDims(hidden=64, ff=96, n_heads=2, n_kv_heads=2)

    def make(d_in, d_out):
        return Linear(d_in, d_out, r=(4, 4, 4), lora_alpha=16, lora_nums=3, blc_weight=1.0, blc_alpha=1, lora_dropout=0.0,
                      loramethod="train", bias=False)

    st = MokaLlamaStack(dims, 2, make).to(torch.bfloat16)
    for n, p in st.named_parameters():
        p.requires_grad = "lora_" in n
    ref = {n: p.detach().clone() for n, p in st.named_parameters() if "lora_" in n}
    dp = attach(st, n_buckets=2)
    assert dp.bucket.n_layers == 2 and len(dp.names) == 2 * 7 * 4
    assert all(n.startswith("layers.0.") for n in dp.names[:28]) and all(n.startswith("layers.1.") for n in dp.names[28:])
    assert dp.offsets == sorted(dp.offsets) and dp.bucket.layer_end[0] == dp.offsets[28]
    for n, o in zip(dp.names, dp.offsets):
        p = dict(st.named_parameters())[n]
        assert p.data_ptr() == dp.work[o:].data_ptr() and torch.equal(p.detach(), ref[n])
        assert torch.equal(dp.master[o:o + p.numel()].view(p.shape), ref[n].float())
    q = st.layers[1].self_attn.q_proj
    assert q._moka_sinks["B"].shape == q.lora_B0.weight.shape and q._moka_sinks["B"].dtype == torch.float32
    assert [tuple(a.shape) for a in q._moka_sinks["A"]] == [tuple(q.lora_A0.weight.shape)] * 3
    assert q._moka_sinks["A"][1].data_ptr() == dp.bucket.flat[dp.offsets[dp.names.index("layers.1.self_attn.q_proj.lora_A1.weight")]:].data_ptr()
    with pytest.raises(_lib.MokaError):
        dp.step()                                   # no CPU path for the optimizer kernel
    dp.detach()
    assert q._moka_sinks is None


class _ToyLayer(torch.nn.Module):
    def __init__(self, d):
        super().__init__()
        self.base = torch.nn.Linear(d, d, bias=True)
        self.norm = torch.nn.LayerNorm(d)
        self.lora_A = torch.nn.Linear(d, 4, bias=False)
        self.lora_B = torch.nn.Linear(4, d, bias=False)

    def forward(self, x):
        return x + torch.tanh(self.base(self.norm(x))) + self.lora_B(self.lora_A(x))


class _ToyStack(torch.nn.Module):
    def __init__(self, d, n):
        super().__init__()
        self.layers = torch.nn.ModuleList(_ToyLayer(d) for _ in range(n))

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x


def _toy(d=24, n=5):
    torch.manual_seed(3)
    st = _ToyStack(d, n)
    for n_, p in st.named_parameters():
        p.requires_grad = "lora_" in n_
    return st


def _toy_run(st, x):
    out = st(x)
    out.square().sum().backward()
    return out.detach().clone(), {n: p.grad.clone() for n, p in st.named_parameters() if p.grad is not None}


def _shard_stack_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from moka_amd.parallel import ShardedFrozenBase
    x = torch.randn(3, 7, 24, generator=torch.Generator().manual_seed(9)).requires_grad_(True)
    ref_out, ref_g = _toy_run(_toy(), x)
    gx_ref = x.grad.clone()
    x.grad = None
    st = _toy()
    full = sum(p.numel() for n, p in st.named_parameters() if "lora_" not in n)
    store = ShardedFrozenBase.shard_stack(st)
    ok = store.shard_bytes() <= (full // world + world * 5) * 4
    ok = ok and all(p.numel() == 0 for n, p in st.named_parameters() if "lora_" not in n)       # the full copies are gone
    for _ in range(2):                                                                         # two steps: buffers recycle cleanly
        for p in st.parameters():
            p.grad = None
        x.grad = None
        out, g = _toy_run(st, x)
        ok = ok and torch.allclose(out, ref_out, rtol=1e-6, atol=1e-6) and torch.allclose(x.grad, gx_ref, rtol=1e-5, atol=1e-6)
        ok = ok and set(g) == set(ref_g) and all(torch.allclose(g[n], ref_g[n], rtol=1e-5, atol=1e-6) for n in g)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_stack_hooks_world2_forward_and_backward():
    """ZeRO-3-style frozen base through the layer hooks (gather before a layer's forward AND before its backward, next layer
    prefetched): outputs, input gradient and adapter gradients equal the unsharded model's, on 2 ranks over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_stack_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = [q.get(timeout=5) for _ in range(2)]
    assert all(ok for _, ok in res), res


def _dp_semantics_worker(rank, world, port, comm_bf16, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from moka_amd.parallel import FlatGradBucket
    from oracle import cases as C
    from oracle import moka_oracle as O
    cd = C.make_case_data("avt_tiny")                       # B = 2 samples: one per rank
    c = cd.case
    M = len(cd.A)
    sizes = [c.d_out * c.r] + [c.r * c.d_in] * M            # dB, dA_0..dA_{M-1} -> one "layer" of a flat bucket
    bucket = FlatGradBucket(sum(sizes), [sum(sizes)], "cpu", n_buckets=1, comm_dtype=torch.bfloat16 if comm_bf16 else None)

    def grads(lo, hi):
        masks = [m[lo:hi] for m in cd.masks]
        rt = O.routing_from_avt_masks(masks)
        y0 = torch.zeros(hi - lo, c.S, c.d_out, dtype=torch.float64)
        _, ctx = O.adapter_forward(cd.x[lo:hi], y0, cd.A, cd.Bw, rt, c.alpha / c.r, [1.0] * M, c.w, c.r)
        _, dA, dB, _ = O.adapter_backward(cd.gy[lo:hi], ctx)
        return torch.cat([dB.reshape(-1)] + [a.reshape(-1) for a in dA]).float()

    bucket.flat.copy_(grads(rank, rank + 1))                # what the weight-gradient kernels of this rank would leave
    bucket.layer_done(0)
    bucket.finish(average=True)
    full = grads(0, c.B) / world                            # gradient of the MEAN loss over the whole batch
    err = ((bucket.flat - full).norm() / full.norm()).item()
    q.put((rank, err))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("comm_bf16", [False, True])
def test_sample_sharding_mean_of_rank_gradients_is_the_full_batch_gradient(comm_bf16):
    """DP semantics of the adapter gradients: the batch is sharded by sample, every rank's dB / dA_m (here from
    the oracle -- the kernels' version of this test is tests/test_gpu_dp.py) go through the flat bucket's all-reduce, and the
    average equals the gradient of the mean loss over the whole batch (samples are independent, lora.py:485 / layer.py:628)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_semantics_worker, args=(r, 2, port, comm_bf16, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = [q.get(timeout=5) for _ in range(2)]
    tol = 2e-2 if comm_bf16 else 1e-6
    assert all(err <= tol for _, err in res), res


def _force_comm_worker(port, q):
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    from moka_amd.parallel import FlatGradBucket
    ends = [40, 100, 164, 200]
    plain = FlatGradBucket(200, ends, "cpu", n_buckets=2)
    forced = FlatGradBucket(200, ends, "cpu", n_buckets=2, force_comm=True)
    calls = []
    forced.on_reduced = lambda lo, hi: calls.append((lo, hi))       # (called behind every shipped bucket, last bucket first)
    g = torch.randn(200, generator=torch.Generator().manual_seed(1))
    out = []
    for b in (plain, forced):
        b.flat.copy_(g)
        for l in range(3, -1, -1):
            b.layer_done(l)
        pending = len(b._pending)
        b.finish(average=True)
        out.append((b.comm, b.world, pending, b.flat.clone()))
    dist.destroy_process_group()
    assert calls == [(100, 200), (0, 100)], calls
    q.put([(c, w, p, t.numpy()) for c, w, p, t in out])


def test_force_comm_runs_the_collectives_in_a_group_of_one_rank():
    """FlatGradBucket(force_comm=True): a process group of ONE rank still ships every bucket through dist.all_reduce (the sum over one
    rank is the identity, the average divides by one) -- the switch that lets the N > 1 communication path run on one device."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_force_comm_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=120)
    p.join(60)
    assert p.exitcode == 0
    (c0, w0, p0, t0), (c1, w1, p1, t1) = res
    assert (c0, w0, p0) == (False, 1, 0) and (c1, w1, p1) == (True, 1, 2)      # two buckets of two layers were really shipped
    assert (t0 == t1).all()


def test_force_comm_needs_a_process_group():
    from moka_amd.parallel import FlatGradBucket
    with pytest.raises(RuntimeError):
        FlatGradBucket(8, [8], "cpu", force_comm=True)


def test_flat_adamw_segments_follow_the_no_decay_ranges():
    """FlatAdamW._segments: a range of the flat buffers cut at the boundaries of the no-decay ranges (biases / norm weights), in order."""
    from moka_amd.parallel import FlatAdamW
    m, g = torch.zeros(64), torch.zeros(64)
    opt = FlatAdamW(m, g, None, lr=1e-3, weight_decay=0.1)
    assert list(opt._segments(0, 64)) == [(0, 64, True)]
    opt.no_decay_ranges = [(8, 16), (32, 40)]
    assert list(opt._segments(0, 64)) == [(0, 8, True), (8, 16, False), (16, 32, True), (32, 40, False), (40, 64, True)]
    assert list(opt._segments(12, 36)) == [(12, 16, False), (16, 32, True), (32, 36, False)]
    assert list(opt._segments(16, 32)) == [(16, 32, True)]
    assert list(opt._segments(8, 16)) == [(8, 16, False)]
    pulled = []
    opt.hyper = lambda: (pulled.append(1), dict(lr=5e-2, betas=(0.8, 0.9), eps=1e-6, weight_decay=0.0))[1]
    opt._pull_hyper()
    assert (opt.lr, opt.betas, opt.eps, opt.weight_decay) == (5e-2, (0.8, 0.9), 1e-6, 0.0) and pulled == [1]


def test_attach_marks_biases_and_norm_weights_as_no_decay_on_cpu():
    """attach(no_decay="hf"): the ranges of the flat buffers that belong to biases / normalisation layers (HF Trainer's rule)."""
    from moka_amd.parallel import attach

    class Proj(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc = torch.nn.Linear(8, 8, bias=True)
            self.norm = torch.nn.LayerNorm(8)

        def forward(self, x):
            return self.norm(self.fc(x))

    model = Proj().to(torch.bfloat16)
    dp = attach(model, optimizer=True, lr=1e-3, weight_decay=0.1, defer_dA=False)
    by = dict(zip(dp.names, zip(dp.offsets, dp.sizes)))
    want = sorted((o, o + (sz + 7) // 8 * 8) for n, (o, sz) in by.items() if n != "fc.weight")
    merged = []
    for lo, hi in want:
        if merged and merged[-1][1] == lo:
            merged[-1] = (merged[-1][0], hi)
        else:
            merged.append((lo, hi))
    assert dp.optimizer.no_decay_ranges == merged
    o, sz = by["fc.weight"]
    assert all(not (lo <= o < hi) for lo, hi in dp.optimizer.no_decay_ranges)
    assert attach(Proj().to(torch.bfloat16), weight_decay=0.1, defer_dA=False, no_decay=None).optimizer.no_decay_ranges == []


def test_tail_bucket_layout():
    """tail_layers = t: the bucket that ships LAST (it holds layer 0; nothing of the backward is left to hide its all-reduce) is layers
    [0, t), the other layers split evenly over the remaining buckets."""
    from moka_amd.parallel import FlatGradBucket
    ends = [10 * (i + 1) for i in range(32)]
    b = FlatGradBucket(320, ends, "cpu", n_buckets=8, tail_layers=1)
    firsts = b.bucket_firsts()
    assert firsts[0] == 0 and firsts[1] == 1 and len(firsts) == 8
    sizes = [len(b.bucket_layers(f)) for f in firsts]
    assert sizes[0] == 1 and sum(sizes) == 32 and max(sizes[1:]) - min(sizes[1:]) <= 5 and sizes[1] == 5
    assert b.bucket_bounds(0) == (0, 10) and b.bucket_bounds(1) == (10, 60) and b.bucket_bounds(firsts[-1])[1] == 320
    assert b.last_bucket_bytes() == 10 * 4
    # default layout unchanged: equal groups
    u = FlatGradBucket(320, ends, "cpu", n_buckets=8)
    assert u.bucket_firsts() == list(range(0, 32, 4)) and u.bucket_bounds(4) == (40, 80) and u.last_bucket_bytes() == 40 * 4
    assert [u.is_bucket_first(l) for l in range(6)] == [True, False, False, False, True, False]


def _payload_worker(rank, world, port, q):
    """The update of a bucket behind its all-reduce (on_reduced) with the bf16 payload: the callback must see the SUMMED gradient widened
    back into the fp32 buffer, once, and finish() must not overwrite what the callback left (here: a zeroed slice)."""
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from moka_amd.parallel import FlatGradBucket
    ends = [40, 100, 164, 200, 260]
    res = {}
    for name, cd in (("fp32", None), ("bf16", torch.bfloat16)):
        b = FlatGradBucket(260, ends, "cpu", n_buckets=3, comm_dtype=cd, tail_layers=1)
        w = torch.zeros(260)
        seen = []

        def upd(lo, hi, b=b, w=w, seen=seen):
            seen.append((lo, hi))
            w[lo:hi] -= 0.5 * b.flat[lo:hi] / world          # "optimizer": plain SGD on the averaged gradient
            b.flat[lo:hi].zero_()
        b.on_reduced = upd
        for step in range(2):
            g = torch.randn(260, generator=torch.Generator().manual_seed(7 * step + rank))
            b.flat.add_(g)
            for l in range(4, -1, -1):
                b.layer_done(l)
            b.finish(average=False)
            assert float(b.flat.abs().max()) == 0.0          # finish() left the zeroed slices alone
        assert seen[:3] == [(164, 260), (40, 164), (0, 40)], seen     # layers [3,5), [1,3), [0,1): the tail bucket last
        res[name] = w
    exp = torch.zeros(260)
    for step in range(2):
        exp -= 0.5 * sum(torch.randn(260, generator=torch.Generator().manual_seed(7 * step + r)) for r in range(world)) / world
    q.put((rank, float((res["fp32"] - exp).abs().max()), float((res["bf16"] - exp).abs().max() / exp.abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_update_behind_a_bf16_payload_world2():
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_payload_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for _, e32, e16 in res:
        assert e32 <= 1e-6 and e16 <= 1e-2, res             # bf16 payload: 2^-9 per addend


def test_geometric_bucket_layout():
    """bucket_sizes spells the layout out (layers per bucket from layer 0 up); geometric_buckets: 1, 3, 9, ... -- the buckets that ship late are small."""
    from moka_amd.parallel import FlatGradBucket, geometric_buckets
    assert geometric_buckets(32) == [1, 3, 9, 19] and geometric_buckets(40) == [1, 3, 9, 27] and geometric_buckets(80) == [1, 3, 9, 27, 40]
    assert geometric_buckets(1) == [1] and geometric_buckets(4) == [1, 3] and all(sum(geometric_buckets(n)) == n for n in range(1, 100))
    ends = [10 * (i + 1) for i in range(32)]
    b = FlatGradBucket(320, ends, "cpu", bucket_sizes=geometric_buckets(32))
    assert b.bucket_firsts() == [0, 1, 4, 13] and [len(b.bucket_layers(f)) for f in b.bucket_firsts()] == [1, 3, 9, 19]
    assert b.bucket_bounds(13) == (130, 320) and b.bucket_bounds(0) == (0, 10) and b.last_bucket_bytes() == 40
    with pytest.raises(ValueError):
        FlatGradBucket(320, ends, "cpu", bucket_sizes=[1, 3, 9])


def test_pass_running_flag_is_cleared_when_the_backward_pass_ends():
    """Builds without torch._C._current_graph_task_id: `_in_backward` falls back to the handle's `_pass_running`, which an engine
    callback clears at the end of the pass that set it (`_bwd_active` stays until finish() / step())."""
    import types
    import torch
    from moka_amd import parallel as P

    dp = types.SimpleNamespace(_bwd_active=False, _pass_running=False)
    seen = []

    class Node(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2

        @staticmethod
        def backward(ctx, g):
            P._mark_pass(dp)
            seen.append((dp._bwd_active, dp._pass_running))
            return g * 2

    x = torch.ones(3, requires_grad=True)
    Node.apply(Node.apply(x)).sum().backward()
    assert seen == [(True, True), (True, True)]
    assert dp._bwd_active and not dp._pass_running            # the pass has ended; the work it reported is still to be finished
    P._mark_pass(dp)                                          # outside a backward pass: nothing to wait for
    assert not dp._pass_running
