"""Data-parallel synchronisation of the adapter gradients (one process per GPU, RCCL over xGMI).

The reference leaves this to DeepSpeed ZeRO-2 (bucketed reduce-scatter of the trainable
gradients, ``VisualText/zero_stage2_config.json:2-10``, ``AudioVisualText/deepspeed/
stage2-offload.json:37-49``).  Only the adapter is trainable (7B, r=16, M=3: 76.4 M parameters),
so here every adapter gradient lives in ONE flat fp32 buffer the weight-gradient kernels
accumulate into directly (``dA_acc`` / ``dB_acc`` of include/moka_hip.h are views of it), and the
buffer is all-reduced in a few contiguous slices ("buckets" of whole decoder layers):

  * the backward walks the layers last -> first; as soon as the launches of a bucket's layers
    are enqueued, ``layer_done`` records an event and starts ``all_reduce`` of that slice on a
    side stream -- RCCL runs while the remaining layers' backward kernels stream HBM;
  * ``finish`` joins the side stream and averages.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): the payload is small (306 MB fp32 for the
whole 7B adapter), so a handful of large slices keeps every collective bandwidth- rather than
latency-bound, and nothing here depends on the backend: the same code runs over ``gloo`` on CPU
tensors (tests/test_parallel_gloo.py).
"""
from __future__ import annotations

import math
import re
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist
from torch import nn


class FlatGradBucket:
    """Flat fp32 gradient buffer + per-layer slice boundaries + bucketed asynchronous all-reduce."""

    def __init__(self, numel: int, layer_end: Sequence[int], device, n_buckets: int = 8,
                 process_group=None, dtype=torch.float32, comm_dtype: Optional[torch.dtype] = None, force_comm: bool = False,
                 tail_layers: Optional[int] = None, bucket_sizes: Optional[Sequence[int]] = None):
        """tail_layers: the backward walks the layers last -> first, so the bucket that holds layer 0 is the one whose all-reduce nothing
        is left to hide; ``tail_layers = t`` makes that bucket layers [0, t) and splits the other layers evenly over the remaining
        ``n_buckets - 1`` buckets (None: ``n_buckets`` equal groups) -- the exposed tail of the step is one small collective.
        bucket_sizes: the layout spelled out -- layers per bucket from layer 0 UPWARD (the first entry is the bucket that ships last), summing
        to the number of layers; overrides n_buckets / tail_layers.  ``geometric_buckets(n)`` = 1, 3, 9, ... : every bucket's all-reduce has
        about a third of its own layers' backward left to hide behind, and there are few buckets (where every bucket is a hipGraph of its own,
        the launch chains meet at every bucket boundary).
        comm_dtype: payload type of the all-reduce (None = the buffer's own fp32).  torch.bfloat16 halves the bytes on
        xGMI (7B r=16: 153 instead of 306 MB per step -- the figure SURVEY.md 8(e) sized): a bucket is rounded to bf16 into a
        staging buffer, summed there and widened back; accumulation across micro-batches stays fp32.
        force_comm: run the collectives even in a process group of ONE rank (default: a single rank short-circuits them).  A
        one-rank RCCL communicator is a real ``ProcessGroupNCCL`` -- its own stream, in-place asynchronous all-reduce, ``wait()`` =
        stream wait -- so the whole communication path (side stream, bucket hooks, ``on_reduced``) can be exercised and priced on one GPU."""
        if not layer_end or layer_end[-1] != numel:
            raise ValueError("layer_end must be increasing offsets ending at numel")
        self.flat = torch.zeros(numel, dtype=dtype, device=device)
        self.comm_dtype = None if comm_dtype in (None, dtype) else comm_dtype
        self._stage = torch.empty(numel, dtype=self.comm_dtype, device=device) if self.comm_dtype is not None else None
        self.layer_end = list(layer_end)
        self.n_layers = len(self.layer_end)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        if force_comm and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("FlatGradBucket(force_comm=True) needs an initialised process group (one rank is enough)")
        self.comm = self.world > 1 or bool(force_comm)      # do the collectives run?
        self.layers_per_bucket = max(1, -(-self.n_layers // max(1, n_buckets)))
        # buckets = contiguous groups of layers [first, end); a bucket ships when its FIRST layer reports (layers arrive last -> first)
        if bucket_sizes is not None:
            sizes_ = [int(v) for v in bucket_sizes]
            if any(v < 1 for v in sizes_) or sum(sizes_) != self.n_layers:
                raise ValueError(f"bucket_sizes {sizes_} must be positive and sum to the {self.n_layers} layers")
            firsts, acc_ = [], 0
            for v in sizes_:
                firsts.append(acc_)
                acc_ += v
            self.layers_per_bucket = max(sizes_)
        elif tail_layers is not None and 0 < int(tail_layers) < self.n_layers and n_buckets > 1:
            t = int(tail_layers)
            per = max(1, -(-(self.n_layers - t) // (n_buckets - 1)))
            firsts = [0] + list(range(t, self.n_layers, per))
            self.layers_per_bucket = per             # (the even groups; the tail bucket is smaller)
        else:
            firsts = list(range(0, self.n_layers, self.layers_per_bucket))
        self._bucket_end = {f: (firsts[i + 1] if i + 1 < len(firsts) else self.n_layers) for i, f in enumerate(firsts)}
        self.is_cuda = self.flat.is_cuda
        self.comm_stream = torch.cuda.Stream(device=device) if (self.is_cuda and self.comm) else None
        self.on_reduced = None       # callable(lo, hi), run on the communication stream behind a bucket's all-reduce (fp32 payload only)
        self._pending: List = []

    # ---------------------------------------------------------------- views
    def layer_slice(self, l: int) -> torch.Tensor:
        lo = self.layer_end[l - 1] if l > 0 else 0
        return self.flat[lo:self.layer_end[l]]

    def is_bucket_first(self, l: int) -> bool:
        """Is l the first (lowest) layer of a bucket, i.e. the layer whose backward completes the bucket?"""
        return l in self._bucket_end

    def bucket_layers(self, l: int) -> range:
        """The layers of the bucket whose FIRST layer is l."""
        return range(l, self._bucket_end[l])

    def bucket_firsts(self) -> List[int]:
        return sorted(self._bucket_end)

    def bucket_bounds(self, l: int):
        """Slice [lo, hi) of the bucket whose FIRST layer is l (buckets are contiguous groups of layers)."""
        hi_layer = self._bucket_end[l]
        lo = self.layer_end[l - 1] if l > 0 else 0
        return lo, self.layer_end[hi_layer - 1]

    def last_bucket_bytes(self) -> int:
        """Payload of the bucket that ships last (the one that holds layer 0): what the step's exposed communication tail moves."""
        lo, hi = self.bucket_bounds(0)
        return (hi - lo) * (self._stage.element_size() if self._stage is not None else self.flat.element_size())

    def zero_(self):
        self.flat.zero_()

    # ---------------------------------------------------------------- backward hooks
    def layer_done(self, l: int):
        """Call after layer l's backward launches are enqueued (layers arrive last -> first)."""
        if not self.comm or not self.is_bucket_first(l):
            return
        lo, hi = self.bucket_bounds(l)
        sl = self.flat[lo:hi]
        if self.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                if self._stage is not None:
                    st = self._stage[lo:hi]
                    st.copy_(sl)                                   # fp32 -> bf16 on the side stream
                    wk = dist.all_reduce(st, group=self.group, async_op=True)
                    if self.on_reduced is not None:
                        wk.wait()
                        sl.copy_(st)                               # the summed bf16 payload widened back, still on the communication stream
                        self.on_reduced(lo, hi)                    # (the slice's update reads the fp32 buffer as with the fp32 payload)
                        self._pending.append((wk, lo, hi, True))
                    else:
                        self._pending.append((wk, lo, hi, False))
                else:
                    wk = dist.all_reduce(sl, group=self.group, async_op=True)
                    if self.on_reduced is not None:
                        wk.wait()                                  # (the communication stream waits, not the host)
                        self.on_reduced(lo, hi)                    # e.g. FlatAdamW.step_range: the bucket's update overlaps the rest of the backward
                    self._pending.append((wk, lo, hi, True))
        else:
            # CPU tensors (gloo; tests): the same order of events, synchronously where a callback needs the sum
            if self._stage is not None:
                st = self._stage[lo:hi]
                st.copy_(sl)
                wk = dist.all_reduce(st, group=self.group, async_op=True)
                if self.on_reduced is not None:
                    wk.wait()
                    sl.copy_(st)
                    self.on_reduced(lo, hi)
                    self._pending.append((wk, lo, hi, True))
                else:
                    self._pending.append((wk, lo, hi, False))
            else:
                wk = dist.all_reduce(sl, group=self.group, async_op=True)
                if self.on_reduced is not None:
                    wk.wait()
                    self.on_reduced(lo, hi)
                self._pending.append((wk, lo, hi, True))

    def finish(self, average: bool = True):
        """Join the outstanding collectives; afterwards ``flat`` holds the (averaged) global gradient."""
        if not self.comm:
            return
        if self.is_cuda:
            done = torch.cuda.Event()
            with torch.cuda.stream(self.comm_stream):
                for wk, lo, hi, widened in self._pending:
                    wk.wait()
                    if self._stage is not None and not widened:
                        self.flat[lo:hi].copy_(self._stage[lo:hi])     # widen the summed payload back
                done.record(self.comm_stream)
            torch.cuda.current_stream(self.flat.device).wait_event(done)
        else:
            for wk, lo, hi, widened in self._pending:
                wk.wait()
                if self._stage is not None and not widened:
                    self.flat[lo:hi].copy_(self._stage[lo:hi])
        self._pending.clear()
        if average and self.world > 1:
            self.flat.div_(self.world)


def geometric_buckets(n_layers: int, tail: int = 1, ratio: float = 3.0) -> List[int]:
    """Layers per gradient bucket from layer 0 upward: tail, ~ratio x tail, ... (the last entry takes what is left): 32 layers -> [1, 3, 9, 19].
    The backward walks the layers last -> first, so the LARGE buckets ship early and overlap with plenty of backward, and the buckets that
    ship late -- little or nothing left to hide them -- are small."""
    sizes, left, cur = [], int(n_layers), max(1, int(tail))
    while left > 0:
        take = min(left, cur)
        if left - take < cur:                        # (what is left would be smaller than this bucket: merge it)
            take = left if sizes else take
        sizes.append(take)
        left -= take
        cur = max(cur + 1, int(round(cur * ratio)))
    return sizes


class FlatAdamW:
    """AdamW on the flat adapter buffers in ONE kernel (``moka_adamw_flat``): gradient averaging (``grad_scale``),
    the decoupled-weight-decay Adam update of the fp32 master copy, the bf16 working copy the kernels read, and
    the zeroing of the gradient buffer for the next accumulation.  Same arithmetic as ``torch.optim.AdamW``
    (tests/test_gpu_parallel.py compares them); the reference leaves this step to DeepSpeed ZeRO-2 / HF Trainer
    (``VisualText/zero_stage2_config.json:2-10``).  There is no CPU path: the buffers must live on the GPU."""

    def __init__(self, master: torch.Tensor, grad: torch.Tensor, work: Optional[torch.Tensor] = None, lr: float = 1e-3,
                 betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        if master.dtype != torch.float32 or grad.dtype != torch.float32 or master.numel() != grad.numel():
            raise TypeError("FlatAdamW: master and grad must be fp32 buffers of the same length")
        if work is not None and (work.dtype != torch.bfloat16 or work.numel() != master.numel()):
            raise TypeError("FlatAdamW: the working copy must be a bf16 buffer of the same length")
        for t in (master, grad) + (() if work is None else (work,)):
            if not t.is_contiguous():
                raise ValueError("FlatAdamW: buffers must be contiguous")
        self.master, self.grad, self.work = master, grad, work
        self.exp_avg = torch.zeros_like(master)
        self.exp_avg_sq = torch.zeros_like(master)
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.t = 0
        # [lo, hi) ranges of the flat buffers that take NO weight decay (biases, norm weights: what HF Trainer's default optimizer
        # excludes, AudioVisualText/trainer.py leaves create_optimizer alone); sorted, disjoint, multiples of 4
        self.no_decay_ranges: List = []
        # callable() -> dict(lr=, betas=, eps=, weight_decay=) evaluated when a step BEGINS (MokaFlatOptimizer installs one that reads
        # param_groups[0]: with the optimizer inside the backward a step begins long before optimizer.step() is called)
        self.hyper = None
        self._state = None                           # 8 floats on the device: the step's coefficients (moka_adamw_begin_dev)

    def _pull_hyper(self) -> None:
        if self.hyper is not None:
            h = self.hyper()
            self.lr, self.eps, self.weight_decay = float(h["lr"]), float(h["eps"]), float(h["weight_decay"])
            self.betas = (float(h["betas"][0]), float(h["betas"][1]))

    def _segments(self, lo: int, hi: int):
        """[lo, hi) cut at the boundaries of the no-decay ranges: (a, b, decays) pieces in order."""
        pos = lo
        for a, b in self.no_decay_ranges:
            a, b = max(a, lo), min(b, hi)
            if a >= b:
                continue
            if a > pos:
                yield pos, a, True
            yield a, b, False
            pos = b
        if pos < hi:
            yield pos, hi, True

    def step(self, grad_scale: float = 1.0, zero_grad: bool = True) -> None:
        from . import _lib
        if not self.master.is_cuda:
            raise _lib.MokaError("moka_amd: FlatAdamW runs as a HIP kernel; the buffers live on %s" % self.master.device)
        self._pull_hyper()
        self.t += 1
        lib = _lib.load()
        stream = torch.cuda.current_stream(self.master.device).cuda_stream
        for a, b, decays in self._segments(0, self.master.numel()):
            _lib.check(lib.moka_adamw_flat(self.master.data_ptr() + 4 * a, None if self.work is None else self.work.data_ptr() + 2 * a,
                                           self.grad.data_ptr() + 4 * a, self.exp_avg.data_ptr() + 4 * a, self.exp_avg_sq.data_ptr() + 4 * a,
                                           b - a, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay if decays else 0.0,
                                           self.t, float(grad_scale), 1 if zero_grad else 0, stream), "moka_adamw_flat")

    # -- the same step in slices, with the step-dependent coefficients in device memory (``moka_adamw_flat_dev``): launches that can be
    #    captured in a hipGraph, or enqueued per gradient bucket while the backward of the earlier layers is still running
    def begin_step(self, device_counter: bool = False) -> None:
        """Count the step and ENQUEUE, on the current stream, the one-thread launch that writes its coefficients {lr / (1 - beta1^t),
        1 / sqrt(1 - beta2^t), 1 - lr * wd} into device memory (``moka_adamw_begin_dev``).  The inputs are launch arguments -- copied when
        the launch is enqueued -- so a host that runs steps ahead of the GPU cannot disturb a step that has not read its coefficients
        yet (a pinned staging buffer, the round-3 form, could).  device_counter: the launch counts the steps itself instead of taking
        t from the host: captured in a hipGraph it advances by one per replay (the caller keeps ``self.t`` in step for bookkeeping).
        NOTE: lr / betas / weight_decay are launch arguments too, so a CAPTURED begin_step replays with the hyper-parameters it was captured with:
        a schedule that changes them needs a re-capture (or the live launch: ``MokaFlatOptimizer`` / ``attach`` never capture it)."""
        from . import _lib
        if not self.master.is_cuda:
            raise _lib.MokaError("moka_amd: FlatAdamW runs as a HIP kernel; the buffers live on %s" % self.master.device)
        if self._state is None:
            self._state = torch.zeros(8, dtype=torch.float32, device=self.master.device)
        self._pull_hyper()
        self.t += 1
        stream = torch.cuda.current_stream(self.master.device).cuda_stream
        _lib.check(_lib.load().moka_adamw_begin_dev(self._state.data_ptr(), self.lr, self.betas[0], self.betas[1], self.weight_decay,
                                                    0 if device_counter else self.t, stream), "moka_adamw_begin_dev")

    def set_device_step(self, t: int) -> None:
        """Make the device-side step counter agree with ``t`` (before capturing / replaying a ``begin_step(device_counter=True)`` launch)."""
        if self._state is None:
            self._state = torch.zeros(8, dtype=torch.float32, device=self.master.device)
        self._state[3:4].view(torch.int32).fill_(int(t))

    def upload_coef(self) -> None:
        """Kept for callers of the round-3 interface: ``begin_step`` now writes the coefficients on the device itself."""
        return None

    def step_range(self, lo: int, hi: int, grad_scale: float = 1.0, zero_grad: bool = True) -> None:
        """The update of parameters [lo, hi) on the current stream, behind ``begin_step()`` in stream order (or behind an event that
        is); lo must be a multiple of 4."""
        from . import _lib
        if lo % 4 or not (0 <= lo <= hi <= self.master.numel()):
            raise ValueError(f"FlatAdamW.step_range: bad range [{lo}, {hi})")
        if self._state is None:
            raise RuntimeError("FlatAdamW.step_range before begin_step()")
        lib = _lib.load()
        stream = torch.cuda.current_stream(self.master.device).cuda_stream
        for a, b, decays in self._segments(lo, hi):
            _lib.check(lib.moka_adamw_flat_dev(self.master.data_ptr() + 4 * a, None if self.work is None else self.work.data_ptr() + 2 * a,
                                               self.grad.data_ptr() + 4 * a, self.exp_avg.data_ptr() + 4 * a, self.exp_avg_sq.data_ptr() + 4 * a,
                                               b - a, self.betas[0], self.betas[1], self.eps, self._state.data_ptr() + (0 if decays else 16),
                                               float(grad_scale), 1 if zero_grad else 0, stream), "moka_adamw_flat_dev")

    def state_dict(self) -> dict:
        return {"step": self.t, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lr": self.lr, "betas": self.betas,
                "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd: dict) -> None:
        self.t = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for k in ("lr", "eps", "weight_decay"):
            if k in sd:
                setattr(self, k, float(sd[k]))
        if "betas" in sd:
            self.betas = (float(sd["betas"][0]), float(sd["betas"][1]))
        if self._state is not None:
            self.set_device_step(self.t)             # (a captured begin_step(device_counter=True) resumes the bias correction at step t + 1, not at 1)


_LAYER_RE = re.compile(r"^(.*?(?:^|\.)layers)\.(\d+)\.")


def _in_backward(dp=None) -> bool:
    """True while an autograd backward pass is running on this thread (activation checkpointing re-runs layer forwards there).
    ``torch._C._current_graph_task_id`` is a private hook: where it is missing, the handle's own flag decides (set by the first
    backward callback of a pass, cleared by an engine callback when that pass ends: ``_mark_pass``)."""
    f = getattr(torch._C, "_current_graph_task_id", None)
    if f is not None:
        return f() != -1
    return bool(dp is not None and dp._pass_running)


def _mark_pass(dp) -> None:
    """Called from inside a backward pass (a deferred-launch node, a layer hook): ``dp._pass_running`` is True until THIS pass ends -- the
    engine runs the queued callback when the graph task completes.  (``dp._bwd_active`` stays set until finish() / step(): it says that a
    pass has reported work, not that one is running.)"""
    dp._bwd_active = True
    if not dp._pass_running:
        dp._pass_running = True

        def _ended():
            dp._pass_running = False
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_ended)
        except RuntimeError:                         # (not inside a backward pass: a direct call from a test or a manual launch)
            dp._pass_running = False


def _decoder_prefix(names: Sequence[str]) -> Optional[str]:
    """The ``<prefix>.layers`` module list that owns most of the adapter parameters: the decoder stack.  Vision / audio
    encoders have ``...encoder.layers.N`` modules with the SAME indices; their parameters must not land in the decoder
    layer's gradient group (that bucket ships when the DECODER layer has finished its backward)."""
    count = {}
    for n in names:
        m = _LAYER_RE.search(n)
        if m:
            count[m.group(1)] = count.get(m.group(1), 0) + 1
    return max(count, key=lambda k: (count[k], -len(k))) if count else None


class AdapterDataParallel:
    """What ``attach`` returns: the flat buffers of a model's trainable parameters, the bucketed gradient all-reduce hooked to
    the backward of its decoder layers, and the fused optimizer step.

        dp = moka_amd.parallel.attach(model)        # after get_peft_model / PeftMixedModel + set_adapter
        for batch in loader:
            loss = model(**batch).loss
            loss.backward()                         # weight-gradient kernels add into dp.bucket.flat; a finished bucket of
            dp.step()                               #   layers is all-reduced (RCCL, side stream) while the backward goes on

    Gradient accumulation: ``with dp.no_sync(): ...backward of every micro-batch but the last...`` (as DDP's ``no_sync``);
    a second backward with ``sync`` on and buckets already shipped raises instead of racing the collective.
    Under HF ``Trainer``: ``MokaFlatOptimizer(dp)`` + ``MokaDPCallback(dp)`` (below)."""

    def __init__(self, model: nn.Module, bucket: FlatGradBucket, master: torch.Tensor, work: torch.Tensor, names: List[str],
                 offsets: List[int], sizes: List[int], optimizer: Optional[FlatAdamW], handles: list):
        self.model, self.bucket, self.master, self.work = model, bucket, master, work
        self.names, self.offsets, self.sizes, self.optimizer, self._handles = names, offsets, sizes, optimizer, handles
        self.kernel_fed: List[str] = []              # parameters whose gradients the weight-gradient kernels write (sinks)
        self.hooked: List[str] = []                  # the other trainable parameters: autograd gradients folded in by a hook
        self._done = set()
        self.sync = True                             # False while accumulating micro-batches: the hooks ship nothing (cf. DDP.no_sync)
        # deferred dA_m launches (attach(defer_dA=True)): closures + the tensors they read, flushed at the end of a decoder layer's
        # backward onto a side stream, where they run beside the NEXT layer's dependency chain (whose rank-space kernels leave most
        # of the chip idle); joined before a bucket is shipped and before the optimizer step
        self._deferred: list = []
        self._side = None
        self._side_busy = False
        # attach(optimizer_in_backward=True): a finished bucket's AdamW slice is enqueued while the backward of the earlier layers is still
        # running (one GPU: on the side stream behind the bucket's deferred dA_m; N > 1: on the communication stream behind its all-reduce)
        self.opt_in_backward = False
        self._opt_begun = False
        self._opt_done: List = []                    # [lo, hi) ranges of the flat buffers already updated in this step
        self._bwd_active = False                     # a backward pass has reported work since the last finish() / step()
        self.unclipped_steps = 0                     # step(max_grad_norm=...) calls that could not clip (optimizer_in_backward)
        self._pass_running = False                   # ... and is still running (``_mark_pass``: cleared by an engine callback at its end)
        # persistent weight shadows (attach(persistent_shadows=True)): (group index, module, lora_B, [lora_A_m], BwT, AT) per adapted projection
        self._shadowed: list = []
        self._group_bounds: list = []                # [lo, hi) of every parameter group (decoder layer) in the flat buffers
        self._graph = None                           # schedule.GraphedTrainStep while it captures: hub stream, chain count, per-chain reports
        # the device-resident part of every adapted projection's dropout seed (moka_opts.seed_dev): 0 in live training -- the seed drawn
        # per call is the whole seed -- rewritten before every replay of a captured step, whose launch arguments are frozen
        self.seed_epoch = torch.zeros(1, dtype=torch.int64, device=bucket.flat.device) if bucket.flat.is_cuda else None

    # ---------------------------------------------------------------- backward side
    def _opt_slice(self, lo: int, hi: int) -> None:
        """AdamW on parameters [lo, hi) on the CURRENT stream (coefficients of the step uploaded on that stream the first time), then the
        weight shadows of the projections whose parameters just changed."""
        if not self._opt_begun:
            self.optimizer.begin_step()              # (hyper-parameters pulled now: MokaFlatOptimizer's param_groups, a scheduler's lr)
            self._opt_begun = True
        self.optimizer.step_range(lo, hi, grad_scale=1.0 / self.bucket.world, zero_grad=True)
        self._opt_done.append((lo, hi))
        self.refresh_shadows(lo, hi)

    def refresh_shadows(self, lo: Optional[int] = None, hi: Optional[int] = None) -> None:
        """Rewrite the persistent weight shadows (BwT, AT) of the adapted projections whose parameters lie in [lo, hi) of the flat buffers
        (default: all) from the current working copies, on the current stream: one ``moka_weight_shadows_batch`` launch per 16 projections."""
        if not self._shadowed:
            return
        from . import functional as F
        pick = [it for it in self._shadowed
                if lo is None or (self._group_bounds[it[0]][0] >= lo and self._group_bounds[it[0]][1] <= hi)]
        by_shape = {}
        for it in pick:
            by_shape.setdefault((it[2].shape[1], len(it[3])), []).append(it)
        with torch.no_grad():
            for (r, _m), items in by_shape.items():
                F.weight_shadows_batch_([it[2].detach() for it in items], [[a.detach() for a in it[3]] for it in items], int(r),
                                        [it[4] for it in items], [it[5] for it in items])

    def _defer(self, fn, tensors, da=None, db=None) -> None:
        """fn: the launch as a closure; da / db = (key, items): the same work described as problems of moka_down_bwd_da_batch (key =
        (routing, r, dropout_p), items = [(dh_kmj, x2, [dA_acc_m], seed)]) or moka_up_bwd_db_batch (key = (routing, r), items =
        [(gy2, hp_kmj, dB_acc)]), so that a decoder layer's weight-gradient launches leave as ONE per kind."""
        _mark_pass(self)
        desc = ("da", da) if da is not None else (("db", db) if db is not None else None)
        self._deferred.append((fn, [t for t in tensors if isinstance(t, torch.Tensor)], desc))
    _defer.accepts_da = True

    def _flush_deferred(self) -> None:
        """Launch what the backward has deferred so far on the side stream, behind everything the main stream has enqueued.  The dA_m
        (and, where dB is a pass of its own, dB) halves that came with a description and share routing, rank and dropout rate go out as one
        batched launch per MOKA_MAX_BATCH projections (a decoder layer of 7 projections: one launch instead of four); everything else as
        it was handed in."""
        if not self._deferred:
            return
        from . import functional as F
        from . import _lib
        dev = self.bucket.flat.device
        if self._graph is not None:
            # a step being captured (schedule.GraphedTrainStep): everything off the chains goes to the capture's ORIGIN stream (the hub) --
            # a dependency between two forked streams crashes hipStreamEndCapture, and nothing on a chain ever waits for the hub
            side = self._graph.hub
        else:
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            side = self._side
        main = torch.cuda.current_stream(dev)
        if side != main:                             # (a one-chain capture is a single list: "off the chain" is then just "later on the same stream")
            side.wait_stream(main)

        def key_of(desc):
            return (desc[0], id(desc[1][0][0])) + tuple(v if not isinstance(v, torch.Tensor) else id(v) for v in desc[1][0][1:])

        groups = {}
        for fn, tensors, desc in self._deferred:
            if desc is not None:
                groups.setdefault(key_of(desc), []).append(desc[1])
        batched = {k for k, v in groups.items() if sum(len(d[1]) for d in v) > 1}
        with torch.cuda.stream(side):
            for fn, tensors, desc in self._deferred:
                if desc is None or key_of(desc) not in batched:
                    fn()
                for t in tensors:
                    t.record_stream(side)            # (the caching allocator must not hand the block out before the side kernel has read it)
            for k in batched:
                items = [it for d in groups[k] for it in d[1]]
                head = groups[k][0][0]
                for i in range(0, len(items), _lib.MOKA_MAX_BATCH):
                    part = items[i:i + _lib.MOKA_MAX_BATCH]
                    if k[0] == "da":
                        rt, r, p, sdev = head
                        F.down_bwd_da_batch_([it[0] for it in part], [it[1] for it in part], rt, r, [it[2] for it in part], p, [it[3] for it in part], seed_dev=sdev)
                    else:
                        rt, r = head
                        F.up_bwd_db_batch_([it[0] for it in part], [it[1] for it in part], rt, r, [it[2] for it in part])
        self._deferred.clear()
        self._side_busy = self._graph is None

    def _join_deferred(self) -> None:
        self._flush_deferred()
        if self._side_busy and self._graph is None:
            torch.cuda.current_stream(self.bucket.flat.device).wait_stream(self._side)
            self._side_busy = False

    def _layer_done(self, l: int) -> None:
        _mark_pass(self)
        self._flush_deferred()                       # the layer's dA_m launches leave for the side stream now
        if self._graph is not None:
            self._graph.layer_done(self, l)          # (capture: the bucket's AdamW slice goes to the hub once EVERY chain has reported the layer)
            return
        if not self.sync or l in self._done:
            return
        self._done.add(l)
        if self.bucket.comm and self.bucket.is_bucket_first(l):
            self._join_deferred()                    # a bucket must not ship before its dA_m have landed
        self.bucket.layer_done(l)                    # (collectives on, opt_in_backward: bucket.on_reduced runs the bucket's AdamW slice behind its all-reduce)
        if self.opt_in_backward and not self.bucket.comm and self.bucket.is_bucket_first(l):
            dev = self.bucket.flat.device
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            self._side.wait_stream(torch.cuda.current_stream(dev))      # the bucket's in-chain gradients (dB at r <= 32, dx-side sinks) are on the main stream
            with torch.cuda.stream(self._side):
                self._opt_slice(*self.bucket.bucket_bounds(l))
            self._side_busy = True

    def _forward_begins(self) -> None:
        """Called from the decoder layers' forward pre-hooks.  A NEW forward while buckets of the previous backward are on their
        way (or already summed) and no step() / finish() in between means gradient accumulation without ``no_sync``: the next
        backward's kernels would add into memory an in-place all-reduce is working on, and those layers would not be shipped
        again.  DDP is merely slower in that situation; here it would be silently wrong, so it is an error."""
        if self._graph is None and self.sync and (self._done or self.bucket._pending) and torch.is_grad_enabled() and not _in_backward(self):
            raise RuntimeError("moka_amd.parallel: a new forward started while gradient buckets of the previous backward are in flight. "
                               "Accumulate micro-batches under `with dp.no_sync():` (sync only on the last one), or call dp.step() / "
                               "dp.finish() after every backward.")

    def no_sync(self):
        """Context manager: the backward hooks ship nothing (gradient accumulation); everything is shipped by the next
        synchronised backward or by ``finish()``."""
        dp = self

        class _NoSync:
            def __enter__(self_inner):
                self_inner.prev, dp.sync = dp.sync, False

            def __exit__(self_inner, *exc):
                dp.sync = self_inner.prev
                return False
        return _NoSync()

    def finish(self, average: bool = True) -> None:
        """Join the all-reduces (every bucket whose hook did not fire -- frozen layers, the parameters outside the decoder
        stack, models without `.layers.N.` modules -- is shipped now).  With ``average`` the flat buffer then holds the mean
        gradient over the ranks."""
        self._join_deferred()
        for l in range(self.bucket.n_layers - 1, -1, -1):
            if l not in self._done and self.bucket.is_bucket_first(l):
                self.bucket.layer_done(l)
        self._done.clear()
        self._bwd_active = False
        self.bucket.finish(average=average)

    def step(self, max_grad_norm: Optional[float] = None) -> Optional[torch.Tensor]:
        """finish() + the fused AdamW kernel (averaging folded into it, gradients left zeroed, working copies refreshed).
        max_grad_norm: clip the l2 norm of the AVERAGED gradient (HF ``max_grad_norm`` / DeepSpeed ``gradient_clipping``,
        ``VisualText/zero_stage2_config.json:36``); the coefficient is folded into the kernel's gradient scale (one host read
        of the norm).  Returns the norm of the averaged gradient when clipping is on."""
        if self.optimizer is None:
            raise RuntimeError("attach(..., optimizer=False): call finish() and run your own optimizer on dp.master / dp.bucket.flat")
        if self.opt_in_backward:
            self.finish(average=False)               # ships what the hooks did not; joins the side / communication streams (and their slices)
            done, pos = sorted(self._opt_done), 0
            for lo, hi in done + [(self.bucket.flat.numel(), self.bucket.flat.numel())]:
                if lo > pos:
                    self._opt_slice(pos, lo)         # whatever no bucket hook covered (parameters outside the decoder stack, frozen layers' groups)
                pos = max(pos, hi)
            self._opt_done.clear()
            self._opt_begun = False
            if max_grad_norm is not None and max_grad_norm > 0:
                # (the buckets of this step were updated inside the backward, before a global norm existed: nothing to refuse any more at this
                #  point, and a step that mutates the parameters and then throws leaves the caller with neither.  MokaFlatOptimizer refuses
                #  the combination up front; a direct caller is told on EVERY call -- a single warning is lost in a training log -- and the
                #  count travels with state_dict())
                import warnings
                self.unclipped_steps += 1
                warnings.warn("attach(optimizer_in_backward=True) updates a bucket before the global gradient norm exists: step(max_grad_norm=%g) "
                              "does NOT clip in this mode (%d step(s) so far)" % (max_grad_norm, self.unclipped_steps), RuntimeWarning, stacklevel=2)
            return None
        self.finish(average=False)
        scale = 1.0 / self.bucket.world
        norm = None
        if max_grad_norm is not None and max_grad_norm > 0:
            norm = self.bucket.flat.norm() * scale
            scale *= min(1.0, float(max_grad_norm) / (float(norm) + 1e-6))
        self.optimizer.step(grad_scale=scale, zero_grad=True)
        self.refresh_shadows()
        return norm

    def grad_norm(self) -> torch.Tensor:
        """l2 norm of the flat gradient AS IT STANDS: the local gradient before finish(), the sum over ranks after
        ``finish(average=False)``, the mean after ``finish()``.  (``step(max_grad_norm=...)`` clips the mean.)"""
        return self.bucket.flat.norm()

    # ---------------------------------------------------------------- parameters / state
    def parameters(self) -> List[torch.nn.Parameter]:
        by = dict(self.model.named_parameters())
        return [by[n] for n in self.names]

    def state_dict(self) -> dict:
        """Everything a resume needs beyond the module's own state_dict: the fp32 master copy (the module parameters are its
        bf16 rounding) and the AdamW moments.  Tensors are references; ``torch.save`` them or ``clone()`` first."""
        return {"names": list(self.names), "offsets": list(self.offsets), "sizes": list(self.sizes), "master": self.master,
                "optimizer": None if self.optimizer is None else self.optimizer.state_dict(),
                "optimizer_in_backward": bool(self.opt_in_backward), "unclipped_steps": int(self.unclipped_steps)}

    def load_state_dict(self, sd: dict) -> None:
        if list(sd["names"]) != self.names or list(sd["offsets"]) != self.offsets:
            raise ValueError("moka_amd.parallel: the checkpoint was taken from a different parameter layout")
        self.master.copy_(sd["master"])
        with torch.no_grad():
            self.work.copy_(self.master)             # bf16 working copies (fp32 parameters are views of the master itself)
        if self.optimizer is not None and sd.get("optimizer") is not None:
            self.optimizer.load_state_dict(sd["optimizer"])
        self.refresh_shadows()

    def detached_state_dict(self, state_dict: Optional[dict] = None) -> dict:
        """``model.state_dict()`` (or the given one) with every tensor that is a view of the flat buffers cloned into its own
        storage: what ``safetensors.torch.save_model`` / ``save_file`` want (they refuse tensors that share one storage)."""
        sd = self.model.state_dict() if state_dict is None else state_dict
        shared = {self.work.untyped_storage().data_ptr(), self.master.untyped_storage().data_ptr()}
        return {k: (v.detach().clone() if isinstance(v, torch.Tensor) and v.untyped_storage().data_ptr() in shared else v)
                for k, v in sd.items()}

    def detach(self) -> None:
        for h in self._handles:
            h.remove()
        for m in self.model.modules():
            if getattr(m, "_moka_sinks", None) is not None:
                m._moka_sinks = None
                m._moka_defer = None
                m._moka_shadows = None
                m._moka_seed_dev = None


def attach(model: nn.Module, n_buckets: int = 8, process_group=None, optimizer: bool = True, lr: float = 1e-4,
           betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
           comm_dtype: Optional[torch.dtype] = None, trainable=None, defer_dA: bool = True,
           optimizer_in_backward: bool = False, force_comm: bool = False, no_decay="hf",
           tail_layers: Optional[int] = None, bucket_sizes=None, persistent_shadows: Optional[bool] = None) -> AdapterDataParallel:
    """Data-parallel training of a MokA-adapted model: one process per GPU, every rank holds the full frozen
    base and the full adapter, batches are sharded by sample, and the only exchange is the trainable-gradient sum.

    * EVERY trainable parameter of ``model`` (``trainable(name, param)``, default ``param.requires_grad``) is re-seated, in
      module order, into flat buffers: an fp32 master copy, ONE flat fp32 gradient buffer, and a bf16 working copy that bf16
      parameters become views of (fp32 parameters -- the reference's adapters follow the base dtype, ``layer.py:124-132`` --
      become views of the master itself).  Both reference scripts train more than the adapter: the Q-Former projectors
      (``AudioVisualText/scripts/finetune/finetune.py:151-160``, ``VisualText/train/train.py:573-579``); they ride in the same
      buffer, the same all-reduce and the same optimizer step;
    * the adapter parameters of the decoder stack are grouped by decoder layer (the ``<prefix>.layers.<i>.`` module list
      that owns most of them; an encoder's ``layers.<i>`` do not count); the adapted projections get views of the gradient
      buffer as *sinks*: their weight-gradient kernels accumulate straight into it (fp32, across micro-batches too) and
      autograd carries no adapter gradients at all;
    * every other trainable parameter gets a post-accumulate-grad hook that adds its autograd gradient into its slice of the
      flat buffer and drops ``.grad``; these sit in a group of their own that ``finish()`` ships (their gradients are the
      last ones a backward produces);
    * a gradient hook on every decoder layer's input reports the layer as finished; ``FlatGradBucket`` starts the RCCL
      all-reduce of a finished bucket of layers on a side stream while the remaining layers' backward runs;
    * ``defer_dA`` (default): only the optimizer needs dA_m, so that half of every ``moka_down_bwd`` leaves the backward's dependency
      chain: the adapted projections hand it to ``dp``, which launches a decoder layer's worth of them on a side stream when the
      layer's backward has been enqueued -- they run beside the next layer's chain -- and joins before a bucket ships / before the
      optimizer step (adapter-only step of the 7B workload: 35.1 -> 34.4 ms);
    * ``step()`` joins, and one kernel (``moka_adamw_flat``) averages (and clips), applies AdamW, refreshes the working
      copies and zeroes the gradients;
    * ``optimizer_in_backward``: the AdamW update of a finished bucket of layers is enqueued at once -- on the side stream behind the
      bucket's deferred dA_m (one GPU) or on the communication stream behind its all-reduce (N > 1, fp32 payload) -- and overlaps the
      backward of the earlier layers; ``step()`` then only updates what no bucket covered.  No gradient clipping in this mode (the
      global norm does not exist yet when the first bucket is updated), and every synchronised backward must be followed by ``step()``.

    * ``tail_layers`` / ``bucket_sizes``: the layout of the gradient buckets (``FlatGradBucket``): a small bucket for the layers whose
      backward runs last, or the whole layout (``"geometric"``: 1, 3, 9, ... layers from the first one up);
    * ``force_comm``: run the bucket all-reduces even in a process group of one rank (``FlatGradBucket``): the communication path
      -- RCCL's own stream, the bucket hooks, the optimizer slices behind the all-reduce -- as N > 1 ranks run it, on one GPU;
    * ``no_decay``: which parameters take no weight decay: ``"hf"`` (default) = what HF ``Trainer``'s default optimizer excludes and
      the reference therefore trains without decay (``AudioVisualText/trainer.py`` does not override ``create_optimizer``): biases and
      the weights of normalisation layers; a callable ``(name, param, module) -> bool``; ``None`` = decay everywhere.

    * ``persistent_shadows`` (default: on with the built-in optimizer): the transposed weight copies the backward kernels read (``BwT``,
      ``AT``: functions of the weights alone) are kept per adapted projection and rewritten where the weights change -- one batched
      launch (``moka_weight_shadows_batch``) behind every optimizer update, per gradient bucket with ``optimizer_in_backward`` -- instead
      of one launch per projection in every forward (7 x n_layers launches off the forward's dependency chain).  Whoever writes the
      adapter weights by other means (``load_state_dict`` on the module, a manual ``copy_``) calls ``dp.refresh_shadows()`` afterwards;
      ``dp.load_state_dict`` does.

    Replaces DeepSpeed ZeRO-2's bucketed reduce-scatter + partitioned optimizer of the reference configurations
    (``VisualText/zero_stage2_config.json:2-10``, ``AudioVisualText/trainer.py:163-218``)."""
    pick = trainable if trainable is not None else (lambda n, p: p.requires_grad)
    named = [(n, p) for n, p in model.named_parameters() if pick(n, p)]
    if not named:
        raise ValueError("attach: the model has no trainable parameters (call get_peft_model / set_adapter first)")
    dev = named[0][1].device
    if any(p.device != dev for _, p in named):
        raise ValueError("attach: the trainable parameters must live on one device (one process per GPU)")
    for n, p in named:
        if p.dtype not in (torch.bfloat16, torch.float32):
            raise TypeError(f"attach: {n} is {p.dtype}; trainable parameters must be bf16 or fp32")
    # which parameters do the kernels feed?  (the adapter matrices of the two mirrors' adapted projections)
    sink_of = {}                                                                  # parameter name -> (module, "B" | index into "A")
    for mod_name, mod in model.named_modules():
        pre = mod_name + "." if mod_name else ""
        if hasattr(mod, "lora_B0") and hasattr(mod, "_plan"):                    # AVT mirror: lora_A0.., lora_B0
            sink_of[f"{pre}lora_B0.weight"] = (mod, "B")
            for i in range(getattr(mod, "lora_num", 0)):
                sink_of[f"{pre}lora_A{i}.weight"] = (mod, i)
        elif hasattr(mod, "lora_A") and hasattr(mod, "_plan") and "text" in getattr(mod, "lora_A", {}):   # VT mirror
            sink_of[f"{pre}lora_B.text.weight"] = (mod, "B")
            sink_of[f"{pre}lora_A.text.weight"] = (mod, 0)
            sink_of[f"{pre}lora_A.image.weight"] = (mod, 1)
    picked = {n for n, _ in named}
    # a projection is kernel-fed only if ALL its adapter matrices are trainable (otherwise autograd + hooks handle what is)
    by_mod = {}
    for n, (mod, slot) in sink_of.items():
        by_mod.setdefault(id(mod), []).append(n)
    fed = set()
    for ns in by_mod.values():
        if all(n in picked for n in ns):
            fed.update(ns)
    prefix = _decoder_prefix([n for n, _ in named if n in fed])
    layer_of = []
    for n, _ in named:
        m = _LAYER_RE.search(n)
        layer_of.append(int(m.group(2)) if (m and n in fed and m.group(1) == prefix) else -1)
    ids = sorted(set(layer_of))
    order = sorted(range(len(named)), key=lambda k: (layer_of[k], k))            # layer by layer, module order inside a layer
    names, offsets, sizes, ends = [], [], [], []
    named_sorted, layer_sorted = [named[k][0] for k in order], [layer_of[k] for k in order]
    off, cur = 0, None
    for k in order:
        if cur is not None and layer_of[k] != cur:
            ends.append(off)
        cur = layer_of[k]
        names.append(named[k][0])
        offsets.append(off)
        sizes.append(named[k][1].numel())
        off += (named[k][1].numel() + 7) // 8 * 8                               # 16-byte aligned bf16 / 32-byte fp32 views
    ends.append(off)
    if bucket_sizes == "geometric":                  # 1, 3, 9, ... groups of layers from the first one up (geometric_buckets)
        bucket_sizes = geometric_buckets(len(ends))
    bucket = FlatGradBucket(off, ends, dev, n_buckets=n_buckets, process_group=process_group, comm_dtype=comm_dtype, force_comm=force_comm,
                            tail_layers=tail_layers, bucket_sizes=bucket_sizes)
    master = torch.zeros(off, dtype=torch.float32, device=dev)
    work = torch.zeros(off, dtype=torch.bfloat16, device=dev)
    by_name = dict(named)
    grad_view = {}
    with torch.no_grad():
        for n, o, sz in zip(names, offsets, sizes):
            p = by_name[n]
            master[o:o + sz].copy_(p.detach().reshape(-1))
            work[o:o + sz].copy_(p.detach().reshape(-1))
            # the module parameter IS the working copy (bf16) / the master (fp32 storage)
            p.data = (work if p.dtype == torch.bfloat16 else master)[o:o + sz].view(p.shape)
            grad_view[n] = bucket.flat[o:o + sz].view(p.shape)
    opt = FlatAdamW(master, bucket.flat, work, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay) if optimizer else None
    if opt is not None and no_decay is not None:
        owner = {}
        for mod_name, mod in model.named_modules():
            for pn, _p in mod.named_parameters(recurse=False):
                owner[(mod_name + "." if mod_name else "") + pn] = mod
        if no_decay == "hf":
            # transformers.Trainer.get_decay_parameter_names: every parameter of a normalisation layer (isinstance against ALL_LAYERNORM_LAYERS) and
            # every parameter whose name contains "bias" takes no decay; modules the table does not know (an RMSNorm of one's own) count by class name
            try:
                from transformers.pytorch_utils import ALL_LAYERNORM_LAYERS as _NORMS
                _norms = tuple(_NORMS)
            except Exception:
                _norms = (nn.LayerNorm,)

            def no_decay(n, p, mod):
                return "bias" in n or isinstance(mod, _norms) or "norm" in type(mod).__name__.lower()
        ranges = []
        for n, o, sz in zip(names, offsets, sizes):
            if no_decay(n, by_name[n], owner.get(n)):
                hi_ = o + (sz + 7) // 8 * 8          # (the padding behind a parameter belongs to it: its values are never read)
                if ranges and ranges[-1][1] == o:
                    ranges[-1] = (ranges[-1][0], hi_)
                else:
                    ranges.append((o, hi_))
        opt.no_decay_ranges = ranges
    dp = AdapterDataParallel(model, bucket, master, work, names, offsets, sizes, opt, [])
    if optimizer_in_backward:
        if opt is None or dev.type != "cuda":
            raise ValueError("attach(optimizer_in_backward=True) needs the built-in optimizer and a GPU")
        # (a bf16 payload is widened back into the fp32 buffer on the communication stream before the slice runs)
        dp.opt_in_backward = True
        if bucket.comm:
            bucket.on_reduced = dp._opt_slice
    # sinks of the adapted projections (both mirrors)
    mods = {}
    for n in fed:
        mod, slot = sink_of[n]
        sk = mods.setdefault(id(mod), (mod, {"B": None, "A": {}}))[1]
        if slot == "B":
            sk["B"] = grad_view[n]
        else:
            sk["A"][slot] = grad_view[n]
    group_of = {n: ids.index(l_) for n, l_ in zip(named_sorted, layer_sorted)}      # parameter name -> its group (decoder layer) in the flat buffers
    if persistent_shadows is None:
        persistent_shadows = bool(optimizer) and dev.type == "cuda"
    for mod, sk in mods.values():
        mod._moka_sinks = {"B": sk["B"], "A": [sk["A"][i] for i in sorted(sk["A"])]}
        if defer_dA and dev.type == "cuda":
            mod._moka_defer = dp._defer
        if dev.type == "cuda":
            mod._moka_seed_dev = dp.seed_epoch
        if persistent_shadows and dev.type == "cuda":
            pn = {slot: n for n, (m_, slot) in sink_of.items() if m_ is mod}
            Bw = by_name[pn["B"]]
            As = [by_name[pn[i]] for i in sorted(k for k in pn if k != "B")]
            if Bw.dtype == torch.bfloat16 and all(a.dtype == torch.bfloat16 for a in As):
                from . import _lib as _L
                RP = _L.rank_pad(Bw.shape[1])
                BwT = torch.empty((RP, Bw.shape[0]), dtype=torch.bfloat16, device=dev)
                AT = torch.empty((len(As), As[0].shape[1], RP), dtype=torch.bfloat16, device=dev)
                mod._moka_shadows = (BwT, AT)
                dp._shadowed.append((group_of[pn["B"]], mod, Bw, As, BwT, AT))
    dp._group_bounds = [(ends[i - 1] if i else 0, ends[i]) for i in range(len(ends))]
    dp.refresh_shadows()
    dp.kernel_fed = [n for n in names if n in fed]
    dp.hooked = [n for n in names if n not in fed]
    # every other trainable parameter: fold the autograd gradient into the flat buffer the moment it has been accumulated
    def fold(view):
        def hook(p):
            if p.grad is None:
                return
            if dp._graph is not None:
                # (capture: the chains run side by side and would race on the read-modify-write of the shared slice: the fold goes to the hub)
                hub, g = dp._graph.hub, p.grad
                if hub != torch.cuda.current_stream(view.device):
                    hub.wait_stream(torch.cuda.current_stream(view.device))
                with torch.cuda.stream(hub):
                    view.add_(g.to(view.dtype).view_as(view))
                g.record_stream(hub)
            else:
                view.add_(p.grad.to(view.dtype).view_as(view))
            p.grad = None
        return hook
    for n in dp.hooked:
        dp._handles.append(by_name[n].register_post_accumulate_grad_hook(fold(grad_view[n])))
    # backward hooks: bucket index l = position of the layer id among the groups that own parameters
    pos = {lid: i for i, lid in enumerate(ids)}
    # "layer l has finished its backward" = the gradient w.r.t. the layer's INPUT has been formed (every adapter node of the
    # layer runs before that: q/k/v feed it, o / gate / up / down sit nearer the loss).  A tensor hook on the input says exactly
    # that; a module full-backward hook does not (it fires at the START of the layer's backward when the input needs no
    # gradient -- the first layer).  Layers whose input carries no gradient are shipped by finish().
    def pre_hook(l):
        def hook(_mod, args, kwargs=None):
            dp._forward_begins()
            x = args[0] if args else None
            if torch.is_grad_enabled() and isinstance(x, torch.Tensor) and x.requires_grad:
                x.register_hook(lambda _g, l=l: dp._layer_done(l))
        return hook
    if prefix is not None:
        for mod_name, mod in model.named_modules():
            if mod_name.startswith(prefix + ".") and mod_name[len(prefix) + 1:].isdigit() and int(mod_name[len(prefix) + 1:]) in pos:
                dp._handles.append(mod.register_forward_pre_hook(pre_hook(pos[int(mod_name[len(prefix) + 1:])])))
    return dp


class MokaFlatOptimizer(torch.optim.Optimizer):
    """``torch.optim.Optimizer``-shaped handle on ``attach``'s flat buffers, so that training loops written around an optimizer
    object -- HF ``Trainer`` (``AudioVisualText/trainer.py:163-218``, ``VisualText/train/train.py:601-617``), LR schedulers --
    drive the data-parallel step:  ``Trainer(model=..., optimizers=(MokaFlatOptimizer(dp), scheduler), callbacks=[MokaDPCallback(dp)])``.

    * ``step()`` = ``dp.step(max_grad_norm)``: join the gradient all-reduce, one fused kernel for averaging / clipping /
      AdamW / working copies / gradient zeroing.  The hyper-parameters are read from ``param_groups[0]`` at every step, so
      schedulers that rewrite ``group["lr"]`` work unchanged;
    * ``zero_grad()`` is a no-op (the kernel leaves the flat gradient zeroed; ``param.grad`` is never populated -- set the
      trainer's own ``max_grad_norm`` to 0 and pass the clip value here);
    * ``state_dict()`` carries the fp32 master copy and the AdamW moments (``dp.state_dict()``)."""

    def __init__(self, dp: AdapterDataParallel, lr: Optional[float] = None, betas=None, eps: Optional[float] = None,
                 weight_decay: Optional[float] = None, max_grad_norm: Optional[float] = None):
        if dp.optimizer is None:
            raise ValueError("MokaFlatOptimizer needs attach(..., optimizer=True)")
        o = dp.optimizer
        defaults = dict(lr=o.lr if lr is None else float(lr), betas=tuple(o.betas if betas is None else betas),
                        eps=o.eps if eps is None else float(eps), weight_decay=o.weight_decay if weight_decay is None else float(weight_decay))
        if dp.opt_in_backward and max_grad_norm is not None and max_grad_norm > 0:
            raise ValueError("MokaFlatOptimizer(max_grad_norm=...) on attach(optimizer_in_backward=True): a bucket is updated inside the "
                             "backward, before the global gradient norm exists -- no clipping in this mode")
        super().__init__([{"params": dp.parameters()}], defaults)
        self.dp, self.max_grad_norm = dp, max_grad_norm
        self.last_grad_norm = None
        # the fused step reads its hyper-parameters from param_groups[0] WHEN IT BEGINS: with the optimizer inside the backward that
        # is the first finished bucket of the backward, long before step() is called -- a scheduler's lr for this step is in
        # param_groups by then (schedulers step after optimizer.step()), the value attach() was given is not
        o.hyper = lambda: self.param_groups[0]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.last_grad_norm = self.dp.step(max_grad_norm=self.max_grad_norm)     # (hyper-parameters: FlatAdamW.hyper, pulled when the step begins)
        return loss

    def zero_grad(self, set_to_none: bool = True) -> None:
        return None

    def state_dict(self) -> dict:
        sd = super().state_dict()
        sd["moka_flat"] = self.dp.state_dict()
        return sd

    def load_state_dict(self, state_dict: dict) -> None:
        sd = dict(state_dict)
        flat = sd.pop("moka_flat", None)
        super().load_state_dict(sd)
        if flat is not None:
            self.dp.load_state_dict(flat)


def keep_out_of_ddp(trainer, dp: AdapterDataParallel) -> None:
    """Under ``torchrun`` HF ``Trainer`` hands the model to ``accelerator.prepare``, which wraps it in ``DistributedDataParallel``; DDP's
    reducer then waits for autograd gradients of the trainable parameters -- gradients that never come, because the kernels write
    them into ``dp``'s flat buffer and ``dp`` all-reduces that itself.  This makes ``accelerator.prepare_model`` return ``dp.model``
    unwrapped (everything else -- device placement of the batches, the optimizer wrapper, gradient accumulation -- is untouched).
    A no-op in a single process."""
    acc = trainer.accelerator
    orig = acc.prepare_model

    def prepare_model(model, *a, **kw):
        if model is dp.model or getattr(model, "module", None) is dp.model:
            return model
        return orig(model, *a, **kw)
    acc.prepare_model = prepare_model


def trainer_callback(dp: AdapterDataParallel):
    """``transformers.TrainerCallback`` that keeps ``dp.sync`` off for every micro-batch of a gradient-accumulation window but
    the last (what ``accelerator.no_sync`` does for a DDP model), so that the bucketed all-reduce overlaps the LAST backward
    and nothing is shipped twice.  Not needed for correctness -- with ``dp.sync = False`` throughout, ``step()`` ships
    everything -- only for the overlap."""
    from transformers import TrainerCallback

    class MokaDPCallback(TrainerCallback):
        def __init__(self):
            self.k = 0

        def on_train_begin(self, args, state, control, **kw):
            dp.sync = args.gradient_accumulation_steps == 1

        def on_step_begin(self, args, state, control, **kw):
            self.k = 0
            dp.sync = args.gradient_accumulation_steps == 1

        def on_substep_end(self, args, state, control, **kw):
            self.k += 1
            dp.sync = self.k == args.gradient_accumulation_steps - 1

    return MokaDPCallback()


def bind_param_grads(params: Sequence[torch.nn.Parameter], bucket: FlatGradBucket, offsets: Sequence[int]):
    """Make ``param.grad`` a view of the flat buffer (dtype must match) so optimizers / HF Trainer see
    ordinary gradients while the kernels and the collective work on the flat storage."""
    for p, off in zip(params, offsets):
        view = bucket.flat[off:off + p.numel()].view_as(p)
        if view.dtype != p.dtype:
            raise TypeError(f"flat bucket is {view.dtype} but parameter is {p.dtype}")
        p.grad = view


class ShardedFrozenBase:
    """Frozen base weights sharded 1/N per rank and gathered one decoder layer at a time.

    The reference runs its 70B configuration under DeepSpeed ZeRO-3 (``VisualText/zero_stage3_config_70b.json:2-13``:
    parameter partitioning with prefetch).  On MI355X the bf16 70B base (140 GB) fits one GPU's 288 GB, so replication is
    the default; this store is for when the memory is wanted for activations instead.  One flat buffer per layer (its
    frozen tensors back to back, padded to a multiple of the world size), each rank keeps its contiguous 1/N slice;
    ``prefetch(l)`` starts ``all_gather_into_tensor`` of layer l into one of two full-size staging buffers on a side
    stream (RCCL over xGMI; one large collective per layer, not one per tensor), ``layer(l)`` waits for it and returns
    the tensors as views of the staging buffer.  Two buffers: layer l + 1 (forward) or l - 1 (backward) streams in while
    layer l computes.  The adapter parameters are never sharded (they are small and live in ``FlatGradBucket``)."""

    def __init__(self, layers: Sequence[Sequence[tuple]], device, process_group=None, dtype=torch.bfloat16):
        """layers[l] = [(name, tensor), ...] full frozen tensors of layer l (consumed: only the local shard is kept)."""
        self.group = process_group
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if on else 1
        self.rank = dist.get_rank(process_group) if on else 0
        self.device = torch.device(device)
        self.meta: List[List[tuple]] = []          # per layer: (name, offset, shape)
        self.shards: List[torch.Tensor] = []
        self.padded: List[int] = []
        for tensors in layers:
            off, meta = 0, []
            for name, t in tensors:
                meta.append((name, off, tuple(t.shape)))
                off += t.numel()
            pad = (off + self.world - 1) // self.world * self.world
            flat = torch.zeros(pad, dtype=dtype, device=self.device)
            for (name, o, shape), (_, t) in zip(meta, tensors):
                flat[o:o + t.numel()].copy_(t.reshape(-1))
            n = pad // self.world
            self.shards.append(flat[self.rank * n:(self.rank + 1) * n].clone())
            self.meta.append(meta)
            self.padded.append(pad)
        cap = max(self.padded) if self.padded else 0
        self.stage = [torch.empty(cap, dtype=dtype, device=self.device) for _ in range(2)]
        self.holds = [-1, -1]                      # layer held by each staging buffer
        self.pending = [None, None]
        self.is_cuda = self.device.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=self.device) if (self.is_cuda and self.world > 1) else None

    def shard_bytes(self) -> int:
        return sum(s.numel() * s.element_size() for s in self.shards)

    def prefetch(self, l: int) -> None:
        """Start gathering layer l into staging buffer l % 2 (no-op if it is already there or on its way).  The buffer is
        a function of the layer alone: tensors handed out for layer l in the forward are valid again -- same storage --
        once layer l has been gathered again for its backward (autograd's saved views stay meaningful)."""
        if l < 0 or l >= len(self.shards):
            return
        slot = l & 1
        if self.holds[slot] == l:
            return
        if self.pending[slot] is not None:         # the buffer's previous gather must have landed before it is reused
            self._wait(slot)
        out = self.stage[slot][:self.padded[l]]
        self.holds[slot] = l
        if self.world == 1:
            out.copy_(self.shards[l])
            return
        if self.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))      # the consumer of the buffer's previous content is enqueued
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self.pending[slot] = dist.all_gather_into_tensor(out, self.shards[l], group=self.group, async_op=True)
        else:
            self.pending[slot] = dist.all_gather_into_tensor(out, self.shards[l], group=self.group, async_op=True)

    def _wait(self, slot: int) -> None:
        wk = self.pending[slot]
        if wk is None:
            return
        if self.is_cuda:
            done = torch.cuda.Event()
            with torch.cuda.stream(self.comm_stream):
                wk.wait()
                done.record(self.comm_stream)
            torch.cuda.current_stream(self.device).wait_event(done)
        else:
            wk.wait()
        self.pending[slot] = None

    def layer(self, l: int, prefetch_next: Optional[int] = None) -> dict:
        """Full tensors of layer l (views of staging buffer l % 2, valid until a layer of the same parity is requested)."""
        self.prefetch(l)
        slot = l & 1
        self._wait(slot)
        if prefetch_next is not None and (prefetch_next & 1) != slot:
            self.prefetch(prefetch_next)
        buf = self.stage[slot]
        return {name: buf[o:o + math.prod(shape)].view(shape) for name, o, shape in self.meta[l]}

    # ------------------------------------------------------------------------------------------------------------
    @classmethod
    def shard_stack(cls, stack: nn.Module, process_group=None) -> "ShardedFrozenBase":
        """ZeRO-3-style partitioning of the FROZEN parameters of a ``MokaLlamaStack`` (or any module with ``.layers``):
        every decoder layer's frozen tensors (everything without ``lora_`` in its name) move into the 1/N-sharded store;
        forward pre-hooks gather layer l (and prefetch l + 1) just before it runs and point the parameters at the staging
        buffer, backward pre-hooks do the same in reverse order (prefetching l - 1) so that the views autograd saved are
        backed by the right data again.  The adapter parameters stay whole on every rank (``attach`` owns them)."""
        layers = list(stack.layers)
        dev = next(stack.parameters()).device
        per_layer, owners = [], []
        for layer in layers:
            items, own = [], []
            for n, p in layer.named_parameters():
                if "lora_" in n or p.requires_grad:
                    continue
                items.append((n, p.detach()))
                own.append((n, p))
            per_layer.append(items)
            owners.append(own)
        dtype = per_layer[0][0][1].dtype
        store = cls(per_layer, dev, process_group=process_group, dtype=dtype)
        for own in owners:                                    # the full copies are gone: only the shards remain
            for _, p in own:
                p.data = torch.empty(0, dtype=p.dtype, device=dev)
        n_layers = len(layers)

        def bind(l, nxt):
            tensors = store.layer(l, prefetch_next=nxt if 0 <= nxt < n_layers else None)
            for n, p in owners[l]:
                p.data = tensors[n]

        store._hooks = []
        for l, layer in enumerate(layers):
            store._hooks.append(layer.register_forward_pre_hook(lambda _m, _a, l=l: bind(l, l + 1)))
            store._hooks.append(layer.register_full_backward_pre_hook(lambda _m, _g, l=l: bind(l, l - 1)))
        return store
