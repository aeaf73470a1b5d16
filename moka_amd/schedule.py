"""Launch schedules of the MokA adapter path: the executor that turns the C entry points of ``include/moka_hip.h`` into a training step.

What the reference leaves to eager PyTorch + DeepSpeed (``AudioVisualText/trainer.py:163-218``, ``VisualText/train/train.py:601-617``: one
module call per projection, autograd's order for the backward, ZeRO-2's hooks for the gradient exchange) is, on this path, a SCHEDULE over
pre-built argument lists:

* ``AdapterUnit``   -- the projections of one decoder layer that read the same input (q/k/v; o; gate/up; down) with the ctypes argument
  lists of every (grouped) entry point built once;
* ``AdapterChain``  -- the units of a whole decoder stack for one part-batch (its own activations, routing, scratch and saved tensors);
* ``run_forward / run_backward / run_shadows`` -- the launch order of one chain: which launches sit on the dependency chain and which leave
  it (the dA_m / dB halves only the optimizer needs, the weight shadows, the AdamW slices of finished gradient buckets);
* ``GraphedAdapterStep`` -- the micro-batch as N independent part-batch chains (nothing in the model mixes tokens of different samples)
  captured as branches of ONE hub-shaped hipGraph -- or, where collectives run between the buckets, one graph per gradient bucket -- that
  share the parameters, the gradient accumulators and the optimizer slices; replayed once per step.  ``bench.py`` only constructs one and
  calls ``step()``; ``GraphedTrainStep`` (below) is the same hub-shaped capture around a whole decoder stack driven through autograd
  (``parallel.attach`` + ``MokaLinearFn``: the trainer path).

The library only enqueues on the stream it is given -- no allocation, no synchronisation, no state -- so its launches capture unchanged.
Everything here is host-side ordering; the arithmetic is the kernels'.  There is no CPU path.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import byref, c_float, c_void_p
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import torch

from . import _lib

E = 2  # bytes per bf16

# kernel families of the step -> what the executor leaves out when a family is ablated (``ScheduleConfig.skip``): the in-schedule
# marginal of a family = step time with everything - step time without that family's launches (tools/ablate.py, bench.py --ablate)
FAMILIES = {
    "down_fwd": {"moka_down_fwd"},                                                          # x . A_m^T                    (moka_xs_kernel / moka_xwm_kernel)
    "up_fwd": {"moka_up_fwd", "moka_up_fwd:fused", "moka_cross_fwd", "moka_cross_fwd:state"},   # interaction + y += hp . B^T  (moka_yx_kernel; three-launch units: + moka_cross_fwd)
    "up_bwd": {"moka_up_bwd", "moka_up_bwd:g"},                                             # the pass over gy: g (+ dB)   (moka_gs_kernel / moka_gy_kernel)
    "cross_bwd": {"moka_cross_bwd"},                                                        # rank-space backward + key rows
    "dx": {"moka_down_bwd:dx"},                                                             # dx += dh . A_m               (moka_expand_kernel / moka_dxt / moka_dxgt)
    "dA": {"moka_down_bwd:dA", "moka_up_bwd:dB"},                                           # what only the optimizer needs: dA_m (+ dB where it is a pass of its own)
    "shadows": {"moka_weight_shadows"},
    "optimizer": {"optimizer"},
    "none": set(),                                                                          # (nothing left out: capture-to-capture spread of the base schedule)
}


@dataclass
class ScheduleConfig:
    """How a step is laid out (the defaults are the measured ones for 7B widths, r <= 32; ``resolve`` fills the "auto" fields)."""
    chains: int = 2                 # part-batches whose launch chains run side by side
    graph: str = "all"              # "all": the whole micro-batch as one hipGraph; "bwd": forward graph + one graph per gradient bucket; "off": live launches
    topology: str = "hub"           # "hub": chains on forked streams, everything off the chains on the capture's origin; "chain": round 4's shape
    defer_da: str = "unit"          # off | main | side | window | layer | bucket | unit   (bench.py --defer-da)
    split_db: bool = False          # dB off the chain too (where it is a pass of its own: r > 32)
    chain_priority: str = "high"    # stream priority of the dependency chain(s)
    fused: bool = True              # two-launch forward (moka_down_fwd -> moka_up_fwd_fused) where the library says it pays
    shadows: str = "opt"            # "opt": weight shadows rewritten behind the optimizer slices; "main": in front of every fused unit
    shadows_batch: bool = True
    chain_first: bool = True        # capture order of a fork's successors (same DAG; the executor follows a node's FIRST out-edge)
    chain_stagger: int = 0          # MB of a fill in front of the later chains' forward (A/B)
    opt_in_backward: bool = True    # the AdamW slice of a gradient bucket inside the backward
    skip: frozenset = frozenset()   # FAMILIES left out of the schedule (ablation: timing only, results are then wrong by construction)

    def skips(self, call: str) -> bool:
        return bool(self.skip) and any(call in FAMILIES[fam] for fam in self.skip)


class AdapterUnit:
    """The projections of one decoder layer that are fed by the same input (q/k/v; o; gate/up; down) with the
    ctypes argument lists of the six (grouped) entry points pre-built.  G = 1 is the per-projection path.

    members: dicts with d_in, d_out, A (list of M [r, d_in] bf16), dA (fp32 sinks), Bw [d_out, r], dB, y [T, d_out] (base output / gy,
    updated in place), h, hp_kmj, BwT, AT (saved forward -> backward); x / dx: the shared input and input gradient [T, d_in]."""

    def __init__(self, label, members, T, r, M, rt, x, dx, scratch, s_in, s_out, w, c, drop_p, seeds, own_dh_kmj=None, fused=False, company=1,
                 seed_dev=None):
        G = len(members)
        # moka_opts.company: how many independent chains run side by side (the pass over gy then sizes its token runs for its share of the CUs;
        # the dx pass of a wide input takes fewer, longer workgroups); moka_opts.seed_dev: the device word every replay's dropout masks follow
        sd = seed_dev.data_ptr() if (seed_dev is not None and drop_p > 0.0) else None
        self.opts = _lib.MokaOpts(None, 0, int(company), sd)
        self.opts_seed = _lib.MokaOpts(seed_dev=sd)                          # (for the calls that take no company hint)
        ob = byref(self.opts) if (company > 1 or sd is not None) else None
        os_ = byref(self.opts_seed) if sd is not None else None
        # per unit: the library's advice for this shape (moka_up_fwd_fused_pays: e.g. not for the 70B widths' single projections)
        self.fused = bool(fused and _lib.up_fwd_fused_pays(T, _lib.ksplit(T, members[0]["d_in"], r, G), [m["d_out"] for m in members], r))
        self.label, self.G, self.T = label, G, T
        self.d_in = members[0]["d_in"]
        self.d_outs = [m["d_out"] for m in members]
        ks_in = _lib.ksplit(T, self.d_in, r, G)
        ks_out = _lib.ksplit_bwd(T, max(self.d_outs), r)
        P = lambda ts: (c_void_p * len(ts))(*[t.data_ptr() for t in ts])          # noqa: E731
        I = lambda vs: (ctypes.c_int * len(vs))(*vs)                              # noqa: E731
        A = P([a for m in members for a in m["A"]])
        dA = P([a for m in members for a in m["dA"]])
        Bw, dB = P([m["Bw"] for m in members]), P([m["dB"] for m in members])
        y = P([m["y"] for m in members])
        h, hp_kmj = P([m["h"] for m in members]), P([m["hp_kmj"] for m in members])
        BwT, AT = P([m["BwT"] for m in members]), P([m["AT"] for m in members])
        part = P([scratch[g]["part"] for g in range(G)])
        hp_tok = P([scratch[g]["hp_tok"] for g in range(G)])
        dh_tok = P([scratch[g]["dh_tok"] for g in range(G)])
        # (deferred dA: the launches run later, beside the next layer's chain: their operand packs cannot sit in the shared scratch)
        dh_kmj = P([(own_dh_kmj[g] if own_dh_kmj is not None else scratch[g]["dh_kmj"]) for g in range(G)])
        ws = P([rt.cross_ws(r, g) for g in range(G)])
        so = (c_float * M)(*s_out)
        sds = (ctypes.c_ulonglong * G)(*seeds)
        do = I(self.d_outs)
        tm = rt.tok_mod.data_ptr()
        self.keep = (members, A, dA, Bw, dB, y, h, hp_kmj, BwT, AT, part, hp_tok, dh_tok, dh_kmj, ws, so, sds, do, x, dx, seed_dev)
        # (defer_da layer: the dA_m halves of a whole decoder layer as one moka_down_bwd_da_batch launch)
        self.da_items = [((own_dh_kmj[g] if own_dh_kmj is not None else scratch[g]["dh_kmj"]), x, self.d_in, members[g]["dA"], seeds[g]) for g in range(G)]
        self.sh_items = [(m["Bw"], m["d_out"], m["A"], self.d_in, m["BwT"], m["AT"]) for m in members]
        self.db_items = [(members[g]["y"], members[g]["hp_kmj"], members[g]["d_out"], members[g]["dB"]) for g in range(G)]
        self.calls = {
            "moka_down_fwd": ("moka_down_fwd_group", (x.data_ptr(), A, tm, part, T, self.d_in, r, M, G, s_in, drop_p, sds, 0, os_)),
            "moka_cross_fwd": ("moka_cross_fwd_group", (part, ks_in, byref(rt.struct), so, Bw, do, A, self.d_in, h, None, hp_tok, hp_kmj,
                                                        BwT, AT, G, r, w, c)),
            "moka_up_fwd": ("moka_up_fwd_group", (hp_tok, Bw, tm, y, T, r, do, G, 0)),
            # fused forward (default): the up-projection computes the interaction itself from the slices (moka_up_fwd_fused) and writes
            # what the BACKWARD reads from the rank space (h, hp_kmj)
            "moka_up_fwd:fused": ("moka_up_fwd_fused_group", (part, ks_in, byref(rt.struct), so, Bw, y, do, h, hp_kmj, G, r, w, c, 0)),
            "moka_cross_fwd:state": ("moka_cross_fwd_group", (part, ks_in, byref(rt.struct), so, Bw, do, A, self.d_in, h, None, None, hp_kmj,
                                                              BwT, AT, G, r, w, c)),
            # the weight shadows the backward reads (BwT, AT): functions of the weights alone -> once per step, off the chain
            "moka_weight_shadows": ("moka_weight_shadows_group", (Bw, do, A, self.d_in, BwT, AT, G, r, M)),
            "moka_up_bwd": ("moka_up_bwd_group", (y, hp_kmj, BwT, tm, so, part, dB, T, r, do, M, G, 0, ob)),
            # the two outputs of moka_up_bwd as separate calls (where dB is a pass of its own anyway, moka_up_bwd_passes() == 2,
            # it leaves the dependency chain like dA_m)
            "moka_up_bwd:g": ("moka_up_bwd_group", (y, hp_kmj, BwT, tm, so, part, None, T, r, do, M, G, 0, ob)),
            "moka_up_bwd:dB": ("moka_up_bwd_group", (y, hp_kmj, BwT, tm, so, None, dB, T, r, do, M, G, 0, ob)),
            "moka_cross_bwd": ("moka_cross_bwd_group", (part, ks_out, h, byref(rt.struct), s_in, None, dh_tok, dh_kmj, ws, G, r, w, c)),
            "moka_down_bwd": ("moka_down_bwd_group", (dh_tok, dh_kmj, x.data_ptr(), AT, tm, dA, dx.data_ptr(), T, self.d_in, r, M, G,
                                                      drop_p, sds, 0, ob)),
            # the two halves of moka_down_bwd as separate calls (either output may be NULL): dx stays on the dependency chain,
            # dA_m is needed by the optimizer only
            "moka_down_bwd:dx": ("moka_down_bwd_group", (dh_tok, dh_kmj, x.data_ptr(), AT, tm, None, dx.data_ptr(), T, self.d_in, r, M, G,
                                                         drop_p, sds, 0, ob)),
            "moka_down_bwd:dA": ("moka_down_bwd_group", (dh_tok, dh_kmj, x.data_ptr(), AT, tm, dA, None, T, self.d_in, r, M, G,
                                                         drop_p, sds, 0, os_)),
        }
        # algorithmic bytes per launch, SURVEY 8(d) split by entry point and summed over the members (the
        # per-projection definition: a group that reads x once is still credited G reads -- the roofline
        # fraction is defined on the reference's per-projection traffic):
        #   down_fwd: read x  E*T*d_in      up_fwd: read+write y  2*E*T*d_out
        #   up_bwd  : read gy E*T*d_out     down_bwd: read x, r+w dx  3*E*T*d_in
        sdo = sum(self.d_outs)
        self.algo = {"moka_down_fwd": E * T * self.d_in * G, "moka_up_fwd": 2 * E * T * sdo, "moka_up_bwd": E * T * sdo,
                     "moka_down_bwd": 3 * E * T * self.d_in * G, "moka_cross_fwd": 3 * 4 * T * r * G, "moka_cross_bwd": 3 * 4 * T * r * G,
                     "moka_weight_shadows": 2 * E * r * (sdo + M * self.d_in * G)}


class _Bag(dict):
    """dict with attribute access (the workload / chain records: tools written against the dict form keep working)."""
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


class AdapterChain(_Bag):
    """One part-batch: units (layer-major, the layer's units in forward order), units_per_layer, rt (routing), T (tokens), layer_da / layer_db
    (argument lists of the batched per-layer dA_m / dB launches), rank, reuse_wait (the pack buffers of layer l + 2 are reused by layer l)."""


class AdapterWorkload(_Bag):
    """chains (list of AdapterChain, sharing parameters / gradient accumulators), master / work / gbuf (flat fp32 master, bf16 working copy,
    fp32 gradient), bucket (parallel.FlatGradBucket), T (tokens of the whole micro-batch), n_params, layer_end, rank."""


def make_layer_batches(units: Sequence[AdapterUnit], per: int, n_layers: int, rt, Tc: int, r: int, M: int, drop_p: float):
    """Argument lists of moka_down_bwd_da_batch / moka_up_bwd_db_batch for every layer of a chain (the layer's units in backward order)."""
    layer_da, layer_db = [], []
    os_ = byref(units[0].opts_seed) if (units and units[0].opts_seed.seed_dev) else None
    for l in range(n_layers):
        items = [it for u in reversed(units[l * per:(l + 1) * per]) for it in u.da_items]
        n = len(items)
        layer_da.append(((c_void_p * n)(*[it[0].data_ptr() for it in items]), (c_void_p * n)(*[it[1].data_ptr() for it in items]),
                         (ctypes.c_int * n)(*[it[2] for it in items]), rt.tok_mod.data_ptr(),
                         (c_void_p * (n * M))(*[a.data_ptr() for it in items for a in it[3]]), n, Tc, r, M, drop_p,
                         (ctypes.c_ulonglong * n)(*[it[4] for it in items]), 0, os_))
        dbi = [it for u in reversed(units[l * per:(l + 1) * per]) for it in u.db_items]
        layer_db.append(((c_void_p * n)(*[it[0].data_ptr() for it in dbi]), (c_void_p * n)(*[it[1].data_ptr() for it in dbi]),
                         (ctypes.c_int * n)(*[it[2] for it in dbi]), rt.tok_mod.data_ptr(), (c_void_p * n)(*[it[3].data_ptr() for it in dbi]),
                         n, Tc, r, M, 0, None))
    return layer_da, layer_db


ENTRY = ["moka_down_fwd", "moka_cross_fwd", "moka_up_fwd", "moka_up_bwd", "moka_cross_bwd", "moka_down_bwd", "moka_weight_shadows"]


class Recorder:
    """HIP-event brackets around launches on the launch stream.  `only` limits which entry points are
    bracketed (bracketing every launch of a step makes the host the bottleneck and distorts the headline)."""

    def __init__(self, only=None, every=1):
        self.only, self.items, self.pool, self.every, self.seen = only, [], [], max(1, int(every)), 0

    def skip(self):
        """Bracket every `every`-th eligible launch (an event record is a packet of its own on the stream: ~2 us each)."""
        self.seen += 1
        return (self.seen % self.every) != 0

    def event(self):
        return self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)

    def reserve(self, n):
        self.pool.extend(torch.cuda.Event(enable_timing=True) for _ in range(n))


_NOSKIP = ScheduleConfig()


def _call(lib, name, u, sp, rec, stream=None, cfg: ScheduleConfig = _NOSKIP):
    """Launch one entry point of unit `u`; bracket it with HIP events (on `stream`, default: torch's current stream, which is the
    launch stream of the bracketed passes) when the recorder asks for it.  "entry:variant" is recorded as "entry"."""
    if cfg.skip and cfg.skips(name):
        return
    sym, args = u.calls[name]
    base = name.split(":")[0]
    if rec is None or (rec.only is not None and base not in rec.only) or rec.skip():
        rc = getattr(lib, sym)(*args, sp)
    else:
        e0, e1 = rec.event(), rec.event()
        e0.record(stream) if stream is not None else e0.record()
        rc = getattr(lib, sym)(*args, sp)
        e1.record(stream) if stream is not None else e1.record()
        rec.items.append((base, u, e0, e1))
    if rc:
        raise _lib.MokaError(lib.moka_last_error().decode())


def run_forward(lib, ch, sp, rec=None, shadows=False, cfg: ScheduleConfig = _NOSKIP):
    """Per unit: down-projection, interaction, up-projection.  Fused units: down-projection -> up-projection with the
    interaction inside (it also writes h and the rank-major hp pack for the backward).  The weight shadows the backward reads (BwT, AT)
    are functions of the weights alone: they are rewritten where the weights change (run_shadows behind the optimizer step), not in
    the forward -- unless `shadows` asks for them in front of every unit (shadows = "main")."""
    for u in ch["units"]:
        if not u.fused:
            _call(lib, "moka_down_fwd", u, sp, rec, cfg=cfg)
            _call(lib, "moka_cross_fwd", u, sp, rec, cfg=cfg)   # (writes its own weight shadows: taking them out gained nothing at rank 64, 82.3 vs 84.1 ms)
            _call(lib, "moka_up_fwd", u, sp, rec, cfg=cfg)
            continue
        if shadows:
            _call(lib, "moka_weight_shadows", u, sp, rec, cfg=cfg)
        _call(lib, "moka_down_fwd", u, sp, rec, cfg=cfg)
        _call(lib, "moka_up_fwd:fused", u, sp, rec, cfg=cfg)


def run_shadows(lib, wl, sp, layers, rec=None, cfg: ScheduleConfig = _NOSKIP):
    """BwT / AT of the given layers' FUSED units (the other units' moka_cross_fwd writes theirs in the forward; all chains share the
    parameters: the first chain's units carry the buffers): one moka_weight_shadows_batch launch per 16 projections (a recorder gets
    the per-unit launches, so that the entry point keeps its line in the table)."""
    if cfg.skips("moka_weight_shadows"):
        return
    ch0 = wl["chains"][0]
    units, per = ch0["units"], ch0["units_per_layer"]
    if rec is not None or not cfg.shadows_batch:
        for l in layers:
            for u in units[l * per:(l + 1) * per]:
                if u.fused:
                    _call(lib, "moka_weight_shadows", u, sp, rec)
        return
    cache = wl.setdefault("_shadow_calls", {})
    key = tuple(layers)
    if key not in cache:
        items = [it for l in layers for u in units[l * per:(l + 1) * per] if u.fused for it in u.sh_items]
        calls = []
        for i in range(0, len(items), _lib.MOKA_MAX_SHADOW_BATCH):
            part = items[i:i + _lib.MOKA_MAX_SHADOW_BATCH]
            n, M = len(part), len(part[0][2])
            calls.append(((c_void_p * n)(*[it[0].data_ptr() for it in part]), (ctypes.c_int * n)(*[it[1] for it in part]),
                          (c_void_p * (n * M))(*[a.data_ptr() for it in part for a in it[2]]), (ctypes.c_int * n)(*[it[3] for it in part]),
                          (c_void_p * n)(*[it[4].data_ptr() for it in part]), (c_void_p * n)(*[it[5].data_ptr() for it in part]), n, ch0["rank"], M))
        cache[key] = calls
    for argl in cache[key]:
        _lib.check(lib.moka_weight_shadows_batch(*argl, sp), "moka_weight_shadows_batch")


def run_backward(lib, ch, sp, n_layers, on_layer_done=None, rec=None, lo=0, defer=None, bucket_opt=None, shadows_after_opt=False, state=None, join=True,
                 flush=False, cfg: ScheduleConfig = _NOSKIP, wl=None):
    """Reverse layer order (layers n_layers-1 .. lo); `on_layer_done(l)` fires after layer l's launches are enqueued.
    defer = (mode, main_stream, side_stream[, split_db, flush_at]): the dA_m halves of a layer's moka_down_bwd calls leave the dependency
    chain (only the optimizer needs them) and are enqueued after the layer's chain -- "main": on the same stream; "side": on a second stream,
    beside the NEXT layer's chain (whose rank-space kernels leave most of the chip idle), joined before the gradients are used.
    state / join: the walk in pieces (the hub capture walks the chains layer by layer, interleaved): `state` carries the buffer-reuse events
    from call to call, join=False leaves the side stream unjoined.  wl: the workload (for run_shadows behind a bucket's optimizer slice)."""
    units, per = ch["units"], ch["units_per_layer"]
    C = lambda name, u, s, r_: _call(lib, name, u, s, r_, cfg=cfg)               # noqa: E731
    if defer is None:
        for l in range(n_layers - 1, lo - 1, -1):
            for u in reversed(units[l * per:(l + 1) * per]):
                C("moka_up_bwd", u, sp, rec)
                C("moka_cross_bwd", u, sp, rec)
                if cfg.skip and (cfg.skips("moka_down_bwd:dx") or cfg.skips("moka_down_bwd:dA")):
                    C("moka_down_bwd:dx", u, sp, rec)
                    C("moka_down_bwd:dA", u, sp, None)
                else:
                    C("moka_down_bwd", u, sp, rec)
            if on_layer_done is not None:
                on_layer_done(l)
        return
    mode, main, side = defer[:3]
    split_db = len(defer) > 3 and defer[3]                       # dB off the chain too (where it is a pass of its own)
    up = "moka_up_bwd:g" if split_db else "moka_up_bwd"
    sps = c_void_p(side.cuda_stream)
    done = state if state is not None else {}                    # layer -> event "its deferred dA launches have finished" (side mode)
    flush_at = None
    per_unit = False
    if mode == "layer":
        mode, batched = "side", True                             # the side schedule with ONE dA launch per layer
    elif mode == "unit":
        # a unit's dA_m leaves for the side stream as soon as its rank-space backward (which writes the packs it reads) has been enqueued,
        # captured behind the unit's dx launch (chain-first).  Hub-shaped graphs only: there a fork per unit does not cut the chain
        mode, batched, per_unit = "side", False, True
    elif mode == "bucket":
        # one fork per gradient BUCKET of layers (every cross-stream edge of a hipGraph costs its replay host time): the batched dA launches
        # of the bucket's layers go out together when its first layer's chain has been enqueued; every layer owns its pack buffers
        mode, batched, flush_at = "side", True, defer[4]
    else:
        batched = False
    skip_da = cfg.skips("moka_down_bwd:dA")
    held = []                                                    # layers whose deferred launches wait for the bucket's flush
    # chain_first (captures only): a layer's side-stream launches are enqueued AFTER the first launch of the next layer's chain.  The DAG is
    # the same; what changes is the order of a fork node's out-edges, and the hipGraph executor (ROCm 7.2) derives its execution streams
    # from a depth-first walk that follows the FIRST out-edge: side-first lets the walk leave the chain at every fork
    reuse = ch.get("reuse_wait", True) and flush_at is None      # (pack buffers of layer l + 2 reused by layer l: the chain waits for that dA)
    late = cfg.chain_first and on_layer_done is None and mode == "side"
    post = done.pop("post", None) if late else None              # (layer, held layers, event on main) of the fork not yet emitted

    def emit(l, held_, ev_main, pending_u):
        side.wait_event(ev_main)
        for u in reversed(units[l * per:(l + 1) * per]):
            if split_db and not batched and not per_unit:
                C("moka_up_bwd:dB", u, sps, None)
            if batched or per_unit or (mode == "window" and u is not pending_u):
                continue                                        # (already out, beside the next unit's rank-space backward)
            C("moka_down_bwd:dA", u, sps, None)
        if batched and not skip_da:
            for ll in held_ + [l]:
                if split_db:
                    _lib.check(lib.moka_up_bwd_db_batch(*ch["layer_db"][ll], sps), "moka_up_bwd_db_batch")
                _lib.check(lib.moka_down_bwd_da_batch(*ch["layer_da"][ll], sps), "moka_down_bwd_da_batch")
        if bucket_opt is not None and mode in ("side", "window"):
            # single GPU: the optimizer step of a gradient bucket as soon as its last dA_m / dB launches are on the side stream -- the
            # update of the finished layers overlaps the backward of the earlier ones (FlatAdamW.step_range, coefficients in device memory)
            opt_, bucket_, scale_ = bucket_opt
            if bucket_.is_bucket_first(l):
                blo, bhi = bucket_.bucket_bounds(l)
                with torch.cuda.stream(side):
                    opt_.step_range(blo, bhi, grad_scale=scale_, zero_grad=True)
                    if shadows_after_opt:
                        # the bucket's weights have just changed: their shadows for the NEXT step's backward, still off the chain
                        run_shadows(lib, wl, sps, bucket_.bucket_layers(l), cfg=cfg)
        ev = torch.cuda.Event()
        ev.record(side)
        done[l] = ev

    for l in range(n_layers - 1, lo - 1, -1):
        if not late and reuse and mode in ("side", "window") and (l + 2) in done:
            main.wait_event(done.pop(l + 2))                     # layer l reuses the pack buffers of layer l + 2
        pending = None
        for u in reversed(units[l * per:(l + 1) * per]):
            C(up, u, sp, rec)
            if post is not None:
                emit(*post)                                      # the layer before's fork, behind this layer's first launch (chain_first)
                post = None
            if late and pending is None and reuse and (l + 2) in done:
                main.wait_event(done.pop(l + 2))                 # (the first writer of the reused pack buffers is this unit's rank-space backward)
            if mode == "window" and pending is not None:
                # the dA of the unit before goes out HERE, so that it starts with this unit's rank-space backward -- the two launches of
                # the chain that leave the memory system idle (a unit's dA moves about as many bytes as that window could)
                side.wait_stream(main)
                C("moka_down_bwd:dA", pending, sps, None)
            C("moka_cross_bwd", u, sp, rec)
            if per_unit:
                ev_u = torch.cuda.Event()
                ev_u.record(main)
            C("moka_down_bwd:dx", u, sp, None)
            if per_unit and cfg.skip and cfg.skips("moka_down_bwd:dx"):
                # (ablation of the dx family: the unit's fork must still see a CHAIN node captured first -- the executor follows a fork's
                #  first out-edge -- so a 256-byte memset stands where the dx launch stood)
                with torch.cuda.stream(main):
                    ch.setdefault("_placeholder", torch.zeros(64, device=main.device)).zero_()
            if per_unit:
                side.wait_event(ev_u)
                if split_db:
                    C("moka_up_bwd:dB", u, sps, None)
                C("moka_down_bwd:dA", u, sps, None)
            pending = u
        if mode == "main":
            for u in reversed(units[l * per:(l + 1) * per]):
                if split_db:
                    C("moka_up_bwd:dB", u, sp, None)
                C("moka_down_bwd:dA", u, sp, None)
        elif flush_at is not None and not flush_at(l) and l > lo:
            held.append(l)                                       # (leaves with its bucket's first layer)
        else:
            ev_main = torch.cuda.Event()
            ev_main.record(main)
            if late:
                post = (l, held, ev_main, pending)
            else:
                emit(l, held, ev_main, pending)
            held = []
        if on_layer_done is not None:
            if mode in ("side", "window") and l in done:
                main.wait_event(done[l])                         # (a bucket must not ship before its dA has landed)
            on_layer_done(l)
    if post is not None:
        if join or flush or lo == 0:
            emit(*post)                                          # nothing follows on the chain
        else:
            done["post"] = post                                  # (the next piece of the walk emits it)
    if mode in ("side", "window") and join:
        main.wait_stream(side)


class HubCapture:
    """One hipGraph in the hub shape: N chains on N forked streams, everything off the chains on the capture's origin stream.

    * hipStreamEndCapture (ROCm 7.2) segfaults on ANY dependency between two streams that are both forks
      (tools/probes/capture_topology.py): every edge has to touch the origin, so the origin is the hub;
    * the executor does not run a graph on the capture's streams: it cuts the DAG into lists by a depth-first walk from the roots
      that follows every node's FIRST out-edge, gives every list a stream of its own and maps those onto a handful of in-order
      hardware queues (a list that waits for another list blocks whatever shares its queue).  Round 4's side-first forks made
      every layer's dA_m a list of its own and cut the chain at every fork.  This graph is SHAPED for that walk: the hub's
      launches are the root's first path (an anchor node captured in front of the forks' first launches), nothing on a chain ever
      waits for the hub (every layer owns its pack buffers: no reuse edges), so the walk yields exactly 1 + N lists.

    Usage:  with HubCapture(graph, n, device, priority) as hc:  ... launches on hc.branch[i] / hc.hub ...
    `root(fn)` runs fn on the hub as the graph's root node (default: a memset of a 256-byte anchor)."""

    def __init__(self, graph: "torch.cuda.CUDAGraph", n_chains: int, device, priority: int = 0, root: Optional[Callable[[], None]] = None,
                 pool=None):
        self.graph, self.device, self.root = graph, device, root
        self.single = n_chains == 0                  # ONE list: everything -- the chain's launches and what is "off the chain" -- on the capture stream
        self.hub = torch.cuda.Stream(device=device, priority=priority if self.single else 0)
        self.branch = [self.hub] if self.single else [torch.cuda.Stream(device=device, priority=priority) for _ in range(n_chains)]
        self.anchor = torch.zeros(64, device=device)
        self._ctx = None
        self._pool = pool

    def __enter__(self):
        kw = {"pool": self._pool} if self._pool is not None else {}
        self._ctx = torch.cuda.graph(self.graph, stream=self.hub, **kw)
        self._ctx.__enter__()
        self.cur = torch.cuda.current_stream()
        if self.root is not None:
            self.root()
        else:
            self.anchor.zero_()                      # (the root)
        if self.single:
            return self
        for st in self.branch:
            st.wait_stream(self.cur)                 # fork
        self.anchor.zero_()                          # the root's FIRST successor is on the hub: the walk runs down the hub before it sees a chain
        return self

    def __exit__(self, et, ev, tb):
        if et is None and not self.single:
            for st in self.branch:
                self.cur.wait_stream(st)             # join
        return self._ctx.__exit__(et, ev, tb)


class GraphedAdapterStep:
    """The adapter fwd + bwd of a decoder stack's adapted projections + the data-parallel step on their gradients, as a replayable schedule.

        step = GraphedAdapterStep(workload, cfg, optimizer=opt, world=world, comm=comm)
        step.capture()              # warm-up launches + hipGraph capture (falls back to live launches if the capture fails)
        for i in range(n): step.step(i)

    workload: AdapterWorkload (chains of AdapterUnit over caller-owned buffers, a parallel.FlatGradBucket);  optimizer: parallel.FlatAdamW on
    the workload's flat buffers or None;  comm: do the gradient collectives run (N > 1 or a forced one-rank group)?"""

    def __init__(self, wl: AdapterWorkload, cfg: ScheduleConfig, n_layers: int, optimizer=None, world: int = 1, comm: bool = False, device=None):
        self.wl, self.cfg, self.L, self.opt, self.world, self.comm = wl, cfg, int(n_layers), optimizer, int(world), bool(comm)
        self.lib = _lib.load()
        self.bucket = wl["bucket"]
        self.dev = device if device is not None else wl["gbuf"].device
        self.main_stream = torch.cuda.current_stream(self.dev)
        self.chains = wl["chains"]
        self.hub = cfg.graph != "off" and (cfg.topology == "hub" or cfg.chains > 1)
        opt = optimizer
        # the optimizer step per gradient bucket INSIDE the backward (off: one launch behind it): needs the side stream of the deferred dA_m
        # (single GPU) or the communication stream behind the bucket's all-reduce (N > 1, fp32 payload)
        self.opt_in_bwd = (opt is not None and cfg.opt_in_backward and not cfg.skips("optimizer") and
                           ((not comm and (cfg.defer_da in ("side", "window", "layer", "bucket", "unit") or cfg.chains > 1) and cfg.graph in ("all", "off"))
                            or comm))
        any_fused = any(u.fused for u in self.chains[0]["units"])
        self.shadows_main = bool(any_fused and cfg.shadows == "main")
        self.shadows_opt = bool(any_fused and cfg.shadows == "opt")
        self.shadows_in_cb = False
        self.fwd_bwd_graph = self.bwd_graphs = self.fwd_graph = None
        self.graph_mode = cfg.graph
        self._keep = []
        self.comm_ev = [] if comm else None
        self.live_side = torch.cuda.Stream(device=self.dev) if cfg.defer_da != "off" else None
        # the device word the units' dropout kernels fold into their seeds (moka_opts.seed_dev): rewritten in front of every step, so a
        # replayed graph -- whose launch arguments, seeds included, are frozen -- still draws a fresh keep mask per step
        self.seed_epoch = wl.get("seed_epoch")
        self._epoch = 0
        if self.opt_in_bwd:
            opt.set_device_step(0)                       # (allocates the device-side coefficient state; no step counted)
            if comm:
                ends = wl["layer_end"]

                def _reduced(blo, bhi):
                    opt.step_range(blo, bhi, grad_scale=1.0 / self.world, zero_grad=True)
                    if self.shadows_opt:                 # (on the communication stream, behind the bucket's update)
                        run_shadows(self.lib, wl, c_void_p(torch.cuda.current_stream().cuda_stream), [l for l in range(self.L) if blo < ends[l] <= bhi], cfg=cfg)
                self.bucket.on_reduced = _reduced
                self.shadows_in_cb = self.shadows_opt
        if self.shadows_opt:
            run_shadows(self.lib, wl, c_void_p(torch.cuda.current_stream().cuda_stream), range(self.L), cfg=cfg)     # the initial weights' shadows
            torch.cuda.synchronize()

    # ------------------------------------------------------------------------------------------------ capture
    def _defer(self, main, side):
        cfg = self.cfg
        return (cfg.defer_da, main, side, cfg.split_db, self.bucket.is_bucket_first) if cfg.defer_da != "off" else None

    def _pieces(self, lo_, hi_):
        # (pieces of the walk: a layer; with defer_da bucket a whole gradient bucket, whose dA_m launches leave together)
        if self.cfg.defer_da == "bucket":
            return [(f, self.bucket.bucket_layers(f).stop) for f in reversed(self.bucket.bucket_firsts()) if lo_ <= f < hi_]
        return [(l, l + 1) for l in range(hi_ - 1, lo_ - 1, -1)]

    def _capture_hub(self, graph, forward, pieces, with_opt):
        """One graph in the hub shape (HubCapture): the walk of the backward is captured piece by piece (a layer; defer_da bucket: a gradient
        bucket), chain by chain, so that the hub's stream order is "piece p of every chain, then the bucket's optimizer slice"."""
        cfg, lib, wl, opt, bucket = self.cfg, self.lib, self.wl, self.opt, self.bucket
        pri = -1 if cfg.chain_priority == "high" else 0

        def root():
            # the step's AdamW coefficients, written on the device by a one-thread launch that counts the steps itself:
            # every replay advances by one, nothing is read from host memory (FlatAdamW.begin_step)
            opt.begin_step(device_counter=True)
            opt.t -= 1                                   # (the capture is not a step)
        hc = HubCapture(graph, len(self.chains), self.dev, priority=pri, root=root if with_opt else None)
        with hc:
            cur, branch = hc.cur, hc.branch
            if forward:
                for ci_, (ch, st) in enumerate(zip(self.chains, branch)):
                    if ci_ and cfg.chain_stagger > 0:
                        # (identical chains that start together march in lockstep -- both in a latency-bound launch at the same
                        #  time; a fill of `chain_stagger` MB in front of the later chains shifts their phase)
                        with torch.cuda.stream(st):
                            self._stagger_buf[:ci_ * cfg.chain_stagger * (1 << 20)].zero_()
                    run_forward(lib, ch, c_void_p(st.cuda_stream), shadows=self.shadows_main, cfg=cfg)
            states = [dict() for _ in branch]
            pend_opt = None

            def hub_opt(lb, evs):
                # the chains add into the same gradient accumulators: the bucket's AdamW slice (and its layers' weight shadows for
                # the next step) goes out on the hub, behind the dA_m launches of the bucket's first layer of EVERY chain
                for ev in evs:
                    cur.wait_event(ev)                   # (the in-chain gradients of the layer: dB rides with the pass over gy)
                blo, bhi = bucket.bucket_bounds(lb)
                opt.step_range(blo, bhi, grad_scale=1.0 / self.world, zero_grad=True)
                if self.shadows_opt:
                    run_shadows(lib, wl, c_void_p(cur.cuda_stream), bucket.bucket_layers(lb), cfg=cfg)
            for pi_, (l, l_hi) in enumerate(pieces):
                for ch, st, stt in zip(self.chains, branch, states):
                    run_backward(lib, ch, c_void_p(st.cuda_stream), l_hi, lo=l, state=stt, join=False, flush=pi_ == len(pieces) - 1,
                                 defer=self._defer(st, cur), cfg=cfg, wl=wl)
                if pend_opt is not None:
                    hub_opt(*pend_opt)                   # (chain-first: behind the first launches of the chains' next piece)
                    pend_opt = None
                if with_opt and bucket.is_bucket_first(l):
                    evs = []
                    for st in branch:
                        ev = torch.cuda.Event()
                        ev.record(st)
                        evs.append(ev)
                    if cfg.chain_first and pi_ < len(pieces) - 1:
                        pend_opt = (l, evs)
                    else:
                        hub_opt(l, evs)
        self._keep.append(hc)                            # (streams kept alive with the graph)

    def capture(self) -> bool:
        """Warm-up launches on a side stream (LDS attributes, lazy module load), then the capture.  Returns True when the step will be
        replayed from hipGraph(s); on a capture failure the step is launched live (every launch of every chain still happens)."""
        cfg, lib, wl, opt, bucket, L, dev = self.cfg, self.lib, self.wl, self.opt, self.bucket, self.L, self.dev
        if cfg.graph == "off":
            return False
        try:
            # the chain's (capture) stream at high priority, the deferred dA / dB stream at normal: when both have workgroups waiting, the
            # dependency chain goes first (7B r = 16, same box twice: 33.43-33.48 -> 33.16-33.18 ms; r = 64: no difference)
            pri = -1 if cfg.chain_priority == "high" else 0
            side = torch.cuda.Stream(device=dev, priority=pri)
            with torch.cuda.stream(side):
                spw = c_void_p(side.cuda_stream)
                for ch in self.chains:
                    run_forward(lib, ch, spw)
                    run_backward(lib, ch, spw, L)        # warm-up on the capture stream
            torch.cuda.synchronize()
            self._stagger_buf = torch.empty(max(1, cfg.chain_stagger * (1 << 20) * max(1, cfg.chains - 1)), dtype=torch.uint8, device=dev)
            if os.environ.get("MOKA_BENCH_FAIL_CAPTURE") == "1":
                raise RuntimeError("MOKA_BENCH_FAIL_CAPTURE=1 (test hook: exercise the live fallback)")
            ch0 = self.chains[0]
            if cfg.graph == "all":
                # (collectives cannot ride inside the graph: capturing the one-rank RCCL all-reduce with torch 2.10 / RCCL 2.26.6 segfaults at
                #  capture time -- measured round 4 -- so N > 1 and a forced one-rank group use one graph per gradient bucket with the hooks between them)
                assert not self.comm, "graph = all: single GPU without collectives only"
                self.fwd_bwd_graph = torch.cuda.CUDAGraph()
                if not self.hub:
                    da_side = torch.cuda.Stream(device=dev)
                    self._keep.append((side, da_side))
                    with torch.cuda.graph(self.fwd_bwd_graph, stream=side):
                        cur = torch.cuda.current_stream()
                        spg = c_void_p(cur.cuda_stream)
                        if self.opt_in_bwd:
                            opt.begin_step(device_counter=True)
                            opt.t -= 1                   # (the capture is not a step)
                        run_forward(lib, ch0, spg, shadows=self.shadows_main, cfg=cfg)
                        run_backward(lib, ch0, spg, L, defer=self._defer(cur, da_side),
                                     bucket_opt=(opt, bucket, 1.0 / self.world) if self.opt_in_bwd else None,
                                     shadows_after_opt=self.shadows_opt and self.opt_in_bwd, cfg=cfg, wl=wl)
                else:
                    self._capture_hub(self.fwd_bwd_graph, True, self._pieces(0, L), self.opt_in_bwd)
            elif not self.hub:
                da_side = torch.cuda.Stream(device=dev)
                self._keep.append((side, da_side))
                self.fwd_graph = torch.cuda.CUDAGraph()  # the forward has no hooks: one graph
                with torch.cuda.graph(self.fwd_graph, stream=side):
                    run_forward(lib, ch0, c_void_p(torch.cuda.current_stream().cuda_stream), shadows=self.shadows_main, cfg=cfg)
                self.bwd_graphs = []
                for lo in reversed(bucket.bucket_firsts()):      # buckets are contiguous groups of layers, walked last -> first
                    hi = bucket.bucket_layers(lo).stop
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        cs = torch.cuda.current_stream()
                        run_backward(lib, ch0, c_void_p(cs.cuda_stream), hi, lo=lo, defer=self._defer(cs, da_side), cfg=cfg, wl=wl)
                    self.bwd_graphs.append((g, lo, hi))
            else:
                # N > 1 with part-batch chains: the forward as one hub-shaped graph, one hub-shaped graph per gradient bucket of the backward (the
                # chains meet at every graph's end: that is where the bucket's all-reduce is handed to RCCL)
                self.fwd_graph = torch.cuda.CUDAGraph()
                self._capture_hub(self.fwd_graph, True, [], False)
                self.bwd_graphs = []
                for lo in reversed(bucket.bucket_firsts()):
                    hi = bucket.bucket_layers(lo).stop
                    g = torch.cuda.CUDAGraph()
                    self._capture_hub(g, False, self._pieces(lo, hi), False)
                    self.bwd_graphs.append((g, lo, hi))
            torch.cuda.synchronize()
            return True
        except Exception as exc:                         # capture is an optimisation, never a requirement
            # (with chains: the part-batches then run one after the other on the one stream -- every launch of the step still happens)
            import sys
            print(f"moka_amd.schedule: hipGraph capture failed ({exc!r}); launching live" +
                  (", the %d chains back to back" % cfg.chains if cfg.chains > 1 else ""), file=sys.stderr)
            self.fwd_bwd_graph = self.bwd_graphs = self.fwd_graph = None
            self.graph_mode = "off (capture failed)"     # (what a report says is what ran)
            torch.cuda.synchronize()
            return False

    # ------------------------------------------------------------------------------------------------ one step
    def step(self, i: int = 0, rec: Optional[Recorder] = None, time_comm: bool = False) -> None:
        cfg, lib, wl, opt, bucket, L = self.cfg, self.lib, self.wl, self.opt, self.bucket, self.L
        main_stream = self.main_stream
        sp = c_void_p(main_stream.cuda_stream)
        if self.seed_epoch is not None:
            self._epoch = (self._epoch + 0x9E3779B97F4A7C15) & 0x7fffffffffffffff       # (a fill kernel on the launch stream: stream-ordered in front of the replay)
            self.seed_epoch.fill_(self._epoch)
        if opt is None or cfg.skips("optimizer"):
            bucket.zero_()                               # (the optimizer kernel leaves the gradient buffer zeroed)
        if self.opt_in_bwd:
            if self.fwd_bwd_graph is None:
                opt.begin_step()                         # this step's coefficients: a one-thread launch on the main stream (launch arguments)
            else:
                opt.t += 1                               # (the captured launch counts on the device; the host keeps the books)
        if self.fwd_bwd_graph is not None:
            self.fwd_bwd_graph.replay()
        else:
            if self.fwd_graph is not None:
                self.fwd_graph.replay()
            else:
                for ch in self.chains:
                    run_forward(lib, ch, sp, rec, shadows=self.shadows_main, cfg=cfg)
            if self.bwd_graphs is not None:
                for g, lo, hi in self.bwd_graphs:
                    g.replay()
                    for l in range(hi - 1, lo - 1, -1):
                        bucket.layer_done(l)             # all-reduce of the finished bucket overlaps the next graphs
            else:
                # (live launches: the chains one after the other; the bucket hooks / optimizer slices ride with the LAST chain's layers -- every
                #  earlier chain's gradients are in front of them in stream order)
                for ci, ch in enumerate(self.chains):
                    last = ci == len(self.chains) - 1
                    run_backward(lib, ch, sp, L, bucket.layer_done if last else None, rec,   # all-reduce of finished layer groups overlaps the rest
                                 defer=self._defer(main_stream, self.live_side),
                                 bucket_opt=(opt, bucket, 1.0 / self.world) if (self.opt_in_bwd and not self.comm and last) else None,
                                 shadows_after_opt=self.shadows_opt and self.opt_in_bwd and not self.comm and last, cfg=cfg, wl=wl)
        skip_opt = opt is None or cfg.skips("optimizer")
        if self.comm_ev is not None and time_comm:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main_stream)
            bucket.finish(average=skip_opt)
            e1.record(main_stream)
            self.comm_ev.append((e0, e1))
        else:
            bucket.finish(average=skip_opt)              # join the all-reduces; the optimizer kernel averages (grad_scale)
        if not skip_opt and not self.opt_in_bwd:
            opt.step(grad_scale=1.0 / self.world, zero_grad=True)
        if self.shadows_opt and not skip_opt and not (self.opt_in_bwd and not self.comm) and not self.shadows_in_cb:
            run_shadows(lib, wl, sp, range(L), cfg=cfg)  # (every weight has changed: the shadows of the whole stack, behind the step)

    @property
    def graphed(self) -> bool:
        return self.fwd_bwd_graph is not None or self.bwd_graphs is not None

    def replay_only(self) -> None:
        """The captured launches alone (no bucket hooks, no bookkeeping): what --verify-graph compares with the live launches."""
        if self.fwd_bwd_graph is not None:
            self.fwd_bwd_graph.replay()
        else:
            self.fwd_graph.replay()
            for g, lo, hi in self.bwd_graphs:
                g.replay()

    def live_pass(self, sp, rec_fwd: Optional[Recorder] = None, rec_bwd: Optional[Recorder] = None) -> None:
        """Every launch of the step live, chain after chain on ONE stream, dA_m / dB in the chain (the reference order of the schedule)."""
        for ch in self.chains:
            run_forward(self.lib, ch, sp, rec_fwd, shadows=self.shadows_main)
            run_backward(self.lib, ch, sp, self.L, None, rec_bwd)

    def replay_host_ms(self, n: int = 3) -> Optional[float]:
        """How long hipGraphLaunch keeps the launching thread for ONE step (idle GPU in front of it, so nothing blocks on a full queue)."""
        if self.fwd_bwd_graph is None:
            return None
        import time
        hs = []
        for _ in range(n):
            torch.cuda.synchronize()
            th = time.perf_counter()
            self.fwd_bwd_graph.replay()
            hs.append((time.perf_counter() - th) * 1e3)
            torch.cuda.synchronize()
            if self.opt_in_bwd:
                self.opt.t += 1
        return round(min(hs), 3)


# ======================================================================================================================================
# the trainer path: a whole decoder stack driven through autograd, captured in the same hub shape
# ======================================================================================================================================
def _split_batch(batch, sizes):
    """dict / tuple / list of tensors (batch dimension 0) -> one such container per part."""
    def cut(t):
        return list(torch.split(t, sizes, dim=0)) if isinstance(t, torch.Tensor) else [t] * len(sizes)
    if isinstance(batch, dict):
        cols = {k: cut(v) for k, v in batch.items()}
        return [{k: cols[k][i] for k in batch} for i in range(len(sizes))]
    if isinstance(batch, (tuple, list)):
        cols = [cut(v) for v in batch]
        return [type(batch)(c[i] for c in cols) for i in range(len(sizes))]
    return cut(batch)


def _tensors(part):
    if isinstance(part, dict):
        return [v for v in part.values() if isinstance(v, torch.Tensor)]
    if isinstance(part, (tuple, list)):
        return [v for v in part if isinstance(v, torch.Tensor)]
    return [part]


def _clone_part(part):
    c = lambda v: v.clone() if isinstance(v, torch.Tensor) else v                 # noqa: E731
    if isinstance(part, dict):
        return {k: c(v) for k, v in part.items()}
    if isinstance(part, (tuple, list)):
        return type(part)(c(v) for v in part)
    return c(part)


class GraphedTrainStep:
    """ONE training step of a model -- forward, backward, deferred weight gradients, fused AdamW, weight shadows -- as ONE hipGraph:
    a single list (``chains=1``, the default: what pays under a frozen base), or ``chains`` part-batches on forked streams in the hub shape.

        dp = moka_amd.parallel.attach(model, ...)                # the reference's get_peft_model(...) model
        step = GraphedTrainStep(dp, lambda part: model(**part).loss, example_batch,
                                routing_fn=lambda part: MokaRouting.from_avt_masks([part["m_t"], part["m_v"], part["m_a"], part["m_q"]]))
        for batch in loader:
            loss = step(batch)                                   # copies the batch into the static inputs, replays

    What the reference runs as ``Trainer.training_step`` + DeepSpeed's hooks + ``optimizer.step()`` (``AudioVisualText/trainer.py:163-218``,
    ``VisualText/train/train.py:601-617``) is here a replay: the step is captured once through autograd (``MokaLinearFn`` and every stock
    PyTorch-ROCm op of the frozen base alike), so the launch order, the part-batch chains and everything ``attach`` takes off the
    dependency chains (dA_m per layer, AdamW slice + weight shadows per gradient bucket) are the graph's.

    * ``step_fn(part) -> loss``: the MEAN loss of the part's samples; the step's loss is the sample-weighted mean of the parts (the gradient
      of the whole micro-batch's mean loss).  Shapes are static: every batch must have the example's shapes;
    * ``routing_fn(part) -> MokaRouting`` (or None for models without masked adapters): compiled from the batch's masks OUTSIDE the graph
      (one host read-back per part) and copied into the ``StaticRouting`` buffers the captured launches point at; inside the captured
      forward the adapters take that routing whatever masks they are handed (``routing.use_routing``);
    * ``chains > 1``: the chains are captured one after the other (the whole forward + backward of part 0, then part 1: an autograd pass
      cannot be interleaved), which only orders the HUB's launches; at replay the chains run side by side.  Beside a frozen base this does
      not pay (a streaming adapter launch next to a hipBLASLt GEMM slows the GEMM by more than it hides) and at the 7B widths the capture
      did not finish within minutes on ROCm 7.2: opt-in, tested on small stacks;
    * the AdamW coefficients are written by a live one-thread launch in front of every replay (``FlatAdamW.begin_step``), so learning-rate
      schedules work; gradient clipping does not exist in this mode (a bucket is updated before the global norm exists);
    * N > 1 (``attach`` under an initialised process group): collectives cannot ride inside a capture (RCCL crashes the runtime), so the graph
      ends with the LOCAL gradient and ONE all-reduce of the flat buffer + the fused AdamW + the shadow rewrite follow it live -- the
      gradient sum is not hidden behind the backward in this mode (the live ``attach`` path overlaps it bucket by bucket);
    * lora_dropout: the per-call seeds are drawn at capture time and frozen with the launch arguments; what varies per replay is the device
      word every dropout kernel folds into its seed (``moka_opts.seed_dev`` = ``dp.seed_epoch``), drawn from torch's CPU generator and
      written in front of every replay: fresh keep masks every step, the same ones in the step's forward and backward.
    dp = None: the same capture around a model without ``attach`` (no optimizer, no hub work): what ``bench.py --e2e`` times the frozen base with."""

    def __init__(self, dp, step_fn: Callable, example_batch, chains: int = 1, routing_fn: Optional[Callable] = None,
                 key_capacity: Optional[int] = None, warmup: int = 2, chain_priority: str = "normal", device=None):
        from .routing import StaticRouting
        self.dp, self.step_fn, self.routing_fn = dp, step_fn, routing_fn
        ts = _tensors(example_batch)
        if not ts:
            raise ValueError("GraphedTrainStep: the example batch holds no tensors")
        B = ts[0].shape[0]
        self.n = max(1, min(int(chains), B))
        self.sizes = [B // self.n + (1 if c < B % self.n else 0) for c in range(self.n)]
        self.weights = [sz / float(B) for sz in self.sizes]
        self.dev = torch.device(device) if device is not None else ts[0].device
        if self.dev.type != "cuda":
            raise _lib.MokaError("GraphedTrainStep captures HIP streams: the batch lives on %s" % self.dev)
        if dp is not None:
            if dp.optimizer is None:
                raise ValueError("GraphedTrainStep needs attach(..., optimizer=True): the fused AdamW slices are part of the captured step")
        # N > 1: collectives cannot be captured (RCCL inside a stream capture crashes the runtime), so the graph ends with the LOCAL gradient
        # in the flat buffer; the all-reduce of the whole buffer and the fused AdamW + shadow rewrite follow it live (three launches + one
        # collective per step: the gradient sum is not hidden behind the backward here -- 306 MB at the 7B widths, ~1-3 ms of a ~330 ms step)
        self.opt_in_graph = dp is not None and not dp.bucket.comm
        self.parts = [_clone_part(p) for p in _split_batch(example_batch, self.sizes)]
        self.rts = [StaticRouting(routing_fn(p), key_capacity) for p in self.parts] if routing_fn is not None else [None] * self.n
        self.pri = -1 if chain_priority == "high" else 0
        self.warmup = int(warmup)
        self.graph, self.loss, self._hc, self._reports, self.hub = None, None, None, {}, None
        self.capture()

    # ---- what AdapterDataParallel calls while this step is being captured
    def layer_done(self, dp, l: int) -> None:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))                 # (the chain's stream: its in-chain gradients of the layer are in front of it)
        evs = self._reports.setdefault(l, [])
        evs.append(ev)
        if self.opt_in_graph and len(evs) == self.n and dp.bucket.is_bucket_first(l):
            if self.hub != torch.cuda.current_stream(self.dev):      # (a one-chain capture is a single list: the reports are in stream order already)
                for e_ in evs:
                    self.hub.wait_event(e_)
            with torch.cuda.stream(self.hub):
                dp._opt_slice(*dp.bucket.bucket_bounds(l))             # (+ the bucket's weight shadows, behind every chain's dA_m of its layers)

    def _run_part(self, c: int):
        from .routing import use_routing
        with use_routing(self.rts[c]):
            loss = self.step_fn(self.parts[c])
            (loss * self.weights[c]).backward()
        return loss.detach() * self.weights[c]

    # ---- state a capture must not leave changed (its warm-up steps and its timed test replays are real optimizer steps)
    def _snapshot(self):
        dp = self.dp
        if dp is None:
            return None
        o = dp.optimizer
        return (dp.master.clone(), dp.work.clone(), o.exp_avg.clone(), o.exp_avg_sq.clone(), o.t)

    def _restore(self, snap) -> None:
        dp = self.dp
        if dp is None:
            return
        o = dp.optimizer
        with torch.no_grad():
            dp.master.copy_(snap[0])
            dp.work.copy_(snap[1])
            o.exp_avg.copy_(snap[2])
            o.exp_avg_sq.copy_(snap[3])
        o.t = snap[4]
        dp.bucket.zero_()
        dp.refresh_shadows()
        torch.cuda.synchronize(self.dev)

    def capture(self, tries: int = 3, accept: float = 1.05) -> None:
        """Warm up, capture, and CHECK the capture: ROCm 7.2 maps the lists of a captured graph (hub, chains) onto a few in-order hardware
        queues, and which ones depends on how many streams the process has created -- every few captures the hub shares a queue with a chain
        and every wait of the hub stalls that chain (measured: the same step 420 instead of 349 ms at the 7B widths, 46 instead of 30 ms for
        the adapter-only schedule).  So a capture is timed on two replays against the live step of the warm-up and redone (new streams, new
        mapping) up to `tries` times while it is more than `accept` x slower than that; the fastest capture is kept.  Model, optimizer state
        and step count are restored after every timed run."""
        import time
        best = None
        self.capture_log = []
        if self.dp is not None and not self.opt_in_graph:
            tries = 1                            # (N > 1: every timed replay is a collective; the ranks must not disagree on how many there are)
        for attempt in range(max(1, int(tries))):
            graph, hc, loss, live_ms = self._capture_once()
            self.graph, self._hc, self.loss = graph, hc, loss
            snap = self._snapshot()
            self.__call__(None)
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            for _ in range(2):
                self.__call__(None)
            torch.cuda.synchronize(self.dev)
            ms = (time.perf_counter() - t0) * 1e3 / 2
            self._restore(snap)
            self.capture_log.append({"attempt": attempt, "replay_ms": round(ms, 3), "live_ms": round(live_ms, 3)})
            if best is None or ms < best[0]:
                best = (ms, graph, hc, loss)
            if ms <= accept * live_ms:
                break
        self.replay_ms, self.graph, self._hc, self.loss = best
        torch.cuda.synchronize(self.dev)

    def _capture_once(self):
        import time
        dp, dev = self.dp, self.dev
        graph = torch.cuda.CUDAGraph()
        # ONE chain: one list.  A hub beside a single chain buys nothing under a frozen base (the streaming launches it would run beside the
        # chain cost the base's GEMMs more than they hide) and makes the replay depend on which hardware queues the runtime maps the two
        # lists onto (measured at the 7B widths: 369 / 411 ms per step from capture to capture, 342 live); a single list has no cross-list
        # wait and replays the same everywhere.
        hc = HubCapture(graph, 0 if self.n == 1 else self.n, dev, priority=self.pri)
        # ---- warm-up: the step live on the very streams of the capture (library handles and workspaces are per stream), state restored afterwards
        snap = self._snapshot()
        o = dp.optimizer if dp is not None else None
        torch.cuda.synchronize(dev)
        live_ms = float("inf")
        for _ in range(max(1, self.warmup)):
            t0 = time.perf_counter()
            for c in range(self.n):
                hc.branch[c].wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(hc.branch[c]):
                    if dp is not None and c < self.n - 1:
                        with dp.no_sync():
                            self._run_part(c)
                    else:
                        self._run_part(c)
                torch.cuda.current_stream(dev).wait_stream(hc.branch[c])
            if dp is not None:
                dp.step()
            torch.cuda.synchronize(dev)
            live_ms = min(live_ms, (time.perf_counter() - t0) * 1e3)
        self._restore(snap)
        # ---- capture
        self._reports = {}
        try:
            if dp is not None:
                dp._graph = self
                dp._opt_begun = True             # (the coefficients come from the live begin_step() in front of every replay)
                dp._opt_done.clear()
                if o._state is None:
                    o.set_device_step(o.t)
            with hc:
                self.hub = hc.cur
                losses = []
                for c in range(self.n):
                    with torch.cuda.stream(hc.branch[c]):
                        losses.append(self._run_part(c))
                        if dp is not None:
                            dp._flush_deferred()     # what the chain's last layers deferred
                if not hc.single:
                    for st in hc.branch:
                        hc.cur.wait_stream(st)
                if dp is not None and self.opt_in_graph:
                    # whatever no bucket hook covered (parameters outside the decoder stack, the first layer when its input carries no
                    # gradient): behind every chain, on the hub
                    done, pos, n_all = sorted(dp._opt_done), 0, dp.bucket.flat.numel()
                    for lo, hi in done + [(n_all, n_all)]:
                        if lo > pos:
                            dp._opt_slice(pos, lo)
                        pos = max(pos, hi)
                loss = torch.stack(losses).sum()
        finally:
            if dp is not None:
                dp._graph = None
                dp._opt_begun = False
                dp._opt_done.clear()
                dp._done.clear()
                dp._bwd_active = False
        torch.cuda.synchronize(dev)
        return graph, hc, loss, live_ms

    recapture = capture

    def __call__(self, batch=None) -> torch.Tensor:
        """Copy `batch` (None: keep the static inputs as they are) into the captured step's inputs, refresh the static routing, replay.
        Returns the step's loss (a tensor the next replay overwrites)."""
        if batch is not None:
            new = _split_batch(batch, self.sizes)
            for c in range(self.n):
                for dst, src in zip(_tensors(self.parts[c]), _tensors(new[c])):
                    if dst.shape != src.shape:
                        raise ValueError(f"GraphedTrainStep: a batch tensor is {tuple(src.shape)}, the captured step was built for {tuple(dst.shape)}")
                    dst.copy_(src, non_blocking=True)
                if self.rts[c] is not None:
                    self.rts[c].load(self.routing_fn(new[c]))
        dp = self.dp
        if dp is not None:
            if dp.seed_epoch is not None:
                from .functional import draw_seed
                dp.seed_epoch.fill_(draw_seed())         # the device part of every dropout seed: this replay's masks (torch.manual_seed controls it)
            if self.opt_in_graph:
                dp.optimizer.begin_step()                # this step's coefficients (lr / betas of NOW), a one-thread launch in front of the replay
        self.graph.replay()
        if dp is not None and not self.opt_in_graph:
            import torch.distributed as dist
            dist.all_reduce(dp.bucket.flat, group=dp.bucket.group)      # (stream-ordered behind the replay; the optimizer kernel averages)
            dp.optimizer.step(grad_scale=1.0 / dp.bucket.world, zero_grad=True)
            dp.refresh_shadows()
        return self.loss
