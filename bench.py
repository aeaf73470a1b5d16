#!/usr/bin/env python3
"""Benchmark of the MokA adapter hot path on MI355X (contract: see the task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--seq S]

One "step" = one pass of the hot path over one batch: adapter forward + backward of all
7 x 32 adapted projections of Llama-2-7B (r = 16, 3 modalities, AVT semantics) on B synthetic
sequences of 2048 tokens per GPU (SURVEY.md 8(d) layout: 16 text | 256 image | 16 text |
128 audio | 64 question | text), then the data-parallel step on the adapter gradients
(bucketed RCCL all-reduce of the flat fp32 gradient buffer on a side stream, launched as soon as
a group of layers has finished its backward and overlapped with the backward of the remaining
layers) and a fused AdamW update of the adapter parameters.  By default the micro-batch runs as TWO part-batch chains (half the sequences
each: nothing in the model mixes tokens of different samples) captured as branches of one hub-shaped hipGraph that share the parameters,
the gradient accumulators and the optimizer slices (--chains).  The frozen base GEMMs
are NOT part of the hot path (they run on stock PyTorch-ROCm); their outputs / input
gradients are the in/out operands of the kernels and are resident in HBM before the clock
starts.

Prints ONE JSON line on rank 0 (metric of BASELINE.json, `roofline` for the dominant kernel,
`cpu_baseline` = the oracle port on the host cores, bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import math
import os
import sys
import time
from ctypes import byref, c_float, c_void_p

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# kernel arguments in device memory instead of host-coherent memory: the documented launch-latency setting of the HIP runtime on
# MI300-class parts (read once, when the runtime initialises; 35.20 -> 35.10 ms per step here, more in the live-launch modes)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
LLAMA7B = dict(d=4096, ff=11008, kv=4096, layers=32)
# --model: the headline is Llama-2-7B; 13b = BASELINE.json configs[3] widths (run it with --rank 64 --seq 4096 --batch 2);
# 70b = configs[4] widths (grouped-query attention: k / v project 8192 -> 1024; zero_stage3_config_70b.json is about the FROZEN
# base, which fits one 288 GB GPU in bf16 -- the adapter path this bench times is the same pure data parallelism)
MODELS = {"7b": LLAMA7B, "13b": dict(d=5120, ff=13824, kv=5120, layers=40), "70b": dict(d=8192, ff=28672, kv=1024, layers=80)}
# the 7 adapted projections of one decoder layer in the reference's call order
# (AudioVisualText/models/modeling_llama.py:326-328,384,222-224): name, d_in, d_out, input id
PROJS = [("q_proj", "d", "d", "hid"), ("k_proj", "d", "kv", "hid"), ("v_proj", "d", "kv", "hid"),
         ("o_proj", "d", "d", "attn"), ("gate_proj", "d", "ff", "hid2"), ("up_proj", "d", "ff", "hid2"),
         ("down_proj", "ff", "d", "act")]
E = 2  # bytes per bf16


def algorithmic_bytes_per_token(dims, r, layers):
    """fwd E(d_in + 2 d_out + 2r), bwd E(d_out + 3 d_in + 3r) per projection."""
    fwd = bwd = 0
    for _, di, do, _ in PROJS:
        di, do = dims[di], dims[do]
        fwd += E * (di + 2 * do + 2 * r)
        bwd += E * (do + 3 * di + 3 * r)
    return fwd * layers, bwd * layers


def synthetic_layout(S):
    """[16 text][256 image/video][16 text][128 audio][64 question][rest text] per 2048 tokens.
    Returns (tok_mod int64[S] with 0 = text, 1 = image, 2 = audio; question bool[S])."""
    k = S / 2048.0
    n_pre, n_img, n_mid, n_aud, n_q = int(16 * k), int(256 * k), int(16 * k), int(128 * k), int(64 * k)
    tok = torch.zeros(S, dtype=torch.int64)
    q = torch.zeros(S, dtype=torch.bool)
    p = n_pre
    tok[p:p + n_img] = 1
    p += n_img + n_mid
    tok[p:p + n_aud] = 2
    p += n_aud
    q[p:p + n_q] = True
    return tok, q


class Unit:
    """The projections of one decoder layer that are fed by the same input (q/k/v; o; gate/up; down) with the
    ctypes argument lists of the six (grouped) entry points pre-built.  G = 1 is the per-projection path."""

    def __init__(self, label, members, T, r, M, rt, x, dx, scratch, s_in, s_out, w, c, drop_p, seeds, own_dh_kmj=None, fused=False, company=1):
        from moka_amd import _lib
        G = len(members)
        # moka_opts.company: how many independent chains run side by side (the pass over gy then sizes its token runs for its share of the CUs;
        # the dx pass of a wide input takes fewer, longer workgroups)
        self.opts = _lib.MokaOpts(None, 0, int(company))
        ob = byref(self.opts) if company > 1 else None
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
        # (--defer-da: the dA launches run a layer later, beside the next layer's chain: their operand packs cannot sit in the shared scratch)
        dh_kmj = P([(own_dh_kmj[g] if own_dh_kmj is not None else scratch[g]["dh_kmj"]) for g in range(G)])
        ws = P([rt.cross_ws(r, g) for g in range(G)])
        so = (c_float * M)(*s_out)
        sd = (ctypes.c_ulonglong * G)(*seeds)
        do = I(self.d_outs)
        tm = rt.tok_mod.data_ptr()
        self.keep = (members, A, dA, Bw, dB, y, h, hp_kmj, BwT, AT, part, hp_tok, dh_tok, dh_kmj, ws, so, sd, do, x, dx)
        # (--defer-da layer: the dA_m halves of a whole decoder layer as one moka_down_bwd_da_batch launch)
        self.da_items = [((own_dh_kmj[g] if own_dh_kmj is not None else scratch[g]["dh_kmj"]), x, self.d_in, members[g]["dA"], seeds[g]) for g in range(G)]
        self.sh_items = [(m["Bw"], m["d_out"], m["A"], self.d_in, m["BwT"], m["AT"]) for m in members]
        self.db_items = [(members[g]["y"], members[g]["hp_kmj"], members[g]["d_out"], members[g]["dB"]) for g in range(G)]
        self.calls = {
            "moka_down_fwd": ("moka_down_fwd_group", (x.data_ptr(), A, tm, part, T, self.d_in, r, M, G, s_in, drop_p, sd, 0)),
            "moka_cross_fwd": ("moka_cross_fwd_group", (part, ks_in, byref(rt.struct), so, Bw, do, A, self.d_in, h, None, hp_tok, hp_kmj,
                                                        BwT, AT, G, r, w, c)),
            "moka_up_fwd": ("moka_up_fwd_group", (hp_tok, Bw, tm, y, T, r, do, G, 0)),
            # --fuse-fwd (default): the up-projection computes the interaction itself from the slices (moka_up_fwd_fused); the rank-space
            # launch only writes what the BACKWARD reads (h, hp_kmj, BwT, AT: hp_tok = NULL) and leaves the dependency chain
            "moka_up_fwd:fused": ("moka_up_fwd_fused_group", (part, ks_in, byref(rt.struct), so, Bw, y, do, h, hp_kmj, G, r, w, c, 0)),
            "moka_cross_fwd:state": ("moka_cross_fwd_group", (part, ks_in, byref(rt.struct), so, Bw, do, A, self.d_in, h, None, None, hp_kmj,
                                                              BwT, AT, G, r, w, c)),
            # the weight shadows the backward reads (BwT, AT): functions of the weights alone -> once per step, off the chain
            "moka_weight_shadows": ("moka_weight_shadows_group", (Bw, do, A, self.d_in, BwT, AT, G, r, M)),
            "moka_up_bwd": ("moka_up_bwd_group", (y, hp_kmj, BwT, tm, so, part, dB, T, r, do, M, G, 0, ob)),
            # the two outputs of moka_up_bwd as separate calls (--defer-db: where dB is a pass of its own anyway, moka_up_bwd_passes() == 2,
            # it leaves the dependency chain like dA_m)
            "moka_up_bwd:g": ("moka_up_bwd_group", (y, hp_kmj, BwT, tm, so, part, None, T, r, do, M, G, 0, ob)),
            "moka_up_bwd:dB": ("moka_up_bwd_group", (y, hp_kmj, BwT, tm, so, None, dB, T, r, do, M, G, 0, ob)),
            "moka_cross_bwd": ("moka_cross_bwd_group", (part, ks_out, h, byref(rt.struct), s_in, None, dh_tok, dh_kmj, ws, G, r, w, c)),
            "moka_down_bwd": ("moka_down_bwd_group", (dh_tok, dh_kmj, x.data_ptr(), AT, tm, dA, dx.data_ptr(), T, self.d_in, r, M, G,
                                                      drop_p, sd, 0, ob)),
            # the two halves of moka_down_bwd as separate calls (either output may be NULL): dx stays on the dependency chain,
            # dA_m is needed by the optimizer only
            "moka_down_bwd:dx": ("moka_down_bwd_group", (dh_tok, dh_kmj, x.data_ptr(), AT, tm, None, dx.data_ptr(), T, self.d_in, r, M, G,
                                                         drop_p, sd, 0, ob)),
            "moka_down_bwd:dA": ("moka_down_bwd_group", (dh_tok, dh_kmj, x.data_ptr(), AT, tm, dA, None, T, self.d_in, r, M, G,
                                                         drop_p, sd, 0, None)),
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


def build_workload(args, dev, lib, bucket_factory, chains=1):
    """`chains` > 1: the micro-batch as that many part-batches (B / chains sequences each) with their own activations, routing,
    scratch and saved tensors, sharing the parameters and the gradient accumulators -- independent chains of launches (nothing in
    the model mixes tokens of different samples), see --chains."""
    from moka_amd import _lib
    from moka_amd.routing import MokaRouting
    vt = args.variant == "vt"
    B, S, r, M = args.batch, args.seq, args.rank, (2 if vt else 3)
    dims, L = MODELS[args.model], args.layers
    d, ff = dims["d"], dims["ff"]
    assert 1 <= chains <= B, "--chains: at most one chain per sequence"
    sizes = [B // chains + (1 if ci < B % chains else 0) for ci in range(chains)]      # (uneven splits: the larger part-batches first)
    T = B * S
    tok, q = synthetic_layout(S)
    if vt:
        # BASELINE.json configs[1]: visual-text -- the audio span becomes text, bool [B,S] masks (VisualText/train/train.py:206-231)
        tok = torch.where(tok == 2, torch.zeros_like(tok), tok)
    RP = _lib.rank_pad(r)
    bf, f32 = torch.bfloat16, torch.float32
    width = lambda k: dims[k]          # noqa: E731
    fused = getattr(args, "fuse_fwd", "off") == "on" and _lib.up_fwd_fused_ok(r)
    args.fused = fused

    # flat parameter / gradient buckets (fp32 master, bf16 working copy, fp32 grads)
    per_layer = sum(M * r * width(di) + r * width(do) for _, di, do, _ in PROJS)
    n_params = per_layer * L
    bucket = bucket_factory(n_params, [per_layer * (l + 1) for l in range(L)])
    gbuf = bucket.flat
    master = torch.empty(n_params, dtype=f32, device=dev)
    work = torch.empty(n_params, dtype=bf, device=dev)
    layer_end, params = [], []
    off = 0
    bound = lambda n: 1.0 / math.sqrt(n)  # noqa: E731  kaiming_uniform(a=sqrt(5))
    for l in range(L):
        row = []
        for pi, (name, di, do, src) in enumerate(PROJS):
            d_in, d_out = width(di), width(do)
            A, dA = [], []
            for m in range(M):
                n = r * d_in
                master[off:off + n].uniform_(-bound(d_in), bound(d_in))
                A.append(work[off:off + n].view(r, d_in))
                dA.append(gbuf[off:off + n].view(r, d_in))
                off += n
            n = d_out * r
            master[off:off + n].normal_(0, 0.02)
            row.append(dict(name=name, d_in=d_in, d_out=d_out, A=A, dA=dA, Bw=work[off:off + n].view(d_out, r), dB=gbuf[off:off + n].view(d_out, r)))
            off += n
        params.append(row)
        layer_end.append(off)
    assert off == n_params
    work.copy_(master)
    assert layer_end == bucket.layer_end

    # units = maximal runs of projections with the same input (--no-group: every projection alone)
    unit_defs = []
    for pi, (name, di, do, src) in enumerate(PROJS):
        if unit_defs and not args.no_group and unit_defs[-1][0] == src and len(unit_defs[-1][1]) < 3:
            unit_defs[-1][1].append(pi)
        else:
            unit_defs.append((src, [pi]))

    s = 16.0 / r
    nset = max(1, min(L, args.distinct))
    chain_list, keep = [], []
    shadow_bufs = {}              # (layer, projection) -> (BwT, AT): functions of the weights alone, so every chain reads the same pair
    for ci in range(chains):
        Bc = sizes[ci]
        Tc = Bc * S
        Tp = _lib.tok_pad(Tc)
        max_ks = max(_lib.ksplit(Tc, ff, r, 1), _lib.ksplit(Tc, d, r, 2), _lib.ksplit_bwd(Tc, ff, r))
        if vt:
            masks = [(tok == 0).reshape(1, S).repeat(Bc, 1).to(dev), (tok == 1).reshape(1, S).repeat(Bc, 1).to(dev), q.reshape(1, S).repeat(Bc, 1).to(dev)]
            rt = MokaRouting.from_vt_masks(*masks)
        else:
            masks = [(tok == m).to(torch.int32).reshape(1, S, 1).repeat(Bc, 1, 1).to(dev) for m in range(3)]
            masks.append(q.to(torch.int32).reshape(1, S, 1).repeat(Bc, 1, 1).to(dev))
            rt = MokaRouting.from_avt_masks(masks)
        # activation buffers: `args.distinct` layer sets cycled (all chains together: each set >> 256 MiB Infinity Cache).  Projections
        # fed by the same tensor (q/k/v <- hid, gate/up <- hid2) share ONE input and ONE input-gradient buffer, as in the
        # decoder (autograd sums their dx).
        sets = []
        for _ in range(nset):
            acts = {k: torch.randn(Tc, width(wk), device=dev, dtype=bf) for k, wk in (("hid", "d"), ("attn", "d"), ("hid2", "d"), ("act", "ff"))}
            dacts = {k: torch.randn(Tc, width(wk), device=dev, dtype=bf) for k, wk in (("hid", "d"), ("attn", "d"), ("hid2", "d"), ("act", "ff"))}
            ys = [torch.randn(Tc, width(do), device=dev, dtype=bf) for _, _, do, _ in PROJS]
            sets.append((acts, dacts, ys))
        # scratch shared by all units of the chain (consumed before the next unit overwrites it), one slot per group member
        # (two sets, alternating from unit to unit: with --fuse-fwd the state launch of unit u reads its slices on a side stream while
        #  unit u + 1 already writes its own)
        scratch2 = [[dict(part=torch.empty(max_ks, Tc, RP, dtype=f32, device=dev), hp_tok=torch.empty(Tp, 2 * RP, dtype=bf, device=dev),
                          dh_tok=torch.empty(Tp, 2 * RP, dtype=bf, device=dev), dh_kmj=torch.empty(M, 2, RP, Tp, dtype=bf, device=dev))
                     for _ in range(3)] for _ in range(2)]
        scratch = scratch2[0]
        units = []
        defer = getattr(args, "defer_da", "off") != "off"
        # (the deferred dA_m launches read a layer's packs a layer -- "bucket": a whole gradient bucket of layers -- later: 2 / L sets of them)
        n_own = L if (getattr(args, "defer_da", "off") == "bucket" or getattr(args, "hub", False)) else 2
        own = [[[torch.empty(M, 2, RP, Tp, dtype=bf, device=dev) for _ in pis] for _, pis in unit_defs] for _ in range(n_own)] if defer else None
        for l in range(L):
            acts, dacts, ys = sets[l % nset]
            members = []
            for pi, pr in enumerate(params[l]):
                # saved forward -> backward, per projection: h (fp32), the rank-major hp pack, the weight shadows
                if (l, pi) not in shadow_bufs:
                    shadow_bufs[(l, pi)] = (torch.empty(RP, pr["d_out"], dtype=bf, device=dev), torch.empty(M, pr["d_in"], RP, dtype=bf, device=dev))
                members.append(dict(pr, y=ys[pi], h=torch.empty(Tc, RP, dtype=f32, device=dev), hp_kmj=torch.empty(2, RP, Tp, dtype=bf, device=dev),
                                    BwT=shadow_bufs[(l, pi)][0], AT=shadow_bufs[(l, pi)][1]))
            for src, pis in unit_defs:
                mem = [members[pi] for pi in pis]
                units.append(Unit("+".join(m["name"].replace("_proj", "") for m in mem), mem, Tc, r, M, rt, acts[src], dacts[src], scratch2[len(units) & 1],
                                  1.0 if vt else s, [s] * M if vt else [1.0] * M, 0.05 if vt else 1.0, 1.0 / math.sqrt(r), args.dropout,
                                  [1000003 * l + pi + 7919 * 104729 * ci for pi in pis],      # every chain its own dropout masks
                                  own_dh_kmj=own[l % n_own][len(units) % len(unit_defs)] if defer else None, fused=fused,
                                  company=chains if getattr(args, "company_hint", "on") == "on" else 1))
        layer_da, layer_db = [], []
        if defer:
            per = len(unit_defs)
            for l in range(L):
                items = [it for u in reversed(units[l * per:(l + 1) * per]) for it in u.da_items]
                n = len(items)
                argl = ((c_void_p * n)(*[it[0].data_ptr() for it in items]), (c_void_p * n)(*[it[1].data_ptr() for it in items]),
                        (ctypes.c_int * n)(*[it[2] for it in items]), rt.tok_mod.data_ptr(),
                        (c_void_p * (n * M))(*[a.data_ptr() for it in items for a in it[3]]), n, Tc, r, M, args.dropout,
                        (ctypes.c_ulonglong * n)(*[it[4] for it in items]), 0, None)
                layer_da.append(argl)
                dbi = [it for u in reversed(units[l * per:(l + 1) * per]) for it in u.db_items]
                layer_db.append(((c_void_p * n)(*[it[0].data_ptr() for it in dbi]), (c_void_p * n)(*[it[1].data_ptr() for it in dbi]),
                                 (ctypes.c_int * n)(*[it[2] for it in dbi]), rt.tok_mod.data_ptr(), (c_void_p * n)(*[it[3].data_ptr() for it in dbi]),
                                 n, Tc, r, M, 0, None))
        chain_list.append(dict(units=units, units_per_layer=len(unit_defs), rt=rt, T=Tc, layer_da=layer_da, layer_db=layer_db, rank=r, reuse_wait=(n_own < L)))
        keep.append((sets, masks, scratch2, own))
    return dict(units=chain_list[0]["units"], units_per_layer=len(unit_defs), rt=chain_list[0]["rt"], layer_da=chain_list[0]["layer_da"], layer_db=chain_list[0]["layer_db"], rank=r, chains=chain_list, master=master, work=work, reuse_wait=chain_list[0]["reuse_wait"],
                gbuf=gbuf, bucket=bucket, T=T, n_params=n_params, layer_end=layer_end, keep=keep)


ENTRY = ["moka_down_fwd", "moka_cross_fwd", "moka_up_fwd", "moka_up_bwd", "moka_cross_bwd", "moka_down_bwd", "moka_weight_shadows"]


LIVE = ("moka_up_fwd",)     # the dominant single-kernel entry point, bracketed inside the timed region


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


def _call(lib, name, u, sp, rec, stream=None):
    """Launch one entry point of unit `u`; bracket it with HIP events (on `stream`, default: torch's current stream, which is the
    launch stream of the bracketed passes) when the recorder asks for it.  "entry:variant" is recorded as "entry"."""
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
        raise RuntimeError(lib.moka_last_error().decode())


def run_forward(lib, wl, sp, rec=None, shadows=False):
    """Per unit: down-projection, interaction, up-projection.  Fused units (--fuse-fwd): down-projection -> up-projection with the
    interaction inside (it also writes h and the rank-major hp pack for the backward).  The weight shadows the backward reads (BwT, AT)
    are functions of the weights alone: they are rewritten where the weights change (run_shadows behind the optimizer step), not in
    the forward -- unless `shadows` asks for them in front of every unit (--shadows main)."""
    for u in wl["units"]:
        if not u.fused:
            _call(lib, "moka_down_fwd", u, sp, rec)
            _call(lib, "moka_cross_fwd", u, sp, rec)       # (writes its own weight shadows: taking them out gained nothing at rank 64, 82.3 vs 84.1 ms)
            _call(lib, "moka_up_fwd", u, sp, rec)
            continue
        if shadows:
            if u.fused:                      # (the other units' moka_cross_fwd writes their shadows in the forward)
                _call(lib, "moka_weight_shadows", u, sp, rec)
        _call(lib, "moka_down_fwd", u, sp, rec)
        _call(lib, "moka_up_fwd:fused", u, sp, rec)


SHADOWS_BATCH = True          # --shadows-batch off: one moka_weight_shadows_group launch per unit (A/B)


def run_shadows(lib, wl, sp, layers, rec=None):
    """BwT / AT of the given layers' FUSED units (the other units' moka_cross_fwd writes theirs in the forward; all chains share the
    parameters: the first chain's units carry the buffers): one moka_weight_shadows_batch launch per 16 projections (a recorder gets
    the per-unit launches, so that the entry point keeps its line in the table)."""
    units, per = wl["units"], wl["units_per_layer"]
    if rec is not None or not SHADOWS_BATCH:
        for l in layers:
            for u in units[l * per:(l + 1) * per]:
                if u.fused:
                    _call(lib, "moka_weight_shadows", u, sp, rec)
        return
    from moka_amd import _lib as _L
    key = ("shadows", tuple(layers))
    if key not in wl:
        items = [it for l in layers for u in units[l * per:(l + 1) * per] if u.fused for it in u.sh_items]
        calls = []
        for i in range(0, len(items), _L.MOKA_MAX_SHADOW_BATCH):
            part = items[i:i + _L.MOKA_MAX_SHADOW_BATCH]
            n, M = len(part), len(part[0][2])
            calls.append(((c_void_p * n)(*[it[0].data_ptr() for it in part]), (ctypes.c_int * n)(*[it[1] for it in part]),
                          (c_void_p * (n * M))(*[a.data_ptr() for it in part for a in it[2]]), (ctypes.c_int * n)(*[it[3] for it in part]),
                          (c_void_p * n)(*[it[4].data_ptr() for it in part]), (c_void_p * n)(*[it[5].data_ptr() for it in part]), n, wl["rank"], M))
        wl[key] = calls
    for argl in wl[key]:
        _L.check(lib.moka_weight_shadows_batch(*argl, sp), "moka_weight_shadows_batch")


def run_backward(lib, wl, sp, n_layers, on_layer_done=None, rec=None, lo=0, defer=None, bucket_opt=None, shadows_after_opt=False, state=None, join=True,
                 flush=False):
    """Reverse layer order (layers n_layers-1 .. lo); `on_layer_done(l)` fires after layer l's launches are enqueued.
    defer = (mode, main_stream, side_stream): the dA_m halves of a layer's moka_down_bwd calls leave the dependency chain (only the
    optimizer needs them) and are enqueued after the layer's chain -- "main": on the same stream; "side": on a second stream,
    beside the NEXT layer's chain (whose rank-space kernels leave most of the chip idle), joined before the gradients are used.
    state / join: the walk in pieces (--chains captures the chains layer by layer, interleaved): `state` carries the buffer-reuse events
    from call to call, join=False leaves the side stream unjoined."""
    units, per = wl["units"], wl["units_per_layer"]
    if defer is None:
        for l in range(n_layers - 1, lo - 1, -1):
            for u in reversed(units[l * per:(l + 1) * per]):
                _call(lib, "moka_up_bwd", u, sp, rec)
                _call(lib, "moka_cross_bwd", u, sp, rec)
                _call(lib, "moka_down_bwd", u, sp, rec)
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
    held = []                                                    # layers whose deferred launches wait for the bucket's flush
    # CHAIN_FIRST (captures only): a layer's side-stream launches are enqueued AFTER the first launch of the next layer's chain.  The DAG is
    # the same; what changes is the order of a fork node's out-edges, and the hipGraph executor (ROCm 7.2) derives its execution streams
    # from a depth-first walk that follows the FIRST out-edge: side-first lets the walk leave the chain at every fork
    reuse = wl.get("reuse_wait", True) and flush_at is None      # (pack buffers of layer l + 2 reused by layer l: the chain waits for that dA)
    late = CHAIN_FIRST and on_layer_done is None and mode == "side"
    post = done.pop("post", None) if late else None              # (layer, held layers, event on main) of the fork not yet emitted

    def emit(l, held_, ev_main, pending_u):
        side.wait_event(ev_main)
        for u in reversed(units[l * per:(l + 1) * per]):
            if split_db and not batched and not per_unit:
                _call(lib, "moka_up_bwd:dB", u, sps, None)
            if batched or per_unit or (mode == "window" and u is not pending_u):
                continue                                        # (already out, beside the next unit's rank-space backward)
            _call(lib, "moka_down_bwd:dA", u, sps, None)
        if batched:
            from moka_amd import _lib as _L
            for ll in held_ + [l]:
                if split_db:
                    _L.check(lib.moka_up_bwd_db_batch(*wl["layer_db"][ll], sps), "moka_up_bwd_db_batch")
                _L.check(lib.moka_down_bwd_da_batch(*wl["layer_da"][ll], sps), "moka_down_bwd_da_batch")
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
                        run_shadows(lib, wl, sps, bucket_.bucket_layers(l))
        ev = torch.cuda.Event()
        ev.record(side)
        done[l] = ev

    for l in range(n_layers - 1, lo - 1, -1):
        if not late and reuse and mode in ("side", "window") and (l + 2) in done:
            main.wait_event(done.pop(l + 2))                     # layer l reuses the pack buffers of layer l + 2
        pending = None
        for u in reversed(units[l * per:(l + 1) * per]):
            _call(lib, up, u, sp, rec)
            if post is not None:
                emit(*post)                                      # the layer before's fork, behind this layer's first launch (CHAIN_FIRST)
                post = None
            if late and pending is None and reuse and (l + 2) in done:
                main.wait_event(done.pop(l + 2))                 # (the first writer of the reused pack buffers is this unit's rank-space backward)
            if mode == "window" and pending is not None:
                # the dA of the unit before goes out HERE, so that it starts with this unit's rank-space backward -- the two launches of
                # the chain that leave the memory system idle (a unit's dA moves about as many bytes as that window could)
                side.wait_stream(main)
                _call(lib, "moka_down_bwd:dA", pending, sps, None)
            _call(lib, "moka_cross_bwd", u, sp, rec)
            if per_unit:
                ev_u = torch.cuda.Event()
                ev_u.record(main)
            _call(lib, "moka_down_bwd:dx", u, sp, None)
            if per_unit:
                side.wait_event(ev_u)
                if split_db:
                    _call(lib, "moka_up_bwd:dB", u, sps, None)
                _call(lib, "moka_down_bwd:dA", u, sps, None)
            pending = u
        if mode == "main":
            for u in reversed(units[l * per:(l + 1) * per]):
                if split_db:
                    _call(lib, "moka_up_bwd:dB", u, sp, None)
                _call(lib, "moka_down_bwd:dA", u, sp, None)
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


CHAIN_FIRST = True            # --capture-order side-first: A/B


# roofline.traffic: HBM bytes per launch of the dominant kernel from the PMC counters -- read from the committed summary of the
# PMC passes of THIS build (tools/pmc_traffic.sh -> profiles/r04_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE
# in separate passes; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, WRITE_SIZE as is), never a constant in here.
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r05_pmc_traffic.json")


def kernel_source_sha256():
    """sha256 over the kernel source + the C header: what a PMC traffic summary is stamped with (tools/pmc_traffic.py)."""
    import hashlib
    h = hashlib.sha256()
    for rel in ("moka_amd/csrc/moka_kernels.hip", "include/moka_hip.h"):
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()


def pmc_traffic_per_launch(T, launches_per_layer):
    """Average HBM bytes per launch of the dominant kernel, scaled to T tokens.  A missing summary leaves the field null (and says so):
    the traffic is measured, never assumed -- and a measured run is never thrown away for it (tests/test_bench_line.py checks that the
    file is committed)."""
    if not os.path.exists(PMC_TRAFFIC_FILE):
        print(f"bench: {PMC_TRAFFIC_FILE} is missing -- run tools/pmc_traffic.sh on the GPU box and commit its pmc_traffic.json there; "
              "roofline.traffic is null in this line", file=sys.stderr)
        return None, "missing: " + os.path.relpath(PMC_TRAFFIC_FILE, ROOT)
    d = json.load(open(PMC_TRAFFIC_FILE))
    rel = os.path.relpath(PMC_TRAFFIC_FILE, ROOT)
    # the summary must have been measured on THESE kernels: it carries the sha256 of the source it was built from, and the library that is
    # loaded must not be older than that source (a kernel change without a re-profile would leave a stale figure on the line)
    from moka_amd import _lib, build as _build
    if d.get("kernel_source_sha256") != kernel_source_sha256():
        print(f"bench: {rel} was measured on other kernels (kernel_source_sha256 differs) -- run tools/pmc_traffic.sh on this build; "
              "roofline.traffic is null in this line", file=sys.stderr)
        return None, "stale: %s was measured on another kernel source" % rel
    if os.path.abspath(_lib.LIB_PATH) == os.path.abspath(_build.OUT) and _build.needs_build():
        print("bench: libmoka_hip.so is older than its source; roofline.traffic is null in this line", file=sys.stderr)
        return None, "stale: the loaded library is older than the kernel source"
    return d["traffic_bytes_per_layer"] * (T / float(d["tokens"])) / launches_per_layer, rel


def usable_cpus() -> int:
    """CPUs this process may really use: min(affinity, cgroup quota).  (The GPU boxes expose 256 logical
    CPUs but run the job under a 16-CPU cgroup quota; 256 OpenMP threads on 16 CPUs is ~60x slower.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, n)


def cpu_model_string() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(args):
    """The oracle port (the CPU restatement of the reference layer, verified equal to it in the build container) on the
    host cores the job may use, on a bounded sample: adapter fwd+bwd of ONE decoder layer's 7 projections for ONE
    sequence, fp32 (the headline value, scaled to the layer count) and bf16 operands, with the per-projection split
    (BASELINE.md section 3).  Baseline only -- never the thing measured or shipped."""
    from oracle import cases as C
    from oracle import moka_oracle as O
    S, r = args.seq, args.rank
    d, ff = MODELS[args.model]["d"], MODELS[args.model]["ff"]
    cores = usable_cpus()
    torch.set_num_threads(cores)
    tok, q = C.build_layout(C.synthetic_sequence_layout(S), S)
    masks = [(tok == m).to(torch.int32).reshape(1, S, 1) for m in range(3)] + [q.to(torch.int32).reshape(1, S, 1)]
    rt = O.routing_from_avt_masks(masks)
    g = torch.Generator().manual_seed(1)
    data = []
    for name, di, do, _ in PROJS:
        d_in, d_out = MODELS[args.model][di], MODELS[args.model][do]
        data.append((name, torch.randn(1, S, d_in, generator=g), torch.randn(1, S, d_out, generator=g),
                     [torch.randn(r, d_in, generator=g) * 0.01 for _ in range(3)], torch.randn(d_out, r, generator=g) * 0.02,
                     torch.randn(1, S, d_out, generator=g)))

    def one_layer(dtype, per=None):
        for name, x, y0, A, Bw, gy in data:
            t0 = time.perf_counter()
            y, ctx = O.adapter_forward(x, y0, A, Bw, rt, 16.0 / r, [1.0] * 3, 1.0, r, dtype=dtype)
            O.adapter_backward(gy, ctx)
            if per is not None:
                per[name] = per.get(name, 0.0) + (time.perf_counter() - t0)

    def timed(dtype, budget):
        # median of >= 10 repeats after >= 3 warm-ups when the budget allows it; the budget bounds the run
        # on slow hosts (the fp32 layer takes ~0.7 s on 16 cores), the repeat count achieved is reported
        t_w = time.perf_counter()
        n_w = 0
        for _ in range(3):
            one_layer(dtype)
            n_w += 1
            if time.perf_counter() - t_w > 0.25 * budget:
                break
        per, laps = {}, []
        t0 = time.perf_counter()
        while True:
            t1 = time.perf_counter()
            one_layer(dtype, per)
            laps.append(time.perf_counter() - t1)
            if (time.perf_counter() - t0 > budget and len(laps) >= 3) or len(laps) >= 50:
                break
            if len(laps) >= 10 and time.perf_counter() - t0 > 0.6 * budget:
                break
        laps.sort()
        n = len(laps)
        med = laps[n // 2] if n % 2 else 0.5 * (laps[n // 2 - 1] + laps[n // 2])
        return med, n, {k: round(v / n * 1e3, 2) for k, v in per.items()}, n_w

    per_layer, n, split, n_w = timed(torch.float32, args.cpu_seconds * 0.7)
    per_layer_bf, n_bf, split_bf, _ = timed(torch.bfloat16, args.cpu_seconds * 0.3)
    return {"value": S / (per_layer * args.layers), "unit": "tokens/s", "cores": cores, "kind": "port", "cpu": cpu_model_string(),
            "sample": f"oracle port (torch fp32), adapter fwd+bwd of 1 decoder layer x 7 projections, 1 sequence of {S} tokens, "
                      f"median of {n} repeats after {n_w} warm-ups, scaled x{args.layers} layers",
            "per_projection_ms_fp32": split,
            "bf16": {"value": S / (per_layer_bf * args.layers), "repeats": n_bf, "per_projection_ms": split_bf}}


def end_to_end(args, dev):
    """Context, not the metric: fwd+bwd of the WHOLE decoder stack (frozen bf16 base on hipBLASLt, stock SDPA / RMSNorm /
    rotary, the seven adapted projections of every layer on the HIP path through moka_amd/decoder.py) against the same
    stack with plain frozen projections, same tokens per GPU.  Embedding table, encoders and LM head are out of scope."""
    from moka_amd.decoder import LlamaDims, MokaLlamaStack
    from moka_amd.peft_hyper import Linear
    B, S, r, L = args.batch, args.seq, args.rank, args.layers
    dims = LlamaDims(hidden=MODELS[args.model]["d"], ff=MODELS[args.model]["ff"], n_heads=MODELS[args.model]["d"] // 128, n_kv_heads=MODELS[args.model]["d"] // 128)
    bf = torch.bfloat16
    tok, q = synthetic_layout(S)
    masks = [(tok == m).to(torch.int32).reshape(1, S, 1).repeat(B, 1, 1).to(dev) for m in range(3)]
    masks.append(q.to(torch.int32).reshape(1, S, 1).repeat(B, 1, 1).to(dev))
    h = torch.randn(B, S, dims.hidden, device=dev, dtype=bf)
    gout = torch.randn(B, S, dims.hidden, device=dev, dtype=bf)

    class Plain(torch.nn.Linear):
        def forward(self, x, *m):
            return super().forward(x)

    def adapted(d_in, d_out):
        m = Linear(d_in, d_out, r=(r, r, r), lora_alpha=16, lora_nums=3, blc_weight=1.0, blc_alpha=1, lora_dropout=args.dropout,
                   loramethod="train", bias=False)
        torch.nn.init.normal_(m.weight, std=0.02)
        torch.nn.init.normal_(m.lora_B0.weight, std=0.02)
        return m

    def plain(d_in, d_out):
        m = Plain(d_in, d_out, bias=False)
        torch.nn.init.normal_(m.weight, std=0.02)
        m.weight.requires_grad = False
        return m

    def timed(make, train_adapter):
        old = torch.get_default_dtype()
        torch.set_default_dtype(bf)
        try:
            with torch.device(dev):
                st = MokaLlamaStack(dims, L, make)
        finally:
            torch.set_default_dtype(old)
        st.train()
        for n, p_ in st.named_parameters():
            p_.requires_grad = train_adapter and "lora_" in n
        params = [p_ for p_ in st.parameters() if p_.requires_grad]
        opt = torch.optim.AdamW(params, lr=1e-4, fused=True) if params else None
        x = h.clone().requires_grad_(True)            # dx reaches the embeddings / projector in the real model

        def step():
            out, _ = st(x, masks)
            out.backward(gout)
            if opt is not None:
                opt.step()
                opt.zero_grad(set_to_none=False)
            x.grad = None

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / n
        del st, opt, params
        torch.cuda.empty_cache()
        return ms

    from moka_amd import functional as MF
    ms_base = timed(plain, False)
    ms_moka = timed(adapted, True)
    # the adapter's x-only / gy-only halves on a side stream beside the frozen base GEMM of the same projection (functional.OVERLAP_BASE,
    # attach(overlap_base=True)): same kernels, same bits
    MF.set_overlap_base(True)
    try:
        ms_ovl = timed(adapted, True)
    finally:
        MF.set_overlap_base(False)
    return {"what": "decoder stack fwd+bwd (+ fused AdamW on the adapter), %d layers, %d x %d tokens, bf16; NOT the metric" % (L, B, S),
            "ms_per_step": round(ms_moka, 2), "tokens_per_s": round(B * S / (ms_moka * 1e-3), 1),
            "frozen_base_only_ms_per_step": round(ms_base, 2), "adapter_share_of_step": round(1.0 - ms_base / ms_moka, 4),
            "overlap_base": {"ms_per_step": round(ms_ovl, 2), "adapter_share_of_step": round(1.0 - ms_base / ms_ovl, 4)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120, help="timed steps (default: ~4 s of GPU work, long enough for an external busy sampler to see it)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4, help="sequences per GPU (reference AVT micro-batch: ft_musicavqa.sh:12-13)")
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--model", choices=tuple(MODELS), default="7b", help="widths / depth of the decoder (the metric is quoted on 7b)")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--distinct", type=int, default=4, help="distinct activation buffer sets cycled over the layers")
    ap.add_argument("--dropout", type=float, default=0.05, help="lora_dropout (both reference scripts train with 0.05)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true")
    ap.add_argument("--variant", choices=("avt", "vt"), default="avt",
                    help="avt: 3 modalities, the headline workload; vt: 2 modalities (BASELINE.json configs[1], 256 image tokens + text)")
    ap.add_argument("--e2e", action="store_true",
                    help="also time the whole decoder stack (frozen base + adapters) through moka_amd/decoder.py and report it as "
                         "`end_to_end` (context only; the metric stays the adapter path)")
    ap.add_argument("--graph", choices=("auto", "off", "bwd", "all"), default="auto",
                    help="hipGraph replay (the library only enqueues on the stream it is given, so its launches capture unchanged): "
                         "all = the whole micro-batch as one graph (single GPU; nothing can be bracketed inside a graph, so `roofline` comes "
                         "from extra live passes after the timed region); bwd = the forward as one graph and one graph per gradient bucket of the "
                         "backward (DP hooks between the graphs); off = every launch live; auto = all on 1 GPU, bwd on N > 1")
    ap.add_argument("--bracket-every", type=int, default=5,
                    help="bracket every n-th launch of the dominant kernel with HIP events inside the timed region (an event record is a "
                         "packet of its own: bracketing all 128 launches of a step costs 0.7 ms of it; 5 is coprime to the 4 unit shapes "
                         "of a layer, so the sample covers them evenly)")
    ap.add_argument("--comm-bf16", action="store_true", help="all-reduce the gradient buckets as bf16 (153 instead of 306 MB per step at 7B r=16; accumulation stays fp32)")
    ap.add_argument("--buckets", type=int, default=0,
                    help="gradient buckets (groups of whole decoder layers, each all-reduced / updated as soon as its layers' backward has been enqueued); "
                         "with collectives every bucket is a hipGraph of its own: fewer buckets = fewer points at which the chains meet.  0 (default) = "
                         "8 equal buckets without collectives, the geometric layout 1 / 3 / 9 / 19 layers (parallel.geometric_buckets) with them")
    ap.add_argument("--tail-layers", type=int, default=-1,
                    help="layers of the gradient bucket that ships last (it holds layer 0: its all-reduce has nothing left to hide behind); the other "
                         "layers split evenly over the remaining buckets (with --buckets N).  -1 (default) / 0: no tail bucket (the default layout with collectives is geometric)")
    ap.add_argument("--no-traffic", action="store_true", help="leave roofline.traffic null instead of reading the PMC summary under profiles/")
    ap.add_argument("--chains", type=int, default=0,
                    help="process the micro-batch as this many part-batches (batch / chains sequences each) whose launch chains are branches of "
                         "the captured graph(s): nothing in the model mixes tokens of different samples, so the chains are independent, and the "
                         "fixed costs of one (kernel boundaries, ramps, the latency-bound rank-space kernels) hide behind the streaming kernels "
                         "of the other; they share the parameters, the gradient accumulators and the optimizer slices.  0 (default) = 2 where the "
                         "batch splits evenly and the step is replayed as graphs, else 1.  Per-kernel durations (`roofline`, `kernels`) are "
                         "taken with the chains back to back on one stream")
    ap.add_argument("--company-hint", choices=("on", "off"), default="on",
                    help="with chains: tell the library how many chains run side by side (moka_opts.company: the pass over gy sizes its token runs for its "
                         "share of the CUs)")
    ap.add_argument("--chain-stagger", type=int, default=0,
                    help="MB of a fill launched in front of the second (third, ...) chain's forward: a phase shift between otherwise identical chains (A/B)")
    ap.add_argument("--defer-da", choices=("auto", "off", "main", "side", "window", "layer", "bucket", "unit"), default="auto",
                    help="the dA_m halves of moka_down_bwd are needed by the optimizer only: layer (default, what moka_amd.parallel.attach does) = "
                         "a layer's worth of them goes out as ONE launch (moka_down_bwd_da_batch: 4 -> 1 launches per layer) on a second stream when "
                         "the layer's chain has been enqueued, and runs beside the next layer's chain (joined before a gradient bucket ships and before "
                         "the optimizer step); side = the same schedule with one launch per unit (round 3); main = those launches on the one stream; "
                         "off = dA_m and dx from one moka_down_bwd call inside the chain; window = a unit's dA_m forked behind the NEXT unit's pass over gy, "
                         "so that it starts with that unit's rank-space backward, the window in which the chain leaves the memory system idle (live "
                         "launches: 35.96 -> 35.16 ms; inside the hipGraph a fork per unit makes the replay host-bound: 51 ms -- an experiment, not a default)")
    ap.add_argument("--opt-in-backward", choices=("on", "off"), default="on",
                    help="the fused AdamW step per gradient bucket inside the backward (behind the bucket's deferred dA / its all-reduce) instead of one launch behind it")
    ap.add_argument("--chain-priority", choices=("auto", "high", "normal"), default="auto",
                    help="stream priority of the captured dependency chain (the deferred dA / dB stream stays at normal priority).  auto = high in the "
                         "one-graph mode (33.5 -> 33.35 ms), normal in --graph bwd: there every bucket graph ends with the chain joining the side "
                         "stream, and a high-priority chain starves the launches it then has to wait for (one GPU: 49.0 against 34.6 ms per step)")
    ap.add_argument("--defer-db", choices=("auto", "on", "off"), default="auto",
                    help="with --defer-da: dB also leaves the dependency chain (auto: where moka_up_bwd_passes() says dB is a pass of its own, r > 32)")
    ap.add_argument("--force-comm", action="store_true",
                    help="single GPU: initialise a ONE-rank RCCL process group and run the gradient collectives through it (FlatGradBucket(force_comm=True)): "
                         "the N > 1 configuration -- --graph bwd, bucket hooks between the graphs, the AdamW slices on the communication stream behind each "
                         "all-reduce -- priced on one GPU (`comm_exposed_ms`), the figure the first multi-GPU run is read against")
    ap.add_argument("--fuse-fwd", choices=("on", "off"), default="on",
                    help="on (default, r <= 32): the up-projection computes the cross-modal interaction itself (moka_up_fwd_fused) and the rank-space "
                         "launch, which then only writes the backward's operands, runs on a side stream off the dependency chain; off: three launches per unit")
    ap.add_argument("--shadows-batch", choices=("on", "off"), default="on",
                    help="with --shadows opt: the shadows of up to 16 projections per launch (moka_weight_shadows_batch; default) or one launch per unit")
    ap.add_argument("--shadows", choices=("opt", "main"), default="opt",
                    help="fused units: where BwT / AT (functions of the weights alone, read by the backward) are written: opt = where the weights change, "
                         "behind the optimizer update (moka_weight_shadows per gradient bucket on the side / communication stream with --opt-in-backward); "
                         "main = in front of every fused unit on the forward's chain (three-launch units: inside moka_cross_fwd)")
    ap.add_argument("--graph-topology", choices=("auto", "hub", "chain"), default="auto",
                    help="shape of the one captured graph: hub (default) = the chain(s) on forked streams, everything off the chains (deferred dA_m, optimizer "
                         "slices, weight shadows) on the capture's origin stream, no edge from the hub back into a chain (every layer owns its pack "
                         "buffers) -- the hipGraph executor then runs 1 + chains lists; chain = round 4's shape (the chain on the origin, a side stream "
                         "forked and joined per layer)")
    ap.add_argument("--capture-order", choices=("chain-first", "side-first"), default="chain-first",
                    help="order in which a fork's two successors are captured (same DAG): the hipGraph executor follows a fork node's FIRST out-edge "
                         "when it cuts the graph into execution streams")
    ap.add_argument("--verify-graph", action="store_true",
                    help="with --no-optimizer: replay the captured step once and run the same launches live, chain after chain on one stream, from the "
                         "same activation state; y / dx must agree bit for bit, the flat gradient to its atomics' spread -> `graph_check`")
    ap.add_argument("--probe-forward", action="store_true",
                    help="also time the forward alone as a hipGraph of its own against the same launches live (HIP events, no profiler) -> `forward_only`")
    ap.add_argument("--no-group", action="store_true",
                    help="launch every projection on its own (the grouped entry points let q/k/v and gate/up share x / dx)")
    args = ap.parse_args()
    if args.layers is None:
        args.layers = MODELS[args.model]["layers"]
    if args.graph == "auto":
        args.graph = "all" if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.force_comm) else "bwd"
    if args.chains == 0:
        # auto: two part-batch chains where the step is replayed as hipGraphs, the batch has two sequences and the launches are short enough to
        # leave gaps (7B widths, rank pad <= 32: 32.2 -> 30.5 ms at r = 16, 41.7 -> 40.9 at r = 32; the 70B widths lose, 160.3 -> 166.1 ms, and
        # so does rank 64 at the 13B widths, 78.6 -> 79.5: their launches fill the chip on their own)
        # (13B widths: r = 16 54.9 -> 48.9 ms, r = 32 63.8 -> 61.2 ms)
        args.chains = 2 if (args.graph != "off" and args.batch >= 2 and args.model in ("7b", "13b") and args.rank <= 32) else 1
    if args.graph_topology == "auto":
        args.graph_topology = "hub" if args.chains > 1 else "chain"      # (one chain: 32.15-32.27 ms in round 4's shape, 32.39 as hub + 1 chain)
    if args.defer_da == "auto":
        # one chain: a layer's dA_m as ONE launch (round 4); two chains: one launch per unit, out as soon as the unit's rank-space backward
        # has been enqueued -- the hub's launches are then short enough to weave between the chains' (same box, per-layer launch / per-unit
        # launches at the layer's end / per unit at once: 30.94 / 30.86 / 30.76 ms; five more pairs layer vs side: 30.65-31.14 vs 30.41-30.84)
        args.defer_da = "unit" if (args.chains > 1 and args.graph != "off") else "layer"
    args.hub = args.graph != "off" and (args.graph_topology == "hub" or args.chains > 1)
    if args.chain_priority == "auto":
        args.chain_priority = "high" if args.graph == "all" else "normal"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become N ranks (one process per GPU) under torch.distributed.run, exactly
        # the command the reference launches with (VisualText/shell/train.sh:62, ft_musicavqa.sh:24: torchrun --nproc_per_node 8)
        import socket
        import subprocess
        backend = os.environ.get("MOKA_BENCH_BACKEND", "nccl")
        if backend == "nccl" and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible (one process per GPU over RCCL; "
                             "MOKA_BENCH_BACKEND=gloo shares a device for a functional check of the N > 1 path, never a measurement)")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.rank_id = rank
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    comm = world > 1 or args.force_comm              # do the gradient collectives run?
    if args.force_comm and world == 1:
        # the N > 1 configuration on ONE GPU: a one-rank RCCL communicator is a real ProcessGroupNCCL (its own stream, in-place
        # asynchronous all-reduce, wait() = stream wait), so --graph bwd, the bucket hooks and the optimizer slices behind the
        # all-reduce run exactly as the driver's multi-GPU launch runs them; what that configuration costs on one GPU is the
        # figure a scaling curve is read against
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group(os.environ.get("MOKA_BENCH_BACKEND", "nccl"), init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                **({"device_id": dev} if os.environ.get("MOKA_BENCH_BACKEND", "nccl") == "nccl" else {}))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl" on ROCm).  MOKA_BENCH_BACKEND=gloo is a functional check of the N > 1 script path on a box with fewer
        # GPUs than ranks (ranks share a device; gloo stages the gradient slices through the host) -- never a measurement.
        backend = os.environ.get("MOKA_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from moka_amd import _lib
    lib = _lib.load()
    _lib.check(lib.moka_device_check(), "moka_device_check")
    from moka_amd.parallel import FlatGradBucket
    if args.chains > 1 and args.defer_da == "window":
        raise SystemExit("--chains > 1 runs with --defer-da off / main / side / layer / bucket")
    # dB leaves the dependency chain with dA_m where the library computes it in a pass of its own anyway (r > 32)
    global SHADOWS_BATCH, CHAIN_FIRST
    SHADOWS_BATCH = args.shadows_batch == "on"
    CHAIN_FIRST = args.capture_order == "chain-first"
    args.split_db = args.defer_da != "off" and (args.defer_db == "on" or (args.defer_db == "auto" and lib.moka_up_bwd_passes(args.rank, 0) == 2))
    # collectives on: geometric buckets -- 1, 3, 9, 19 layers from layer 0 up.  The backward walks the layers last -> first: the big buckets
    # ship early with plenty of backward left to hide their all-reduce, the bucket nothing is left to hide is ONE layer, and there are four
    # points (not eight) at which the per-bucket graphs make the chains meet (one GPU, one-rank RCCL: 32.0-32.5 -> 31.3 ms)
    from moka_amd.parallel import geometric_buckets
    tail = args.tail_layers if args.tail_layers >= 0 else 0
    sizes = None
    if args.buckets == 0 and comm and tail == 0:
        sizes = geometric_buckets(args.layers)
    nb = args.buckets if args.buckets > 0 else 8
    wl = build_workload(args, dev, lib, lambda n, ends: FlatGradBucket(n, ends, dev, n_buckets=nb, comm_dtype=torch.bfloat16 if args.comm_bf16 else None,
                                                                       force_comm=args.force_comm, tail_layers=tail or None, bucket_sizes=sizes),
                        chains=args.chains)
    T = wl["T"]
    torch.cuda.synchronize()

    main_stream = torch.cuda.current_stream()
    bucket = wl["bucket"]
    opt = None
    if not args.no_optimizer:
        # AdamW on the flat buffers as ONE kernel of the library (moka_adamw_flat): gradient averaging, update of the fp32 master,
        # bf16 working copy for the next forward and zeroing of the gradient buffer in a single pass (34 B / parameter)
        from moka_amd.parallel import FlatAdamW
        opt = FlatAdamW(wl["master"], bucket.flat, wl["work"], lr=1e-4)
    L = args.layers
    # the optimizer step per gradient bucket INSIDE the backward (off: one launch behind it): needs the side stream of the deferred dA_m
    # (single GPU) or the communication stream behind the bucket's all-reduce (N > 1, fp32 payload)
    # (--chains N: every chain defers its dA_m to a side stream of its own; the slice of a bucket goes out on one more stream once the
    #  bucket's layers have landed in EVERY chain)
    opt_in_bwd = (opt is not None and args.opt_in_backward == "on" and
                  ((not comm and (args.defer_da in ("side", "window", "layer", "bucket", "unit") or args.chains > 1) and args.graph in ("auto", "all", "off"))
                   or comm))
    # fused forward: the weight shadows are rewritten where the weights change ("opt": behind the optimizer -- the bucket's slice on the
    # side / communication stream with --opt-in-backward, the one launch behind the backward otherwise) or in front of every unit ("main")
    shadows_main = bool(args.fused and args.shadows == "main")
    shadows_opt = bool(args.fused and args.shadows == "opt")
    shadows_in_cb = False
    if opt_in_bwd:
        opt.set_device_step(0)                       # (allocates the device-side coefficient state; no step counted)
        if comm:
            ends = wl["layer_end"]

            def _reduced(blo, bhi):
                opt.step_range(blo, bhi, grad_scale=1.0 / world, zero_grad=True)
                if shadows_opt:                      # (on the communication stream, behind the bucket's update)
                    run_shadows(lib, wl, c_void_p(torch.cuda.current_stream().cuda_stream), [l for l in range(L) if blo < ends[l] <= bhi])
            bucket.on_reduced = _reduced
            shadows_in_cb = shadows_opt
    if shadows_opt:
        run_shadows(lib, wl, c_void_p(torch.cuda.current_stream().cuda_stream), range(L))     # the initial weights' shadows
        torch.cuda.synchronize()

    records = Recorder(only=LIVE, every=args.bracket_every)
    records.reserve(2 * len(wl["units"]) * args.steps + 16)

    # hipGraphs (the library only enqueues on the stream it is given -- no allocation, no synchronisation -- so its launches
    # capture unchanged).  "bwd": the forward as one graph and the backward launches of every gradient bucket (4 layers: 80 kernels)
    # as one graph each; the bucket hooks (RCCL all-reduce of a finished bucket) run between the graphs exactly as between live
    # layers.  "all": the whole micro-batch as one graph (single GPU).  Nothing can be bracketed inside a graph: `roofline` then
    # comes from the extra live passes behind the timed region.
    fwd_bwd_graph, bwd_graphs, fwd_graph = None, None, None
    if args.graph != "off":
        try:
            # the chain's (capture) stream at high priority, the deferred dA / dB stream at normal: when both have workgroups waiting, the
            # dependency chain goes first (7B r = 16, same box twice: 33.43-33.48 -> 33.16-33.18 ms; r = 64: no difference)
            side = torch.cuda.Stream(device=dev, priority=-1 if args.chain_priority == "high" else 0)
            with torch.cuda.stream(side):
                spw = c_void_p(side.cuda_stream)
                for ch in wl["chains"]:
                    run_forward(lib, ch, spw)
                    run_backward(lib, ch, spw, L)   # warm-up on the capture stream (LDS attributes, lazy module load)
            torch.cuda.synchronize()
            pri = -1 if args.chain_priority == "high" else 0
            anchor = torch.zeros(64, device=dev)
            stagger_buf = torch.empty(max(1, args.chain_stagger * (1 << 20) * max(1, args.chains - 1)), dtype=torch.uint8, device=dev)

            def capture_hub(graph, forward, pieces, with_opt):
                """One graph in the hub shape: the N part-batch chains on N forked streams, everything off the chains (every chain's deferred
                dA_m / dB launches, the optimizer slices, the weight shadows) on the capture's origin stream.
                * hipStreamEndCapture (ROCm 7.2) segfaults on ANY dependency between two streams that are both forks
                  (tools/probes/capture_topology.py): every edge has to touch the origin, so the origin is the hub;
                * the executor does not run a graph on the capture's streams: it cuts the DAG into lists by a depth-first walk from the roots
                  that follows every node's FIRST out-edge, gives every list a stream of its own and maps those onto a handful of in-order
                  hardware queues (a list that waits for another list blocks whatever shares its queue).  Round 4's side-first forks made
                  every layer's dA_m a list of its own and cut the chain at every fork.  This graph is SHAPED for that walk: the hub's
                  launches are the root's first path (an anchor node captured in front of the forks' first launches), nothing on a chain ever
                  waits for the hub (every layer owns its pack buffers: no reuse edges), so the walk yields exactly 1 + N lists;
                * the walk of the backward is captured piece by piece (a layer; --defer-da bucket: a gradient bucket), chain by chain, so that
                  the hub's stream order is "piece p of every chain, then the bucket's optimizer slice"."""
                hub = torch.cuda.Stream(device=dev)
                branch = [torch.cuda.Stream(device=dev, priority=pri) for _ in wl["chains"]]
                with torch.cuda.graph(graph, stream=hub):
                    cur = torch.cuda.current_stream()
                    if with_opt:
                        # the step's AdamW coefficients, written on the device by a one-thread launch that counts the steps itself:
                        # every replay advances by one, nothing is read from host memory (FlatAdamW.begin_step)
                        opt.begin_step(device_counter=True)
                        opt.t -= 1                       # (the capture is not a step)
                    else:
                        anchor.zero_()                   # (the root)
                    for st in branch:
                        st.wait_stream(cur)              # fork
                    anchor.zero_()                       # the root's FIRST successor is on the hub: the walk runs down the hub before it sees a chain
                    if forward:
                        for ci_, (ch, st) in enumerate(zip(wl["chains"], branch)):
                            if ci_ and args.chain_stagger > 0:
                                # (identical chains that start together march in lockstep -- both in a latency-bound launch at the same
                                #  time; a fill of `--chain-stagger` MB in front of the later chains shifts their phase)
                                with torch.cuda.stream(st):
                                    stagger_buf[:ci_ * args.chain_stagger * (1 << 20)].zero_()
                            run_forward(lib, ch, c_void_p(st.cuda_stream), shadows=shadows_main)
                    states = [dict() for _ in branch]
                    pend_opt = None

                    def hub_opt(lb, evs):
                        # the chains add into the same gradient accumulators: the bucket's AdamW slice (and its layers' weight shadows for
                        # the next step) goes out on the hub, behind the dA_m launches of the bucket's first layer of EVERY chain
                        for ev in evs:
                            cur.wait_event(ev)           # (the in-chain gradients of the layer: dB rides with the pass over gy)
                        blo, bhi = bucket.bucket_bounds(lb)
                        opt.step_range(blo, bhi, grad_scale=1.0 / world, zero_grad=True)
                        if shadows_opt:
                            run_shadows(lib, wl, c_void_p(cur.cuda_stream), bucket.bucket_layers(lb))
                    for pi_, (l, l_hi) in enumerate(pieces):
                        for ch, st, stt in zip(wl["chains"], branch, states):
                            run_backward(lib, ch, c_void_p(st.cuda_stream), l_hi, lo=l, state=stt, join=False, flush=pi_ == len(pieces) - 1,
                                         defer=(args.defer_da, st, cur, args.split_db, bucket.is_bucket_first) if args.defer_da != "off" else None)
                        if pend_opt is not None:
                            hub_opt(*pend_opt)           # (chain-first: behind the first launches of the chains' next piece)
                            pend_opt = None
                        if with_opt and bucket.is_bucket_first(l):
                            evs = []
                            for st in branch:
                                ev = torch.cuda.Event()
                                ev.record(st)
                                evs.append(ev)
                            if CHAIN_FIRST and pi_ < len(pieces) - 1:
                                pend_opt = (l, evs)
                            else:
                                hub_opt(l, evs)
                    for st in branch:
                        cur.wait_stream(st)              # join
                return hub, branch                       # (kept alive with the graph)

            def pieces_of(lo_, hi_):
                # (pieces of the walk: a layer; with --defer-da bucket a whole gradient bucket, whose dA_m launches leave together)
                if args.defer_da == "bucket":
                    return [(f, bucket.bucket_layers(f).stop) for f in reversed(bucket.bucket_firsts()) if lo_ <= f < hi_]
                return [(l, l + 1) for l in range(hi_ - 1, lo_ - 1, -1)]

            graph_keep = []
            if os.environ.get("MOKA_BENCH_FAIL_CAPTURE") == "1":
                raise RuntimeError("MOKA_BENCH_FAIL_CAPTURE=1 (test hook: exercise the live fallback)")
            if args.graph == "all":
                # (collectives cannot ride inside the graph: capturing the one-rank RCCL all-reduce with torch 2.10 / RCCL 2.26.6 segfaults at
                #  capture time -- measured round 4 -- so N > 1 and --force-comm use one graph per gradient bucket with the hooks between them)
                assert not comm, "--graph all: single GPU without collectives only"
                fwd_bwd_graph = torch.cuda.CUDAGraph()
                if not args.hub:
                    da_side = torch.cuda.Stream(device=dev)
                    with torch.cuda.graph(fwd_bwd_graph, stream=side):
                        cur = torch.cuda.current_stream()
                        spg = c_void_p(cur.cuda_stream)
                        if opt_in_bwd:
                            opt.begin_step(device_counter=True)
                            opt.t -= 1                   # (the capture is not a step)
                        run_forward(lib, wl, spg, shadows=shadows_main)
                        run_backward(lib, wl, spg, L, defer=(args.defer_da, cur, da_side, args.split_db, bucket.is_bucket_first) if args.defer_da != "off" else None,
                                     bucket_opt=(opt, bucket, 1.0 / world) if opt_in_bwd else None, shadows_after_opt=shadows_opt and opt_in_bwd)
                else:
                    graph_keep.append(capture_hub(fwd_bwd_graph, True, pieces_of(0, L), opt_in_bwd))
            elif not args.hub:
                da_side = torch.cuda.Stream(device=dev)
                fwd_graph = torch.cuda.CUDAGraph()       # the forward has no hooks: one graph
                with torch.cuda.graph(fwd_graph, stream=side):
                    run_forward(lib, wl, c_void_p(torch.cuda.current_stream().cuda_stream), shadows=shadows_main)
                bwd_graphs = []
                for lo in reversed(bucket.bucket_firsts()):      # buckets are contiguous groups of layers, walked last -> first
                    hi = bucket.bucket_layers(lo).stop
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        cs = torch.cuda.current_stream()
                        run_backward(lib, wl, c_void_p(cs.cuda_stream), hi, lo=lo,
                                     defer=(args.defer_da, cs, da_side, args.split_db, bucket.is_bucket_first) if args.defer_da != "off" else None)
                    bwd_graphs.append((g, lo, hi))
            else:
                # N > 1 with part-batch chains: the forward as one hub-shaped graph, one hub-shaped graph per gradient bucket of the backward (the
                # chains meet at every graph's end: that is where the bucket's all-reduce is handed to RCCL)
                fwd_graph = torch.cuda.CUDAGraph()
                graph_keep.append(capture_hub(fwd_graph, True, [], False))
                bwd_graphs = []
                for lo in reversed(bucket.bucket_firsts()):
                    hi = bucket.bucket_layers(lo).stop
                    g = torch.cuda.CUDAGraph()
                    graph_keep.append(capture_hub(g, False, pieces_of(lo, hi), False))
                    bwd_graphs.append((g, lo, hi))
            torch.cuda.synchronize()
        except Exception as exc:                         # capture is an optimisation, never a requirement
            # (with chains: the part-batches then run one after the other on the one stream -- every launch of the step still happens)
            print(f"bench: hipGraph capture failed ({exc!r}); launching live" + (", the %d chains back to back" % args.chains if args.chains > 1 else ""), file=sys.stderr)
            fwd_bwd_graph, bwd_graphs, fwd_graph = None, None, None
            args.graph = "off (capture failed)"          # (what the line reports is what ran)
            torch.cuda.synchronize()

    # N > 1: how long the main stream stands still in bucket.finish() (the part of the all-reduce the backward did not hide)
    comm_ev = [] if comm else None

    live_side = torch.cuda.Stream(device=dev) if args.defer_da != "off" else None

    def step(i, rec=None):
        sp = c_void_p(main_stream.cuda_stream)
        if opt is None:
            bucket.zero_()                           # (the optimizer kernel leaves the gradient buffer zeroed)
        if opt_in_bwd:
            if fwd_bwd_graph is None:
                opt.begin_step()                     # this step's coefficients: a one-thread launch on the main stream (launch arguments)
            else:
                opt.t += 1                           # (the captured launch counts on the device; the host keeps the books)
        if fwd_bwd_graph is not None:
            fwd_bwd_graph.replay()
        else:
            if fwd_graph is not None:
                fwd_graph.replay()
            else:
                for ch in wl["chains"]:
                    run_forward(lib, ch, sp, rec, shadows=shadows_main)
            if bwd_graphs is not None:
                for g, lo, hi in bwd_graphs:
                    g.replay()
                    for l in range(hi - 1, lo - 1, -1):
                        bucket.layer_done(l)         # all-reduce of the finished bucket overlaps the next graphs
            else:
                # (live launches: the chains one after the other; the bucket hooks / optimizer slices ride with the LAST chain's layers -- every
                #  earlier chain's gradients are in front of them in stream order)
                for ci, ch in enumerate(wl["chains"]):
                    last = ci == len(wl["chains"]) - 1
                    run_backward(lib, ch, sp, L, bucket.layer_done if last else None, rec,   # all-reduce of finished layer groups overlaps the rest
                                 defer=(args.defer_da, main_stream, live_side, args.split_db, bucket.is_bucket_first) if args.defer_da != "off" else None,
                                 bucket_opt=(opt, bucket, 1.0 / world) if (opt_in_bwd and not comm and last) else None,
                                 shadows_after_opt=shadows_opt and opt_in_bwd and not comm and last)
        if comm_ev is not None and i >= args.warmup:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main_stream)
            bucket.finish(average=opt is None)
            e1.record(main_stream)
            comm_ev.append((e0, e1))
        else:
            bucket.finish(average=opt is None)       # join the all-reduces; the optimizer kernel averages (grad_scale)
        if opt is not None and not opt_in_bwd:
            opt.step(grad_scale=1.0 / world, zero_grad=True)
        if shadows_opt and opt is not None and not (opt_in_bwd and not comm) and not shadows_in_cb:
            run_shadows(lib, wl, sp, range(L))       # (every weight has changed: the shadows of the whole stack, behind the step)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, records if rank == 0 else None)     # HIP events bracket every launch of the timed steps
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = tt.item()
    ms_per_step = el * 1e3 / args.steps
    tokens_per_s = world * T * args.steps / el

    # host side of a replay: how long hipGraphLaunch keeps the launching thread for ONE step (idle GPU in front of it, so nothing blocks on
    # a full queue) against the step on the GPU -- a multi-branch graph is replayed node by node, and a step whose replay takes the host
    # longer than the GPU needs is host-bound
    replay_host_ms = None
    if fwd_bwd_graph is not None and rank == 0:
        hs = []
        for _ in range(3):
            torch.cuda.synchronize()
            th = time.perf_counter()
            fwd_bwd_graph.replay()
            hs.append((time.perf_counter() - th) * 1e3)
            torch.cuda.synchronize()
            if opt_in_bwd:
                opt.t += 1
        replay_host_ms = round(min(hs), 3)

    # --verify-graph: the captured schedule (chains on forked streams, deferred dA_m on the hub, whatever the executor makes of it) against the
    # same launches live, one chain after the other on ONE stream, from the same activation state: y and dx bit for bit (deterministic
    # kernels), the flat gradient to the spread of its fp32 atomics
    graph_check = None
    if args.verify_graph and rank == 0:
        if opt is not None or (fwd_bwd_graph is None and bwd_graphs is None):
            raise SystemExit("--verify-graph compares gradients: run it with --no-optimizer and a graph mode")
        mut = [t for k in wl["keep"] for (acts, dacts, ys) in k[0] for t in list(dacts.values()) + list(ys)]
        snap = [t.clone() for t in mut]

        def restore():
            for t, s_ in zip(mut, snap):
                t.copy_(s_)
            bucket.zero_()
        restore()
        torch.cuda.synchronize()
        if fwd_bwd_graph is not None:
            fwd_bwd_graph.replay()
        else:
            fwd_graph.replay()
            for g, lo, hi in bwd_graphs:
                g.replay()
        torch.cuda.synchronize()
        g_graph = bucket.flat.clone()
        got = [t.clone() for t in mut]
        restore()
        sp_v = c_void_p(torch.cuda.current_stream().cuda_stream)
        for ch in wl["chains"]:
            run_forward(lib, ch, sp_v, None, shadows=shadows_main)
            run_backward(lib, ch, sp_v, L, None, None)
        torch.cuda.synchronize()
        den = float(bucket.flat.abs().max())
        graph_check = {"activations_bit_identical": all(torch.equal(a_, b_) for a_, b_ in zip(got, mut)), "tensors_compared": len(mut),
                       "grad_max_abs_diff_over_max": float((g_graph - bucket.flat).abs().max()) / max(den, 1e-30),
                       "grad_max_abs": den, "grad_nonzero_frac": float((bucket.flat != 0).float().mean()),
                       "what": "graph replay vs the same launches live, chain after chain on one stream, from the same activation state"}
        restore()
        del got, g_graph

    out = None
    if rank == 0:
        fwd_b, bwd_b = algorithmic_bytes_per_token(MODELS[args.model], args.rank, args.layers)
        algo_gbs = (fwd_b + bwd_b) * T / (ms_per_step * 1e-3) / 1e9
        # per-launch durations from the HIP events recorded on the launch stream inside the timed region
        def collect(items):
            tot = {n: 0.0 for n in ENTRY}
            cnt = {n: 0 for n in ENTRY}
            byt = {n: 0 for n in ENTRY}
            per_shape = {}
            for n, u, e0, e1 in items:
                ms = e0.elapsed_time(e1)
                tot[n] += ms
                cnt[n] += 1
                byt[n] += u.algo[n]
                key = (n, u.label, u.d_in, tuple(u.d_outs))
                a_, b_, _ = per_shape.get(key, (0.0, 0, 0))
                per_shape[key] = (a_ + ms, b_ + 1, u.algo[n])
            return tot, cnt, byt, per_shape
        sp_ = c_void_p(torch.cuda.current_stream().cuda_stream)
        units_all = wl["units"]
        roof_behind = not records.items
        if not records.items:
            # graph replay: nothing can be bracketed inside the timed region -> the dominant entry point is bracketed (every n-th
            # launch, as in the live mode) in extra live passes right behind it, same buffers, same kernel sequence
            for _ in range(min(args.steps, 3)):
                for ch in wl["chains"]:
                    run_forward(lib, ch, sp_, records, shadows=shadows_main)
                    run_backward(lib, ch, sp_, L, None, None)
            torch.cuda.synchronize()
        tot, cnt, byt, per_shape = collect(records.items)         # the dominant entry point (LIVE)
        # every entry point, in one extra untimed pass (full bracketing would perturb the timed region)
        extra = Recorder()
        for ch in wl["chains"]:
            run_forward(lib, ch, sp_, extra, shadows=shadows_main)
            run_backward(lib, ch, sp_, L, None, extra)
        if shadows_opt:
            run_shadows(lib, wl, sp_, range(L), extra)
        torch.cuda.synchronize()
        tot_x, cnt_x, byt_x, per_shape_x = collect(extra.items)
        # the forward alone, replayed as a hipGraph of its own against the same launches live: HIP events around whole passes, no
        # profiler (profiles/README.md: do the 8-25 us gaps rocprofv3 shows in front of the forward kernels of a graph replay exist?)
        fwd_only = None
        if args.probe_forward and args.chains == 1:
            try:
                pst = torch.cuda.Stream(device=dev, priority=-1 if args.chain_priority == "high" else 0)
                with torch.cuda.stream(pst):
                    run_forward(lib, wl, c_void_p(pst.cuda_stream), shadows=shadows_main)
                torch.cuda.synchronize()
                fg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(fg, stream=pst):
                    run_forward(lib, wl, c_void_p(torch.cuda.current_stream().cuda_stream), shadows=shadows_main)
                torch.cuda.synchronize()

                def _timed(fn, n=6):
                    fn()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(n):
                        fn()
                    b.record()
                    torch.cuda.synchronize()
                    return a.elapsed_time(b) / n
                g_ms = _timed(fg.replay)
                l_ms = _timed(lambda: run_forward(lib, wl, sp_, None, shadows=shadows_main))
                fwd_only = {"graph_ms": round(g_ms, 3), "live_ms": round(l_ms, 3), "launches": sum(2 if u.fused else 3 for u in units_all),
                            "what": "forward pass alone, 6 passes between two HIP events: replay of a forward-only hipGraph vs the same launches live"}
                del fg
            except Exception as exc:                 # a probe, never a requirement
                fwd_only = {"error": repr(exc)}
        live_items = records.items
        table = {}
        for (n, label, di, dos), (ms, c_, nb) in sorted(per_shape_x.items()):
            avg = ms / c_
            table[f"{n}[{label}: {di}->{'/'.join(str(v) for v in dos)}]"] = {"avg_ms": round(avg, 4), "algo_GBps": round(nb / (avg * 1e-3) / 1e9, 1)}
        # the dominant kernel: the largest entry point of a pass that is ONE kernel launch
        # (moka_up_fwd -> moka_yt_kernel<RP> for the batched launches, moka_expand_kernel<.., true> for single projections; grouped units
        #  run their members in one launch, grid z)
        # the dominant KERNEL: every launch of moka_up_fwd is moka_yx_kernel<RP> where the whole stack runs the fused forward (the headline);
        # otherwise the entry point is served by several kernels and the line names them
        RPk = _lib.rank_pad(args.rank)
        all_fused = all(u.fused for u in units_all)
        any_fused = any(u.fused for u in units_all)
        kern_name = ("moka_yx_kernel<%d>" % RPk) if all_fused else (
            ("moka_yx_kernel<%d> (units %s) + " % (RPk, ", ".join(sorted({u.label for u in units_all if u.fused}))) if any_fused else "") +
            "moka_yt_kernel<%d> / moka_expand_kernel<%d,NQ,true> (three-launch units)" % (RPk, RPk))
        dom = "moka_up_fwd"
        dom_bytes = byt[dom]
        dom_avg_ms = tot[dom] / cnt[dom]
        achieved = dom_bytes / cnt[dom] / (dom_avg_ms * 1e-3) / 1e9
        # the bytes THIS implementation has to move per step (grouped x / dx counted once per group, the deferred dA's second read of x
        # counted; rank-space tensors and weights left out as in the contract figure): forward x + y read-modify-write, backward gy +
        # dx read-modify-write + x again
        per_layer = wl["units_per_layer"]
        actual_b = 0
        for u in units_all[:per_layer]:
            sdo = sum(u.d_outs)
            actual_b += E * T * (u.d_in + 2 * sdo) + E * T * (sdo + 2 * u.d_in + u.d_in)
        actual_b *= args.layers
        actual_gbs = actual_b / (ms_per_step * 1e-3) / 1e9
        traffic, traffic_src = None, None
        if not args.no_traffic and (args.model, args.rank, args.variant) == ("7b", 16, "avt") and not args.no_group:
            # (the PMC passes profile the headline workload; per launch = per layer / the layer's up-projection launches)
            traffic, traffic_src = pmc_traffic_per_launch(T, wl["units_per_layer"])
            if traffic is not None:
                traffic = round(traffic / args.chains)   # (the PMC passes profile whole-batch launches; traffic is linear in the tokens)
            if args.chains > 1 and traffic is not None:
                traffic_src += " / %d (launches of %d tokens)" % (args.chains, T // args.chains)
            elif traffic is not None:
                traffic_src += " (measured on launches of %d tokens, scaled)" % json.load(open(PMC_TRAFFIC_FILE))["tokens"]
        out = {
            "metric": "tokens/sec/GPU Llama-2-7B MokA r=16 seq2048 bf16; adapter HBM %roofline" if (args.model, args.rank, args.seq) == ("7b", 16, 2048)
                      else "tokens/sec/GPU Llama-2-%s MokA r=%d seq%d bf16; adapter HBM %%roofline" % (args.model.upper(), args.rank, args.seq),
            "value": round(tokens_per_s, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # `value` is the whole-job aggregate over the N GPUs (driver contract); the metric's per-GPU figure beside it
            "tokens_per_s_per_gpu": round(tokens_per_s / world, 1), "aggregate_tokens_per_s": round(tokens_per_s, 1),
            "comm_exposed_ms": (round(sum(a.elapsed_time(b) for a, b in comm_ev) / max(1, len(comm_ev)), 4) if comm_ev else 0.0),
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "Llama-2-%s dims, MokA r=%d %s, adapter fwd+bwd of 7x%d projections, "
                                   "seq=%d (%s), lora_dropout %g, batch %d seq/GPU, %s, "
                                   "+ DP grad all-reduce (RCCL) + fused AdamW on adapter params"
                                   % (args.model.upper(), args.rank, "M=2 (VT semantics)" if args.variant == "vt" else "M=3 (AVT semantics)", args.layers, args.seq,
                                      ("%d image + %d question + text" % (args.seq // 8, args.seq // 32)) if args.variant == "vt"
                                      else ("%d image + %d audio + %d question + text" % (args.seq // 8, args.seq // 16, args.seq // 32)),
                                      args.dropout, args.batch,
                                      ("one launch set per projection" if args.no_group else "q/k/v and gate/up through the grouped entry points")
                                      + ("" if args.chains == 1 else ", as %d independent part-batch chains on %d streams" % (args.chains, args.chains))),
                       "tokens_per_gpu_per_step": T, "layers": args.layers, "rank": args.rank, "parallelism": f"dp{world}"},
            "distributed": {"world_size": world, "dist_world_size": dist.get_world_size() if (comm and dist.is_initialized()) else 1,
                            "backend": (dist.get_backend() if (comm and dist.is_initialized()) else None),
                            "force_comm": bool(args.force_comm),
                            "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
                            "grad_payload": "%s payload of the fp32 flat bucket, %d buckets, all-reduce on a side stream overlapped with the backward" % ("bf16" if args.comm_bf16 else "fp32", len(bucket.bucket_firsts())),
                            "bucket_layers": [len(bucket.bucket_layers(f)) for f in bucket.bucket_firsts()], "last_bucket_bytes": bucket.last_bucket_bytes(),
                            "adapter_params": wl["n_params"]},
            "graph": args.graph, "graph_replay_host_ms": replay_host_ms, "graph_topology": ("hub" if args.hub else "chain") if not args.graph.startswith("off") else None,
            "graph_check": graph_check,
            "fused_forward": ("all units" if all(u.fused for u in units_all) else ("units " + ", ".join(sorted({u.label for u in units_all if u.fused})) if any(u.fused for u in units_all) else False)) if args.fused else False,
            "chains": args.chains,
            "defer_dA": args.defer_da,
            "defer_dB": bool(args.split_db), "chain_priority": args.chain_priority, "optimizer_in_backward": bool(opt_in_bwd),
            # (what this line's schedule is and is not: ADVICE r04)
            "schedule": "bench.py's launch schedule over the C ABI (part-batch chains / deferred dA_m on a hub stream / one hipGraph, persistent weight "
                        "shadows rewritten behind the optimizer slices); moka_amd.parallel.attach + MokaLinearFn run ONE chain through autograd and "
                        "relaunch the shadows every forward: --e2e measures that path",
            "adapter_hbm_roofline_frac": round(algo_gbs / world / HBM_PEAK_GBS, 4),
            "adapter_algorithmic_GBps_per_gpu": round(algo_gbs / world, 1),
            # what the bus really carries: this implementation's own bytes (x / dx of a group once, x a second time for the deferred dA)
            "adapter_actual_bytes_per_step": int(actual_b), "adapter_actual_GBps_per_gpu": round(actual_gbs, 1),
            "adapter_actual_hbm_frac": round(actual_gbs / HBM_PEAK_GBS, 4),
            "roofline": {"bound": "hbm", "kernel": kern_name, "entry_point": "moka_up_fwd_fused" if all_fused else "moka_up_fwd",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": round(dom_bytes / cnt[dom]),
                         "avg_launch_ms": round(dom_avg_ms, 4), "launches_timed": cnt[dom],
                         "where": ("live passes right behind the timed region (the timed region replays a hipGraph: nothing can be bracketed inside it)"
                                   if roof_behind else "HIP events inside the timed region (every %d-th launch)" % args.bracket_every),
                         "note": (None if args.chains == 1 else
                                  "launches of %d tokens timed ALONE, the %d chains back to back on one stream; in the step they run beside the other chain's "
                                  "launches (not observable inside a graph; rocprofv3 serialises the dispatches) -- adapter_hbm_roofline_frac is the step's figure"
                                  % (T // args.chains, args.chains))},
            "entry_point_ms_per_pass": {n: round(tot_x[n], 3) for n in ENTRY},
            "forward_only": fwd_only,
            "kernels": table,
        }
        if world == 1 and args.e2e:
            del wl, opt, records
            torch.cuda.empty_cache()
            out["end_to_end"] = end_to_end(args, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        final_line = json.dumps(out)
    if world > 1:
        dist.barrier()
    if comm:
        dist.destroy_process_group()
    if rank == 0 and out is not None:
        # the ONE JSON line, as the last thing on stdout: RCCL writes its version banner through C stdio, which is flushed at exit --
        # behind a line printed from Python -- so C stdio is flushed first and the line goes straight to file descriptor 1
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.write(1, (final_line + "\n").encode())


if __name__ == "__main__":
    main()
