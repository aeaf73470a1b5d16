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
    if os.environ.get("MOKA_BENCH_LAYOUT") == "aligned":       # diagnostics only (tools/experiments): the same spans, every boundary on a multiple of 128 tokens
        n_pre, n_mid, n_img, n_aud = 0, 0, (n_img + 127) // 128 * 128, (n_aud + 127) // 128 * 128
    tok = torch.zeros(S, dtype=torch.int64)
    q = torch.zeros(S, dtype=torch.bool)
    p = n_pre
    tok[p:p + n_img] = 1
    p += n_img + n_mid
    tok[p:p + n_aud] = 2
    p += n_aud
    q[p:p + n_q] = True
    return tok, q


from moka_amd import schedule as SCH  # noqa: E402  (the launch schedule lives in the package: bench.py builds synthetic buffers, constructs
#                                                      a GraphedAdapterStep over them and times its step())

Unit = SCH.AdapterUnit


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
    T = B * S
    tok, q = synthetic_layout(S)
    if vt:
        # BASELINE.json configs[1]: visual-text -- the audio span becomes text, bool [B,S] masks (VisualText/train/train.py:206-231)
        tok = torch.where(tok == 2, torch.zeros_like(tok), tok)
    by_class = getattr(args, "chain_split", "sample") == "class" and chains == 2
    if by_class:
        # chains INSIDE the samples: text tokens that are neither query nor key rows interact with nothing
        # (lora.py:497-499 masks the update to video / audio rows; keys are the question span), so the token blocks that hold query / key
        # rows -- here: every sequence's leading blocks up to the end of the question span -- form chain 0 (with the rank-space launches'
        # real work), the text-only rest of every sequence chain 1 (a routing without keys).  Same tokens, same sums; also for B = 1.
        span = int(max(int(torch.nonzero(q).max()) if bool(q.any()) else -1, int(torch.nonzero(tok > 0).max()) if bool((tok > 0).any()) else -1)) + 1
        S0 = min(S, (span + 127) // 128 * 128)
        assert 0 < S0 < S and bool((tok[S0:] == 0).all()) and not bool(q[S0:].any()), "--chain-split class: no text-only tail in this layout"
        parts = [(B, S0, tok[:S0], q[:S0]), (B, S - S0, None, None)]
    else:
        assert 1 <= chains <= B, "--chains: at most one chain per sequence"
        # (uneven splits: the larger part-batches first)
        parts = [(B // chains + (1 if ci < B % chains else 0), S, tok, q) for ci in range(chains)]
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
    # the device word the dropout kernels fold into their seeds (moka_opts.seed_dev): the executor rewrites it in front of every step, so the
    # replays of a captured graph -- seeds frozen with the launch arguments -- draw fresh keep masks like the training steps they stand for
    seed_epoch = torch.zeros(1, dtype=torch.int64, device=dev) if (args.dropout > 0 and getattr(args, "seed_dev", "on") == "on") else None
    nset = max(1, min(L, args.distinct))
    chain_list, keep = [], []
    shadow_bufs = {}              # (layer, projection) -> (BwT, AT): functions of the weights alone, so every chain reads the same pair
    for ci in range(chains):
        Bc, Sc, tok_c, q_c = parts[ci]
        Tc = Bc * Sc
        Tp = _lib.tok_pad(Tc)
        max_ks = max(_lib.ksplit(Tc, ff, r, 1), _lib.ksplit(Tc, d, r, 2), _lib.ksplit_bwd(Tc, ff, r))
        if tok_c is None:
            masks = None
            rt = MokaRouting.plain(Bc, Sc, dev, M)                   # text tokens only, no key rows: the adapter of modality 0, no interaction
        elif vt:
            masks = [(tok_c == 0).reshape(1, Sc).repeat(Bc, 1).to(dev), (tok_c == 1).reshape(1, Sc).repeat(Bc, 1).to(dev), q_c.reshape(1, Sc).repeat(Bc, 1).to(dev)]
            rt = MokaRouting.from_vt_masks(*masks)
        else:
            masks = [(tok_c == m).to(torch.int32).reshape(1, Sc, 1).repeat(Bc, 1, 1).to(dev) for m in range(3)]
            masks.append(q_c.to(torch.int32).reshape(1, Sc, 1).repeat(Bc, 1, 1).to(dev))
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
                                  company=chains if getattr(args, "company_hint", "on") == "on" else 1, seed_dev=seed_epoch))
        layer_da, layer_db = SCH.make_layer_batches(units, len(unit_defs), L, rt, Tc, r, M, args.dropout) if defer else ([], [])
        chain_list.append(SCH.AdapterChain(units=units, units_per_layer=len(unit_defs), rt=rt, T=Tc, layer_da=layer_da, layer_db=layer_db, rank=r, reuse_wait=(n_own < L)))
        keep.append((sets, masks, scratch2, own))
    return SCH.AdapterWorkload(units=chain_list[0]["units"], units_per_layer=len(unit_defs), rt=chain_list[0]["rt"], layer_da=chain_list[0]["layer_da"], layer_db=chain_list[0]["layer_db"], rank=r, chains=chain_list, master=master, work=work, reuse_wait=chain_list[0]["reuse_wait"],
                             gbuf=gbuf, bucket=bucket, T=T, n_params=n_params, layer_end=layer_end, keep=keep, seed_epoch=seed_epoch)


ENTRY = SCH.ENTRY
LIVE = ("moka_up_fwd",)     # the dominant single-kernel entry point, bracketed inside the timed region
Recorder, run_forward, run_backward, run_shadows = SCH.Recorder, SCH.run_forward, SCH.run_backward, SCH.run_shadows      # (tools/ written against bench.*)


# roofline.traffic: HBM bytes per launch of the dominant kernel from the PMC counters -- read from the committed summary of the
# PMC passes of THIS build (tools/pmc_traffic.sh -> profiles/r04_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE
# in separate passes; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, WRITE_SIZE as is), never a constant in here.
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")


def kernel_source_sha256():
    """sha256 over every file the library is built from (the translation units, their two headers, the C header): what a PMC traffic summary
    is stamped with (tools/pmc_traffic.py)."""
    import hashlib
    from moka_amd import build as _build
    h = hashlib.sha256()
    for path in _build.sources():
        h.update(open(path, "rb").read())
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

    def build(make, train_adapter):
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
        return st

    def clock(step, n=3, warm=2):
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / n

    batch = {"h": h, "gout": gout, "m_t": masks[0], "m_v": masks[1], "m_a": masks[2], "m_q": masks[3]}
    captures = []                                    # per captured step: the timed capture attempts (GraphedTrainStep.capture_log)

    def stack_loss(st):
        # (dx reaches the embeddings / projector in the real model: the stack's input carries a gradient; the "loss" is <out, gout>, whose
        #  backward feeds gout into the last layer like out.backward(gout) does)
        def f(part):
            x = part["h"].detach().requires_grad_(True)
            out, _ = st(x, [part["m_t"], part["m_v"], part["m_a"], part["m_q"]])
            return (out.float() * part["gout"].float()).sum() / out.shape[0]
        return f

    def live(make, mode, **attach_kw):
        """mode: "base" (frozen stack only) | "autograd" (adapters as plain autograd nodes + torch's fused AdamW: no attach) |
        "attach" (moka_amd.parallel.attach: gradient sinks, persistent weight shadows, the fused flat AdamW; defer_dA as given)."""
        st = build(make, mode != "base")
        f = stack_loss(st)
        if mode == "attach":
            from moka_amd.parallel import attach
            dp = attach(st, n_buckets=8, lr=1e-4, **attach_kw)

            def step():
                f(batch).backward()
                dp.step()
        else:
            params = [p_ for p_ in st.parameters() if p_.requires_grad]
            opt = torch.optim.AdamW(params, lr=1e-4, fused=True) if params else None

            def step():
                f(batch).backward()
                if opt is not None:
                    opt.step()
                    opt.zero_grad(set_to_none=False)
        ms = clock(step)
        del st, step, f
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        return ms

    def graphed(make, with_adapter, chains, **attach_kw):
        """The whole step as ONE hub-shaped hipGraph captured through autograd (moka_amd.schedule.GraphedTrainStep)."""
        from moka_amd.parallel import attach
        from moka_amd.routing import MokaRouting
        st = build(make, with_adapter)
        dp = attach(st, n_buckets=8, lr=1e-4, **attach_kw) if with_adapter else None
        gs = SCH.GraphedTrainStep(dp, stack_loss(st), batch, chains=chains,
                                  routing_fn=(lambda p_: MokaRouting.from_avt_masks([p_["m_t"], p_["m_v"], p_["m_a"], p_["m_q"]])) if with_adapter else None)
        ms = clock(lambda: gs(None), n=5)
        captures.append({"adapter": bool(with_adapter), "chains": chains, **attach_kw, "tries": gs.capture_log})
        del gs, dp, st
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        return ms

    def say(msg):
        print("bench --e2e: " + msg, file=sys.stderr, flush=True)

    res = {"what": "decoder stack fwd+bwd (+ AdamW on the adapter), %d layers, %d x %d tokens, bf16; NOT the metric.  adapter_ms = the stack's step "
                   "minus the frozen stack's step in the SAME launch mode" % (L, B, S)}
    base_live = live(plain, "base")
    say("frozen base, live: %.2f ms" % base_live)
    auto_live = live(adapted, "autograd")
    say("autograd nodes + torch AdamW, live: %.2f ms" % auto_live)
    att_live = live(adapted, "attach", defer_dA=False)
    say("attach(defer_dA=False), live: %.2f ms" % att_live)
    att_defer = live(adapted, "attach", defer_dA=True)
    say("attach(defer_dA=True), live: %.2f ms" % att_defer)
    res["live"] = {"frozen_base_only_ms": round(base_live, 2),
                   "autograd_nodes_torch_adamw_ms": round(auto_live, 2), "autograd_nodes_adapter_ms": round(auto_live - base_live, 2),
                   "attach_ms": round(att_live, 2), "attach_adapter_ms": round(att_live - base_live, 2),
                   "attach_defer_dA_ms": round(att_defer, 2), "attach_defer_dA_adapter_ms": round(att_defer - base_live, 2),
                   "note": "defer_dA puts the dA_m launches on a side stream beside the NEXT layer's launches: beside the frozen base's hipBLASLt GEMMs the "
                           "streaming kernel costs the GEMMs more than it hides (the same finding as round 5's overlap_base, removed this round)"}
    for ch in ([1, 2] if args.e2e_chains2 else [1]):
        if ch > B:
            continue
        try:
            b_ms = graphed(plain, False, ch)
            say("frozen base, graphed, %d chain(s): %.2f ms" % (ch, b_ms))
            a_ms = graphed(adapted, True, ch, defer_dA=False)
            say("attach(defer_dA=False), graphed, %d chain(s): %.2f ms" % (ch, a_ms))
            d_ms = graphed(adapted, True, ch, defer_dA=True)
            say("attach(defer_dA=True), graphed, %d chain(s): %.2f ms" % (ch, d_ms))
            res["graphed_chains%d" % ch] = {"frozen_base_only_ms": round(b_ms, 2), "attach_ms": round(a_ms, 2), "adapter_ms": round(a_ms - b_ms, 2),
                                            "attach_defer_dA_ms": round(d_ms, 2), "adapter_share_of_step": round(1.0 - b_ms / a_ms, 4)}
        except Exception as exc:                     # (a capture the runtime refuses is reported, not fatal: the live figures stand)
            res["graphed_chains%d" % ch] = {"error": repr(exc)[:300]}
            torch.cuda.synchronize()
    res["captures"] = captures
    g1 = res.get("graphed_chains1", {})
    # the trainer path's figure: the captured step where it exists (host-independent), else the live attach() loop -- each against the
    # frozen stack in the SAME launch mode
    if "attach_ms" in g1:
        a_ms, b_ms = min(g1["attach_ms"], g1["attach_defer_dA_ms"]), g1["frozen_base_only_ms"]
        mode = "moka_amd.schedule.GraphedTrainStep (attach + MokaLinearFn captured as one single-list hipGraph)"
    else:
        a_ms, b_ms = min(att_live, att_defer), base_live
        mode = "moka_amd.parallel.attach, live launches"
    res.update({"ms_per_step": round(a_ms, 2), "tokens_per_s": round(B * S / (a_ms * 1e-3), 1), "frozen_base_only_ms_per_step": round(b_ms, 2),
                "adapter_ms": round(a_ms - b_ms, 2), "adapter_share_of_step": round(1.0 - b_ms / a_ms, 4), "mode": mode})
    return res


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
    ap.add_argument("--e2e-chains2", action="store_true",
                    help="with --e2e: also capture the whole stack as TWO part-batch chains (tests/test_gpu_trainer.py runs that shape on a small stack; "
                         "at the 7B widths the capture did not finish within minutes on ROCm 7.2 -- run it under `timeout`)")
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
    ap.add_argument("--chain-split", choices=("sample", "class"), default="sample",
                    help="how --chains 2 cuts the micro-batch: sample (default) = half the sequences each; class = by token class INSIDE every sequence: "
                         "chain 0 the leading token blocks up to the end of the question span (all query / key rows: the interaction lives there), chain 1 "
                         "the text-only rest (no key rows, no interaction) -- also splits a batch of ONE sequence")
    ap.add_argument("--seed-dev", choices=("on", "off"), default="on",
                    help="on: every step's dropout masks follow a device word rewritten in front of the step (moka_opts.seed_dev), so graph replays are "
                         "distinct training steps; off: the seeds are launch arguments only (a replay repeats the capture's masks; A/B)")
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
    ap.add_argument("--ablate", default="dominant",
                    help="in-schedule marginals: dominant (default) = the dominant kernel family only -> roofline.in_schedule; all / a comma list of "
                         "moka_amd.schedule.FAMILIES -> also the `ablation` table; off")
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
        # (rank pad 64, one chain + side stream: every launch fills the chip, the side stream's dA_m / dB launches are 13 ms of a 78 ms step, and a
        #  high-priority chain only delays them: normal 77.37 / 77.52 against high 78.05 / 78.00 ms, two same-box pairs, tools/experiments/r06/run_j.sh)
        args.chain_priority = "high" if (args.graph == "all" and args.rank <= 32) else "normal"

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

    # The schedule is the package's (moka_amd/schedule.py): part-batch chains as branches of one hub-shaped hipGraph, dA_m / dB / optimizer
    # slices / weight shadows on the hub, per-bucket graphs where collectives run between them.  bench.py builds the synthetic buffers,
    # constructs the executor and times its step().
    def make_cfg(skip=()):
        return SCH.ScheduleConfig(chains=args.chains, graph=args.graph, topology=args.graph_topology, defer_da=args.defer_da, split_db=bool(args.split_db),
                                chain_priority=args.chain_priority, fused=bool(args.fused), shadows=args.shadows, shadows_batch=args.shadows_batch == "on",
                                chain_first=args.capture_order == "chain-first", chain_stagger=args.chain_stagger,
                                opt_in_backward=args.opt_in_backward == "on", skip=frozenset(skip))
    sched = SCH.GraphedAdapterStep(wl, make_cfg(), L, optimizer=opt, world=world, comm=comm, device=dev)
    opt_in_bwd, shadows_main, shadows_opt = sched.opt_in_bwd, sched.shadows_main, sched.shadows_opt

    records = Recorder(only=LIVE, every=args.bracket_every)
    records.reserve(2 * len(wl["units"]) * args.steps + 16)

    sched.capture()                                  # (falls back to live launches, and says so, if the capture fails)
    args.graph = sched.graph_mode                    # (what the line reports is what ran)
    fwd_bwd_graph, bwd_graphs = sched.fwd_bwd_graph, sched.bwd_graphs

    def step(i, rec=None):
        sched.step(i, rec, time_comm=i >= args.warmup)

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
    replay_host_ms = sched.replay_host_ms() if rank == 0 else None

    # --verify-graph: the captured schedule (chains on forked streams, deferred dA_m on the hub, whatever the executor makes of it) against the
    # same launches live, one chain after the other on ONE stream, from the same activation state: y and dx bit for bit (deterministic
    # kernels), the flat gradient to the spread of its fp32 atomics
    graph_check = None
    if args.verify_graph and rank == 0 and opt is not None:
        # the DEFAULT schedule -- chains sharing gradient accumulators, AdamW slices and shadow rewrites on the hub behind
        # per-branch events, chain-first deferred optimizer slices -- against the same steps launched live from identical state: K steps
        # each way from the same master / moments / activations; a missing hub -> chain edge (a shadow rewrite or the gradient zeroing
        # racing a slower chain) shows up as different weights
        if not sched.graphed:
            raise SystemExit("--verify-graph: the step is not replayed from a graph")
        K = 3
        mut = [t for k in wl["keep"] for (acts, dacts, ys) in k[0] for t in list(dacts.values()) + list(ys)]
        snap = [t.clone() for t in mut]
        st0 = (wl["master"].clone(), wl["work"].clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.t)
        live_exec = SCH.GraphedAdapterStep(wl, SCH.ScheduleConfig(**{**make_cfg().__dict__, "graph": "off"}), L, optimizer=opt, world=world, comm=comm, device=dev)

        def restore():
            for t, s_ in zip(mut, snap):
                t.copy_(s_)
            wl["master"].copy_(st0[0]); wl["work"].copy_(st0[1]); opt.exp_avg.copy_(st0[2]); opt.exp_avg_sq.copy_(st0[3])
            opt.t = st0[4]
            opt.set_device_step(st0[4])
            bucket.zero_()
            run_shadows(lib, wl, c_void_p(torch.cuda.current_stream().cuda_stream), range(L))
            torch.cuda.synchronize()
        sh_units = [u for u in wl["units"] if u.fused][:8]
        results = []
        for ex in (sched, live_exec):
            restore()
            ex._epoch = 0                            # (the same sequence of dropout epochs either way)
            for i_ in range(K):
                ex.step(i_)
            torch.cuda.synchronize()
            results.append((wl["master"].clone(), wl["work"].clone(), [m["BwT"].clone() for u in sh_units for m in u.keep[0]], bucket.flat.clone(),
                            [t.clone() for t in mut[:8]]))
        (mg, wg, sg, fg, ag), (ml, wl_, sl, fl, al) = results
        graph_check = {"what": "%d optimizer steps of the captured default schedule vs the same steps launched live (chains back to back, dA_m / AdamW slices "
                               "on a side stream), from identical master / moments / activations" % K,
                       "steps": K, "master_rel_diff": float((mg - ml).norm() / ml.norm()), "master_max_abs_diff": float((mg - ml).abs().max()),
                       "work_max_abs_diff": float((wg.float() - wl_.float()).abs().max()),
                       "shadows_equal_work": all(bool(torch.equal(a_, b_)) for a_, b_ in zip(sg, sl)) if float((wg.float() - wl_.float()).abs().max()) == 0.0 else None,
                       "shadow_max_abs_diff": max([float((a_.float() - b_.float()).abs().max()) for a_, b_ in zip(sg, sl)] or [0.0]),
                       "grad_left_zero": float(fg.abs().max()) == 0.0 and float(fl.abs().max()) == 0.0,
                       "activations_max_rel_diff": max(float((a_.float() - b_.float()).norm() / b_.float().norm().clamp_min(1e-20)) for a_, b_ in zip(ag, al))}
        restore()
        del results, live_exec
    elif args.verify_graph and rank == 0:
        if fwd_bwd_graph is None and bwd_graphs is None:
            raise SystemExit("--verify-graph: the step is not replayed from a graph")
        mut = [t for k in wl["keep"] for (acts, dacts, ys) in k[0] for t in list(dacts.values()) + list(ys)]
        snap = [t.clone() for t in mut]

        def restore():
            for t, s_ in zip(mut, snap):
                t.copy_(s_)
            bucket.zero_()
        restore()
        torch.cuda.synchronize()
        sched.replay_only()
        torch.cuda.synchronize()
        g_graph = bucket.flat.clone()
        got = [t.clone() for t in mut]
        restore()
        sp_v = c_void_p(torch.cuda.current_stream().cuda_stream)
        sched.live_pass(sp_v)
        torch.cuda.synchronize()
        den = float(bucket.flat.abs().max())
        graph_check = {"activations_bit_identical": all(torch.equal(a_, b_) for a_, b_ in zip(got, mut)), "tensors_compared": len(mut),
                       "grad_max_abs_diff_over_max": float((g_graph - bucket.flat).abs().max()) / max(den, 1e-30),
                       "grad_max_abs": den, "grad_nonzero_frac": float((bucket.flat != 0).float().mean()),
                       "what": "graph replay vs the same launches live, chain after chain on one stream, from the same activation state"}
        restore()
        del got, g_graph

    # In-schedule marginals: the SAME captured schedule with one kernel family's launches left out (ScheduleConfig.skip:
    # the executor simply does not enqueue them -- no rebuilt library, no edited source), timed like the headline; marginal = base - without.
    # The default line carries the dominant family's (`roofline.in_schedule`); --ablate all / a list: the whole table (`ablation`).
    ablation = None
    if rank == 0 and world == 1 and not comm and sched.fwd_bwd_graph is not None and args.ablate != "off" and not args.verify_graph:
        fams = ["up_fwd"] if args.ablate == "dominant" else ([f for f in SCH.FAMILIES if f != "none"] if args.ablate == "all" else [f.strip() for f in args.ablate.split(",")])
        if opt is None:
            fams = [f for f in fams if f != "optimizer"]
        n_ab = max(5, min(args.steps, 20))

        def timed_schedule(skip, accept_ms=None, tries=3):
            # (every capture creates new streams, and which in-order hardware queues the executor maps a graph's lists onto depends on how many
            #  exist: every few captures the two chains of a graph share a queue and the step takes ~46 instead of ~30 ms.  A schedule is therefore
            #  captured up to `tries` times and the FASTEST capture counts -- the mapping is the runtime's, not the schedule's)
            best = None
            for _ in range(tries):
                sc = SCH.GraphedAdapterStep(wl, make_cfg(skip), L, optimizer=opt, world=world, comm=comm, device=dev)
                if not sc.capture():
                    return None
                for i_ in range(3):
                    sc.step(i_)
                torch.cuda.synchronize()
                t_ = time.perf_counter()
                for i_ in range(n_ab):
                    sc.step(i_)
                torch.cuda.synchronize()
                ms_ = (time.perf_counter() - t_) * 1e3 / n_ab
                best = ms_ if best is None else min(best, ms_)
                del sc
                if accept_ms is not None and best <= accept_ms:
                    break
            return best
        fam_bytes = {"down_fwd": lambda u: u.algo["moka_down_fwd"], "up_fwd": lambda u: u.algo["moka_up_fwd"], "up_bwd": lambda u: u.algo["moka_up_bwd"],
                     "cross_bwd": lambda u: u.algo["moka_cross_bwd"], "dx": lambda u: 2 * E * u.T * u.d_in * u.G, "dA": lambda u: E * u.T * u.d_in * u.G,
                     "shadows": lambda u: u.algo["moka_weight_shadows"] / max(1, args.chains), "optimizer": lambda u: 0, "none": lambda u: 0}
        rows = {}
        for fam in fams:
            try:
                base_ms = timed_schedule((), accept_ms=1.02 * ms_per_step)   # (re-measured beside every ablated schedule: the pair shares the box's clock state)
                wo_ms = timed_schedule((fam,), accept_ms=0.0)
            except Exception as exc:                 # (context for the line, never a reason to lose it: the headline has been measured)
                print(f"bench: in-schedule ablation of `{fam}` failed ({exc!r}); roofline.in_schedule is null in this line", file=sys.stderr)
                torch.cuda.synchronize()
                break
            if base_ms is None or wo_ms is None:
                continue
            nbytes = sum(fam_bytes[fam](u) for ch in wl["chains"] for u in ch["units"]) if fam != "optimizer" else 34 * wl["n_params"]
            marg = base_ms - wo_ms
            rows[fam] = {"base_ms": round(base_ms, 3), "without_ms": round(wo_ms, 3), "marginal_ms": round(marg, 3), "algorithmic_bytes": int(nbytes),
                         "achieved_GBps": round(nbytes / (marg * 1e-3) / 1e9, 1) if marg > 0 else None,
                         "frac": round(nbytes / (marg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if marg > 0 else None}
        ablation = {"how": "step time of the captured schedule minus the step time of the same schedule captured WITHOUT the family's launches "
                           "(moka_amd.schedule.ScheduleConfig.skip), %d timed steps each after 3 warm-up steps, base re-measured beside every row, fastest of "
                           "up to 3 captures per schedule (the runtime's queue mapping of a capture varies); "
                           "algorithmic bytes per step (SURVEY 8(d)) / marginal time" % n_ab,
                    "families": rows}

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
                sched.live_pass(sp_, records, None)
            torch.cuda.synchronize()
        tot, cnt, byt, per_shape = collect(records.items)         # the dominant entry point (LIVE)
        # every entry point, in one extra untimed pass (full bracketing would perturb the timed region)
        extra = Recorder()
        sched.live_pass(sp_, extra, extra)
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
            "comm_exposed_ms": (round(sum(a.elapsed_time(b) for a, b in sched.comm_ev) / max(1, len(sched.comm_ev)), 4) if sched.comm_ev else 0.0),
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
            "schedule": "moka_amd.schedule.GraphedAdapterStep over the C ABI (part-batch chains / deferred dA_m on a hub stream / one hipGraph, persistent "
                        "weight shadows rewritten behind the optimizer slices) on caller-owned buffers; the autograd path (moka_amd.parallel.attach + "
                        "MokaLinearFn under a decoder stack, GraphedTrainStep) is what --e2e measures",
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
                                  % (T // args.chains, args.chains)),
                         # the dominant family INSIDE the timed schedule (both chains, hub launches and all): marginal of leaving its launches out
                         "in_schedule": (dict(kernel=kern_name, family="up_fwd", **ablation["families"]["up_fwd"], how=ablation["how"])
                                         if (ablation and "up_fwd" in ablation["families"]) else None)},
            "entry_point_ms_per_pass": {n: round(tot_x[n], 3) for n in ENTRY},
            "forward_only": fwd_only,
            "kernels": table,
        }
        if ablation is not None and args.ablate != "dominant":
            out["ablation"] = ablation
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
