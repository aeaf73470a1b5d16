#!/usr/bin/env python3
"""Copy what a round's measurement run left under gpurun_out/ into profiles/ (tracked) under the round's prefix.

    python tools/round_collect.py r04

Takes the bench lines of gpurun_out/<tag>/ (one JSON object per file: the line that starts with '{'), the kernel trace / timeline, the
PMC traffic summary, and digests the raw counter tables of gpurun_out/<tag>_mfma[_r64]/summary.md and gpurun_out/<tag>_stall/summary.md
into profiles/<tag>_pmc_mfma_lds.md and profiles/<tag>_pmc_stall.md (derived ratios first, the raw tables behind them)."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(path):
    hdr = None
    for l in open(path):
        c = [x.strip() for x in l.strip().strip('|').split('|')]
        if l.startswith('| moka kernel'):
            hdr = c
            continue
        if l.startswith('|---') or hdr is None:
            continue
        yield dict(zip(hdr, c))


def main(tag):
    src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
    for f in sorted(glob.glob(os.path.join(src, "bench*.json"))):
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if not lines:
            print("no JSON line in", f)
            continue
        json.dump(json.loads(lines[-1]), open(os.path.join(dst, f"{tag}_{os.path.basename(f)}"), "w"))
    # ---- the fixed term of the step, tracked as a first-class number: step = fixed_ms + ms_per_sequence * (sequences per GPU), least
    #      squares over the batch sizes a schedule was run on (one chain: b1 / b2 / b4 / b8; two part-batch chains: b2 / b3 / b4 / b8)
    def line(name):
        f = os.path.join(src, name + ".json")
        if not os.path.exists(f):
            return None
        ls = [l for l in open(f).read().splitlines() if l.startswith("{")]
        return json.loads(ls[-1]) if ls else None
    fits = {}
    for label, names in (("one chain (round 4's graph shape)", {1: "bench_b1", 2: "bench_b2_chains1", 4: "bench_b4_chains1", 8: "bench_b8_chains1"}),
                         ("two part-batch chains (the default where the batch has two sequences)", {2: "bench_b2", 3: "bench_b3", 4: "bench_driver_command", 8: "bench_b8"})):
        pts = [(bsz, line(n)["ms_per_step"]) for bsz, n in sorted(names.items()) if line(n) is not None]
        if len(pts) >= 2:
            n_ = len(pts)
            sx, sy = sum(p[0] for p in pts), sum(p[1] for p in pts)
            sxx, sxy = sum(p[0] * p[0] for p in pts), sum(p[0] * p[1] for p in pts)
            slope = (n_ * sxy - sx * sy) / (n_ * sxx - sx * sx)
            fits[label] = {"points_batch_ms": pts, "ms_per_sequence": round(slope, 3), "fixed_ms": round((sy - slope * sx) / n_, 3),
                           "slope_alone_frac_of_8TBps": round(17.305e6 * 2048 / (slope * 1e-3) / 8e12, 4)}
    if fits:
        json.dump({"what": "step time = fixed_ms + ms_per_sequence x sequences per GPU (least squares over the batch sizes of one run of tools/round_profile.sh, one box)",
                   "fits": fits}, open(os.path.join(dst, f"{tag}_fixed_term.json"), "w"), indent=1)
        print(json.dumps(fits, indent=1))
    for name in ("kernel_trace.md", "timeline.md", "pmc_fetch_size.md", "pmc_write_size.md", "pmc_traffic.json"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
    # ---- MFMA / LDS
    parts = [(os.path.join(ROOT, "gpurun_out", f"{tag}_mfma", "summary.md"), "r = 16"), (os.path.join(ROOT, "gpurun_out", f"{tag}_mfma_r64", "summary.md"), "r = 64")]
    if all(os.path.exists(p) for p, _ in parts):
        out = [f"# rocprofv3 --pmc MFMA / VALU / LDS counter passes on the {tag} build (`tools/pmc_mfma.sh`: one `--pmc` group per pass with `--kernel-trace` only)", "",
               "r = 16: `bench.py --layers 4 --steps 1 --warmup 1 --graph off` (7B widths, 8192 tokens); r = 64: `--model 13b --rank 64 --seq 4096 --batch 2 --layers 2`.",
               "MFMA utilisation = `SQ_VALU_MFMA_BUSY_CYCLES` / (4 x `SQ_BUSY_CU_CYCLES`) (four SIMDs per CU); VALU / LDS = share of `SQ_WAVE_CYCLES` (quad-cycles) with a VALU / LDS",
               "instruction active; bank conflicts = `SQ_LDS_BANK_CONFLICT` / `SQ_LDS_IDX_ACTIVE`.  The north star asks for the up-projection: `moka_yx_kernel<16>` (the fused",
               "interaction + up-projection, every launch of `moka_up_fwd_fused`) -- the contraction is free, the kernel is bound by the read-modify-write of y -- and",
               "`moka_yt_kernel<64>` at rank 64.", "",
               "| kernel | grid (threads) | avg us | MFMA busy | VALU active | LDS active | LDS bank conflicts |", "|---|---:|---:|---:|---:|---:|---:|"]
        for path, t in parts:
            for r in rows(path):
                try:
                    busy, cu = float(r['SQ_VALU_MFMA_BUSY_CYCLES']), float(r['SQ_BUSY_CU_CYCLES'])
                    valu, wc, lds = float(r['SQ_ACTIVE_INST_VALU']), float(r['SQ_WAVE_CYCLES']), float(r['SQ_ACTIVE_INST_LDS'])
                    bc = float(r['SQ_LDS_BANK_CONFLICT']) / max(1.0, float(r['SQ_LDS_IDX_ACTIVE']))
                except Exception:
                    continue
                if 'adamw' in r['moka kernel'] or 'shadows' in r['moka kernel']:
                    continue
                out.append(f"| {r['moka kernel']} ({t}) | {r['grid (threads)']} | {r['avg us']} | {busy / (4 * cu):.3f} | {valu / wc:.3f} | {lds / wc:.3f} | {bc:.3f} |")
        for path, t in parts:
            out += ["", f"## raw counters, {t}", ""] + open(path).read().splitlines()
        open(os.path.join(dst, f"{tag}_pmc_mfma_lds.md"), "w").write("\n".join(out) + "\n")
    # ---- stall (the headline configuration and, when it was run, the rank-64 one: tools/pmc_stall.sh <tag>_stall_r64 "--model 13b --rank 64 ...")
    for suffix, cmd in (("", "--layers 2"), ("_r64", "--model 13b --rank 64 --seq 4096 --batch 2 --layers 2 --defer-da off")):
        path = os.path.join(ROOT, "gpurun_out", f"{tag}_stall{suffix}", "summary.md")
        if not os.path.exists(path):
            continue
        out = [f"# rocprofv3 --pmc stall / memory-pipeline passes on the {tag} build (`tools/pmc_stall.sh {tag}_stall{suffix} \"{cmd}\"`: `bench.py {cmd} --steps 1 --warmup 1 --graph off`, one pass per counter group)", "",
               "Fractions of `SQ_WAVE_CYCLES` (quad-cycles): wait_any = `SQ_WAIT_ANY` (s_waitcnt / barrier), wait_inst = `SQ_WAIT_INST_ANY` (waiting to issue), active = `SQ_ACTIVE_INST_ANY`;",
               "latency = `TCP_TCC_READ_REQ_LATENCY` / `TCP_TCC_READ_REQ` (cycles per L2 read request); L2 hit share = `TCC_HIT` / (`TCC_HIT` + `TCC_MISS`).", "",
               "| kernel | grid (threads) | avg us | wait_any | wait_inst | active | of which VALU | latency | L2 hit share |", "|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
        for r in rows(path):
            try:
                wc, wa, wi, ac, va = (float(r[k]) for k in ('SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU'))
                lat = float(r['TCP_TCC_READ_REQ_LATENCY_sum']) / max(1.0, float(r['TCP_TCC_READ_REQ_sum']))
                hit = float(r['TCC_HIT_sum']) / max(1.0, float(r['TCC_HIT_sum']) + float(r['TCC_MISS_sum']))
            except Exception:
                continue
            if 'adamw' in r['moka kernel'] or 'shadows' in r['moka kernel']:
                continue
            out.append(f"| {r['moka kernel']} | {r['grid (threads)']} | {r['avg us']} | {wa / wc:.2f} | {wi / wc:.2f} | {ac / wc:.2f} | {va / wc:.2f} | {lat:.0f} | {hit:.2f} |")
        out += ["", "## raw counters", ""] + open(path).read().splitlines()
        open(os.path.join(dst, f"{tag}_pmc_stall{suffix}.md"), "w").write("\n".join(out) + "\n")
    for f in sorted(glob.glob(os.path.join(dst, f"{tag}_bench*.json"))):
        d = json.load(open(f))
        print(os.path.basename(f), d["value"], d["ms_per_step"], d.get("adapter_hbm_roofline_frac"), d.get("adapter_actual_hbm_frac"), d["roofline"]["achieved"], d.get("comm_exposed_ms"))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r04")
