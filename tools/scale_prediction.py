#!/usr/bin/env python3
"""profiles/<tag>_scale_prediction.json: what the first multi-GPU run of bench.py should show, written BEFORE it exists so that SCALE_rNN.json
can be diffed against it.  Inputs: the one-GPU line and the --force-comm line of a round's profile set
(gpurun_out/<tag>/bench_driver_command.json, bench_forcecomm.json).      python tools/scale_prediction.py r06"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line(path):
    ls = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(ls[-1])


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", tag)
    one, fc = line(os.path.join(src, "bench_driver_command.json")), line(os.path.join(src, "bench_forcecomm.json"))
    T = one["config"]["tokens_per_gpu_per_step"]
    d = fc["distributed"]
    sizes_mb = []
    per_layer = d["adapter_params"] / sum(d["bucket_layers"]) * 4 / 1e6
    for n in d["bucket_layers"]:
        sizes_mb.append(round(n * per_layer, 2))
    # ring all-reduce of S bytes on N ranks moves 2 (N - 1) / N * S per rank; bus bandwidths of this class of part over xGMI (7 links x ~153 GB/s,
    # point to point): 150-300 GB/s for 30-200 MB messages at N = 8, one link pair (50-100 GB/s) at N = 2
    busbw = {2: (50.0, 100.0), 4: (100.0, 200.0), 8: (150.0, 300.0)}
    out = {"what": "prediction for bench.py --gpus N (weak scaling, %d tokens per GPU per step, fp32 gradient payload, geometric buckets, two part-batch chains, one hub-shaped "
                   "graph per gradient bucket), written before any multi-GPU run exists" % T,
           "inputs": {"one_gpu_ms": one["ms_per_step"], "force_comm_one_rank_ms": fc["ms_per_step"], "force_comm_exposed_ms": fc["comm_exposed_ms"],
                      "bucket_layers_from_layer0_up": d["bucket_layers"], "bucket_MB_fp32": sizes_mb, "last_bucket_bytes": d["last_bucket_bytes"]},
           "assumptions": ["the backward walks the layers last -> first: a bucket ships when its first (lowest) layer has finished, the big buckets early",
                           "every bucket but the last has 2-3 x its own all-reduce time of backward left behind it, so it hides as long as RCCL's kernels get CUs beside two chip-filling chains",
                           "exposed = the tail bucket (one layer) + its AdamW slice (~20 us)", "RCCL contending with the chains for CUs / HBM is what a one-GPU run cannot show: anything beyond the band below is that"],
           "per_n": {}}
    for n, (lo, hi) in busbw.items():
        f = 2.0 * (n - 1) / n
        t = [[round(s * f / hi, 3), round(s * f / lo, 3)] for s in sizes_mb]            # ms (MB / (GB/s) = ms)
        tail = t[0]
        ms = [round(fc["ms_per_step"] + tail[0], 2), round(fc["ms_per_step"] + tail[1] + 0.1, 2)]
        out["per_n"][str(n)] = {"bucket_allreduce_ms_lo_hi": t, "comm_exposed_ms": [round(tail[0], 3), round(tail[1] + 0.02, 3)], "ms_per_step": ms,
                                "tokens_per_s_aggregate": [round(n * T / (ms[1] * 1e-3)), round(n * T / (ms[0] * 1e-3))],
                                "fraction_of_the_one_gpu_line": [round(one["ms_per_step"] / ms[1], 3), round(one["ms_per_step"] / ms[0], 3)]}
    out["per_n"]["1"] = {"ms_per_step": [one["ms_per_step"], one["ms_per_step"]], "tokens_per_s_aggregate": [round(one["value"]), round(one["value"])]}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_scale_prediction.json"), "w"), indent=1)
    print(json.dumps(out["per_n"], indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r06")
