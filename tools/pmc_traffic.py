#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from the PMC counters, per the HBM / rocprofv3 section of
/opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass),
values in KiB per dispatch, FETCH_SIZE doubled (gfx950 reports half of the bytes of a wide coalesced read), WRITE_SIZE as is.

    python tools/pmc_traffic.py <fetch_results.db> <write_results.db> <layers> <T> > profiles/rNN_pmc_traffic.json

Sums the corrected bytes of every dispatch of `moka_yx_kernel<..>` (and `moka_yt_kernel<..>` / `moka_expand_kernel<.., true, ..>`: the y += hp Bw^T
kernels behind moka_up_fwd / moka_up_fwd_fused) and divides by the number of decoder layers the profiled run covered: bench.py turns that into bytes per
launch (a layer has 4 up-projection launches: q+k+v, o, gate+up, down) next to its algorithmic figure."""
import hashlib
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha256():
    """What ties a traffic summary to a build: the sha256 over the kernel source and the C header (bench.py recomputes it and refuses a
    summary measured on other kernels)."""
    sys.path.insert(0, ROOT)
    from moka_amd import build as _build
    h = hashlib.sha256()
    for path in _build.sources():
        h.update(open(path, "rb").read())
    return h.hexdigest()


def total(path, counter, pattern):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, grid_size, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name, grid_size",
                       (counter,)).fetchall()
    out, n, per_grid = 0.0, 0, {}
    for name, grid, c, v in rows:
        if re.search(pattern, name):
            out += v * 1024.0
            n += c
            per_grid[int(grid)] = {"dispatches": c, "avg_MiB": v / c / 1024.0}
    return out, n, per_grid


def main(fetch_db, write_db, layers, T, launches_per_layer=4):
    pat = r"moka_expand_kernel<\d+, \d+, true|moka_yt_kernel<|moka_yx_kernel<"          # the forms of the y += hp Bw^T pass (column- / token-owning / fused with the interaction)
    fb, fn, fg = total(fetch_db, "FETCH_SIZE", pat)
    wb, wn, wg = total(write_db, "WRITE_SIZE", pat)
    assert fn == wn and fn > 0 and fn % (layers * launches_per_layer) == 0, (fn, wn)
    passes = fn // (layers * launches_per_layer)          # forward passes the profiled run made (warm-up, timed, bracketed extras)
    per_layer = (2.0 * fb + wb) / (layers * passes)
    print(json.dumps({
        "kernel": "moka_yx_kernel<RP> (moka_up_fwd_fused; moka_yt_kernel / moka_expand_kernel<RP,NQ,true> where a unit runs the three-launch forward)",
        "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py; FETCH_SIZE x 2 "
                  "(gfx950 unit correction of MI355X_MICROARCH.md) + WRITE_SIZE; summed over the launches of one decoder layer",
        "kernel_source_sha256": source_sha256(),
        "library_sha256": hashlib.sha256(open(os.path.join(ROOT, "moka_amd", "libmoka_hip.so"), "rb").read()).hexdigest(),
        "tokens": T, "layers_profiled": layers, "forward_passes_profiled": passes, "dispatches": fn,
        "traffic_bytes_per_layer": per_layer,
        "fetch_x2_bytes_per_layer": 2.0 * fb / (layers * passes), "write_bytes_per_layer": wb / (layers * passes),
        "by_grid_fetch_MiB": fg, "by_grid_write_MiB": wg,
    }, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
