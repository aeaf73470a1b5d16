#!/usr/bin/env python3
"""Concurrency of a rocprofv3 kernel trace (rocpd sqlite): how long 0 / 1 / 2 / 3+ kernels were in flight over a window of the trace,
which kernel families run alone, and per family the duration it has WITH its company.

    python tools/rocpd_overlap.py <results.db> [first_fraction last_fraction]      (default 0.15 0.5: graph replays of bench.py)
"""
import re
import sqlite3
import subprocess
import sys
from collections import defaultdict


def demangle(n):
    try:
        out = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        out = n
    return re.sub(r"\(.*\)$", "", out).replace("void ", "")


def main(path, lo=0.15, hi=0.5):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    qcol = next((c for c in cols if c in ("queue_id", "queue", "stream_id", "stream")), None)
    sel = f"start, end, {name_col}" + (f", {qcol}" if qcol else "")
    rows = cur.execute(f"select {sel} from kernels order by start").fetchall()
    rows = rows[int(len(rows) * lo):int(len(rows) * hi)]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for i, r in enumerate(rows):
        ev.append((r[0], 1, i))
        ev.append((r[1], -1, i))
    ev.sort()
    level_t = defaultdict(float)
    alone = defaultdict(float)        # family -> time it was the only kernel in flight
    active = set()
    prev = ev[0][0]
    for t, d, i in ev:
        dt = t - prev
        if dt > 0:
            level_t[min(len(active), 3)] += dt
            if len(active) == 1:
                alone[rows[next(iter(active))][2]] += dt
        prev = t
        if d > 0:
            active.add(i)
        else:
            active.discard(i)
    span = t1 - t0
    print(f"# concurrency over dispatches {lo:.0%}..{hi:.0%} of `{path}`: {len(rows)} dispatches, {span/1e6:.3f} ms" + (f", queue column `{qcol}`" if qcol else ""))
    print("\n| kernels in flight | ms | share |\n|---|---:|---:|")
    for k in sorted(level_t):
        print(f"| {k}{'+' if k == 3 else ''} | {level_t[k]/1e6:.3f} | {level_t[k]/span:.3f} |")
    dur, cnt = defaultdict(float), defaultdict(int)
    qs = defaultdict(set)
    for r in rows:
        dur[r[2]] += r[1] - r[0]
        cnt[r[2]] += 1
        if qcol:
            qs[r[2]].add(r[3])
    print("\n| kernel | calls | total ms | avg us | alone ms | queues |\n|---|---:|---:|---:|---:|---|")
    for n in sorted(dur, key=lambda k: -dur[k])[:24]:
        print(f"| `{demangle(n)[:70]}` | {cnt[n]} | {dur[n]/1e6:.3f} | {dur[n]/cnt[n]/1e3:.2f} | {alone[n]/1e6:.3f} | {sorted(qs[n]) if qcol else ''} |")


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0], float(a[1]) if len(a) > 1 else 0.15, float(a[2]) if len(a) > 2 else 0.5)
