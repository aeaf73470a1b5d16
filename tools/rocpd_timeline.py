#!/usr/bin/env python3
"""Per-dispatch timeline of a rocprofv3 rocpd sqlite result: for every kernel name the summed duration AND the summed gap to
the end of the previous dispatch (idle time in front of it), over the last `--tail` fraction of the trace.

    python tools/rocpd_timeline.py <results.db> [tail_fraction [skip_fraction]]

skip_fraction: leave out that last part of the trace (bench.py runs bracketed LIVE passes behind the timed graph replays: they use the
in-chain schedule; a window of graph replays only is e.g. tail 0.8, skip 0.5).
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


def main(path, tail=0.5, skip=0.0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select start, end, {name_col} from kernels order by start").fetchall()
    rows = rows[int(len(rows) * (1 - tail)):int(len(rows) * (1 - skip))]
    dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
    prev_end = rows[0][0]
    for s, e, n in rows:
        dur[n] += e - s
        gap[n] += max(0, s - prev_end)
        cnt[n] += 1
        prev_end = max(prev_end, e)
    span = rows[-1][1] - rows[0][0]
    print(f"# timeline of the dispatches from {1 - tail:.0%} to {1 - skip:.0%} of `{path}`: {len(rows)} dispatches over {span/1e6:.3f} ms "
          f"(kernels {sum(dur.values())/1e6:.3f} ms, gaps {sum(gap.values())/1e6:.3f} ms)\n")
    print("| kernel | calls | total ms | avg us | gap before: total ms | avg us |")
    print("|---|---:|---:|---:|---:|---:|")
    for n in sorted(dur, key=lambda k: -dur[k])[:30]:
        print(f"| `{demangle(n)[:80]}` | {cnt[n]} | {dur[n]/1e6:.3f} | {dur[n]/cnt[n]/1e3:.2f} | {gap[n]/1e6:.3f} | {gap[n]/cnt[n]/1e3:.2f} |")


def raw(path, n=60, frac=0.5):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    print(cols)
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select start, end, {name_col} from kernels order by start").fetchall()
    k = int(len(rows) * frac)
    rows = rows[k:k + n]
    t0 = rows[0][0]
    pe = t0
    for s, e, nm in rows:
        print(f"{(s - t0)/1e3:10.2f} us  dur {(e - s)/1e3:8.2f}  gap {(s - pe)/1e3:7.2f}  {demangle(nm)[:70]}")
        pe = e


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "raw":
        raw(sys.argv[1], int(sys.argv[3]) if len(sys.argv) > 3 else 60, float(sys.argv[4]) if len(sys.argv) > 4 else 0.5)
    else:
        main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5, float(sys.argv[3]) if len(sys.argv) > 3 else 0.0)
