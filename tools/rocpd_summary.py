#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2) rocpd sqlite result into a per-kernel stats table (markdown).

    python tools/rocpd_summary.py gpurun_out/prof1/r1_results.db > profiles/<name>.md
"""
import re
import sqlite3
import subprocess
import sys


def demangle(n):
    try:
        out = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        out = n
    out = re.sub(r"\(.*\)$", "", out)
    return out.replace("void ", "")


def main(path, top=25):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary of `{path}`\n")
    print(f"total kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, c, s, a, mn, mx in rows[:top]:
        print(f"| `{demangle(n)[:90]}` | {c} | {s/1e6:.3f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*s/total:.1f} |")


def by_grid(path, lo=0.0, hi=1.0):
    """Per (kernel, grid) rows -- the projections of a layer differ in grid -- over the dispatches between the fractions lo..hi of the trace."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    gcols = [c for c in cols if c.startswith("grid")]
    wcols = [c for c in cols if c.startswith("workgroup")]
    t0, t1 = cur.execute("select min(start), max(end) from kernels").fetchone()
    a, b = t0 + lo * (t1 - t0), t0 + hi * (t1 - t0)
    g = " * ".join(gcols) if gcols else "0"
    w = " * ".join(wcols) if wcols else "0"
    rows = cur.execute(f"select {name_col}, {g}, {w}, count(*), sum(end-start), avg(end-start), min(end-start) from kernels where start >= ? and end <= ? "
                       f"group by {name_col}, {g} order by sum(end-start) desc", (a, b)).fetchall()
    total = sum(r[4] for r in rows)
    print(f"# per (kernel, grid) rows of `{path}`, dispatches between {lo:.2f} and {hi:.2f} of the trace\n")
    print("| kernel | grid (threads) | workgroup | calls | total ms | avg us | min us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    for n, gs, ws, c, s_, av, mn in rows[:60]:
        print(f"| `{demangle(n)[:90]}` | {gs} | {ws} | {c} | {s_/1e6:.3f} | {av/1e3:.2f} | {mn/1e3:.2f} | {100*s_/total:.1f} |")


def sequence(path, lo=0.9, n=120):
    """The dispatches from the fraction `lo` of the trace on, in start order: name, grid, duration, idle time in front."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    g = " * ".join(c for c in cols if c.startswith("grid")) or "0"
    t0, t1 = cur.execute("select min(start), max(end) from kernels").fetchone()
    rows = cur.execute(f"select start, end, {name_col}, {g} from kernels where start >= ? order by start limit ?", (t0 + lo * (t1 - t0), int(n))).fetchall()
    print(f"# dispatch sequence of `{path}` from {lo:.3f} of the trace\n")
    print("| # | kernel | grid (threads) | us | idle before us |")
    print("|---:|---|---:|---:|---:|")
    prev = None
    for k, (st, en, nm, gs) in enumerate(rows):
        print(f"| {k} | `{demangle(nm)[:70]}` | {gs} | {(en - st)/1e3:.2f} | {((st - prev)/1e3 if prev else 0):.2f} |")
        prev = en


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "seq":
        sequence(sys.argv[1], *(float(v) for v in sys.argv[3:5]))
    elif len(sys.argv) > 2 and sys.argv[2] == "bygrid":
        by_grid(sys.argv[1], *(float(v) for v in sys.argv[3:5]))
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
