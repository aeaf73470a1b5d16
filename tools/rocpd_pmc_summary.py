#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd sqlite result (run where the db lives).

    python tools/rocpd_pmc_summary.py <results.db> [FETCH_SIZE|WRITE_SIZE]

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  gfx950 note (/opt/skills/guides/MI355X_MICROARCH.md,
section HBM): FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so the
`x2` column is the one to compare with a byte count; WRITE_SIZE is uncalibrated."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    return n[:70]


def main(path, counter="FETCH_SIZE"):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, count(*), avg(value), avg(duration), min(grid_size), max(grid_size) from counters_collection "
                       "where counter_name = ? group by kernel_name order by sum(value) desc", (counter,)).fetchall()
    print(f"| kernel | dispatches | avg {counter} MiB | x2 (gfx950 read correction) MiB | avg us |")
    print("|---|---:|---:|---:|---:|")
    for n, c, v, d, g0, g1 in rows[:14]:
        print(f"| `{short(n)}` | {c} | {v/1024:.2f} | {2*v/1024:.2f} | {d/1e3:.1f} |")
    # split the moka kernels by grid size (distinguishes the 4096- and 11008-wide launches)
    rows = cur.execute("select kernel_name, grid_size, count(*), avg(value), avg(duration) from counters_collection "
                       "where counter_name = ? and kernel_name like '%moka_%' group by kernel_name, grid_size order by kernel_name, grid_size", (counter,)).fetchall()
    print(f"\n| moka kernel | grid (threads) | dispatches | avg {counter} MiB | x2 MiB | avg us |")
    print("|---|---:|---:|---:|---:|---:|")
    for n, g, c, v, d in rows:
        print(f"| `{short(n)}` | {g} | {c} | {v/1024:.2f} | {2*v/1024:.2f} | {d/1e3:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "FETCH_SIZE")
