#!/usr/bin/env python3
"""Per-kernel averages of every PMC counter found in one or more rocprofv3 rocpd sqlite results.

    python tools/rocpd_pmc_multi.py <results.db> [<results.db> ...]

One row per (kernel, grid size); one column per counter (value per dispatch, summed over the
instances the profiler reports) + the average duration."""
import re
import sqlite3
import sys


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "")[:64]


def main(paths):
    table, counters = {}, []
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        for n, g, cn, c, v, d in cur.execute(
                "select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                "where kernel_name like '%moka_%' group by kernel_name, grid_size, counter_name"):
            if cn not in counters:
                counters.append(cn)
            row = table.setdefault((short(n), g), {"n": c, "us": d / 1e3})
            row[cn] = v
    print("| moka kernel | grid (threads) | dispatches | avg us | " + " | ".join(counters) + " |")
    print("|---|---:|---:|---:|" + "---:|" * len(counters))
    for (n, g), row in sorted(table.items()):
        print(f"| `{n}` | {g} | {row['n']} | {row['us']:.1f} | " + " | ".join(f"{row.get(c, float('nan')):.4g}" for c in counters) + " |")


if __name__ == "__main__":
    main(sys.argv[1:])
