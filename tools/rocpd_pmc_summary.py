#!/usr/bin/env python
"""Per-kernel averages of every PMC counter in a rocprofv3 rocpd sqlite DB -> CSV on stdout.

    python tools/rocpd_pmc_summary.py /tmp/x_results.db [--by-grid] [name-substring ...] > gpurun_out/pmc.csv

--by-grid: one row per (kernel, grid size) -- the launches of one GEMM kernel over different shapes come out separately.
"""
import re
import sqlite3
import sys


def main(path, filters, by_grid=False):
    db = sqlite3.connect(path)
    gcols = []
    if by_grid:
        cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
        gcols = [c for c in cols if re.search(r"grid", c, re.I)]
    sel = "".join(f", {c}" for c in gcols)
    rows = db.execute(f"select kernel_name, counter_name{sel}, count(*), avg(value), min(value), max(value), avg(end-start) "
                      f"from counters_collection group by kernel_name, counter_name{sel} order by 1, 2").fetchall()
    print("kernel,counter," + "".join(f"{c}," for c in gcols) + "dispatches,avg,min,max,avg_dispatch_us")
    for r in rows:
        n, cn = r[0], r[1]
        g = r[2:2 + len(gcols)]
        c, a, mn, mx, d = r[2 + len(gcols):]
        short = re.sub(r"\(.*", "", n)
        if filters and not any(f in short for f in filters):
            continue
        print(f"\"{short}\",{cn}," + "".join(f"{x}," for x in g) + f"{c},{a:.1f},{mn:.1f},{mx:.1f},{d / 1e3:.2f}")


if __name__ == "__main__":
    args = [a for a in sys.argv[2:] if a != "--by-grid"]
    main(sys.argv[1], args, by_grid="--by-grid" in sys.argv[2:])
