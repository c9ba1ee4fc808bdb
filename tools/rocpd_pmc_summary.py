#!/usr/bin/env python
"""Per-kernel averages of every PMC counter in a rocprofv3 rocpd sqlite DB -> CSV on stdout.

    python tools/rocpd_pmc_summary.py /tmp/x_results.db [name-substring ...] > gpurun_out/pmc.csv
"""
import re
import sqlite3
import sys


def main(path, filters):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(end-start) "
                      "from counters_collection group by kernel_name, counter_name order by 1, 2").fetchall()
    print("kernel,counter,dispatches,avg,min,max,avg_dispatch_us")
    for n, cn, c, a, mn, mx, d in rows:
        short = re.sub(r"\(.*", "", n)
        if filters and not any(f in short for f in filters):
            continue
        print(f"\"{short}\",{cn},{c},{a:.1f},{mn:.1f},{mx:.1f},{d / 1e3:.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
