#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace into a per-kernel stats CSV, like `--stats` prints.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.csv
"""
import re
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("kernel,calls,total_us,avg_us,min_us,max_us,pct")
    for n, c, s, a, mn, mx in rows:
        short = re.sub(r"\(.*", "", n)
        print(f"\"{short}\",{c},{s / 1e3:.1f},{a / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100.0 * s / total:.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
