#!/usr/bin/env python
"""Per-kernel stats of a rocprofv3 (rocpd sqlite) kernel trace, split by GRID SIZE -- the launches of one GEMM kernel over different
shapes (QKV / out_proj / linear1 / linear2 of a NAR layer differ in their number of column tiles) come out as separate rows.

    python tools/rocpd_by_grid.py x_results.db [name-substring ...] > by_grid.csv"""
import re
import sqlite3
import sys


def main(path, filters):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    gcols = [c for c in cols if re.search(r"grid|workgroup", c, re.I)]
    if not gcols:
        print("no grid columns; columns are:", cols)
        return
    sel = ", ".join(gcols)
    rows = db.execute(f"select {name_col}, {sel}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                      f"group by {name_col}, {sel} order by 1, {len(gcols) + 3} desc").fetchall()
    print("kernel," + ",".join(gcols) + ",calls,total_us,avg_us,min_us,max_us")
    for r in rows:
        short = re.sub(r"\(.*", "", r[0])
        if filters and not any(f in short for f in filters):
            continue
        g = r[1:1 + len(gcols)]
        c, s, a, mn, mx = r[1 + len(gcols):]
        print(f"\"{short}\"," + ",".join(str(x) for x in g) + f",{c},{s / 1e3:.1f},{a / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
