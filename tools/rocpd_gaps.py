#!/usr/bin/env python
"""Where a decode step's time goes: per-kernel durations AND the gaps between consecutive kernels of a rocprofv3 kernel trace
(rocpd sqlite).  The AR step is a chain of dependent launches: its wall time is sum(durations) + sum(gaps).

    python tools/rocpd_gaps.py x_results.db [--window dec_sample_kernel] > gaps.csv

Kernels are ordered by start time; for each kernel name: calls, average duration, average gap to the NEXT kernel's start (start of the
next minus end of this one; negative = overlap), and the share of the summed (duration + gap).  With --window NAME only the
launches between consecutive occurrences of kernel NAME are kept per window (one decode step = from one sampler to the next) and the
average window length is printed too."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(.*", "", n)
    return n.replace("void vx::", "").replace("vx::", "").replace("(anonymous namespace)::", "")


def main(path, window):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    rows = [(short(n), s, e) for n, s, e in rows]
    stats = {}
    wins = []
    last_w = None
    for i, (n, s, e) in enumerate(rows):
        nxt = rows[i + 1][1] if i + 1 < len(rows) else None
        gap = (nxt - e) if nxt is not None else 0
        if gap > 200_000:          # > 200 us: a host-side pause (sync, phase change), not part of the chain
            gap = None
        d = stats.setdefault(n, [0, 0.0, 0.0, 0])
        d[0] += 1
        d[1] += e - s
        if gap is not None:
            d[2] += gap
            d[3] += 1
        if window and window in n:
            if last_w is not None and s - last_w < 5_000_000:
                wins.append(s - last_w)
            last_w = s
    tot = sum(v[1] + v[2] for v in stats.values()) or 1
    print("kernel,calls,avg_us,avg_gap_to_next_us,total_ms_incl_gaps,pct")
    for n, (c, dur, gap, gc) in sorted(stats.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print(f"\"{n}\",{c},{dur / c / 1e3:.2f},{(gap / gc / 1e3) if gc else 0:.2f},{(dur + gap) / 1e6:.1f},{100.0 * (dur + gap) / tot:.1f}")
    if wins:
        wins.sort()
        print(f"# windows between '{window}' launches: n={len(wins)} median {wins[len(wins) // 2] / 1e3:.1f} us mean {sum(wins) / len(wins) / 1e3:.1f} us")


if __name__ == "__main__":
    w = None
    args = sys.argv[1:]
    if "--window" in args:
        i = args.index("--window")
        w = args[i + 1]
        del args[i:i + 2]
    main(args[0], w)
