#!/usr/bin/env python
"""Prints ROWS_TABLE of examples/c_bench.c: (prompt frames, prompt text ids, model language id) of rows 0..31 of bench.make_rows,
so the C client runs the geometry of the headline workload without numpy.  tests/test_abi_c.py compares the two.
   python tools/c_bench_rows.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

LANG = {"en": 0, "zh": 1, "ja": 2}          # models/vallex.py:439-443


def table(n=32):
    return [(int(r["prompt"].shape[0]), int(r["enroll"]), LANG[r["text_language"]]) for r in bench.make_rows(0, n)]


if __name__ == "__main__":
    t = table()
    print("static const int ROWS_TABLE[32][3] = {")
    for i in range(0, 32, 8):
        print("    " + ", ".join("{%d, %d, %d}" % v for v in t[i:i + 8]) + ("," if i < 24 else "};"))
