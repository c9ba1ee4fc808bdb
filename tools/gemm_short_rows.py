#!/usr/bin/env python
"""f16x2 GEMM tile choice by row count: 256 x 256 (kernel 8), 256 x 128 (kernel 7) and the 128 x 128 tiles of the
short-row-set kernel (kernel 9, two 4-wave workgroups per CU) on the four NAR shapes, M from one utterance (983) to a full batch.  The
max |difference| to the fp32-MFMA kernel must be the same number for all three (the per-element accumulation order does not depend on
the tile shape).   python tools/gemm_short_rows.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vallex_amd  # noqa: E402

eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
for M in (384, 983, 1966, 2949, 4915, 7864, 11796, 15728, 31616):
    for (N, K) in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
        a = eng.bench_gemm(M, N, K, 7, 10)
        b = eng.bench_gemm(M, N, K, 9, 10)
        c = eng.bench_gemm(M, N, K, 8, 10)
        d = eng.bench_gemm(M, N, K, 6, 10)
        e = eng.bench_gemm(M, N, K, 10, 10)                   # 128 x 128 with two LDS stages forced (kernel 9 takes four when tiles <= 256)
        same = "SAME" if a[1] == b[1] == c[1] == e[1] else "DIFFERENT"
        best = min((a[0], "256x128"), (b[0], "128x128"), (c[0], "256x256"))
        print(f"M={M:5d} N={N:5d} K={K:5d}  256x128 {a[0]:7.1f}  128x128 {b[0]:7.1f} (2 stages {e[0]:7.1f})  256x256 {c[0]:7.1f}  auto {d[0]:7.1f} us   best {best[1]}  "
              f"auto/best {d[0] / best[0]:.2f}  {same} max-diff {a[1]:.2e}", flush=True)
