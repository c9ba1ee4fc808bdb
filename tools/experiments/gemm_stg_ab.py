#!/usr/bin/env python
"""A/B of the staggered wave-group schedule of gemm_f16x2's 256 x 256 tile (template parameter STG, kernel id 14) against the plain
one (15), interleaved rounds in ONE process, on the four NAR shapes at M = 31616 (and the trimmed last layer's M = 19200):
    python tools/gemm_stg_ab.py [rounds]
Both give bit-identical sums (the max |diff| to the fp32-MFMA kernel is printed and must be equal)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vallex_amd  # noqa: E402

eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for M in (31616, 19200):
    for (N, K) in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
        res, diff = {}, {}
        for r in range(rounds):
            for k in (15, 14):
                us, md = eng.bench_gemm(M, N, K, k, 5)
                res.setdefault(k, []).append(us)
                diff[k] = md
        names = {15: "plain", 14: "staggered"}
        print(f"M={M:5d} N={N:5d} K={K:5d}: " + "  |  ".join(
            f"{names[k]}: min {min(v):7.1f} med {sorted(v)[len(v) // 2]:7.1f} us {2.0 * M * N * K / min(v) / 1e6:6.1f} TF diff {diff[k]:.3e}"
            for k, v in sorted(res.items(), reverse=True)) + f"  |  staggered / plain = {min(res[14]) / min(res[15]):.3f}", flush=True)
