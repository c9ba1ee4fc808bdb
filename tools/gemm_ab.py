#!/usr/bin/env python
"""A/B of the two candidate full-sequence GEMM arithmetics on the four NAR shapes (M = 31616 packed rows), interleaved
rounds in ONE process (cdna guide rule 24): bf16x3 (6 bf16 MFMAs per block) vs f16x2 (3 f16 MFMAs per block).
   python tools/gemm_ab.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vallex_amd  # noqa: E402

eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
M = int(os.environ.get("GEMM_M", 31616))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
kernels = [(0, "f32"), (2, "x3-dma"), (7, "f16x2 256x128"), (8, "f16x2 256x256")]
for (N, K) in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
    res = {k: [] for k, _ in kernels}
    diff = {}
    for r in range(rounds):
        for k, _ in kernels:
            us, md = eng.bench_gemm(M, N, K, k, 5)
            res[k].append(us)
            diff[k] = md
    print(f"N={N:5d} K={K:5d}  " + "  |  ".join(
        f"{name}: min {min(res[k]):8.1f} med {sorted(res[k])[len(res[k]) // 2]:8.1f} us {2.0 * M * N * K / min(res[k]) / 1e6:6.1f} TF diff-vs-f32 {diff[k]:.2e}"
        for k, name in kernels), flush=True)
