#!/usr/bin/env python
"""Kernel-development aid: time the full-sequence GEMM kernels on the NAR shapes (M = 31616 packed rows).
   python tools/gemm_bench.py            # on an MI355X"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vallex_amd  # noqa: E402
from vallex_amd import _capi  # noqa: E402

# the probes live in the tools-only build: python vall-e-x_amd/_build.py --dev
_capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev", "libvallex_hip.so")

eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 31616
for (N, K) in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
    row = [f"N={N:5d} K={K:5d}"]
    # 6 = the product's tile choice, 7 / 8 = 256 x 128 / 256 x 256 tiles forced; 61-64 = probes of the 256 x 128 kernel;
    # 74 / 75 = probes 14 / 15: the 256 x 256 kernel with the LDS-DMA instructions issued between the MFMAs (behind the first eight /
    # behind every sixth; same sums as 8)
    for k, name in ((2, "x3-dma"), (6, "f16x2"), (7, "h2-256x128"), (8, "h2-256x256"), (74, "h2-256x256-dma-early"),
                    (79, "h2-256x256-dma-b2b(r02)"), (75, "h2-256x256-dma-spread"), (80, "spread-first-half"), (81, "spread-3/4"), (82, "spread-first-third"), (76, "h2-256x256-dma-spread-frag2"), (77, "h2-256x256-noDMA"), (78, "h2-256x256-noMFMA"),
                    (61, "h2-noDMA"),
                    (62, "h2-noMFMA"), (63, "h2-nofrag"), (64, "h2-nobarrier")):
        us, md = eng.bench_gemm(M, N, K, k, 5)
        row.append(f"{name}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF diff {md:.2e}")
    print("  |  ".join(row), flush=True)
