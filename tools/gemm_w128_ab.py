#!/usr/bin/env python
"""Kernel-development aid (tools-only build: python vall-e-x_amd/_build.py --dev): the 128 x 128 per-wave-tile f16x2 GEMM (probe 23:
4 waves, 256 accumulator registers, two fragment sets, mid-tile rendezvous, requests two K tiles ahead) against the product's
256 x 256 kernel (8 waves of 64 x 128) on the four NAR shapes, interleaved rounds in one process, on random and on all-zero operands
(quiet operands take the power limit away: what is left is the schedule), with the shader clock held under each.
    python tools/gemm_w128_ab.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vallex_amd  # noqa: E402
from vallex_amd import _capi  # noqa: E402

_capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev", "libvallex_hip.so")
eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
M = 31616
for data in ("random", "zero"):
    os.environ["VX_BENCH_GEMM_DATA"] = data
    for (N, K) in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
        for r in range(rounds):
            row = [f"{data:6s} N={N:5d} K={K:5d}"]
            for k, name in ((8, "product 256x256 / 8 waves"), (83, "probe 23: 4 waves x 128x128")):
                us, md, mhz = eng.bench_gemm_clock(M, N, K, k, 6)
                row.append(f"{name}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF  {mhz:5.0f} MHz  diff {md:.3e}")
            print("  |  ".join(row), flush=True)
