#!/usr/bin/env python
"""Kernel-development aid (dev library): shader-clock stamps of wave 0 of the first 256 workgroups of gemm_f16x2 around two
consecutive k-steps in steady state -- how long a k-step takes and where (waiting at the rendezvous for the operand DMA,
issuing the next DMA, reading fragments, issuing the 24 + 24 MFMAs).
   python tools/gemm_timeline.py [M N K]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vallex_amd  # noqa: E402
from vallex_amd import _capi  # noqa: E402

_capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev", "libvallex_hip.so")
eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (30400, 3072, 1024)
us, md = eng.bench_gemm(M, N, K, 6, 3)
print(f"gemm_f16x2 M={M} N={N} K={K}: {us:.1f} us per launch, {2.0 * M * N * K / us / 1e6:.1f} TF")
st = np.zeros(256 * 16, np.uint64)
eng.lib.vx_dev_gemm_stamps(st.ctypes.data_as(C.POINTER(C.c_uint64)))
st = st.reshape(256, 16).astype(np.int64)
names = ["wait for stage (vmcnt + barrier)", "issue DMA of the next stage", "issue 16 fragment reads", "issue 48 MFMAs"]
for half, base in (("even k-step", 0), ("odd k-step", 5)):
    d = [st[:, base + 1] - st[:, base], st[:, base + 2] - st[:, base + 1], st[:, base + 3] - st[:, base + 2],
         st[:, base + 4] - st[:, base + 3]]
    print(f"{half}: " + "  |  ".join(f"{n}: {x.mean():6.0f} clk (p10 {np.percentile(x, 10):5.0f}, p90 {np.percentile(x, 90):5.0f})"
                                    for n, x in zip(names, d)))
per = (st[:, 9] - st[:, 0]) / 2.0
print(f"k-step: {per.mean():.0f} clk (p10 {np.percentile(per, 10):.0f}, p90 {np.percentile(per, 90):.0f}); "
      f"ideal matrix-pipe time 2 waves x 24 MFMA x 32 clk = 1536 clk")
tot = st[:, 10] - st[:, 12]
epi = st[:, 11] - st[:, 10]
print(f"k loop of a tile ({K // 32} k-steps): {tot.mean():.0f} clk = {tot.mean() / (K // 32):.0f} per k-step incl. fill; epilogue {epi.mean():.0f} clk")
