#!/usr/bin/env python
"""dev build only: shader clock sampled WHILE each f16x2 GEMM variant runs (VX_BENCH_CLOCK=1 makes vx_bench_gemm start a
one-wave probe kernel on a second stream that counts s_memtime ticks per 100 MHz reference tick)."""
import os
import sys

os.environ["VX_BENCH_CLOCK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vallex_amd  # noqa: E402
from vallex_amd import _capi  # noqa: E402

_capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev", "libvallex_hip.so")
eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
for k, name in ((0, "f32"), (2, "x3-dma"), (8, "h2-256x256"), (75, "h2-256x256-dma-spread"), (77, "h2-256x256-noDMA"),
                (78, "h2-256x256-noMFMA"), (61, "h2-noDMA"), (62, "h2-noMFMA"), (8, "h2-256x256")):
    us, md = eng.bench_gemm(31616, 3072, 1024, k, 20)
    print(f"{name}: {us:8.1f} us", flush=True)
