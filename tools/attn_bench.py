#!/usr/bin/env python
"""Kernel-development aid: time the full-sequence attention kernels (fp32 MFMA and bf16x3) and their component probes
at the NAR shape (32 sequences x 988 rows) and a ragged-edge shape, and report the bf16x3 kernel's max abs difference
to the fp32 kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vallex_amd
from vallex_amd import _capi

# the probes live in the tools-only build: python vall-e-x_amd/_build.py --dev
_capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev", "libvallex_hip.so")  # noqa: E402

eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
for B, L in ((32, 988), (8, 77)):
    flops = 4.0 * B * L * L * 1024
    for causal in (False, True):
        for base, kname in ((0, "f32"), (10, "x3 ")):
            row = [f"B={B} L={L} causal={int(causal)} {kname}"]
            probes = ((0, "full"), (1, "no-staging"), (2, "no-mfma"), (3, "no-softmax"))
            if base == 10:
                probes += ((4, "no-lds-store"), (5, "no-global-load"))
            for v, name in probes:
                us, md = eng.bench_attn(B, L, causal, base + v, 5)
                row.append(f"{name}: {us:8.1f} us ({flops / us / 1e6:6.1f} TF)" + (f" diff {md:.3g}" if v == 0 else ""))
            print("  |  ".join(row), flush=True)
