#!/usr/bin/env python
"""Kernel-development aid: time attn_full and its component probes at the NAR shape (32 sequences x 988 rows)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vallex_amd  # noqa: E402

eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
B, L = 32, 988
flops = 4.0 * B * L * L * 1024
for causal in (False, True):
    row = [f"causal={int(causal)}"]
    for v, name in ((0, "full"), (1, "no-staging"), (2, "no-mfma"), (3, "no-softmax")):
        us = eng.bench_attn(B, L, causal, v, 5)
        row.append(f"{name}: {us:8.1f} us ({flops / us / 1e6:6.1f} TF)")
    print("  |  ".join(row), flush=True)
