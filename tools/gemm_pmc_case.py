#!/usr/bin/env python
"""One launch set of each bf16x3 GEMM kernel at the NAR QKV shape, for a rocprofv3 --pmc pass (tools/rocpd_pmc_summary.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vallex_amd  # noqa: E402

eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
for k in (1, 2, 3):
    us, md = eng.bench_gemm(31616, 3072, 1024, k, 2)
    print(k, us, md, flush=True)
