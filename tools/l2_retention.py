#!/usr/bin/env python
"""Does a weight-streaming decode GEMM find in L2 what the previous launch read?  (vx_bench_kernel 2 / 3: the same 12.6 MB GEMM back to
back with plain / non-temporal loads; 1: the four different GEMMs of a layer = 50 MB, no reuse possible.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import vallex_amd  # noqa: E402,F401
from oracle import synth  # noqa: E402
from vallex_amd.models.vallex import VALLE  # noqa: E402

m = VALLE(1024, 16, 2, norm_first=True, add_prenet=False, prefix_mode=1, share_embedding=True, nar_scale_factor=1.0, prepend_bos=True,
          num_quantizers=8, engine_max_batch=32, engine_max_text=64, engine_max_prompt=64, engine_max_new=16)
m.to("cuda:0").load_state_dict(synth.vallex_state_dict(2, 0), strict=True)
rows = []
for i in range(32):
    a, t = synth.synth_prompt(20, 5, seed=i)
    rows.append(dict(text=np.concatenate([t[0], synth.synth_text(10, i)]), prompt=a[0], enroll=5, prompt_language="en", text_language="en"))
m.inference_batch(rows, top_k=1, force_eos_at=4)
for which, name in ((1, "4 different GEMMs of a layer (cold)"), (2, "same QKV GEMM back to back (cache hits)")):
    for _ in range(2):
        us, by = m.engine.bench_kernel(which, 200, 0)
        print(f"{name}: {us:.2f} us per launch, {by / us / 1e3:.0f} GB/s", flush=True)
