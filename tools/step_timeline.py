#!/usr/bin/env python
"""Kernel-development aid: where does the time of ONE decode layer go -- inside the kernels or between them?
The dev library (python vall-e-x_amd/_build.py --dev) stamps the 100 MHz wall clock in thread 0 of every workgroup of the
decode kernels; this script replays a one-layer step graph (QKV | dec_attn | reduce+LN | linear1 | linear2 | reduce+LN |
predict | sampler) on the state of a real batch-32 run and prints the timeline of the last replay.
   python tools/step_timeline.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import vallex_amd  # noqa: E402,F401
from vallex_amd import _capi  # noqa: E402

_capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev", "libvallex_hip.so")
from oracle import synth  # noqa: E402
from vallex_amd.models.vallex import VALLE  # noqa: E402

FR = 300
m = VALLE(1024, 16, 12, norm_first=True, add_prenet=False, prefix_mode=1, share_embedding=True, nar_scale_factor=1.0,
          prepend_bos=True, num_quantizers=8, engine_max_batch=32, engine_max_text=256, engine_max_prompt=320,
          engine_max_new=FR + 8)
m.to("cuda:0").load_state_dict(synth.vallex_state_dict(12, 0, eos_gain=0.0), strict=True)
eng = m.engine
batch = m.make_batch(bench.make_rows(0, 32))
eng.infer(batch, top_k=10, seed=1, force_eos_at=FR, sync_every=16)
eng.ar_prefill(batch)                               # live rows again (the run above ended with every row finished)
for _ in range(3):
    eng.ar_step(np.full(32, 5, np.int32))
us, _ = eng.bench_kernel(3, 20, 0)
print(f"one-layer step graph: {us:.1f} us per replay (20 replays back to back)")
st = np.zeros(8 * 512 * 8, np.uint64)
eng.lib.vx_dev_stamps(st.ctypes.data_as(C.POINTER(C.c_uint64)))
st = st.reshape(8, 512, 8).astype(np.int64)
names = ["QKV", "linear2", "predict", "linear1", "reduce+LN<16>", "reduce+LN<8>", "dec_attn", "sampler"]
order = [0, 6, 4, 3, 1, 5, 2, 7]                  # launch order inside the step
t0 = min(int(st[k, :, 0][st[k, :, 0] > 0].min()) for k in order if (st[k, :, 0] > 0).any())
print(f"{'kernel':14s} {'WGs':>4s} | first start  last start | stamp offsets from the kernel's first start (avg / max over workgroups), us")
prev_end = None
for k in order:
    live = st[k, :, 0] > 0
    if not live.any():
        continue
    s = st[k][live]
    first = int(s[:, 0].min())
    cols = []
    for j in range(1, 8):
        v = s[:, j][s[:, j] > 0]
        if v.size:
            cols.append(f"s{j}: {(v.mean() - first) / 100:5.2f}/{(v.max() - first) / 100:5.2f}")
    end = max(int(s[:, j].max()) for j in range(8))
    gap = "" if prev_end is None else f"  gap since previous kernel's last stamp: {(first - prev_end) / 100:5.2f}"
    print(f"{names[k]:14s} {int(live.sum()):4d} | {(first - t0) / 100:8.2f}   {(s[:, 0].max() - t0) / 100:8.2f}   | " + "  ".join(cols) + gap)
    prev_end = end
