#!/usr/bin/env python
"""Kernel-development aid: the inside of dec_attn_qkv_kernel (one row: norm1 + QKV + split attention in one launch).  The dev library
(python vall-e-x_amd/_build.py --dev) stamps the 100 MHz wall clock in thread 0 of every workgroup: 0 start, 1 x = norm1(h) in LDS
(weight rows and the first K/V tile requested), 2 q ready, 3 K/V stream done, 4 partial written.  Replays a one-layer step graph on the
state of a real one-row run.
   python tools/step_timeline_b1.py [rows]"""
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

ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 1
FR = 300
m = VALLE(1024, 16, 12, norm_first=True, add_prenet=False, prefix_mode=1, share_embedding=True, nar_scale_factor=1.0,
          prepend_bos=True, num_quantizers=8, engine_max_batch=ROWS, engine_max_text=256, engine_max_prompt=320,
          engine_max_new=FR + 8)
m.to("cuda:0").load_state_dict(synth.vallex_state_dict(12, 0, eos_gain=0.0), strict=True)
eng = m.engine
batch = m.make_batch(bench.make_rows(0, ROWS))
eng.infer(batch, top_k=10, seed=1, force_eos_at=FR, sync_every=16)
eng.ar_prefill(batch)
for _ in range(3):
    eng.ar_step(np.full(ROWS, 5, np.int32))
us, _ = eng.bench_kernel(3, 20, 0)
print(f"one-layer step graph, {ROWS} row(s): {us:.1f} us per replay (20 replays back to back)")
st = np.zeros(8 * 512 * 8, np.uint64)
eng.lib.vx_dev_stamps(st.ctypes.data_as(C.POINTER(C.c_uint64)))
st = st.reshape(8, 512, 8).astype(np.int64)
s = st[6][st[6, :, 0] > 0]
first = int(s[:, 0].min())
print(f"dec_attn_qkv: {len(s)} workgroups, launch spread {(s[:, 0].max() - first) / 100:.2f} us")
for j, what in ((1, "x in LDS, requests out"), (2, "q ready"), (3, "stream done"), (4, "partial written")):
    v = s[:, j][s[:, j] > 0]
    if v.size:
        print(f"  stamp {j} ({what:24s}): avg {(v.mean() - first) / 100:6.2f}  min {(v.min() - first) / 100:6.2f}  max {(v.max() - first) / 100:6.2f} us after the first workgroup's start")
for k, nm in ((1, "linear2 / out_proj (N=1024 skinny GEMM)"), (3, "linear1"), (2, "predict"), (7, "sampler")):
    t = st[k][st[k, :, 0] > 0]
    if len(t):
        print(f"{nm}: first start {(t[:, 0].min() - first) / 100:7.2f} us, last stamp {(t.max() - first) / 100:7.2f} us")
