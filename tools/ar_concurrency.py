#!/usr/bin/env python
"""Experiment: does the latency-bound AR decode chain profit from a SECOND independent chain running beside it?
One engine with 32 rows (the product configuration) vs two engines (contexts, streams, host threads) with 16 rows each,
running at the same time.  Prints AR milliseconds per 600-frame batch and aggregate AR tokens/s.
   python tools/ar_concurrency.py"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import vallex_amd  # noqa: E402,F401
from oracle import synth  # noqa: E402
from vallex_amd.models.vallex import VALLE  # noqa: E402

FR = 600
sd = synth.vallex_state_dict(12, 0, eos_gain=0.0)


def mk(rows):
    m = VALLE(1024, 16, 12, norm_first=True, add_prenet=False, prefix_mode=1, share_embedding=True, nar_scale_factor=1.0,
              prepend_bos=True, num_quantizers=8, engine_max_batch=rows, engine_max_text=256, engine_max_prompt=320,
              engine_max_new=FR + 8)
    m.to("cuda:0").load_state_dict(sd, strict=True)
    return m, m.engine


def run(m, eng, rows, out, key, barrier=None):
    b = m.make_batch(rows)
    eng.infer(b, top_k=10, seed=1, force_eos_at=FR, sync_every=16)          # warm-up (graph capture)
    if barrier:
        barrier.wait()
    t0 = time.perf_counter()
    eng.infer(b, top_k=10, seed=2, force_eos_at=FR, sync_every=16)
    out[key] = (time.perf_counter() - t0, eng.last_stats())


rows = bench.make_rows(0, 32)
res = {}
m32, e32 = mk(32)
run(m32, e32, rows, res, "one32")
w, st = res["one32"]
print(f"1 x 32 rows: AR {st['ar_ms']:.1f} ms, NAR {st['nar_ms']:.1f} ms, wall {w * 1e3:.1f} ms -> {32 * FR / st['ar_ms'] * 1e3:.0f} AR tok/s", flush=True)
del m32, e32
ma, ea = mk(16)
mb, eb = mk(16)
bar = threading.Barrier(2)
ta = threading.Thread(target=run, args=(ma, ea, rows[:16], res, "a", bar))
tb = threading.Thread(target=run, args=(mb, eb, rows[16:], res, "b", bar))
ta.start(); tb.start(); ta.join(); tb.join()
for k in ("a", "b"):
    w, st = res[k]
    print(f"2 x 16 rows, engine {k}: AR {st['ar_ms']:.1f} ms, NAR {st['nar_ms']:.1f} ms, wall {w * 1e3:.1f} ms", flush=True)
ar = max(res["a"][1]["ar_ms"], res["b"][1]["ar_ms"])
print(f"2 x 16 concurrent: AR phase {ar:.1f} ms -> {32 * FR / ar * 1e3:.0f} AR tok/s aggregate", flush=True)
# alone, for reference
run(ma, ea, rows[:16], res, "a_alone")
w, st = res["a_alone"]
print(f"1 x 16 rows alone: AR {st['ar_ms']:.1f} ms -> {16 * FR / st['ar_ms'] * 1e3:.0f} AR tok/s", flush=True)
