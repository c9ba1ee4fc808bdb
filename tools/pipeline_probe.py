#!/usr/bin/env python
"""Experiment: two contexts on ONE GPU, each running whole batches (AR decode, then the 7 NAR stages, then Vocos) from its
own host thread, half a period apart -- does the matrix-bound NAR phase of one batch hide under the latency/HBM-bound AR
decode of the other?  Prints milliseconds per 32-row batch for one context alone and for the two together.
   python tools/pipeline_probe.py [iterations]
   VX_CU_MASK_0=<64 hex> VX_CU_MASK_1=<64 hex> python tools/pipeline_probe.py      # with a static CU partition"""
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sd = synth.vallex_state_dict(12, 0, eos_gain=0.0)
vsd = synth.vocos_state_dict(2)


def mk():
    m = VALLE(1024, 16, 12, norm_first=True, add_prenet=False, prefix_mode=1, share_embedding=True, nar_scale_factor=1.0,
              prepend_bos=True, num_quantizers=8, engine_max_batch=32, engine_max_text=256, engine_max_prompt=320,
              engine_max_new=FR + 8)
    m.to("cuda:0").load_state_dict(sd, strict=True)
    m.load_vocos_state_dict(vsd)
    return m, m.engine


def one(eng, b, seed):
    codes = eng.infer(b, top_k=10, seed=seed, force_eos_at=FR, sync_every=16)
    eng.vocos_decode(codes, 2)
    return eng.last_stats()


def loop(eng, b, n, delay, log, key):
    time.sleep(delay)
    for i in range(n):
        t0 = time.perf_counter()
        st = one(eng, b, 10 + i)
        log.append((key, i, time.perf_counter() - t0, st["ar_ms"], st["nar_ms"]))


N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if os.environ.get("PROBE_MASKS", "1") == "1":
    per = 256 // N
    for i in range(N):
        os.environ[f"VX_CU_MASK_{i}"] = f"{((1 << per) - 1) << (per * i):064x}"
ctx = [mk() for _ in range(N)]
bat = [m.make_batch(bench.make_rows(32 * i, 32)) for i, (m, _) in enumerate(ctx)]
for (m, e), b in zip(ctx, bat):
    one(e, b, 1)                                     # warm-up: graph capture
log = []
t0 = time.perf_counter()
loop(ctx[0][1], bat[0], K, 0.0, log, "alone")
alone = (time.perf_counter() - t0) / K
print(f"context 0 alone ({256 // N if os.environ.get('PROBE_MASKS', '1') == '1' else 256} CUs): {alone * 1e3:7.1f} ms per batch  "
      f"({32 * FR / 75 / alone:6.1f} audio-s/s); AR {sum(r[3] for r in log) / K:.1f}  NAR {sum(r[4] for r in log) / K:.1f}", flush=True)
log = []
th = [threading.Thread(target=loop, args=(ctx[i][1], bat[i], K, 0.35 * i, log, f"c{i}")) for i in range(N)]
t0 = time.perf_counter()
for t in th:
    t.start()
for t in th:
    t.join()
wall = time.perf_counter() - t0
print(f"{N} contexts together: {wall / (N * K) * 1e3:7.1f} ms per batch ({N * K * 32 * FR / 75 / wall:6.1f} audio-s/s incl. ramp)", flush=True)
for r in sorted(log):
    print(f"    {r[0]} #{r[1]}: wall {r[2] * 1e3:7.1f}  AR {r[3]:7.1f}  NAR {r[4]:7.1f}", flush=True)
mid = [r for r in log if 0 < r[1] < K - 1]
if mid:
    cyc = sum(r[2] for r in mid) / len(mid)
    print(f"steady state: {cyc * 1e3:.1f} ms per cycle per context -> {cyc / N * 1e3:.1f} ms per batch "
          f"({N * 32 * FR / 75 / cyc:.1f} audio-s/s)", flush=True)
