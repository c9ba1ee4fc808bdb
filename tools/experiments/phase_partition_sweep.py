#!/usr/bin/env python
"""Serving throughput of several 32-row batches in flight on one MI355X, by how the contexts share the CUs:
  one      one context, every CU (the headline configuration)
  slices   n contexts, each confined to 256 / n CUs for ALL its phases (vx_config.cu_mask alone; `bench.py --contexts n` until round 4)
  phases   n contexts decode on the same X-CU partition (own streams), ONE shared stream on the other 256 - X CUs takes the NAR
           stages and the Vocos head of whichever batch has finished decoding (vx_set_matrix_stream, ABI 5)
The contexts are built once; partitions are changed at run time (vx_set_cu_mask).  Same workload, fences and pass accounting as
bench.py's throughput mode (bench.contexts_measure).
   python tools/experiments/phase_partition_sweep.py [--steps 4] [--settings one,slices:2,phases:3:112,...]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import vallex_amd  # noqa: E402,F401
from oracle import synth  # noqa: E402
from vallex_amd._capi import MatrixStream, cu_partition, cu_split  # noqa: E402
from vallex_amd.models.vallex import VALLE  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=4, help="timed passes per context")
ap.add_argument("--frames", type=int, default=bench.FRAMES)
ap.add_argument("--settings", default="one,slices:2,phases:2:128,phases:3:96,phases:3:112,phases:3:128,phases:3:144,"
                                      "phases:4:112,phases:4:128,one")
args = ap.parse_args()
settings = [s.split(":") for s in args.settings.split(",")]
nmax = max([int(s[1]) for s in settings if len(s) > 1] + [1])

import torch  # noqa: E402
torch.cuda.set_device(0)
sd, vsd = synth.vallex_state_dict(bench.NUM_LAYERS, 0, eos_gain=0.0), synth.vocos_state_dict(2)
ctxs = []
for i in range(nmax):
    m = VALLE(1024, 16, bench.NUM_LAYERS, norm_first=True, add_prenet=False, prefix_mode=1, share_embedding=True,
              nar_scale_factor=1.0, prepend_bos=True, num_quantizers=8, engine_max_batch=bench.ROWS_PER_GPU, engine_max_text=256,
              engine_max_prompt=320, engine_max_new=max(args.frames, 64) + 8)
    m.to("cuda:0").load_state_dict(sd, strict=True)
    m.load_vocos_state_dict(vsd)
    ctxs.append((m.engine, m.make_batch(bench.make_rows(bench.ROWS_PER_GPU * i, bench.ROWS_PER_GPU)), m))
    bench.log(f"context {i} ready")


def sync():
    for e, _, _ in ctxs:
        e.synchronize()
    torch.cuda.synchronize()


for st in settings:
    kind, n = st[0], int(st[1]) if len(st) > 1 else 1
    ms = None
    for e, _, _ in ctxs:
        e.set_matrix_stream(None)
        e.set_cu_mask(0)
    if kind == "slices":
        for (e, _, _), mask in zip(ctxs[:n], cu_partition(n)):
            e.set_cu_mask(mask)
    elif kind == "phases":
        dec, mat = cu_split(int(st[2]))
        ms = MatrixStream(0, mat)
        for e, _, _ in ctxs[:n]:
            e.set_cu_mask(dec)
            e.set_matrix_stream(ms)
    elapsed, res = bench.contexts_measure([(e, b) for e, b, _ in ctxs[:n]], sync, args.frames, args.steps, 1,
                                          stagger=0.7 / n if n > 1 else 0.0)
    frames = sum(r[1] for r in res)
    out = dict(setting=":".join(st), contexts=n, passes=len(res), audio_s_per_s=round(frames / 75.0 / elapsed, 1),
               ms_per_pass=round(elapsed * 1e3 / len(res), 1), ar_ms_in_context=round(sum(r[2] for r in res) / len(res), 1),
               nar_ms_in_context=round(sum(r[3] for r in res) / len(res), 1))
    print(json.dumps(out), flush=True)
    for e, _, _ in ctxs:
        e.set_matrix_stream(None)
    if ms is not None:
        ms.close()
