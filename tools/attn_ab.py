#!/usr/bin/env python
"""A/B of a build variant of the library on the full-sequence attention kernels (and the NAR phase of the bench workload):
    python vall-e-x_amd/_build.py "--variant=nopk:attn_full_h2.hip,attn_full_x3.hip,attn_full.hip:-Xclang -target-feature -Xclang -packed-fp32-ops"
    python tools/attn_ab.py nopk
Each library runs in its own process (the binding loads one library per process)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import vallex_amd
from vallex_amd import _capi
lib = sys.argv[1]
if lib != "product":
    _capi.LIB_PATH = os.path.join(%r, "tools", "devx_" + lib, "libvallex_hip.so")
eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
for (b, L, causal) in ((32, 988, False), (32, 384, True), (8, 1326, False), (1, 983, False)):
    for var, name in ((20, "h2"), (10, "x3"), (0, "f32")):
        us, md = eng.bench_attn(b, L, causal, var, 10)
        print(f"{lib:8s} {name:4s} batch {b:2d} L {L:4d} {'causal' if causal else 'full  '}: {us:8.1f} us  max|diff to fp32| {md:.2e}", flush=True)
''' % (ROOT, ROOT)

for rnd in range(2):
    for lib in ["product"] + sys.argv[1:]:
        subprocess.run([sys.executable, "-c", CHILD, lib], check=False)
