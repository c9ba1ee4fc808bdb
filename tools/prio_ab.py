#!/usr/bin/env python
"""A/B of the wave-priority variants (guide T5) of the two MFMA kernels of the NAR phase, interleaved rounds in ONE process:
    python vall-e-x_amd/_build.py --dev && python tools/prio_ab.py [rounds]
attn_full_h2: variant 20 = no priority (product), 21 = s_setprio(1) around every MFMA, 22 = static priority for the odd workgroups,
23 = priority through the PV phase.  gemm_f16x2 256 x 256: kernel 8 = none, 12 = static priority for waves 4-7, 13 = around every
k16 step's MFMAs.  Same results in every variant (max |diff| to the fp32 kernel is printed)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vallex_amd  # noqa: E402
from vallex_amd import _capi  # noqa: E402

_capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev", "libvallex_hip.so")
eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3

for (b, L, causal) in ((32, 988, False), (32, 384, True), (8, 1326, False), (1, 983, False)):
    res, diff = {}, {}
    for r in range(rounds):
        for var in (20, 21, 22, 23):
            us, md = eng.bench_attn(b, L, causal, var, 10)
            res.setdefault(var, []).append(us)
            diff[var] = md
    print(f"attn_full_h2 batch {b:2d} L {L:4d} {'causal' if causal else 'full  '}: " + "  |  ".join(
        f"prio{var - 20}: min {min(v):7.1f} med {sorted(v)[len(v) // 2]:7.1f} us diff {diff[var]:.1e}" for var, v in sorted(res.items())),
        flush=True)

M = int(os.environ.get("GEMM_M", 31616))
for (N, K) in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
    res, diff = {}, {}
    for r in range(rounds):
        for k in (8, 12, 13):
            us, md = eng.bench_gemm(M, N, K, k, 5)
            res.setdefault(k, []).append(us)
            diff[k] = md
    names = {8: "none", 12: "static w4-7", 13: "per k16"}
    print(f"gemm_f16x2 256x256 N={N:5d} K={K:5d}: " + "  |  ".join(
        f"{names[k]}: min {min(v):7.1f} med {sorted(v)[len(v) // 2]:7.1f} us {2.0 * M * N * K / min(v) / 1e6:6.1f} TF diff {diff[k]:.1e}"
        for k, v in sorted(res.items())), flush=True)
