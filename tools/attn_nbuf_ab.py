#!/usr/bin/env python
"""Round 6: the fp32 full-sequence attention with two LDS buffers (two barriers per tile, rounds 1-5) against three (one barrier per
tile), interleaved in one process at the NAR shape (32 sequences x 988 rows), the prefill shape (32 x 384, prefix-LM mask) and a
ragged one; max |difference| of each against the two-buffer kernel must be exactly 0.
    python tools/attn_nbuf_ab.py [rounds]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vallex_amd  # noqa: E402

eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for B, L, causal in ((32, 988, False), (32, 384, True), (8, 77, False), (5, 1301, True)):
    flops = 4.0 * B * L * L * 1024
    t = {8: [], 9: []}
    for r in range(rounds):
        for v in (8, 9):
            us, md = eng.bench_attn(B, L, causal, v, 5)
            t[v].append(us)
            assert md == 0.0, (B, L, causal, v, md)
    m2, m3 = statistics.median(t[8]), statistics.median(t[9])
    print(f"B={B} L={L} causal={int(causal)}: two buffers {m2:8.1f} us ({flops / m2 / 1e6:6.1f} TF)   three buffers {m3:8.1f} us "
          f"({flops / m3 / 1e6:6.1f} TF)   ratio {m3 / m2:.4f}   max|diff| 0", flush=True)
