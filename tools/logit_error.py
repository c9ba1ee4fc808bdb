#!/usr/bin/env python
"""Measured distance of the engine's logits to the LIVE reference's, per arithmetic of the full-sequence path (MI355X).

    python tools/logit_error.py [out.json]            (default gpurun_out/r03_logit_error.json; copy it to profiles/)

For each of vx_config.arith = f16x2 (default) / bf16x3 / f32 and each 12-layer golden (six 600-frame BASELINE-shape rows, the
1024-frame cap row, the second sentence of the sliding-window chain, two trained-like-weight rows) it records
  * ids_equal: all T x 8 ids of a full vx_infer run equal the reference's,
  * ar_max_abs: max |logit - reference| over the teacher-forced decode steps 0, 50, 100, ... (the reference's own ids are fed),
  * ar_flips: greedy rows: decode steps (of all T) whose arg-max is not the reference's id,
  * nar_max_abs[7]: the same per NAR stage over the first 16 generated rows,
  * phases_rerun_in_f32: f16x2 range-guard fallbacks taken,
next to the reference's own decision margins stored in the golden (smallest top-2 logit gap per greedy AR step / smallest
distance of the draw to a CDF boundary for top-k rows; smallest top-2 gap per NAR stage) and the logit scale.  The cached decode
step is exact fp32 in every mode, so the AR figures differ between modes only through the prefill (K/V cache, first logits).
Test infrastructure: uses the goldens and tests/_util.py; never part of the product path."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path):
    from oracle.make_golden import CHAIN_CASES, FULL_CASES, FULL_LOGIT_EVERY, LONG_CASES, TRAINED_CASES, chain_second
    from tests._util import case_model, golden, inputs_row, nar_logit_error, teacher_forced_logit_error
    kw = dict(max_new=1032, max_prompt=576, max_text=256, max_batch=2)
    jobs = []           # (name, case, golden, row, uniforms, codes key)
    for n in sorted(FULL_CASES) + sorted(LONG_CASES):
        c = dict(FULL_CASES, **LONG_CASES)[n]
        jobs.append((n, c, golden(n)) + inputs_row(c))
    cc = CHAIN_CASES["nl12_chain2_zh"]
    gc = golden("nl12_chain2_zh")
    jobs.append(("nl12_chain2_zh (sentence 2)", cc, gc) + inputs_row(dict(cc, useed=cc["useed2"]), inputs=chain_second(cc, gc["codes1"])))
    for n in sorted(TRAINED_CASES):
        jobs.append((n, TRAINED_CASES[n], golden(n)) + inputs_row(TRAINED_CASES[n]))
    res = {"how": __doc__.split("\n\n")[1].strip(), "logit_every": FULL_LOGIT_EVERY, "modes": {}, "goldens": {}}
    for n, c, g, row, us in jobs:
        res["goldens"][n] = dict(frames=int(g["codes"].shape[1]), top_k=c["top_k"], weights="trained-like" if c.get("trained") else "default init",
                                 longest_context=int(len(row["text"]) + 1 + row["prompt"].shape[0] + g["codes"].shape[1]),
                                 ref_ar_margin_min=float(g["ar_margin"].min()),
                                 ref_ar_margin_kind="top-2 logit gap" if c["top_k"] == 1 else "distance of u to the nearest CDF boundary (probability)",
                                 ref_nar_margin_min=[float(v) for v in g["nar_margin"]],
                                 ar_logit_abs_max=float(np.abs(g["ar_logits"]).max()), nar_logit_abs_max=float(np.abs(g["nar_logits"]).max()))
    for arith in ("f16x2", "bf16x3", "f32"):
        rows = {}
        for n, c, g, row, us in jobs:
            t0 = time.time()
            m = case_model(c, arith=arith, debug_taps=True, **kw)
            out = m.inference_batch([row], top_k=c["top_k"], uniforms=None if us is None else us[:, None], force_eos_at=c["force_eos_at"])[0]
            fb = m.engine.last_fallbacks()
            ar, flips = teacher_forced_logit_error(m, row, g, FULL_LOGIT_EVERY)
            codes, nar = nar_logit_error(m, row, g)
            rows[n] = dict(ids_equal=bool(out.shape == g["codes"][0].shape and (out == g["codes"][0]).all()),
                           ids_differing=int((out != g["codes"][0]).sum()) if out.shape == g["codes"][0].shape else -1,
                           ar_max_abs=ar, ar_flips=flips if c["top_k"] == 1 else None, nar_max_abs=nar,
                           nar_ids_equal=bool((codes == g["codes"][0]).all()), phases_rerun_in_f32=fb["prefill"] + fb["nar"])
            print(f"[{arith}] {n}: ids_equal {rows[n]['ids_equal']} AR {ar:.2e} NAR {max(nar):.2e} ({time.time() - t0:.1f} s)", flush=True)
        res["modes"][arith] = dict(engine=list(m.engine.arith_mode()), rows=rows,
                                   ar_max_abs=max(r["ar_max_abs"] for r in rows.values()),
                                   nar_max_abs=max(max(r["nar_max_abs"]) for r in rows.values()),
                                   all_ids_equal=all(r["ids_equal"] for r in rows.values()))
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)
    print("wrote", out_path)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r03_logit_error.json"))
