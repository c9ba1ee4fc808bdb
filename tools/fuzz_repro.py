#!/usr/bin/env python
"""Re-run ONE trial of tools/fuzz_soak.py (same seeded rows) and explain a differing row: the engine's ids for it (in the batch and
alone), and the ORACLE's decision margin at the first differing (frame, codebook) -- for a NAR codebook the gap between the two
largest logits of that stage's predict layer at that frame.  A gap at the level of fp32 reassociation noise (~1e-5 of logits of
size ~10) is a coin the reference does not decide reproducibly either; a large gap is a bug.
    python tools/fuzz_repro.py TRIAL [ROW]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from oracle import synth  # noqa: E402
from oracle.vallex_oracle import VallexOracle  # noqa: E402
from tests._util import get_model  # noqa: E402
from tools.fuzz_soak import EOS_GAIN, NL, SEED, trial_rows  # noqa: E402


def main():
    trial = int(sys.argv[1])
    rows, cols, top_k, temperature, cap = trial_rows(trial)
    m = get_model(NL, SEED, EOS_GAIN, max_new=64, max_prompt=256, max_text=128, max_batch=32)
    orc = VallexOracle(synth.vallex_state_dict(NL, SEED, EOS_GAIN), NL)
    outs = m.inference_batch(rows, top_k=top_k, temperature=temperature, uniforms=np.stack(cols, axis=1), force_eos_at=cap)
    print(f"trial {trial}: batch {len(rows)} top_k {top_k} T {temperature} cap {cap}; arithmetic {m.engine.arith_mode()}", flush=True)
    want = [int(sys.argv[2])] if len(sys.argv) > 2 else range(len(rows))
    for i in want:
        r, u = rows[i], cols[i]
        taps = {}
        ref = orc.inference(r["text"][None], np.array([len(r["text"])]), r["prompt"][None], r["enroll"], top_k=top_k, temperature=temperature,
                            prompt_language=r["prompt_language"], text_language=r["text_language"], uniforms=u, force_eos_at=cap, taps=taps)[0]
        alone = m.inference_batch([r], top_k=top_k, temperature=temperature, uniforms=u[:, None], force_eos_at=cap)[0]
        same_b = outs[i].shape == ref.shape and np.array_equal(outs[i], ref)
        same_a = alone.shape == ref.shape and np.array_equal(alone, ref)
        if same_b and same_a and len(sys.argv) <= 2:
            continue
        print(f"row {i}: S {len(r['text'])} Tp {r['prompt'].shape[0]} T {ref.shape[0]}  in batch == oracle: {same_b}  alone == oracle: {same_a}")
        for name, got in (("batch", outs[i]), ("alone", alone)):
            if got.shape != ref.shape:
                print(f"  {name}: {got.shape[0]} frames vs {ref.shape[0]}")
                continue
            d = np.argwhere(got != ref)
            for (t, q) in d[:6]:
                t, q = int(t), int(q)
                if q == 0:
                    lg = taps["ar_logits"][t].reshape(-1).double()
                else:
                    lg = taps["nar_logits"][q - 1][t].reshape(-1).double()
                top = torch.topk(lg, 3)
                print(f"  {name}: frame {t} codebook {q}: engine {int(got[t, q])} oracle {int(ref[t, q])}; oracle top-3 logits "
                      f"{[round(float(v), 6) for v in top.values]} at ids {[int(v) for v in top.indices]} -> top-2 gap {float(top.values[0] - top.values[1]):.3e}")


if __name__ == "__main__":
    main()
