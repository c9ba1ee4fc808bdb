"""Soak version of tests/test_gpu_fuzz.py (MI355X): seeded random ragged batches of 1 .. 32 rows against the CPU oracle, row by row,
for a wall-clock budget.  Batches of every size walk through the three decode chains (<= 4 rows, 5 .. 31, 32); prompts of 0 .. 200
frames, texts of 1 .. 40 ids, three languages, top-k 10 / greedy / unfiltered multinomial with injected uniforms, EOS-friendly
weights so that rows end at different steps.  A differing row is reported with the decision margin of the oracle at the first
differing step (a margin below the engine's logit distance is a coin the reference does not decide either) and the run goes on.

    python tools/fuzz_soak.py --seconds 540 --first-trial 100 > gpurun_out/fuzz_soak.log
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from oracle import synth  # noqa: E402
from oracle.vallex_oracle import VallexOracle  # noqa: E402
from tests._util import get_model  # noqa: E402

NL, SEED, EOS_GAIN = 2, 12, 2.5


def first_step_margin(orc, r, u, top_k, temperature, step):
    """distance of the uniform to the nearest CDF boundary (sampling) / top-2 logit gap (greedy) at AR step `step` of the oracle"""
    taps = {}
    orc.inference(r["text"][None], np.array([len(r["text"])]), r["prompt"][None], r["enroll"], top_k=top_k, temperature=temperature,
                  prompt_language=r["prompt_language"], text_language=r["text_language"], uniforms=u, force_eos_at=step + 1, taps=taps)
    lg = taps["ar_logits"][step].reshape(-1).double()
    if top_k == 1:
        t2 = torch.topk(lg, 2).values
        return float(t2[0] - t2[1])
    if temperature != 1.0:
        lg = lg / temperature
    if top_k > 0:
        kth = torch.topk(lg, top_k).values[-1]
        lg = torch.where(lg < kth, torch.full_like(lg, -float("inf")), lg)
    cdf = torch.cumsum(torch.softmax(lg, 0), 0)
    return float(torch.min(torch.abs(cdf / cdf[-1] - float(u[step]))))


def trial_rows(trial):
    """the seeded random ragged batch of one trial: (rows, uniform columns, top_k, temperature, cap) -- shared with tools/fuzz_repro.py"""
    rng = np.random.default_rng(9000 + trial)
    batch = int(rng.choice([1, 2, 3, 4, int(rng.integers(5, 32)), int(rng.integers(5, 32)), 32]))
    mode = int(rng.integers(0, 4))
    top_k, temperature = ((10, 1.0), (1, 1.0), (-100, 1.0), (10, 0.8))[mode]
    cap = int(rng.integers(8, 49))
    rows, cols = [], []
    for i in range(batch):
        tp = int(rng.choice([0, 1, 2, int(rng.integers(3, 201))]))
        sp = 0 if tp == 0 else int(rng.integers(1, 41))
        nt = int(rng.integers(1, 41))
        a, t = synth.synth_prompt(tp, sp, seed=int(rng.integers(1, 1 << 30)))
        txt = np.concatenate([t[0], synth.synth_text(nt, int(rng.integers(1, 1 << 30)))])
        rows.append(dict(text=txt, prompt=a[0], enroll=sp, prompt_language=("en", "zh", "ja")[int(rng.integers(0, 3))],
                         text_language=("en", "zh", "ja")[int(rng.integers(0, 3))]))
        cols.append(synth.uniforms(4096, 1, int(rng.integers(1, 1 << 30)))[:, 0])
    return rows, cols, top_k, temperature, cap


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=540.0)
    ap.add_argument("--first-trial", type=int, default=100)
    args = ap.parse_args()
    m = get_model(NL, SEED, EOS_GAIN, max_new=64, max_prompt=256, max_text=128, max_batch=32)
    orc = VallexOracle(synth.vallex_state_dict(NL, SEED, EOS_GAIN), NL)
    t0 = time.time()
    trial, rows_done, bad, by_chain = args.first_trial, 0, 0, {"<=4": 0, "5..31": 0, "32": 0}
    while time.time() - t0 < args.seconds:
        rows, cols, top_k, temperature, cap = trial_rows(trial)
        batch = len(rows)
        outs = m.inference_batch(rows, top_k=top_k, temperature=temperature, uniforms=np.stack(cols, axis=1), force_eos_at=cap)
        fb = m.engine.last_fallbacks()
        lens, diffs = [], []
        for i, (r, u) in enumerate(zip(rows, cols)):
            ref = orc.inference(r["text"][None], np.array([len(r["text"])]), r["prompt"][None], r["enroll"], top_k=top_k,
                                temperature=temperature, prompt_language=r["prompt_language"], text_language=r["text_language"],
                                uniforms=u, force_eos_at=cap)[0]
            lens.append(ref.shape[0])
            if outs[i].shape != ref.shape or not np.array_equal(outs[i], ref):
                n = min(outs[i].shape[0], ref.shape[0])
                d = np.argwhere(outs[i][:n] != ref[:n])
                step = int(d[0][0]) if len(d) else n
                cb = int(d[0][1]) if len(d) else -1
                if cb <= 0:
                    mg = first_step_margin(orc, r, u, top_k, temperature, step)
                else:      # a NAR codebook: the oracle's own gap between the two largest logits of that stage at that frame (argmax decision)
                    taps = {}
                    orc.inference(r["text"][None], np.array([len(r["text"])]), r["prompt"][None], r["enroll"], top_k=top_k,
                                  temperature=temperature, prompt_language=r["prompt_language"], text_language=r["text_language"],
                                  uniforms=u, force_eos_at=cap, taps=taps)
                    t2 = torch.topk(taps["nar_logits"][cb - 1][step].reshape(-1).double(), 2).values
                    mg = float(t2[0] - t2[1])
                diffs.append((i, outs[i].shape[0], ref.shape[0], step, cb, mg))
        rows_done += batch
        by_chain["<=4" if batch <= 4 else ("32" if batch == 32 else "5..31")] += 1
        print(f"trial {trial}: batch {batch:2d} top_k {top_k:4d} T {temperature} cap {cap:2d} lengths {min(lens)}..{max(lens)} "
              f"fallbacks {fb['prefill']}/{fb['nar']} -> {'OK' if not diffs else 'DIFF ' + str(diffs)}", flush=True)
        bad += len(diffs)
        trial += 1
    print(f"SOAK: {trial - args.first_trial} batches ({by_chain}), {rows_done} rows, {bad} differing rows, {time.time() - t0:.0f} s", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
