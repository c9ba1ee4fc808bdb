#!/usr/bin/env python
"""CPU baseline provenance (round 6): the reference's OWN VALLE.inference (imported from /root/reference; the tree
is only present in the builder's container, never on the GPU box) on row 0 of bench.py's workload -- 12 layers, the bench's seeded
weights, top-k 10 with the bench's injected uniforms (default_rng(1234)), EOS forced at 600 frames -- timed beside the oracle port
that bench.py's cpu_baseline leg times on the GPU box's host cores.

    python tools/cpu_reference.py [--frames 600] [--threads N]

Writes  profiles/r06_cpu_reference.json   kind "reference": cores, seconds per phase, audio-s/s, the port timed in the same
                                          process, port / reference ratio, ids_equal(reference, port)
        tests/golden/bench_row0.npz       ids (600, 8) of the LIVE reference for that row (+ the AR decision margins): bench.py's
                                          parity block compares the engine's ids against these as well as against the oracle
The hooks are the ones oracle/make_golden.py uses (topk_sampling is looked up at call time, models/vallex.py:569; torch.multinomial
replaced by the inverse CDF over the injected uniforms)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    import bench
    from oracle import synth
    from oracle.vallex_oracle import VallexOracle, inverse_cdf_sample
    cores = args.threads or bench.usable_cores()
    torch.set_num_threads(cores)
    frames = args.frames
    sd = synth.vallex_state_dict(bench.NUM_LAYERS, 0, eos_gain=0.0)
    r = bench.make_rows(0, 1)[0]
    us = np.random.default_rng(1234).random(frames + 1).astype(np.float32)

    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import models.vallex as V
    from models.vallex import VALLE
    m = VALLE(1024, 16, bench.NUM_LAYERS, norm_first=True, add_prenet=False, prefix_mode=1, share_embedding=True, nar_scale_factor=1.0,
              prepend_bos=True, num_quantizers=8).eval()
    res = m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    st = {"step": 0, "t_nar": None, "margin": []}
    orig_sampling, orig_multinomial = V.topk_sampling, torch.multinomial

    def hooked(logits, top_k=10, top_p=1.0, temperature=1.0):
        tok, lp = orig_sampling(logits, top_k=top_k, top_p=top_p, temperature=temperature)
        if st["step"] >= frames:
            tok = torch.full_like(tok, synth.EOS_ID)
        st["step"] += 1
        return tok, lp

    def multinomial(probs, num_samples=1, **kw):
        cdf = torch.cumsum(probs[0].double(), 0)
        st["margin"].append(float(torch.min(torch.abs(cdf / cdf[-1] - float(us[st["step"]])))))
        return torch.tensor([[inverse_cdf_sample(probs[0], float(us[st["step"]]))]], dtype=torch.long)

    orig_nar = m.nar_decoder.forward

    def nar_forward(*a, **k):
        if st["t_nar"] is None:
            st["t_nar"] = time.perf_counter()
        return orig_nar(*a, **k)

    m.nar_decoder.forward = nar_forward
    V.topk_sampling, torch.multinomial = hooked, multinomial
    text = torch.from_numpy(r["text"][None]).to(torch.int32)
    try:
        with torch.no_grad():
            t0 = time.perf_counter()
            codes = m.inference(text, torch.IntTensor([text.shape[-1]]), torch.from_numpy(r["prompt"][None]).to(torch.int32),
                                enroll_x_lens=r["enroll"], top_k=10, temperature=1.0, prompt_language=r["prompt_language"],
                                text_language=r["text_language"])
            t1 = time.perf_counter()
    finally:
        V.topk_sampling, torch.multinomial = orig_sampling, orig_multinomial
    ref_codes = codes[0].numpy().astype(np.int64)
    T = ref_codes.shape[0]
    ref_s = dict(ar=round(st["t_nar"] - t0, 2), nar=round(t1 - st["t_nar"], 2), total=round(t1 - t0, 2))
    print("reference:", ref_codes.shape, ref_s, flush=True)

    # the port, same process, same threads (what bench.py times on the GPU box)
    orc = VallexOracle(sd, bench.NUM_LAYERS)
    textl = torch.from_numpy(r["text"].astype(np.int64))
    prompts = torch.from_numpy(r["prompt"].astype(np.int64))
    with torch.no_grad():
        p0 = time.perf_counter()
        gen = orc.ar_generate(textl, prompts[:, 0], r["enroll"], r["prompt_language"], r["text_language"], 10, 1.0, us, frames, None)
        p1 = time.perf_counter()
        pcodes = orc.nar_generate(textl, prompts, gen, r["enroll"], r["prompt_language"], r["text_language"], None)
        p2 = time.perf_counter()
    port_s = dict(ar=round(p1 - p0, 2), nar=round(p2 - p1, 2), total=round(p2 - p0, 2))
    same = pcodes.shape == ref_codes.shape and bool((np.asarray(pcodes) == ref_codes).all())
    print("port:", port_s, "ids equal to the reference's:", same, flush=True)
    out = dict(kind="reference", what="the reference's own VALLE.inference (models/vallex.py:458-686, imported from /root/reference) on row 0 of "
                                      "bench.py's workload: 12 layers, S=%d, Tp=%d, %d frames, top-k 10, injected uniforms; no Vocos (the pip "
                                      "package is absent offline)" % (len(r["text"]), r["prompt"].shape[0], T),
               cores=cores, box="builder container (%d usable cores); the GPU box has no /root/reference" % bench.usable_cores(),
               frames=T, seconds=ref_s, value=round(T / 75.0 / (t1 - t0), 4), unit="audio-seconds/s", ar_tokens_per_s=round(T / ref_s["ar"], 2),
               port=dict(kind="port", seconds=port_s, value=round(T / 75.0 / (p2 - p0), 4)),
               port_over_reference=round((t1 - t0) / (p2 - p0), 3),
               ids_equal_reference_vs_port=same, ids_digest=bench.ids_digest([ref_codes]),
               min_ar_decision_margin=float(np.min(st["margin"])) if st["margin"] else None,
               torch=torch.__version__)
    json.dump(out, open(os.path.join(ROOT, "profiles", "r06_cpu_reference.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bench_row0.npz"), codes=ref_codes.astype(np.int16),
                        ar_margin=np.asarray(st["margin"], np.float64), frames=np.int64(frames))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
