"""Generate tests/golden/*.npz by running the LIVE reference (/root/reference).

Run in the build container only (the reference tree does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden

What it does, per case in CASES:
  1. builds the reference `VALLE` (models/vallex.py:405) with the inference-time
     constructor args of utils/generation.py:67-78 (num_layers per case),
  2. `load_state_dict(strict=True)` of oracle.synth.vallex_state_dict -> proves the
     synthetic dict has the reference's exact key/shape layout,
  3. calls the reference's own `VALLE.inference` (models/vallex.py:458) on CPU,
     with optional hooks that the survey used:
       - `models.vallex.topk_sampling` wrapper that records logits and forces EOS
         at a chosen step (emulates a trained model's termination),
       - `torch.multinomial` replaced by inverse-CDF over injected uniforms,
  4. stores token ids (1,T,8), per-step AR logits (first few steps) and NAR
     stage-0 logits so tests can check the oracle -- and through it the HIP
     engine -- without the reference being present.

The committed fixtures are small (ids + a few logit rows); weights are re-created
from the seed on any box.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")

from . import synth                                 # noqa: E402
from .vallex_oracle import inverse_cdf_sample       # noqa: E402

# name -> dict(num_layers, seed, eos_gain, preset | (tp,sp), n_text, lang, top_k, force_eos_at, uniforms_seed)
CASES = {
    # 2-layer, natural EOS via eos_gain, greedy
    "nl2_greedy_eos": dict(num_layers=2, seed=0, eos_gain=2.5, preset="librispeech_1", n_text=12, lang="en",
                           top_k=1, force_eos_at=None, useed=None),
    # 2-layer, runs into the 16*S cap (S = 8+6)
    "nl2_greedy_cap": dict(num_layers=2, seed=1, eos_gain=1.0, synth_prompt=(20, 8), n_text=6, lang="zh",
                           prompt_lang="zh", top_k=1, force_eos_at=None, useed=None),
    # 2-layer, forced EOS at 40, ja prompt, per-token text languages (code-switch list)
    "nl2_force40_mixlang": dict(num_layers=2, seed=2, eos_gain=1.0, preset="cafe", n_text=20, lang="mix",
                                top_k=1, force_eos_at=40, useed=None),
    # 2-layer, top-k=10 with injected uniforms
    "nl2_topk10": dict(num_layers=2, seed=3, eos_gain=1.0, preset="paimon", n_text=16, lang="en",
                       top_k=10, force_eos_at=48, useed=1234),
    # 2-layer, unfiltered multinomial (API default top_k=-100) with injected uniforms, temperature 0.8
    "nl2_full_multinomial": dict(num_layers=2, seed=4, eos_gain=1.0, synth_prompt=(30, 10), n_text=10, lang="en",
                                 prompt_lang="en", top_k=-100, temperature=0.8, force_eos_at=32, useed=77),
    # best_of=3 beams (UI path, launch-ui.py:294 uses 5): top-k sampling, natural EOS at different lengths per beam
    "nl2_bestof3": dict(num_layers=2, seed=5, eos_gain=1.6, preset="paimon", n_text=10, lang="en", top_k=10,
                        force_eos_at=60, useed=4321, best_of=3),
    "nl2_bestof3_worst": dict(num_layers=2, seed=5, eos_gain=1.6, preset="paimon", n_text=10, lang="en", top_k=10,
                              force_eos_at=60, useed=4321, best_of=3, length_penalty=0.7, return_worst=True),
    # full 12-layer model, BASELINE config-1 shape cut to 24 frames (keeps the fixture cheap to re-verify)
    "nl12_c1_short": dict(num_layers=12, seed=0, eos_gain=1.0, preset="librispeech_1", n_text=100, lang="en",
                          top_k=1, force_eos_at=24, useed=None),
}

# "sharp" attention (attn_gain 3: score std ~4.5 instead of ~0.5): a test bed that reacts to K/V and to the score arithmetic,
# which the default random init (nearly uniform softmax) does not.  Kept apart from CASES: the GPU parity suite iterates CASES.
SHARP_CASES = {
    "nl2_sharp_greedy": dict(num_layers=2, seed=8, eos_gain=1.0, attn_gain=3.0, preset="librispeech_1", n_text=14, lang="en",
                             top_k=1, force_eos_at=48, useed=None),
    "nl2_sharp_topk10": dict(num_layers=2, seed=9, eos_gain=1.0, attn_gain=3.0, preset="cafe", n_text=12, lang="ja",
                             top_k=10, force_eos_at=40, useed=555),
}

# VALLE.continual (models/vallex.py:688-787): text ids + a full (T, 8) code matrix; NAR stages only
CONTINUAL_CASES = {
    "nl2_continual": dict(num_layers=2, seed=6, eos_gain=1.0, n_text=14, frames=61),          # prefix_len = 30
    "nl2_continual_long": dict(num_layers=2, seed=7, eos_gain=1.0, n_text=9, frames=470),     # prefix_len = 225 (3 s cap)
}

CODE2LANG = {0: "zh", 1: "ja", 2: "en"}      # macros.py:15-19 via utils/generation.py:114-115


def load_preset(name):
    d = np.load(os.path.join(GOLD, "presets", name + ".npz"))
    return d["audio_tokens"], d["text_tokens"], CODE2LANG[int(d["lang_code"])]


def case_inputs(c):
    if "preset" in c:
        a, t, pl = load_preset(c["preset"])
    else:
        a, t = synth.synth_prompt(*c["synth_prompt"], seed=c["seed"])
        pl = c["prompt_lang"]
    txt = synth.synth_text(c["n_text"], c["seed"])[None]
    text = np.concatenate([t, txt], -1)
    if c["lang"] == "mix":
        langs = [("en", "zh", "ja")[i % 3] for i in range(c["n_text"])]
    else:
        langs = c["lang"]
    return a, t, text, pl, langs


def run_reference(c):
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import models.vallex as V
    from models.vallex import VALLE

    m = VALLE(1024, 16, c["num_layers"], norm_first=True, add_prenet=False, prefix_mode=1,
              share_embedding=True, nar_scale_factor=1.0, prepend_bos=True, num_quantizers=8).eval()
    sd = synth.vallex_state_dict(c["num_layers"], c["seed"], c["eos_gain"], c.get("attn_gain", 1.0))
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    a, t, text, pl, langs = case_inputs(c)

    rec = {"logits": [], "step": 0}
    orig_sampling = V.topk_sampling
    orig_multinomial = torch.multinomial
    nbeam = c.get("best_of", 1)
    us = None if c["useed"] is None else synth.uniforms(4096, nbeam, c["useed"])

    def hooked(logits, top_k=10, top_p=1.0, temperature=1.0):
        if rec["step"] < 8:
            rec["logits"].append(logits[0].detach().clone().numpy())
        tok, lp = orig_sampling(logits, top_k=top_k, top_p=top_p, temperature=temperature)
        if c["force_eos_at"] is not None and rec["step"] >= c["force_eos_at"]:
            tok = torch.full_like(tok, synth.EOS_ID)
        rec["step"] += 1
        return tok, lp

    def multinomial(probs, num_samples=1, **kw):
        if us is None:
            return orig_multinomial(probs, num_samples, **kw)
        return torch.tensor([[inverse_cdf_sample(probs[i], float(us[rec["step"], i]))] for i in range(probs.shape[0])],
                            dtype=torch.long)

    nar_logits = []
    orig_nar_pred = m.nar_predict_layers[0].forward

    def nar_pred(xx):
        out = orig_nar_pred(xx)
        nar_logits.append(out.detach().clone().numpy())
        return out

    m.nar_predict_layers[0].forward = nar_pred
    V.topk_sampling = hooked
    torch.multinomial = multinomial
    try:
        with torch.no_grad():
            codes = m.inference(torch.from_numpy(text).to(torch.int32), torch.IntTensor([text.shape[-1]]),
                                torch.from_numpy(a).to(torch.int32), enroll_x_lens=t.shape[-1],
                                top_k=c["top_k"], temperature=c.get("temperature", 1.0),
                                prompt_language=pl, text_language=langs, best_of=nbeam,
                                length_penalty=c.get("length_penalty", 1.0), return_worst=c.get("return_worst", False))
    finally:
        V.topk_sampling = orig_sampling
        torch.multinomial = orig_multinomial
    return dict(codes=codes.numpy().astype(np.int64),
                ar_logits=np.stack(rec["logits"]).astype(np.float32),
                nar_logits0=nar_logits[0][0, :16].astype(np.float32))


def continual_inputs(c):
    text = synth.synth_text(c["n_text"], c["seed"])[None]
    y, _ = synth.synth_prompt(c["frames"], 1, seed=c["seed"] + 100)
    return text, y


def run_reference_continual(c):
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from models.vallex import VALLE

    m = VALLE(1024, 16, c["num_layers"], norm_first=True, add_prenet=False, prefix_mode=1,
              share_embedding=True, nar_scale_factor=1.0, prepend_bos=True, num_quantizers=8).eval()
    sd = synth.vallex_state_dict(c["num_layers"], c["seed"], c["eos_gain"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    text, y = continual_inputs(c)
    with torch.no_grad():
        codes = m.continual(torch.from_numpy(text).to(torch.int32), torch.IntTensor([text.shape[-1]]),
                            torch.from_numpy(y).to(torch.int64))
    return dict(codes=codes.numpy().astype(np.int64))


def main(only=None):
    os.makedirs(GOLD, exist_ok=True)
    for name, c in CONTINUAL_CASES.items():
        if only and name not in only:
            continue
        out = run_reference_continual(c)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
        print(name, out["codes"].shape, out["codes"][0, :4, 1], flush=True)
    for name, c in list(CASES.items()) + list(SHARP_CASES.items()):
        if only and name not in only:
            continue
        out = run_reference(c)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
        print(name, out["codes"].shape, out["codes"][0, :4, 0], flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or None)
