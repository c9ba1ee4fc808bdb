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

# EOS as the very FIRST sample.  models/vallex.py:579-582 raises SyntaxError("well trained model shouldn't reach here.") only if
# prompts.shape[1] == y.shape[1]; with prepend_bos=True (utils/generation.py:76) y carries the extra BOS, the condition is never
# true, and the live reference returns an EMPTY (1, 0, 8) tensor -- pinned here so that the mirror does not "fix" it.
EDGE_CASES = {
    "nl2_eos_first": dict(num_layers=2, seed=0, eos_gain=2.5, preset="librispeech_1", n_text=12, lang="en", top_k=1,
                          force_eos_at=0, useed=None),
    # no audio prompt and no prompt text at all (utils/generation.py:121-123: zeros([1, 0, 8]), enroll_x_lens = 0)
    "nl2_no_prompt": dict(num_layers=2, seed=0, eos_gain=2.5, synth_prompt=(0, 0), prompt_lang="en", n_text=11, lang="en", top_k=1,
                          force_eos_at=9, useed=None),
    # the smallest input: one prompt frame, one prompt text id, one text id
    "nl2_minimal": dict(num_layers=2, seed=3, eos_gain=1.0, synth_prompt=(1, 1), prompt_lang="ja", n_text=1, lang="zh", top_k=1,
                        force_eos_at=5, useed=None),
}

# The largest enrolment the reference accepts: make_prompt refuses audio over 15 s (utils/prompt_making.py:60-61) -> 1125 frames; with the
# longest preset transcript (160 ids) + 96 text ids: S = 256, prefill over 256 + 1 + 1125 = 1382 positions, NAR stages over 1421 rows --
# the longest single sequence of any fixture (2 layers keep it cheap; the kernels do not know the layer count).
MAX_CASES = {
    "nl2_max_prompt": dict(num_layers=2, seed=4, eos_gain=1.0, synth_prompt=(1125, 160), prompt_lang="en", n_text=96, lang="zh",
                           top_k=10, force_eos_at=40, useed=99),
}

# BASELINE C1/C2/C3 shape at FULL length: 12 layers, preset prompt + 100 phoneme ids, EOS forced at
# 600 frames (8.0 s) => Ltot = S + Tp + 600 ~ 983 for librispeech_1.  Weights = the bench weights (seed 0, eos_gain 0: the EOS
# logit is exactly 0, never the arg-max and never inside the top-10, so no run ends early).  All six share ONE weight set so
# the GPU test can put several rows in one `inference_batch` call (packed rows >= 1024 => the DMA bf16x3 GEMM and attn_full_x3
# at L ~ 983 are what gets compared with the reference).  One live-reference run is ~20-40 s of CPU.
FULL_CASES = {
    "nl12_full_en_greedy": dict(num_layers=12, seed=0, eos_gain=0.0, preset="librispeech_1", n_text=100, lang="en", top_k=1,
                                force_eos_at=600, useed=None, full=True),
    "nl12_full_zh_greedy": dict(num_layers=12, seed=0, eos_gain=0.0, preset="paimon", n_text=100, lang="zh", top_k=1,
                                force_eos_at=600, useed=None, full=True, text_seed=1),
    "nl12_full_ja_greedy": dict(num_layers=12, seed=0, eos_gain=0.0, preset="cafe", n_text=100, lang="ja", top_k=1,
                                force_eos_at=600, useed=None, full=True, text_seed=2),
    "nl12_full_en_topk10": dict(num_layers=12, seed=0, eos_gain=0.0, preset="librispeech_1", n_text=100, lang="en", top_k=10,
                                force_eos_at=600, useed=1234, full=True, text_seed=3),
    "nl12_full_zh_topk10": dict(num_layers=12, seed=0, eos_gain=0.0, preset="paimon", n_text=100, lang="zh", top_k=10,
                                force_eos_at=600, useed=2345, full=True, text_seed=4),
    "nl12_full_ja_topk10": dict(num_layers=12, seed=0, eos_gain=0.0, preset="cafe", n_text=100, lang="ja", top_k=10,
                                force_eos_at=600, useed=3456, full=True, text_seed=5),
}
# Round 3 -- the regimes the six FULL_CASES do not reach (all 12 layers, bench weights unless stated):
#  * TRAINED_CASES: weights with the statistics of a trained checkpoint (synth.trained_like_state_dict: heavy-tailed weights,
#    LayerNorm gains U(0.5, 4), massive FFN channels / residual dimensions, decisive AR logits);
#  * nl12_cap1024_en: S = 58 + 6 = 64 text ids, nothing ever emits EOS -> the run ends at the reference's own cap
#    (models/vallex.py:575-578: y.shape[1] - prompts.shape[1] > 16 * x_lens.max()) with exactly 1024 frames, contexts to 1314
#    (BASELINE config 3: "padded to 1024 codec tokens");
#  * CHAIN_CASES: two consecutive sentences of generate_audio_from_long_text's sliding window (utils/generation.py:229-274):
#    sentence 2 is prompted by ALL frames of sentence 1 (encoded_frames[:, :, -NUM_QUANTIZERS:] keeps every frame) and by its
#    text (text_tokens[:, enroll_x_lens:]), the prompt LANGUAGE stays the first prompt's (lang_pr is never updated) ->
#    S = 200, Tp = 563, contexts to 200 + 1 + 563 + 563 = 1327 (BASELINE config 5).
TRAINED_CASES = {
    "nl12_trained_en_greedy": dict(num_layers=12, seed=0, eos_gain=0.0, trained=True, preset="librispeech_1", n_text=100, lang="en",
                                   top_k=1, force_eos_at=600, useed=None, full=True, text_seed=6),
    "nl12_trained_zh_topk10": dict(num_layers=12, seed=0, eos_gain=0.0, trained=True, preset="paimon", n_text=100, lang="zh",
                                   top_k=10, force_eos_at=600, useed=4567, full=True, text_seed=7),
}
# The Gradio UI's own call (launch-ui.py:285-295): unfiltered multinomial (top_k=-100, temperature 1), best_of=5 beams ranked by
# sum(logp) / len (models/vallex.py:583-594), here on the full 12-layer model with trained-like weights whose EOS logit is live
# (eos_gain 1.85): four beams end by themselves after 9, 9, 8 and 13 frames, the fifth runs into the forced EOS at 120 (which only
# bounds the fixture); finished beams keep emitting EOS while the others go on (:572-573).  The best beam is a 9-frame one;
# `_worst` (return_worst=True) returns the 120-frame one.  The uniforms seed was chosen among twelve for the largest decision margin
# of the inverse-CDF draws (min over all beams and steps of |cdf - u| = 9.2e-4; seeds with 1e-6 exist and pin nothing).
UI_CASES = {
    "nl12_ui_bestof5_ja": dict(num_layers=12, seed=0, eos_gain=1.85, trained=True, preset="cafe", n_text=40, lang="ja", top_k=-100,
                               temperature=1.0, best_of=5, force_eos_at=120, useed=8643, text_seed=11),
    "nl12_ui_bestof5_ja_worst": dict(num_layers=12, seed=0, eos_gain=1.85, trained=True, preset="cafe", n_text=40, lang="ja",
                                     top_k=-100, temperature=1.0, best_of=5, force_eos_at=120, useed=8643, text_seed=11,
                                     return_worst=True),
}
LONG_CASES = {
    "nl12_cap1024_en": dict(num_layers=12, seed=0, eos_gain=0.0, preset="librispeech_1", n_text=6, lang="en", top_k=1,
                            force_eos_at=None, useed=None, full=True, text_seed=8),
}
CHAIN_FRAMES = 563             # 7.5 s per sentence (bench.py --long-text)
CHAIN_CASES = {
    # sentence 1: the paimon preset (zh) + 100 ids; sentence 2: built from sentence 1's output by chain_second()
    "nl12_chain2_zh": dict(num_layers=12, seed=0, eos_gain=0.0, preset="paimon", n_text=100, lang="zh", top_k=10,
                           force_eos_at=CHAIN_FRAMES, useed=5678, full=True, text_seed=9, text_seed2=10, useed2=6789),
}
# Operands outside the fp16 range of the f16x2 kernels, same function bit for bit (synth.out_of_range_state_dict): the base case's
# golden must come out again.  (base case, kind)
RANGE_CASES = {
    "nl2_range_ffn": ("nl2_topk10", "ffn"),
    "nl2_range_v": ("nl2_sharp_greedy", "v"),
    "nl2_range_k": ("nl2_sharp_topk10", "k"),
}
# Precision, not range (round 4): one weight per projection tensor at 1000 x the init bound (synth.outlier_state_dict) -- the per-tensor
# power-of-two scale of the f16x2 weight planes is then set by the outlier and every ordinary weight sits 10 bits lower in its
# fp16 head + tail pair.  A different function than the base weights, so it has its own live-reference golden (with margins).
OUTLIER_CASES = {
    "nl2_outlier1000": dict(num_layers=2, seed=3, eos_gain=1.0, outlier=1000.0, preset="paimon", n_text=16, lang="en", top_k=10,
                            force_eos_at=48, useed=1234, full=True),
    "nl2_outlier1000_greedy": dict(num_layers=2, seed=5, eos_gain=1.0, outlier=1000.0, preset="librispeech_1", n_text=14, lang="en",
                                   top_k=1, force_eos_at=48, useed=None, full=True),
}
FULL_LOGIT_EVERY = 50          # AR logits are stored for steps 0, 50, ..., 550 (+ the forced-EOS step is not stored)

# Shapes and languages of ALL 41 reference presets (presets/*.npz: frames, prompt text ids, lang_code zh 0 / ja 1 / en 2) -- metadata
# only; the prompt CONTENT of these cases is synthetic (any ids 0..1023 exercise the same code).  Covers prompts from 161 to 758
# frames and 18 to 160 prompt text ids in the three prompt languages, each with a different text language.
PRESET_SHAPES = [
    ("acou_1", 225, 58, 2), ("acou_2", 225, 30, 2), ("acou_3", 225, 28, 2), ("acou_4", 225, 40, 2), ("alan", 749, 160, 1),
    ("amused", 310, 32, 2), ("anger", 324, 71, 2), ("babara", 162, 46, 0), ("bronya", 265, 45, 0), ("cafe", 331, 59, 1),
    ("dingzhen", 263, 67, 0), ("disgust", 596, 57, 2), ("emo_amused", 225, 24, 2), ("emo_anger", 225, 31, 2),
    ("emo_neutral", 225, 49, 2), ("emo_sleepy", 225, 23, 2), ("emotion_sleepiness", 500, 54, 2), ("en2zh_tts_1", 663, 142, 2),
    ("en2zh_tts_2", 361, 36, 2), ("en2zh_tts_3", 313, 74, 2), ("en2zh_tts_4", 654, 148, 2), ("esta", 602, 117, 1),
    ("fuxuan", 758, 129, 1), ("librispeech_1", 225, 58, 2), ("librispeech_2", 225, 29, 2), ("librispeech_3", 225, 56, 2),
    ("librispeech_4", 225, 51, 2), ("neutral", 308, 69, 2), ("paimon", 198, 38, 0), ("rosalia", 161, 40, 0), ("seel", 191, 68, 0),
    ("sleepiness", 500, 54, 2), ("vctk_1", 225, 43, 2), ("vctk_2", 225, 34, 2), ("vctk_3", 225, 33, 2), ("vctk_4", 225, 18, 2),
    ("yaesakura", 177, 41, 0), ("zh2en_tts_1", 345, 85, 0), ("zh2en_tts_2", 294, 71, 0), ("zh2en_tts_3", 258, 76, 0),
    ("zh2en_tts_4", 500, 112, 0),
]
PRESET_SHAPE_SEED, PRESET_SHAPE_FRAMES = 21, 10


def preset_shape_case(i):
    """case dict (same schema as CASES) of preset-shape i: 2 layers, greedy, 12 text ids, forced EOS at 10 frames"""
    name, tp, sp, code = PRESET_SHAPES[i]
    return dict(num_layers=2, seed=PRESET_SHAPE_SEED, eos_gain=0.0, synth_prompt=(tp, sp), prompt_seed=500 + i, n_text=12,
                text_seed=700 + i, lang=("en", "zh", "ja")[i % 3], prompt_lang=CODE2LANG[code], top_k=1,
                force_eos_at=PRESET_SHAPE_FRAMES, useed=None)


# VALLE.continual (models/vallex.py:688-787): text ids + a full (T, 8) code matrix; NAR stages only
CONTINUAL_CASES = {
    "nl2_continual": dict(num_layers=2, seed=6, eos_gain=1.0, n_text=14, frames=61),          # prefix_len = 30
    "nl2_continual_long": dict(num_layers=2, seed=7, eos_gain=1.0, n_text=9, frames=470),     # prefix_len = 225 (3 s cap)
}

# the same on the full model with trained-like weights (tests/test_gpu_trained_like.py; the oracle side runs with VX_SLOW=1)
CONTINUAL12_CASES = {
    "nl12_continual_trained": dict(num_layers=12, seed=0, eos_gain=0.0, trained=True, n_text=60, frames=450),   # prefix_len = 225
}

CODE2LANG = {0: "zh", 1: "ja", 2: "en"}      # macros.py:15-19 via utils/generation.py:114-115


def load_preset(name):
    d = np.load(os.path.join(GOLD, "presets", name + ".npz"))
    return d["audio_tokens"], d["text_tokens"], CODE2LANG[int(d["lang_code"])]


def case_inputs(c):
    if "preset" in c:
        a, t, pl = load_preset(c["preset"])
    else:
        a, t = synth.synth_prompt(*c["synth_prompt"], seed=c.get("prompt_seed", c["seed"]))
        pl = c["prompt_lang"]
    txt = synth.synth_text(c["n_text"], c.get("text_seed", c["seed"]))[None]
    text = np.concatenate([t, txt], -1)
    if c["lang"] == "mix":
        langs = [("en", "zh", "ja")[i % 3] for i in range(c["n_text"])]
    else:
        langs = c["lang"]
    return a, t, text, pl, langs


def case_state_dict(c):
    """the weights of a case (tests and the generator build them the same way)"""
    if c.get("trained"):
        return synth.trained_like_state_dict(c["num_layers"], c["seed"], eos_gain=c["eos_gain"])
    sd = synth.vallex_state_dict(c["num_layers"], c["seed"], c["eos_gain"], c.get("attn_gain", 1.0))
    if c.get("range_kind"):
        sd = synth.out_of_range_state_dict(sd, c["num_layers"], c["range_kind"])
    if c.get("outlier"):
        sd = synth.outlier_state_dict(sd, c["num_layers"], c["outlier"])
    return sd


def all_cases():
    """every named single-call case (name -> dict); RANGE_CASES are their base case + `range_kind`"""
    out = {}
    for grp in (CASES, SHARP_CASES, EDGE_CASES, MAX_CASES, FULL_CASES, TRAINED_CASES, LONG_CASES, UI_CASES, OUTLIER_CASES):
        out.update(grp)
    for name, (base, kind) in RANGE_CASES.items():
        out[name] = dict(out[base], range_kind=kind)
    return out


def chain_second(c, codes1):
    """inputs of sentence 2 of a CHAIN case from sentence 1's output (1, T, 8): utils/generation.py:264-266 -- audio prompt =
    all generated frames, text prompt = sentence 1's own text, then the new sentence's ids; prompt language unchanged."""
    a1, t1, text1, pl, langs = case_inputs(c)
    own = text1[:, t1.shape[-1]:]
    new = synth.synth_text(c["n_text"], c["text_seed2"])[None]
    return np.asarray(codes1, np.int64), own, np.concatenate([own, new], -1), pl, langs


def run_reference(c, inputs=None, useed=None, sd=None):
    """`sd`: explicit weights instead of the case's synthetic recipe (tools/verify_checkpoint.py runs a real checkpoint through this)"""
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import models.vallex as V
    from models.vallex import VALLE

    m = VALLE(1024, 16, c["num_layers"], norm_first=True, add_prenet=False, prefix_mode=1,
              share_embedding=True, nar_scale_factor=1.0, prepend_bos=True, num_quantizers=8).eval()
    sd = sd if sd is not None else case_state_dict(c)
    missing = m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    a, t, text, pl, langs = inputs if inputs is not None else case_inputs(c)
    if useed is not None:
        c = dict(c, useed=useed)

    full = bool(c.get("full"))
    rec = {"logits": [], "step": 0, "margin": []}
    orig_sampling = V.topk_sampling
    orig_multinomial = torch.multinomial
    nbeam = c.get("best_of", 1)
    us = None if c["useed"] is None else synth.uniforms(4096, nbeam, c["useed"])

    def hooked(logits, top_k=10, top_p=1.0, temperature=1.0):
        if (rec["step"] % FULL_LOGIT_EVERY == 0) if full else (rec["step"] < 8):
            rec["logits"].append(logits[0].detach().clone().numpy())
        if full and c["top_k"] == 1:
            # decision margin of the reference itself: gap between its two largest logits.  A step whose gap is below the
            # fp32 reassociation noise (~1e-5 here) is one the reference does not decide reproducibly either
            # (its logits move by 7e-7 with the thread count).
            top2 = torch.topk(logits[0], 2).values
            rec["margin"].append(float(top2[0] - top2[1]))
        tok, lp = orig_sampling(logits, top_k=top_k, top_p=top_p, temperature=temperature)
        if c["force_eos_at"] is not None and rec["step"] >= c["force_eos_at"]:
            tok = torch.full_like(tok, synth.EOS_ID)
        rec["step"] += 1
        return tok, lp

    def multinomial(probs, num_samples=1, **kw):
        if us is None:
            return orig_multinomial(probs, num_samples, **kw)
        if full:
            # margin of the inverse-CDF decision: distance of u * total to the nearest CDF boundary, in probability units
            cdf = torch.cumsum(probs[0].double(), 0)
            rec["margin"].append(float(torch.min(torch.abs(cdf / cdf[-1] - float(us[rec["step"], 0])))))
        return torch.tensor([[inverse_cdf_sample(probs[i], float(us[rec["step"], i]))] for i in range(probs.shape[0])],
                            dtype=torch.long)

    nar_logits = []
    nar_margin = []

    def hook_stage(j):
        orig = m.nar_predict_layers[j].forward

        def nar_pred(xx):
            out = orig(xx)
            nar_logits.append(out.detach().clone().numpy())
            if full:
                t2 = torch.topk(out[0], 2, dim=-1).values
                nar_margin.append(float((t2[:, 0] - t2[:, 1]).min()))
            return out

        m.nar_predict_layers[j].forward = nar_pred

    for j in range(7 if full else 1):
        hook_stage(j)
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
    if full:
        return dict(codes=codes.numpy().astype(np.int64),
                    ar_logits=np.stack(rec["logits"]).astype(np.float32),                     # steps 0, 50, ..., 550
                    nar_logits=np.stack([l[0, :16] for l in nar_logits]).astype(np.float32),  # [7 stages][16 frames][1024]
                    ar_margin=np.array(rec["margin"], np.float64),                            # per AR step (see hooks)
                    nar_margin=np.array(nar_margin, np.float64))                              # per stage: min top-2 gap
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
    sd = case_state_dict(c)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    text, y = continual_inputs(c)
    with torch.no_grad():
        codes = m.continual(torch.from_numpy(text).to(torch.int32), torch.IntTensor([text.shape[-1]]),
                            torch.from_numpy(y).to(torch.int64))
    return dict(codes=codes.numpy().astype(np.int64))


def make_preset_shapes():
    """one fixture for all 41 preset shapes: codes [41][10][8] + the first AR logits row of each"""
    codes, logits = [], []
    for i in range(len(PRESET_SHAPES)):
        out = run_reference(preset_shape_case(i))
        assert out["codes"].shape == (1, PRESET_SHAPE_FRAMES, 8), (PRESET_SHAPES[i], out["codes"].shape)
        codes.append(out["codes"][0].astype(np.int16))
        logits.append(out["ar_logits"][0])
        print(PRESET_SHAPES[i], out["codes"][0, :3, 0], flush=True)
    np.savez_compressed(os.path.join(GOLD, "preset_shapes.npz"), codes=np.stack(codes), ar_logits0=np.stack(logits).astype(np.float32))


def main(only=None):
    os.makedirs(GOLD, exist_ok=True)
    if only and "preset_shapes" in only:
        make_preset_shapes()
        return
    for name, c in list(CONTINUAL_CASES.items()) + list(CONTINUAL12_CASES.items()):
        if (only and name not in only) or (not only and name in CONTINUAL12_CASES):
            continue
        out = run_reference_continual(c)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
        print(name, out["codes"].shape, out["codes"][0, :4, 1], flush=True)
    import time
    for name, c in CHAIN_CASES.items():
        if not only or name not in only:
            continue                                   # slow: only when named
        t0 = time.time()
        o1 = run_reference(c)
        o2 = run_reference(c, inputs=chain_second(c, o1["codes"]), useed=c["useed2"])
        out = dict(codes1=o1["codes"], **o2)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
        print(name, o1["codes"].shape, "->", o2["codes"].shape, f"{time.time() - t0:.1f}s min AR margin {o2['ar_margin'].min():.3e}, "
              f"min NAR margin {o2['nar_margin'].min():.3e}", flush=True)
    for name, (base, kind) in RANGE_CASES.items():
        if not only or name not in only:
            continue
        # no new fixture: the rescaled weights are the same function, the LIVE reference must return the base golden
        out = run_reference(dict(all_cases()[name]))
        gold = np.load(os.path.join(GOLD, base + ".npz"))
        assert np.array_equal(out["codes"], gold["codes"]), (name, "live reference differs from the base golden")
        print(name, "live reference == golden of", base, "| max |logit diff|", float(np.abs(out["ar_logits"] - gold["ar_logits"]).max()),
              flush=True)
    for name, c in (list(CASES.items()) + list(SHARP_CASES.items()) + list(EDGE_CASES.items()) + list(MAX_CASES.items())
                    + list(FULL_CASES.items())
                    + list(TRAINED_CASES.items()) + list(LONG_CASES.items()) + list(UI_CASES.items())
                    + list(OUTLIER_CASES.items())):
        if only and name not in only:
            continue
        if c.get("full") and not only and os.path.exists(os.path.join(GOLD, name + ".npz")):
            continue                                   # full-length runs are slow: regenerate only when named
        t0 = time.time()
        out = run_reference(c)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
        extra = ""
        if c.get("full"):
            extra = f" min AR margin {out['ar_margin'].min():.3e}, min NAR margin {out['nar_margin'].min():.3e}"
        print(name, out["codes"].shape, out["codes"][0, :4, 0], f"{time.time() - t0:.1f}s" + extra, flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or None)
