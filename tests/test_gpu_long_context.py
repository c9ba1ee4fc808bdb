"""GPU: live-reference parity in the regimes BASELINE configs 3 and 5 name and the 600-frame goldens do not reach.

  * `nl12_cap1024_en`: S = 64 text ids and no EOS -> the run ends at the reference's own cap, `y.shape[1] - prompts.shape[1] >
    16 * x_lens.max()` (models/vallex.py:575-578), after exactly 1024 frames; decode contexts to 1314.
  * `nl12_chain2_zh`: two consecutive sentences of the sliding window of generate_audio_from_long_text
    (utils/generation.py:229-274): sentence 2 is prompted by ALL 563 frames and the text of sentence 1 -> S = 200, Tp = 563,
    decode contexts to 1327 (what `bench.py --long-text` times).
  * the three EDGE_CASES fixtures (no prompt at all, the smallest possible input, EOS as the first sample) directly against the
    reference's stored output.
Token ids bit-exact on all 8 codebooks, alone and inside an 8-row batch; teacher-forced AR logits at every 50th step and the NAR
logits of all 7 stages within the tolerances measured in profiles/r03_logit_error.json (written next to each assert)."""
import numpy as np
import pytest

from oracle import synth
from oracle.make_golden import CHAIN_CASES, EDGE_CASES, FULL_LOGIT_EVERY, LONG_CASES, MAX_CASES, chain_second
from tests._util import assert_codes, case_model, golden, inputs_row, nar_logit_error, teacher_forced_logit_error

pytestmark = pytest.mark.gpu

KW = dict(max_new=1032, max_prompt=576, max_text=256, max_batch=8)
AR_TOL, NAR_TOL = 2e-5, 6e-4          # abs; logits have std ~0.6 (AR) / ~25 (NAR); measured 5.6e-6 / 1.8e-4 (r03_logit_error.json)


def _filler_rows(n, S, Tp, seed0, lang="en"):
    rows = []
    for i in range(n):
        a, t = synth.synth_prompt(Tp, S // 3, seed=seed0 + i)
        rows.append(dict(text=np.concatenate([t[0], synth.synth_text(S - S // 3, seed0 + i)]), prompt=a[0], enroll=S // 3,
                         prompt_language=lang, text_language=lang))
    return rows


def test_reference_cap_16S_at_1024_frames_alone_and_in_a_batch():
    name = "nl12_cap1024_en"
    c = LONG_CASES[name]
    g = golden(name)
    assert g["codes"].shape == (1, 1024, 8)
    row, _ = inputs_row(c)
    assert len(row["text"]) == 64
    m = case_model(c, **KW)
    out = m.inference_batch([row], top_k=1)[0]                     # no forced EOS: the 16*S rule ends the row
    assert_codes(name, out, g)
    # 8 rows of 64 text ids (ragged prompts): every row runs to the cap; the golden row sits in slot 5
    rows = _filler_rows(7, 64, 180, 31_000)
    for i, r in enumerate(rows):
        r["prompt"] = r["prompt"][: 150 + 20 * i]
    rows.insert(5, row)
    outs = m.inference_batch(rows, top_k=1)
    assert all(o.shape == (1024, 8) for o in outs)
    assert_codes(name + " (row 5 of 8)", outs[5], g)


def test_cap_case_logits_at_contexts_to_1314():
    name = "nl12_cap1024_en"
    c = LONG_CASES[name]
    g = golden(name)
    row, _ = inputs_row(c)
    m = case_model(c, debug_taps=True, **dict(KW, max_batch=2))
    worst, flips = teacher_forced_logit_error(m, row, g, FULL_LOGIT_EVERY)
    assert flips == 0, f"{flips} of 1024 greedy decisions differ (min reference margin {g['ar_margin'].min():.2e})"
    assert worst <= AR_TOL, worst
    codes, errs = nar_logit_error(m, row, g)
    assert max(errs) <= NAR_TOL, errs
    np.testing.assert_array_equal(codes, g["codes"][0])
    print(f"{name}: AR max |logit - ref| {worst:.2e}; NAR per stage {['%.1e' % e for e in errs]}")


def _chain_rows(c, g, codes1):
    row2, us2 = inputs_row(dict(c, useed=c["useed2"]), inputs=chain_second(c, codes1))
    assert len(row2["text"]) == 200 and row2["prompt"].shape[0] == codes1.shape[1]
    return row2, us2


def test_sliding_window_chain_contexts_to_1327():
    name = "nl12_chain2_zh"
    c = CHAIN_CASES[name]
    g = golden(name)
    row1, us1 = inputs_row(c)
    m = case_model(c, **KW)
    out1 = m.inference_batch([row1], top_k=c["top_k"], uniforms=us1[:, None], force_eos_at=c["force_eos_at"])[0]
    assert_codes(name + " sentence 1", out1, g, key="codes1")
    # sentence 2 is built from the ENGINE's own sentence-1 output, like the product's sliding window does
    row2, us2 = _chain_rows(c, g, out1[None])
    out2 = m.inference_batch([row2], top_k=c["top_k"], uniforms=us2[:, None], force_eos_at=c["force_eos_at"])[0]
    assert_codes(name + " sentence 2", out2, g)
    # ... and as row 2 of an 8-row batch of second sentences (same prompt length, other texts and draws)
    rows, cols = [], []
    for i in range(8):
        if i == 2:
            rows.append(row2)
            cols.append(us2)
        else:
            txt = np.concatenate([row2["text"][:100], synth.synth_text(100, 52_000 + i)])
            rows.append(dict(row2, text=txt, prompt=np.roll(row2["prompt"], 7 * (i + 1), axis=0)))
            cols.append(synth.uniforms(4096, 1, 53_000 + i)[:, 0])
    outs = m.inference_batch(rows, top_k=c["top_k"], uniforms=np.stack(cols, axis=1), force_eos_at=c["force_eos_at"])
    assert_codes(name + " sentence 2 (row 2 of 8)", outs[2], g)


def test_chain_second_sentence_logits():
    name = "nl12_chain2_zh"
    c = CHAIN_CASES[name]
    g = golden(name)
    row2, _ = _chain_rows(c, g, g["codes1"])
    m = case_model(c, debug_taps=True, **dict(KW, max_batch=2))
    worst, _ = teacher_forced_logit_error(m, row2, g, FULL_LOGIT_EVERY)
    assert worst <= AR_TOL, worst
    codes, errs = nar_logit_error(m, row2, g)
    assert max(errs) <= NAR_TOL, errs
    np.testing.assert_array_equal(codes, g["codes"][0])
    print(f"{name}: AR max |logit - ref| {worst:.2e}; NAR per stage {['%.1e' % e for e in errs]}")


@pytest.mark.parametrize("name", sorted(EDGE_CASES))
def test_edge_fixtures_from_the_live_reference(name):
    c = EDGE_CASES[name]
    g = golden(name)
    row, _ = inputs_row(c)
    m = case_model(c, max_new=64, max_prompt=400, max_text=128, max_batch=4)
    out = m.inference(row["text"][None], np.array([len(row["text"])]), row["prompt"][None], row["enroll"], top_k=c["top_k"],
                      prompt_language=row["prompt_language"], text_language=row["text_language"], force_eos_at=c["force_eos_at"])
    assert tuple(out.shape) == g["codes"].shape, (name, tuple(out.shape), g["codes"].shape)
    np.testing.assert_array_equal(out.numpy(), g["codes"])
    if g["codes"].shape[1]:
        m2 = case_model(c, debug_taps=True, max_new=64, max_prompt=400, max_text=128, max_batch=4)
        m2.engine.ar_prefill(m2.make_batch([row]))
        np.testing.assert_allclose(m2.engine.ar_logits()[0], g["ar_logits"][0], atol=3e-4, rtol=0)


def test_largest_enrolment_alone_and_in_a_batch():
    """the longest prompt the reference enrols (15 s = 1125 frames, utils/prompt_making.py:60-61) with 256 text ids: causal prefill
    over 1382 positions, decode contexts to 1422, the seven NAR stages over 1421 rows -- the longest single sequence of any fixture
    (2 layers); the live reference's ids alone (small-batch chain) and as row 3 of 5 (context-split chain)"""
    name = "nl2_max_prompt"
    c = MAX_CASES[name]
    g = golden(name)
    row, us = inputs_row(c)
    assert len(row["text"]) == 256 and row["prompt"].shape == (1125, 8)
    m = case_model(c, max_new=64, max_prompt=1152, max_text=256, max_batch=8)
    out = m.inference_batch([row], top_k=c["top_k"], uniforms=us[:, None], force_eos_at=c["force_eos_at"])[0]
    assert_codes(name, out, g)
    rows = _filler_rows(4, 90, 300, 71_000, "ja")
    rows.insert(3, row)
    cols = [synth.uniforms(4096, 1, 72_000 + i)[:, 0] for i in range(5)]
    cols[3] = us
    outs = m.inference_batch(rows, top_k=c["top_k"], uniforms=np.stack(cols, axis=1), force_eos_at=c["force_eos_at"])
    assert_codes(f"{name} as row 3 of 5", outs[3], g)
    assert m.engine.last_fallbacks()["lifetime"] == 0
