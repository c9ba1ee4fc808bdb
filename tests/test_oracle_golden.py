"""CPU: the oracle restatement vs the committed outputs of the LIVE reference
(tests/golden/*.npz, made by oracle/make_golden.py from /root/reference)."""
import os

import numpy as np
import pytest

from oracle import synth
from oracle.make_golden import (CASES, CONTINUAL_CASES, EDGE_CASES, FULL_CASES, FULL_LOGIT_EVERY, GOLD, PRESET_SHAPES, SHARP_CASES,
                                case_inputs, continual_inputs, preset_shape_case)
from oracle.vallex_oracle import VallexOracle

FAST = [n for n in CASES if n.startswith("nl2_")]


def run_oracle(name, taps=None):
    c = CASES[name]
    sd = synth.vallex_state_dict(c["num_layers"], c["seed"], c["eos_gain"])
    orc = VallexOracle(sd, c["num_layers"])
    a, t, text, pl, langs = case_inputs(c)
    nb = c.get("best_of", 1)
    us = None if c["useed"] is None else synth.uniforms(4096, nb, c["useed"])
    if us is not None and nb == 1:
        us = us[:, 0]
    return orc.inference(text, np.array([text.shape[-1]]), a, t.shape[-1], top_k=c["top_k"],
                         temperature=c.get("temperature", 1.0), prompt_language=pl, text_language=langs,
                         uniforms=us, force_eos_at=c["force_eos_at"], taps=taps, best_of=nb,
                         length_penalty=c.get("length_penalty", 1.0), return_worst=c.get("return_worst", False))


@pytest.mark.parametrize("name", FAST)
def test_oracle_matches_reference_tokens(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    taps = {}
    codes = run_oracle(name, taps)
    assert codes.shape == g["codes"].shape
    np.testing.assert_array_equal(codes, g["codes"])          # bit-exact ids, all 8 codebooks
    if CASES[name].get("best_of", 1) == 1:
        ar = np.stack([l.numpy() for l in taps["ar_logits"][: g["ar_logits"].shape[0]]])
        np.testing.assert_allclose(ar, g["ar_logits"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(taps["nar_logits"][0][:16].numpy(), g["nar_logits0"], atol=5e-3, rtol=0)


def test_oracle_matches_reference_12_layers():
    name = "nl12_c1_short"
    g = np.load(os.path.join(GOLD, name + ".npz"))
    np.testing.assert_array_equal(run_oracle(name), g["codes"])


def test_synthetic_state_dict_layout():
    sd = synth.vallex_state_dict(12, 0)
    assert len(sd) == 374
    n_ar = sum(v.size for k, v in sd.items() if k.startswith("ar_decoder."))
    assert n_ar == 151_156_736
    assert sd["nar_predict_layers.0.weight"] is sd["nar_audio_embeddings.2.word_embeddings.weight"]


@pytest.mark.parametrize("name", sorted(CONTINUAL_CASES))
def test_oracle_continual_matches_reference(name):
    """VALLE.continual (models/vallex.py:688-787): NAR-only continuation, no language embedding, prefix = first half of y
    capped at 225 frames.  Golden = the live reference's own continual()."""
    c = CONTINUAL_CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))["codes"]
    orc = VallexOracle(synth.vallex_state_dict(c["num_layers"], c["seed"], c["eos_gain"]), c["num_layers"])
    text, y = continual_inputs(c)
    out = orc.continual(text, np.array([text.shape[-1]]), y)
    prefix_len = min(int(y.shape[1] * 0.5), 225)
    assert out.shape == g.shape == (1, y.shape[1] - prefix_len, 8)
    np.testing.assert_array_equal(out, g)
    np.testing.assert_array_equal(out[0, :, 0], y[0, prefix_len:, 0])      # first codebook is passed through


def test_oracle_continual_matches_reference_12_trained_like_layers():
    """the same on the full model with trained-like weights (225 given + 225 continued frames); smallest arg-max margin of the
    seven stages 2.8e-2, logits to |220|"""
    from oracle.make_golden import CONTINUAL12_CASES, case_state_dict
    name = "nl12_continual_trained"
    c = CONTINUAL12_CASES[name]
    text, y = continual_inputs(c)
    out = VallexOracle(case_state_dict(c), c["num_layers"]).continual(text, np.array([text.shape[-1]]), y)
    np.testing.assert_array_equal(out, np.load(os.path.join(GOLD, name + ".npz"))["codes"])


def test_oracle_matches_reference_at_the_largest_enrolment():
    """15 s of prompt (1125 frames, utils/prompt_making.py:60-61) + 256 text ids: prefill over 1382 positions, NAR over 1421 rows"""
    from oracle.make_golden import MAX_CASES
    c = MAX_CASES["nl2_max_prompt"]
    g = np.load(os.path.join(GOLD, "nl2_max_prompt.npz"))
    a, t, text, pl, langs = case_inputs(c)
    assert a.shape == (1, 1125, 8) and text.shape == (1, 256)
    taps = {}
    codes = _run_case(c, taps=taps)
    np.testing.assert_array_equal(codes, g["codes"])
    ar = np.stack([l.numpy() for l in taps["ar_logits"][: g["ar_logits"].shape[0]]])
    np.testing.assert_allclose(ar, g["ar_logits"], atol=2e-4, rtol=0)


@pytest.mark.parametrize("name", sorted(SHARP_CASES))
def test_oracle_matches_reference_tokens_sharp_attention(name):
    """attn_gain 3 weights (peaky, trained-looking attention): this test bed reacts to K/V precision and to the score
    arithmetic, which the default random init does not.  Golden = the live reference."""
    c = SHARP_CASES[name]
    orc = VallexOracle(synth.vallex_state_dict(c["num_layers"], c["seed"], c["eos_gain"], c["attn_gain"]), c["num_layers"])
    a, t, text, pl, langs = case_inputs(c)
    us = None if c["useed"] is None else synth.uniforms(4096, 1, c["useed"])[:, 0]
    codes = orc.inference(text, np.array([text.shape[-1]]), a, t.shape[-1], top_k=c["top_k"], prompt_language=pl,
                          text_language=langs, uniforms=us, force_eos_at=c["force_eos_at"])
    np.testing.assert_array_equal(codes, np.load(os.path.join(GOLD, name + ".npz"))["codes"])


# one full-length case by default (~40 s on 8 cores); VX_SLOW=1 runs all six
FULL_DEFAULT = ["nl12_full_ja_topk10"]


@pytest.mark.parametrize("name", sorted(FULL_CASES) if os.environ.get("VX_SLOW") == "1" else FULL_DEFAULT)
def test_oracle_matches_reference_full_length(name):
    """BASELINE C1-C3 shape (12 layers, preset + 100 ids, 600 frames, Ltot ~ 983): oracle ids == live reference ids for all
    600 x 8 tokens, AR logits at every 50th step and the NAR logits of all 7 stages within fp32 reassociation distance."""
    c = FULL_CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    assert g["codes"].shape == (1, 600, 8)
    orc = VallexOracle(synth.vallex_state_dict(c["num_layers"], c["seed"], c["eos_gain"]), c["num_layers"])
    a, t, text, pl, langs = case_inputs(c)
    us = None if c["useed"] is None else synth.uniforms(4096, 1, c["useed"])[:, 0]
    taps = {}
    codes = orc.inference(text, np.array([text.shape[-1]]), a, t.shape[-1], top_k=c["top_k"], prompt_language=pl,
                          text_language=langs, uniforms=us, force_eos_at=c["force_eos_at"], taps=taps)
    np.testing.assert_array_equal(codes, g["codes"])
    # the golden holds steps 0, 50, ..., 600 (step 600 is the sampling call that was forced to EOS)
    n = min(len(taps["ar_logits"]), 601)
    ar = np.stack([taps["ar_logits"][i].numpy() for i in range(0, n, FULL_LOGIT_EVERY)])
    np.testing.assert_allclose(ar, g["ar_logits"][: len(ar)], atol=2e-4, rtol=0)
    assert len(ar) >= 12
    for st in range(7):
        np.testing.assert_allclose(taps["nar_logits"][st][:16].numpy(), g["nar_logits"][st], atol=5e-3, rtol=0)


def test_oracle_matches_reference_on_preset_shapes():
    """the shapes / languages of the reference's 41 presets (metadata only, synthetic content): oracle ids == live reference ids.
    A spread of 9 by default (shortest, longest, every language), all 41 with VX_SLOW=1."""
    g = np.load(os.path.join(GOLD, "preset_shapes.npz"))
    assert g["codes"].shape == (len(PRESET_SHAPES), 10, 8)
    pick = range(len(PRESET_SHAPES)) if os.environ.get("VX_SLOW") == "1" else (0, 4, 7, 9, 17, 22, 29, 35, 40)
    orc = None
    for i in pick:
        c = preset_shape_case(i)
        if orc is None:
            orc = VallexOracle(synth.vallex_state_dict(c["num_layers"], c["seed"], c["eos_gain"]), c["num_layers"])
        a, t, text, pl, langs = case_inputs(c)
        assert a.shape[1] == PRESET_SHAPES[i][1] and t.shape[1] == PRESET_SHAPES[i][2]
        taps = {}
        codes = orc.inference(text, np.array([text.shape[-1]]), a, t.shape[-1], top_k=1, prompt_language=pl, text_language=langs,
                              force_eos_at=c["force_eos_at"], taps=taps)
        np.testing.assert_array_equal(codes[0], g["codes"][i].astype(np.int64), err_msg=PRESET_SHAPES[i][0])
        np.testing.assert_allclose(taps["ar_logits"][0].numpy(), g["ar_logits0"][i], atol=2e-4, rtol=0)


def test_eos_as_first_sample_returns_an_empty_result_like_the_live_reference():
    """models/vallex.py:579-582 would raise SyntaxError("well trained model shouldn't reach here.") if the stop rule fired with
    `prompts.shape[1] == y.shape[1]` -- never true with prepend_bos=True (y carries the BOS): the LIVE reference returns an
    empty (1, 0, 8) tensor (golden nl2_eos_first), and so do the oracle and the mirror (GPU: test_edge_cases_... (b))."""
    c = EDGE_CASES["nl2_eos_first"]
    g = np.load(os.path.join(GOLD, "nl2_eos_first.npz"))
    assert g["codes"].shape == (1, 0, 8) and g["nar_logits0"].shape[0] == 0
    orc = VallexOracle(synth.vallex_state_dict(c["num_layers"], c["seed"], c["eos_gain"]), c["num_layers"])
    a, t, text, pl, langs = case_inputs(c)
    taps = {}
    codes = orc.inference(text, np.array([text.shape[-1]]), a, t.shape[-1], top_k=1, prompt_language=pl, text_language=langs,
                          force_eos_at=0, taps=taps)
    assert codes.shape == (1, 0, 8)
    np.testing.assert_allclose(taps["ar_logits"][0].numpy(), g["ar_logits"][0], atol=2e-4, rtol=0)


@pytest.mark.parametrize("name", ["nl2_no_prompt", "nl2_minimal"])
def test_oracle_matches_reference_on_smallest_inputs(name):
    """no audio prompt / prompt text at all (utils/generation.py:121-123) and the smallest possible input (one prompt frame, one
    prompt id, one text id): oracle ids == the live reference's, first AR logits and NAR stage-0 logits within fp32 distance"""
    c = EDGE_CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    orc = VallexOracle(synth.vallex_state_dict(c["num_layers"], c["seed"], c["eos_gain"]), c["num_layers"])
    a, t, text, pl, langs = case_inputs(c)
    assert a.shape[1] == c["synth_prompt"][0] and t.shape[1] == c["synth_prompt"][1]
    taps = {}
    codes = orc.inference(text, np.array([text.shape[-1]]), a, t.shape[-1], top_k=1, prompt_language=pl, text_language=langs,
                          force_eos_at=c["force_eos_at"], taps=taps)
    np.testing.assert_array_equal(codes, g["codes"])
    n = g["ar_logits"].shape[0]
    np.testing.assert_allclose(np.stack([l.numpy() for l in taps["ar_logits"][:n]]), g["ar_logits"], atol=2e-4, rtol=0)
    np.testing.assert_allclose(taps["nar_logits"][0][:16].numpy(), g["nar_logits0"], atol=5e-3, rtol=0)


# ---- round 3: operands beyond the fp16 range, trained-like weights, the 16*S cap at 1024 frames, the sliding-window chain ----
def _run_case(c, inputs=None, useed="case", taps=None):
    from oracle.make_golden import case_state_dict
    orc = VallexOracle(case_state_dict(c), c["num_layers"])
    a, t, text, pl, langs = inputs if inputs is not None else case_inputs(c)
    useed = c["useed"] if useed == "case" else useed
    us = None if useed is None else synth.uniforms(4096, 1, useed)[:, 0]
    return orc.inference(text, np.array([text.shape[-1]]), a, t.shape[-1], top_k=c["top_k"], temperature=c.get("temperature", 1.0),
                         prompt_language=pl, text_language=langs, uniforms=us, force_eos_at=c["force_eos_at"], taps=taps)


def test_out_of_range_rescaling_is_the_same_function():
    """synth.out_of_range_state_dict (channels x 2^12, read-out x 2^-12) does not change ONE bit of the fp32 function: the
    oracle on the rescaled weights reproduces the base golden's ids and its logits exactly (the live reference does too:
    oracle/make_golden.py RANGE_CASES, difference 0.0) -- while the hidden activations / values / keys are now far outside
    fp16 range (what tests/test_gpu_range_fallback.py needs)."""
    from oracle.make_golden import RANGE_CASES, all_cases, case_state_dict
    import torch
    for name, (base, kind) in sorted(RANGE_CASES.items()):
        c = all_cases()[name]
        t_new, t_old = {}, {}
        codes = _run_case(c, taps=t_new)
        np.testing.assert_array_equal(codes, np.load(os.path.join(GOLD, base + ".npz"))["codes"], err_msg=name)
        _run_case(all_cases()[base], taps=t_old)
        for x, y in zip(t_new["ar_logits"], t_old["ar_logits"]):
            assert torch.equal(x, y), name
        sd = case_state_dict(c)
        if kind == "ffn":
            assert np.abs(sd["ar_decoder.layers.0.linear1.weight"][:64]).max() > 100.0
    # the rescaled operands really are out of range: |relu(W1 LN(x) + b1)| for a unit-variance row
    x = torch.randn(64, 1024)
    w = torch.from_numpy(case_state_dict(all_cases()["nl2_range_ffn"])["nar_decoder.layers.0.linear1.weight"])
    assert float((x @ w.T).abs().max()) > 2047.0


def test_oracle_matches_reference_with_outlier_weights():
    """synth.outlier_state_dict: one weight per projection tensor at 1000 x the init bound (the per-tensor scale of the f16x2
    weight planes is then set by the outlier; tests/test_gpu_range_fallback.py runs the engine on these).  Golden = live reference."""
    from oracle.make_golden import OUTLIER_CASES, case_state_dict
    for name, c in sorted(OUTLIER_CASES.items()):
        sd = case_state_dict(c)
        w = sd["nar_decoder.layers.1.linear1.weight"]
        assert np.abs(w).max() > 900.0 * np.median(np.abs(w)), name                 # one element dominates the tensor
        g = np.load(os.path.join(GOLD, name + ".npz"))
        np.testing.assert_array_equal(_run_case(c), g["codes"], err_msg=name)
        assert g["ar_margin"].min() > 1e-4 and g["nar_margin"].min() > 1e-3       # the reference decides these steps robustly


SLOW = os.environ.get("VX_SLOW") == "1"


@pytest.mark.skipif(not SLOW, reason="12 layers x 600-1024 frames on the CPU oracle: ~1 min each; VX_SLOW=1 (the GPU suite runs all of them)")
@pytest.mark.parametrize("name", ["nl12_trained_en_greedy", "nl12_trained_zh_topk10", "nl12_cap1024_en"])
def test_oracle_matches_reference_round3_long(name):
    from oracle.make_golden import LONG_CASES, TRAINED_CASES
    c = dict(TRAINED_CASES, **LONG_CASES)[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    taps = {}
    codes = _run_case(c, taps=taps)
    np.testing.assert_array_equal(codes, g["codes"])
    ar = np.stack([taps["ar_logits"][i].numpy() for i in range(0, min(len(taps["ar_logits"]), g["codes"].shape[1] + 1), FULL_LOGIT_EVERY)])
    scale = max(1.0, float(np.abs(g["ar_logits"]).max()))
    np.testing.assert_allclose(ar, g["ar_logits"][: len(ar)], atol=2e-4 * scale, rtol=0)


@pytest.mark.skipif(not SLOW, reason="two 12-layer 563-frame oracle runs; VX_SLOW=1")
def test_oracle_matches_reference_sliding_window_chain():
    from oracle.make_golden import CHAIN_CASES, chain_second
    c = CHAIN_CASES["nl12_chain2_zh"]
    g = np.load(os.path.join(GOLD, "nl12_chain2_zh.npz"))
    c1 = _run_case(c)
    np.testing.assert_array_equal(c1, g["codes1"])
    np.testing.assert_array_equal(_run_case(c, inputs=chain_second(c, c1), useed=c["useed2"]), g["codes"])


@pytest.mark.skipif(not SLOW, reason="five 12-layer beams on the CPU oracle (~30 s); VX_SLOW=1 (the GPU suite runs it)")
def test_oracle_matches_reference_ui_call_best_of_5():
    """launch-ui.py:285-295 on 12 trained-like layers: the best and the worst of five unfiltered-multinomial beams"""
    from oracle.make_golden import UI_CASES, case_state_dict
    c = UI_CASES["nl12_ui_bestof5_ja"]
    orc = VallexOracle(case_state_dict(c), c["num_layers"])
    a, t, text, pl, langs = case_inputs(c)
    us = synth.uniforms(4096, c["best_of"], c["useed"])
    for name, worst in (("nl12_ui_bestof5_ja", False), ("nl12_ui_bestof5_ja_worst", True)):
        taps = {}
        codes = orc.inference(text, np.array([text.shape[-1]]), a, t.shape[-1], top_k=c["top_k"], temperature=c["temperature"],
                              prompt_language=pl, text_language=langs, uniforms=us, force_eos_at=c["force_eos_at"], taps=taps,
                              best_of=c["best_of"], return_worst=worst)
        np.testing.assert_array_equal(codes, np.load(os.path.join(GOLD, name + ".npz"))["codes"], err_msg=name)


def test_round3_fixture_shapes():
    """what the committed fixtures hold (cheap; the content is compared on the GPU and, with VX_SLOW=1, by the oracle)"""
    from oracle.make_golden import CHAIN_CASES, CHAIN_FRAMES, LONG_CASES, TRAINED_CASES, chain_second
    g = np.load(os.path.join(GOLD, "nl12_cap1024_en.npz"))
    c = LONG_CASES["nl12_cap1024_en"]
    a, t, text, pl, langs = case_inputs(c)
    assert text.shape[-1] == 64 and g["codes"].shape == (1, 16 * 64, 8)          # models/vallex.py:575-578
    assert g["ar_logits"].shape == (1024 // FULL_LOGIT_EVERY + 1, 1025) and g["nar_logits"].shape == (7, 16, 1024)
    assert int(g["codes"].max()) < 1024
    g = np.load(os.path.join(GOLD, "nl12_chain2_zh.npz"))
    c = CHAIN_CASES["nl12_chain2_zh"]
    a2, t2, text2, pl2, _ = chain_second(c, g["codes1"])
    assert g["codes1"].shape == g["codes"].shape == (1, CHAIN_FRAMES, 8)
    assert a2.shape == (1, CHAIN_FRAMES, 8) and t2.shape == (1, 100) and text2.shape == (1, 200) and pl2 == "zh"
    assert text2.shape[-1] + 1 + 2 * CHAIN_FRAMES == 1327                        # longest decode context
    for n in TRAINED_CASES:
        g = np.load(os.path.join(GOLD, n + ".npz"))
        assert g["codes"].shape == (1, 600, 8) and float(np.abs(g["ar_logits"]).max()) > 20.0      # decisive logits
    # the UI call: the best beam ended by itself, the worst ran to the forced end; same weights, same draws
    best, worst = (np.load(os.path.join(GOLD, n + ".npz"))["codes"] for n in ("nl12_ui_bestof5_ja", "nl12_ui_bestof5_ja_worst"))
    assert best.shape == (1, 9, 8) and worst.shape == (1, 120, 8) and int(max(best.max(), worst.max())) < 1024


def test_oracle_matches_reference_on_bench_row0():
    """the utterance bench.py's cpu_baseline leg computes on the oracle (and its parity block compares the engine against): row 0 of
    the bench workload, 12 layers, 600 frames, top-k 10 with the injected uniforms -- the oracle's ids against the ids of the LIVE
    reference for the same utterance (tests/golden/bench_row0.npz, tools/cpu_reference.py).  AR only past the first 64 frames would
    not be cheaper: the NAR stages need all 600."""
    import torch
    import bench
    g = np.load(os.path.join(GOLD, "bench_row0.npz"))
    frames = int(g["frames"])
    r = bench.make_rows(0, 1)[0]
    us = np.random.default_rng(1234).random(frames + 1).astype(np.float32)
    orc = VallexOracle(synth.vallex_state_dict(bench.NUM_LAYERS, 0, eos_gain=0.0), bench.NUM_LAYERS)
    text = torch.from_numpy(r["text"].astype(np.int64))
    prompts = torch.from_numpy(r["prompt"].astype(np.int64))
    with torch.no_grad():
        gen = orc.ar_generate(text, prompts[:, 0], r["enroll"], r["prompt_language"], r["text_language"], 10, 1.0, us, frames, None)
        codes = orc.nar_generate(text, prompts, gen, r["enroll"], r["prompt_language"], r["text_language"], None)
    np.testing.assert_array_equal(np.asarray(codes), g["codes"].astype(np.int64))
