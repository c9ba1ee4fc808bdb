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
