"""GPU parity tests: the HIP engine (through the C ABI) vs the committed outputs of the live reference
(tests/golden) and vs the CPU oracle on the same seeded inputs.  Bit-exact token ids; logits/activations within
fp32-summation-order tolerance (written at each assert)."""
import os

import numpy as np
import pytest

from oracle import synth
from oracle.make_golden import (CASES, CONTINUAL_CASES, PRESET_SHAPES, SHARP_CASES, case_inputs, continual_inputs,
                                preset_shape_case)
from oracle.vallex_oracle import VallexOracle, VocosOracle
from tests._util import case_row, get_model, golden

pytestmark = pytest.mark.gpu

NL2 = [n for n in CASES if n.startswith("nl2_")]


def _oracle(c):
    return VallexOracle(synth.vallex_state_dict(c["num_layers"], c["seed"], c["eos_gain"]), c["num_layers"])


def test_ar_prefill_activations_and_logits():
    """kernel-level: embedding rows, every prefill layer output, final-norm + predict logits."""
    name = "nl2_greedy_eos"
    c, row, _ = case_row(name)
    m = get_model(c["num_layers"], c["seed"], c["eos_gain"], debug_taps=True)
    eng = m.engine
    eng.ar_prefill(m.make_batch([row]))
    import torch
    orc = _oracle(c)
    taps = {}
    h, kv, S = orc.ar_prefill(torch.from_numpy(row["text"].astype(np.int64)), torch.from_numpy(row["prompt"][:, 0].astype(np.int64)),
                              row["enroll"], row["prompt_language"], row["text_language"], taps)
    L = taps["ar_prefill_in"].shape[0]
    got_in = eng.read_tap("ar_prefill_in", L * 1024).reshape(L, 1024)
    np.testing.assert_allclose(got_in, taps["ar_prefill_in"].numpy(), atol=1e-6, rtol=0)   # gathers + 2 adds
    for l in range(c["num_layers"]):
        got = eng.read_tap(f"ar_layer_out.{l}", L * 1024).reshape(L, 1024)
        ref = taps["ar_layer_out"][l].numpy()
        err = np.abs(got - ref).max()
        assert err < 2e-4 * max(1.0, np.abs(ref).max()), (l, err)       # fp32 reassociation over K<=4096
    logits = eng.ar_logits()[0]
    ref = orc.ar_logits(h).numpy()
    np.testing.assert_allclose(logits, ref, atol=2e-4, rtol=0)
    np.testing.assert_allclose(logits, golden(name)["ar_logits"][0], atol=3e-4, rtol=0)     # live reference
    assert int(np.argmax(logits)) == int(golden(name)["codes"][0, 0, 0])


@pytest.mark.parametrize("name", ["nl2_greedy_eos", "nl2_force40_mixlang"])
def test_ar_teacher_forced_steps(name):
    """feed the reference's own tokens; every cached decode step must reproduce its logits / argmax."""
    c, row, _ = case_row(name)
    g = golden(name)
    m = get_model(c["num_layers"], c["seed"], c["eos_gain"], debug_taps=True)
    eng = m.engine
    eng.ar_prefill(m.make_batch([row]))
    codes0 = g["codes"][0, :, 0]
    for t in range(len(codes0)):
        lg = eng.ar_logits()[0]
        if t < g["ar_logits"].shape[0]:
            np.testing.assert_allclose(lg, g["ar_logits"][t], atol=3e-4, rtol=0)
        assert int(np.argmax(lg)) == int(codes0[t]), f"step {t}"
        eng.ar_step(np.array([codes0[t]], np.int32))


@pytest.mark.parametrize("name", NL2)
def test_infer_matches_reference_tokens(name):
    """VALLE.inference drop-in: all 8 codebooks bit-exact vs the live reference run (tests/golden)."""
    c, row, us = case_row(name)
    m = get_model(c["num_layers"], c["seed"], c["eos_gain"])
    out = m.inference(row["text"][None], np.array([len(row["text"])]), row["prompt"][None], row["enroll"],
                      top_k=c["top_k"], temperature=c.get("temperature", 1.0), prompt_language=row["prompt_language"],
                      text_language=row["text_language"], uniforms=us, force_eos_at=c["force_eos_at"],
                      best_of=c.get("best_of", 1), length_penalty=c.get("length_penalty", 1.0),
                      return_worst=c.get("return_worst", False))
    g = golden(name)["codes"]
    assert tuple(out.shape) == g.shape
    np.testing.assert_array_equal(out.numpy(), g)


def test_all_41_preset_shapes_match_reference():
    """prompt shapes and languages of every reference preset (161..758 frames, 18..160 prompt text ids, zh/ja/en prompts with a
    different text language each): bit-exact ids against the live reference, each row alone AND all 41 in one call (two
    micro-batches of the engine, ragged lengths)."""
    g = golden("preset_shapes")
    c0 = preset_shape_case(0)
    m = get_model(c0["num_layers"], c0["seed"], c0["eos_gain"], max_new=64, max_prompt=800, max_text=256, max_batch=41)
    rows = []
    for i in range(len(PRESET_SHAPES)):
        c = preset_shape_case(i)
        a, t, text, pl, langs = case_inputs(c)
        rows.append(dict(text=text[0], prompt=a[0], enroll=t.shape[-1], prompt_language=pl, text_language=langs))
    for i in (0, 4, 9, 22, 29, 40):
        out = m.inference_batch([rows[i]], top_k=1, force_eos_at=c0["force_eos_at"])[0]
        np.testing.assert_array_equal(out, g["codes"][i].astype(np.int64), err_msg=PRESET_SHAPES[i][0])
    outs = m.inference_batch(rows, top_k=1, force_eos_at=c0["force_eos_at"])
    for i, o in enumerate(outs):
        np.testing.assert_array_equal(o, g["codes"][i].astype(np.int64), err_msg=PRESET_SHAPES[i][0])


def test_infer_12_layers_matches_reference():
    name = "nl12_c1_short"
    c, row, us = case_row(name)
    m = get_model(12, c["seed"], c["eos_gain"])
    out = m.inference(row["text"][None], np.array([len(row["text"])]), row["prompt"][None], row["enroll"], top_k=1,
                      prompt_language=row["prompt_language"], text_language=row["text_language"],
                      force_eos_at=c["force_eos_at"])
    np.testing.assert_array_equal(out.numpy(), golden(name)["codes"])


def test_nar_logits_and_codes():
    name = "nl2_greedy_eos"
    c, row, _ = case_row(name)
    g = golden(name)
    m = get_model(c["num_layers"], c["seed"], c["eos_gain"], debug_taps=True)
    codes = m.engine.nar(m.make_batch([row]), [g["codes"][0, :, 0].astype(np.int32)])[0]
    np.testing.assert_array_equal(codes, g["codes"][0])
    T = g["codes"].shape[1]
    lg = m.engine.read_tap("nar_logits0", T * 1024).reshape(T, 1024)
    np.testing.assert_allclose(lg[:16], g["nar_logits0"], atol=5e-3, rtol=0)   # logits std ~25, K=1024


def test_ragged_batch_rows_equal_their_batch1_runs():
    """distinct utterances in one call (what BASELINE configs 3/4 need; no reference equivalent): row i of a ragged
    batch == the reference run on row i alone.  All rows share weights, so use cases with the same seed/weights."""
    base = CASES["nl2_greedy_eos"]
    m = get_model(base["num_layers"], base["seed"], base["eos_gain"])
    orc = _oracle(base)
    rows, refs = [], []
    for i, (tp, sp, nt, lang) in enumerate([(20, 6, 9, "en"), (57, 11, 5, "zh"), (3, 2, 14, "ja"), (33, 9, 7, "en")]):
        a, t = synth.synth_prompt(tp, sp, seed=50 + i)
        txt = np.concatenate([t[0], synth.synth_text(nt, 50 + i)])
        rows.append(dict(text=txt, prompt=a[0], enroll=sp, prompt_language=lang, text_language=lang))
        refs.append(orc.inference(txt[None], np.array([len(txt)]), a, sp, top_k=1, prompt_language=lang,
                                  text_language=lang, force_eos_at=30 + 3 * i)[0])
    # per-row force_eos is a scalar in the ABI: run rows with their own cap one by one AND all together with EOS gain
    outs = [m.inference_batch([r], top_k=1, force_eos_at=30 + 3 * i)[0] for i, r in enumerate(rows)]
    for o, r in zip(outs, refs):
        np.testing.assert_array_equal(o, r)
    refs_b = [orc.inference(r["text"][None], np.array([len(r["text"])]), r["prompt"][None], r["enroll"], top_k=1,
                            prompt_language=r["prompt_language"], text_language=r["text_language"], force_eos_at=25)[0]
              for r in rows]
    outs_b = m.inference_batch(rows, top_k=1, force_eos_at=25)
    for o, r in zip(outs_b, refs_b):
        np.testing.assert_array_equal(o, r)


def test_vocos_head_matches_oracle():
    """waveform RMS error <= 1e-4 (north_star tolerance) vs the torch restatement of the pip `vocos` arithmetic."""
    m = get_model(2, 0, 2.5, vocos=True)
    rng = np.random.default_rng(5)
    codes = [rng.integers(0, 1024, size=(T, 8), dtype=np.int64) for T in (37, 5, 64)]
    got = m.engine.vocos_decode(codes, 2)
    orc = VocosOracle(synth.vocos_state_dict(2))
    for c, a in zip(codes, got):
        ref = orc.decode_codes(c[None], 2)[0]
        assert a.shape == ref.shape == (c.shape[0] * 320,)
        rms_err = float(np.sqrt(np.mean((a - ref) ** 2)))
        assert rms_err <= 1e-4, rms_err
        assert abs(np.sqrt(np.mean(a ** 2)) - np.sqrt(np.mean(ref ** 2))) <= 1e-4


def test_vocos_long_input_is_windowed_bit_identically():
    """an input longer than the Vocos arena (the reference decodes a whole long text in one call, utils/generation.py:271-273)
    is decoded in overlapping windows: bit-identical to a single pass of an engine whose arena holds it, and within the
    north_star tolerance of the CPU restatement."""
    rng = np.random.default_rng(11)
    codes = [rng.integers(0, 1024, size=(T, 8), dtype=np.int64) for T in (1300, 40)]
    small = get_model(2, 0, 2.5, vocos=True, max_new=64, max_batch=2)          # arena: 512 frames per pass -> 4 windows for row 0
    got = small.engine.vocos_decode(codes, 2)
    big = get_model(2, 0, 2.5, vocos=True, max_new=1400, max_batch=2)          # arena holds both rows in one pass
    ref = big.engine.vocos_decode(codes, 2)
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)
    orc = VocosOracle(synth.vocos_state_dict(2)).decode_codes(codes[0][None], 2)[0]
    assert float(np.sqrt(np.mean((got[0] - orc) ** 2))) <= 1e-4


def test_generate_audio_api_end_to_end():
    """utils.generation drop-in: preset .npz in, float32 waveform out; deterministic via injected uniforms, checked
    against reference-arithmetic run on the CPU (oracle AR+NAR, Vocos restatement)."""
    import os
    from vallex_amd.utils import generation as G
    sd = synth.vallex_state_dict(2, 11)
    vsd = synth.vocos_state_dict(2)
    G.preload_models(state_dict=sd, vocos_state_dict=vsd, num_layers=2, max_new=320, max_prompt=400, max_text=256,
                     max_batch=4)
    preset = os.path.join(os.path.dirname(__file__), "golden", "presets", "paimon.npz")
    ids = synth.synth_text(14, 3)
    us = synth.uniforms(64, 1, 99)[:, 0]
    wav = G.generate_audio(ids, prompt=preset, language="en", uniforms=us, force_eos_at=24)
    assert wav.dtype == np.float32 and wav.ndim == 1 and wav.shape[0] == 24 * 320 and np.isfinite(wav).all()
    d = np.load(preset)
    text = np.concatenate([d["text_tokens"][0], ids])[None]
    orc = VallexOracle(sd, 2)
    codes = orc.inference(text, np.array([text.shape[1]]), d["audio_tokens"], d["text_tokens"].shape[1], top_k=-100,
                          temperature=1.0, prompt_language="zh", text_language="en", uniforms=us, force_eos_at=24)
    ref = VocosOracle(vsd).decode_codes(codes, 2)[0]
    assert float(np.sqrt(np.mean((wav - ref) ** 2))) <= 1e-4
    # long-text, fixed-prompt mode: two pre-split "sentences" of ids
    G.rng = np.random.default_rng(0)
    wav2 = G.generate_audio_from_long_text([ids, synth.synth_text(9, 4)], prompt=preset, language="en",
                                           mode="fixed-prompt", uniforms=us, force_eos_at=10)
    assert wav2.shape[0] == 2 * 10 * 320 and np.isfinite(wav2).all()


def test_generate_audio_batch_equals_per_utterance_calls():
    """the batch form (one inference_batch + one Vocos call; SURVEY section 8 f4 service path) returns, per utterance, exactly
    what `generate_audio` returns for that utterance alone"""
    import os
    from vallex_amd.utils import generation as G
    sd = synth.vallex_state_dict(2, 11)
    G.preload_models(state_dict=sd, vocos_state_dict=synth.vocos_state_dict(2), num_layers=2, max_new=320, max_prompt=400,
                     max_text=256, max_batch=4)
    pdir = os.path.join(os.path.dirname(__file__), "golden", "presets")
    texts = [synth.synth_text(12, 31), synth.synth_text(7, 32), synth.synth_text(15, 33)]
    prompts = [os.path.join(pdir, "paimon.npz"), None, os.path.join(pdir, "cafe.npz")]
    langs = ["en", "zh", "ja"]
    us = synth.uniforms(64, 3, 123)
    batch = G.generate_audio_batch(texts, prompts=prompts, language=langs, uniforms=us, force_eos_at=20)
    for i in range(3):
        alone = G.generate_audio(texts[i], prompt=prompts[i], language=langs[i], uniforms=us[:, i], force_eos_at=20)
        np.testing.assert_array_equal(batch[i], alone)


def test_long_text_sliding_window_carries_prompt():
    """generate_audio_from_long_text(mode='sliding-window') with the carry branch always taken (launch-ui.py:493
    behaviour): chunk k+1 is prompted by ALL frames and the text ids of chunk k (utils/generation.py:264-266)."""
    import os
    from vallex_amd.utils import generation as G
    sd = synth.vallex_state_dict(2, 11)
    vsd = synth.vocos_state_dict(2)
    G.preload_models(state_dict=sd, vocos_state_dict=vsd, num_layers=2, max_new=320, max_prompt=400, max_text=256,
                     max_batch=4)
    preset = os.path.join(os.path.dirname(__file__), "golden", "presets", "paimon.npz")
    s1, s2 = synth.synth_text(9, 21), synth.synth_text(7, 22)

    class Always:
        def random(self):
            return 0.0
    G.rng = Always()
    # generate_audio* fix top_k=-100 like the reference; make the draw deterministic with injected uniforms instead
    us = synth.uniforms(64, 1, 5)[:, 0]
    wav = G.generate_audio_from_long_text([s1, s2], prompt=preset, language="en", mode="sliding-window", uniforms=us,
                                          force_eos_at=12)
    d = np.load(preset)
    orc = VallexOracle(sd, 2)
    t1 = np.concatenate([d["text_tokens"][0], s1])[None]
    c1 = orc.inference(t1, np.array([t1.shape[1]]), d["audio_tokens"], d["text_tokens"].shape[1], top_k=-100,
                       prompt_language="zh", text_language="en", uniforms=us, force_eos_at=12)
    t2 = np.concatenate([s1, s2])[None]
    c2 = orc.inference(t2, np.array([t2.shape[1]]), c1, len(s1), top_k=-100, prompt_language="zh", text_language="en",
                       uniforms=us, force_eos_at=12)
    ref = VocosOracle(vsd).decode_codes(np.concatenate([c1, c2], axis=1), 2)[0]
    assert wav.shape == ref.shape
    assert float(np.sqrt(np.mean((wav - ref) ** 2))) <= 1e-4


def test_edge_cases_no_prompt_empty_result_and_two_microbatches():
    """(a) no audio prompt at all (utils/generation.py:121-123: empty prompts, enroll_x_lens = 0);
    (b) EOS as the very first sample -> the reference returns an EMPTY (1,0,8) tensor (prepend_bos path, see oracle);
    (c) 35 rows = two AR micro-batches of the engine (32 + 3): every row still equals its batch-1 run."""
    base = CASES["nl2_greedy_eos"]
    m = get_model(base["num_layers"], base["seed"], base["eos_gain"], max_batch=40)
    orc = _oracle(base)
    # (a)
    txt = synth.synth_text(11, 77)
    empty = np.zeros((1, 0, 8), np.int64)
    ref = orc.inference(txt[None], np.array([11]), empty, 0, top_k=1, prompt_language="en", text_language="en",
                        force_eos_at=9)
    out = m.inference(txt[None], np.array([11]), empty, 0, top_k=1, prompt_language="en", text_language="en",
                      force_eos_at=9)
    np.testing.assert_array_equal(out.numpy(), ref)
    # (b)
    a, t = synth.synth_prompt(12, 4, seed=5)
    text = np.concatenate([t[0], txt])[None]
    out0 = m.inference(text, np.array([text.shape[1]]), a, 4, top_k=1, prompt_language="en", text_language="en",
                       force_eos_at=0)
    assert tuple(out0.shape) == (1, 0, 8)
    assert orc.inference(text, np.array([text.shape[1]]), a, 4, top_k=1, prompt_language="en", text_language="en",
                         force_eos_at=0).shape == (1, 0, 8)
    assert m.engine.vocos_decode([out0.numpy()[0]], 2)[0].shape == (0,) if m._vocos_sd is not None else True
    # (c)
    rows = []
    for i in range(35):
        ap, tp = synth.synth_prompt(5 + (i * 3) % 17, 2 + i % 5, seed=300 + i)
        tx = np.concatenate([tp[0], synth.synth_text(4 + i % 6, 300 + i)])
        rows.append(dict(text=tx, prompt=ap[0], enroll=tp.shape[1], prompt_language=("en", "zh", "ja")[i % 3],
                         text_language=("en", "zh", "ja")[(i + 1) % 3]))
    outs = m.inference_batch(rows, top_k=1, force_eos_at=6)
    for i in (0, 13, 31, 32, 34):
        r = rows[i]
        ref = orc.inference(r["text"][None], np.array([len(r["text"])]), r["prompt"][None], r["enroll"], top_k=1,
                            prompt_language=r["prompt_language"], text_language=r["text_language"], force_eos_at=6)[0]
        np.testing.assert_array_equal(outs[i], ref)


def test_engine_rejects_bad_arguments():
    import vallex_amd
    base = CASES["nl2_greedy_eos"]
    m = get_model(base["num_layers"], base["seed"], base["eos_gain"])
    a, t = synth.synth_prompt(8, 3, seed=1)
    long_text = np.full(300, 7, np.int32)                     # > max_text (256)
    with pytest.raises(vallex_amd.VallexHipError):
        m.inference_batch([dict(text=long_text, prompt=a[0], enroll=3, prompt_language="en", text_language="en")], top_k=1)
    bad = a[0].copy()
    bad[0, 0] = 5000                                          # codebook id out of range
    with pytest.raises(vallex_amd.VallexHipError):
        m.inference_batch([dict(text=t[0], prompt=bad, enroll=3, prompt_language="en", text_language="en")], top_k=1)
    with pytest.raises(vallex_amd.VallexHipError):
        m.inference_batch([dict(text=t[0], prompt=a[0], enroll=3, prompt_language="en", text_language="en")], top_k=1,
                          temperature=0.0)


@pytest.mark.parametrize("name", sorted(CONTINUAL_CASES))
def test_continual_matches_reference(name):
    """`VALLE.continual` (models/vallex.py:688-787) through vx_nar with language id -1 (= no language embedding):
    bit-exact ids against the live reference's own continual() output."""
    c = CONTINUAL_CASES[name]
    m = get_model(c["num_layers"], c["seed"], c["eos_gain"])
    text, y = continual_inputs(c)
    out = m.continual(text, np.array([text.shape[-1]]), y)
    out = out.numpy() if hasattr(out, "numpy") else out
    np.testing.assert_array_equal(out, golden(name)["codes"])


@pytest.mark.parametrize("name", sorted(SHARP_CASES))
def test_infer_matches_reference_tokens_sharp_attention(name):
    """The attn_gain-3 goldens (peaky attention: sensitive to K/V precision and score arithmetic) through the C ABI."""
    c = SHARP_CASES[name]
    m = get_model(c["num_layers"], c["seed"], c["eos_gain"], attn_gain=c["attn_gain"])
    a, t, text, pl, langs = case_inputs(c)
    us = None if c["useed"] is None else synth.uniforms(4096, 1, c["useed"])[:, 0]
    out = m.inference(text, np.array([text.shape[-1]]), a, t.shape[-1], top_k=c["top_k"], prompt_language=pl, text_language=langs,
                      uniforms=us, force_eos_at=c["force_eos_at"])
    out = out.numpy() if hasattr(out, "numpy") else out
    np.testing.assert_array_equal(out, golden(name)["codes"])
