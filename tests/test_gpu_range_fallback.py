"""GPU: operands outside the fp16 range of the f16x2 kernels (|activation|, |q|/8, |k|, |v| >= 2047).

The reference computes in fp32 and puts no bound on the FFN hidden activations (modules/transformer.py:371-373) or on q / k / v
(modules/activation.py:144-166).  `synth.out_of_range_state_dict` rescales channels by 2^12 and their read-out by 2^-12: in fp32
that is THE SAME function bit for bit (oracle/make_golden.py RANGE_CASES: the live reference returns the base case's golden, logit
difference 0.0), but the hidden activations / values / keys are now ~3000-5000.  The engine must notice on the device, re-run
the affected phase on the exact-fp32 kernels by itself and return the reference's ids -- the call must not fail, and must not
need an environment variable."""
import numpy as np
import pytest

from oracle.make_golden import RANGE_CASES, all_cases
from tests._util import assert_codes, case_model, golden, inputs_row

pytestmark = pytest.mark.gpu

ALL = all_cases()


def _run(m, c, rows, us):
    return m.inference_batch(rows, top_k=c["top_k"], temperature=c.get("temperature", 1.0),
                             uniforms=None if us is None else us, force_eos_at=c["force_eos_at"])


@pytest.mark.parametrize("name", sorted(RANGE_CASES))
def test_out_of_range_operands_fall_back_to_fp32(name):
    base, kind = RANGE_CASES[name]
    c = ALL[name]
    row, us = inputs_row(c)
    m = case_model(c)                                   # default arithmetic: f16x2 projections + attention
    assert m.engine.arith_mode() == ("f16x2", "f16x2")
    out = _run(m, c, [row], None if us is None else us[:, None])[0]
    fb = m.engine.last_fallbacks()
    assert fb["prefill"] >= 1 and fb["nar"] >= 1, (name, fb)      # both stacks carry the rescaled channels
    assert_codes(name, out, golden(base))
    # the same row next to an ordinary-length second row (two rows per call), and once more: the flag was cleared
    row2 = dict(row, text=row["text"][: max(row["enroll"] + 3, len(row["text"]) - 4)])
    us2 = None if us is None else np.stack([us, us[::-1]], axis=1)
    outs = _run(m, c, [row, row2], us2)
    assert_codes(name + " (row 0 of 2)", outs[0], golden(base))
    assert m.engine.last_fallbacks()["prefill"] >= 1


def test_only_the_phase_that_needs_it_is_rerun():
    """AR stack rescaled, NAR stack untouched (and vice versa): exactly that phase falls back"""
    from oracle import synth
    from tests import _util
    base, kind = RANGE_CASES["nl2_range_ffn"]
    c = ALL[base]
    row, us = inputs_row(c)
    sd = synth.vallex_state_dict(c["num_layers"], c["seed"], c["eos_gain"])
    for stacks, want in ((("ar",), (1, 0)), (("nar",), (0, 1))):
        m = _util.VALLE(1024, 16, c["num_layers"], norm_first=True, add_prenet=False, prefix_mode=1, share_embedding=True,
                        nar_scale_factor=1.0, prepend_bos=True, num_quantizers=8, engine_max_new=320, engine_max_prompt=400,
                        engine_max_text=256, engine_max_batch=4)
        m.to("cuda:0").load_state_dict(synth.out_of_range_state_dict(sd, c["num_layers"], kind, stacks=stacks), strict=True)
        out = _run(m, c, [row], None if us is None else us[:, None])[0]
        fb = m.engine.last_fallbacks()
        assert (fb["prefill"], fb["nar"]) == want, (stacks, fb)
        assert_codes(f"{base} / {stacks}", out, golden(base))


def test_in_range_weights_never_fall_back_and_fp32_mode_has_no_guard():
    base, kind = RANGE_CASES["nl2_range_v"]
    c = ALL[base]
    row, us = inputs_row(c)
    m = case_model(c)
    out = _run(m, c, [row], None if us is None else us[:, None])[0]
    assert m.engine.last_fallbacks() == dict(prefill=0, nar=0, lifetime=0)
    assert_codes(base, out, golden(base))
    c2 = ALL["nl2_range_v"]
    m32 = case_model(c2, arith="f32")
    assert m32.engine.arith_mode() == ("f32", "f32")
    out = _run(m32, c2, [row], None if us is None else us[:, None])[0]
    assert m32.engine.last_fallbacks()["lifetime"] == 0
    assert_codes("nl2_range_v in fp32 mode", out, golden(base))


def test_fallback_with_best_of_beams_and_continual():
    """the two other callers of the guarded phases: best_of = 3 (ONE shared prefill broadcast to three decode rows, re-run in fp32
    before the beams start) and VALLE.continual (NAR only, vx_nar)"""
    from oracle import synth
    from oracle.make_golden import CASES, CONTINUAL_CASES, continual_inputs
    from tests import _util
    for name in ("nl2_bestof3", "nl2_bestof3_worst"):
        c = CASES[name]
        _, row, us = _util.case_row(name)
        sd = synth.out_of_range_state_dict(synth.vallex_state_dict(c["num_layers"], c["seed"], c["eos_gain"]), c["num_layers"], "ffn")
        m = _util.VALLE(1024, 16, c["num_layers"], norm_first=True, add_prenet=False, prefix_mode=1, share_embedding=True,
                        nar_scale_factor=1.0, prepend_bos=True, num_quantizers=8, engine_max_new=320, engine_max_prompt=400,
                        engine_max_text=256, engine_max_batch=4)
        m.to("cuda:0").load_state_dict(sd, strict=True)
        out = m.inference(row["text"][None], np.array([len(row["text"])]), row["prompt"][None], row["enroll"], top_k=c["top_k"],
                          prompt_language=row["prompt_language"], text_language=row["text_language"], uniforms=us,
                          force_eos_at=c["force_eos_at"], best_of=c["best_of"], length_penalty=c.get("length_penalty", 1.0),
                          return_worst=c.get("return_worst", False))
        fb = m.engine.last_fallbacks()
        assert fb["prefill"] == 1 and fb["nar"] == 1, fb
        np.testing.assert_array_equal(out.numpy(), golden(name)["codes"])
    cc = CONTINUAL_CASES["nl2_continual"]
    sd = synth.out_of_range_state_dict(synth.vallex_state_dict(cc["num_layers"], cc["seed"], cc["eos_gain"]), cc["num_layers"], "v")
    m = _util.VALLE(1024, 16, cc["num_layers"], norm_first=True, add_prenet=False, prefix_mode=1, share_embedding=True,
                    nar_scale_factor=1.0, prepend_bos=True, num_quantizers=8, engine_max_new=320, engine_max_prompt=400,
                    engine_max_text=256, engine_max_batch=4)
    m.to("cuda:0").load_state_dict(sd, strict=True)
    text, y = continual_inputs(cc)
    out = m.continual(text, np.array([text.shape[-1]]), y)
    assert m.engine.last_fallbacks() == dict(prefill=0, nar=1, lifetime=1)
    np.testing.assert_array_equal(np.asarray(out), golden("nl2_continual")["codes"])


def test_outlier_weights_lose_no_ids_and_need_no_fallback():
    """PRECISION rather than range: one weight per projection tensor at 1000 x the init bound.  Each weight tensor's f16x2 planes
    are scaled from its own max |w|, so the outlier pushes every ordinary weight 10 bits down in its fp16 head + tail pair (tails
    towards fp16 subnormals).  Ids must still be the live reference's, logits within the goldens' margins, and the range guard
    must stay silent (nothing is out of RANGE)."""
    from oracle.make_golden import OUTLIER_CASES
    from tests._util import nar_logit_error, teacher_forced_logit_error
    for name, c in sorted(OUTLIER_CASES.items()):
        g = golden(name)
        row, us = inputs_row(c)
        for arith in ("f16x2", "f32"):
            m = case_model(c, arith=arith, debug_taps=True)
            out = _run(m, c, [row], None if us is None else us[:, None])[0]
            assert m.engine.last_fallbacks()["lifetime"] == 0, (name, arith)
            assert_codes(f"{name} [{arith}]", out, g)
            worst, flips = teacher_forced_logit_error(m, row, g, 50)
            assert worst <= 1e-4, (name, arith, worst)                   # reference's own smallest AR margin here: 1.7e-4
            if c["top_k"] == 1:
                assert flips == 0, (name, arith, flips)
            codes, errs = nar_logit_error(m, row, g)
            assert max(errs) <= 2e-3, (name, arith, errs)                # smallest NAR margin: 2.8e-2
            assert_codes(f"{name} NAR [{arith}]", codes, g)


def test_a_model_that_keeps_leaving_the_range_goes_straight_to_fp32():
    """sticky fallback: after two raises of a phase kind the context runs that kind on the fp32 kernels directly -- every call
    still returns the reference's ids and still reports the phases it ran in fp32"""
    base, kind = RANGE_CASES["nl2_range_ffn"]
    c = ALL["nl2_range_ffn"]
    row, us = inputs_row(c)
    m = case_model(c)
    life = m.engine.last_fallbacks()["lifetime"]
    for call in range(4):
        out = _run(m, c, [row], None if us is None else us[:, None])[0]
        fb = m.engine.last_fallbacks()
        assert (fb["prefill"], fb["nar"]) == (1, 1), (call, fb)
        assert fb["lifetime"] == life + 2 * (call + 1), (call, fb)
        assert_codes(f"nl2_range_ffn call {call}", out, golden(base))


def test_sticky_mode_is_reported_warned_about_and_can_be_left():
    """ABI 5 (vx_fallback_state / vx_fallback_reset): the context says when it is in sticky mode, Engine.infer warns at the moment
    the mode engages (not only at the first raise), a reset leaves it at once, and the consecutive-raise count starts again"""
    import warnings
    base, kind = RANGE_CASES["nl2_range_ffn"]
    c = ALL["nl2_range_ffn"]
    row, us = inputs_row(c)
    m = case_model(c)
    e = m.engine
    e.fallback_reset()
    st = e.fallback_state()
    assert (st["prefill"], st["nar"]) == (False, False), st
    t0 = st["times_engaged"]
    e._fb_warned, e._sticky_seen = False, t0
    u = None if us is None else us[:, None]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = _run(m, c, [row], u)[0]                       # first raise of each kind: re-run, not sticky yet
        assert_codes("sticky call 0", out, golden(base))
        st = e.fallback_state()
        assert (st["prefill"], st["nar"], st["times_engaged"]) == (False, False, t0), st
        assert any("left the fp16 range" in str(x.message) for x in w) and not any("ENGAGED" in str(x.message) for x in w)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = _run(m, c, [row], u)[0]                       # second CONSECUTIVE raise of each kind: both kinds go sticky
        assert_codes("sticky call 1", out, golden(base))
        st = e.fallback_state()
        assert (st["prefill"], st["nar"], st["times_engaged"]) == (True, True, t0 + 2), st
        assert any("ENGAGED" in str(x.message) for x in w), [str(x.message) for x in w]
    e.fallback_reset()
    st = e.fallback_state()
    assert (st["prefill"], st["nar"], st["times_engaged"]) == (False, False, t0 + 2), st
    out = _run(m, c, [row], u)[0]                           # after the reset one raise is not enough again
    assert_codes("sticky call 2", out, golden(base))
    assert e.fallback_state()["prefill"] is False and e.last_fallbacks()["prefill"] == 1


def test_a_clean_phase_resets_the_consecutive_count():
    """two raises that are NOT consecutive (a clean pass of the kind between them) never engage sticky mode: the AR-only
    rescaled model raises in the prefill and runs clean NAR phases, so its NAR count stays 0 and only the prefill goes sticky;
    an in-range model never moves either count"""
    from oracle import synth
    from tests import _util
    base, kind = RANGE_CASES["nl2_range_ffn"]
    c = ALL[base]
    row, us = inputs_row(c)
    u = None if us is None else us[:, None]
    sd = synth.vallex_state_dict(c["num_layers"], c["seed"], c["eos_gain"])
    m = _util.VALLE(1024, 16, c["num_layers"], norm_first=True, add_prenet=False, prefix_mode=1, share_embedding=True,
                    nar_scale_factor=1.0, prepend_bos=True, num_quantizers=8, engine_max_new=320, engine_max_prompt=400,
                    engine_max_text=256, engine_max_batch=4)
    m.to("cuda:0").load_state_dict(synth.out_of_range_state_dict(sd, c["num_layers"], kind, stacks=("ar",)), strict=True)
    for call in range(3):
        assert_codes(f"ar-only call {call}", _run(m, c, [row], u)[0], golden(base))
        st = m.engine.fallback_state()
        assert st["nar"] is False and st["prefill"] is (call >= 1), (call, st)
    m2 = case_model(c)                                      # the in-range base model
    for call in range(3):
        assert_codes(f"in-range call {call}", _run(m2, c, [row], u)[0], golden(base))
    assert m2.engine.fallback_state() == dict(prefill=False, nar=False, times_engaged=0)
    assert m2.engine.last_fallbacks()["lifetime"] == 0
