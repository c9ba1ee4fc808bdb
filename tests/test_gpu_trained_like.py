"""GPU: weights with the statistics of a TRAINED checkpoint (the real vallex-checkpoint.pt of utils/generation.py:79-83 is not
available offline): heavy-tailed projection weights, LayerNorm gains U(0.5, 4), massive FFN channels and residual dimensions,
decisive AR logits (oracle/synth.py: trained_like_state_dict).  Two 12-layer, 600-frame runs of the LIVE reference on them
(oracle/make_golden.py TRAINED_CASES) must come back bit for bit in every arithmetic of the full-sequence path -- f16x2 (the
default), bf16x3 and exact fp32 -- and the default must get there WITHOUT falling back to fp32 (the operands of these weights
stay inside the fp16 range: the format's margin, not the fallback, carries them)."""
import numpy as np
import pytest

from oracle import synth
from oracle.make_golden import CONTINUAL12_CASES, FULL_LOGIT_EVERY, TRAINED_CASES, UI_CASES, case_inputs, continual_inputs
from tests._util import assert_codes, case_model, golden, inputs_row, nar_logit_error, teacher_forced_logit_error

pytestmark = pytest.mark.gpu

KW = dict(max_new=608, max_prompt=400, max_text=256, max_batch=4)
# abs tolerances on logits with std ~9 (AR) / ~48 (NAR) and |max| ~35 / ~220; measured on MI355X 6.1e-5 / 6.2e-4 in every
# arithmetic mode (profiles/r03_logit_error.json); the reference's smallest margins here are 8.5e-3 (AR) / 1.8e-3 (NAR)
AR_TOL, NAR_TOL = 2e-4, 2e-3


@pytest.mark.parametrize("arith", ["f16x2", "bf16x3", "f32"])
@pytest.mark.parametrize("name", sorted(TRAINED_CASES))
def test_trained_like_weights_match_the_live_reference(name, arith):
    c = TRAINED_CASES[name]
    g = golden(name)
    row, us = inputs_row(c)
    m = case_model(c, arith=arith, **KW)
    assert m.engine.arith_mode() == (arith, arith)
    out = m.inference_batch([row], top_k=c["top_k"], uniforms=None if us is None else us[:, None],
                            force_eos_at=c["force_eos_at"])[0]
    fb = m.engine.last_fallbacks()
    assert_codes(f"{name} [{arith}]", out, g)
    assert fb["prefill"] == 0 and fb["nar"] == 0, f"{arith}: the run needed the fp32 fallback {fb}"


@pytest.mark.parametrize("arith", ["f16x2", "f32"])
def test_trained_like_logits(arith):
    name = "nl12_trained_en_greedy"
    c = TRAINED_CASES[name]
    g = golden(name)
    row, _ = inputs_row(c)
    m = case_model(c, arith=arith, debug_taps=True, **KW)
    worst, flips = teacher_forced_logit_error(m, row, g, FULL_LOGIT_EVERY)
    assert flips == 0, f"{flips} of 600 greedy decisions differ (min reference margin {g['ar_margin'].min():.2e})"
    assert worst <= AR_TOL, worst
    codes, errs = nar_logit_error(m, row, g)
    assert max(errs) <= NAR_TOL, errs
    np.testing.assert_array_equal(codes, g["codes"][0])
    print(f"{name} [{arith}]: AR max |logit - ref| {worst:.2e}; NAR per stage {['%.1e' % e for e in errs]}")


@pytest.mark.parametrize("name,nrows,slot", [("nl12_trained_en_greedy", 32, 11), ("nl12_trained_zh_topk10", 8, 6)])
def test_trained_like_row_inside_a_batch(name, nrows, slot):
    """the same golden rows through the other two decode chains: 32 rows (one context split, out_proj fused into dec_attn, 16 head
    slabs) and 8 rows (context-split dec_attn + combine + separate out_proj) -- alone they take the small-batch chain"""
    c = TRAINED_CASES[name]
    g = golden(name)
    row, us = inputs_row(c)
    m = case_model(c, max_new=608, max_prompt=400, max_text=256, max_batch=32)
    rows, cols = [], []
    rng = np.random.default_rng(5)
    for i in range(nrows):
        if i == slot:
            rows.append(row)
            cols.append(us if us is not None else np.zeros(4096, np.float32))
            continue
        tp, sp = int(rng.integers(150, 300)), int(rng.integers(20, 80))
        a, t = synth.synth_prompt(tp, sp, seed=61_000 + i)
        lang = ("en", "zh", "ja")[i % 3]
        rows.append(dict(text=np.concatenate([t[0], synth.synth_text(100, 61_000 + i)]), prompt=a[0], enroll=sp, prompt_language=lang,
                         text_language=lang))
        cols.append(synth.uniforms(4096, 1, 62_000 + i)[:, 0])
    outs = m.inference_batch(rows, top_k=c["top_k"], uniforms=None if us is None else np.stack(cols, axis=1),
                             force_eos_at=c["force_eos_at"])
    assert m.engine.last_fallbacks()["lifetime"] == 0
    assert_codes(f"{name} as row {slot} of {nrows}", outs[slot], g)


@pytest.mark.parametrize("name", sorted(UI_CASES))
def test_ui_call_best_of_5_matches_the_live_reference(name):
    """launch-ui.py:285-295: top_k=-100 (unfiltered multinomial), temperature 1, best_of=5, on 12 trained-like layers with a live
    EOS logit -- four beams end by themselves at different steps, one runs on; the reference's pick (best, and the worst with
    return_worst) must come back bit for bit.  Five beams decode on the context-split chain (5..31 rows)."""
    c = UI_CASES[name]
    a, t, text, pl, langs = case_inputs(c)
    m = case_model(c, max_new=128, max_prompt=700, max_text=256, max_batch=8)
    us = synth.uniforms(4096, c["best_of"], c["useed"])
    out = m.inference(text, np.array([text.shape[-1]]), a, t.shape[-1], top_k=c["top_k"], temperature=c["temperature"],
                      prompt_language=pl, text_language=langs, uniforms=us, force_eos_at=c["force_eos_at"], best_of=c["best_of"],
                      return_worst=c.get("return_worst", False))
    g = golden(name)["codes"]
    assert tuple(out.shape) == g.shape, (tuple(out.shape), g.shape)
    np.testing.assert_array_equal(out.numpy(), g)
    assert m.engine.last_fallbacks()["lifetime"] == 0



@pytest.mark.parametrize("arith", ["f16x2", "f32"])
def test_continual_on_trained_like_weights(arith):
    """`VALLE.continual` (models/vallex.py:688-787) on the 12 trained-like layers: 225 given + 225 continued frames, the seven NAR
    stages only; the live reference's ids (smallest arg-max margin 2.8e-2 on logits to |220|)"""
    name = "nl12_continual_trained"
    c = CONTINUAL12_CASES[name]
    m = case_model(c, arith=arith, **KW)
    text, y = continual_inputs(c)
    out = m.continual(text, np.array([text.shape[-1]]), y)
    out = out.numpy() if hasattr(out, "numpy") else out
    np.testing.assert_array_equal(out, golden(name)["codes"])
    assert m.engine.last_fallbacks()["nar"] == 0
