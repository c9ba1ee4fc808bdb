"""GPU parity at the shapes bench.py times: 12 layers, a reference preset prompt + 100
phoneme ids, 600 frames (8.0 s), Ltot = S + Tp + 600 ~ 860-1090, decode contexts up to ~1090 -- against the committed outputs of
the LIVE reference (tests/golden/nl12_full_*.npz, made by oracle/make_golden.py FULL_CASES; weights = the bench weights).

Why these exist: with >= 2 such rows in one call the packed row count is >= 1024, so the projections run on
the large-M GEMM path (engine.hip proj(): `gemm_f16x2`) and the NAR attention (`attn_full_h2`) at L ~ 983 -- the kernels that
dominate the bench -- and `dec_attn` walks contexts 260 .. 1090.  The short goldens never reach any of that.

Bars: token ids bit-exact (all 8 codebooks, all 600 frames); AR logits (every 50th step, teacher-forced) within 2e-5 abs of
the reference's; NAR logits of all 7 stages (first 16 generated rows) within 6e-4 abs (logits std ~25, K = 1024).  Measured on
MI355X (profiles/r03_logit_error.json, all three arithmetic modes alike): AR <= 5.6e-6, NAR <= 1.8e-4; the reference's own smallest
decision margins in these goldens are 3.8e-5 (AR top-2 logit gap) and 5e-4 (NAR), so the bars sit BELOW the margins.
"""
import numpy as np
import pytest

from oracle import synth
from oracle.make_golden import FULL_CASES, FULL_LOGIT_EVERY, case_inputs
from tests._util import get_model, golden

pytestmark = pytest.mark.gpu

AR_TOL, NAR_TOL = 2e-5, 6e-4

GREEDY = [n for n in FULL_CASES if FULL_CASES[n]["top_k"] == 1]
TOPK = [n for n in FULL_CASES if FULL_CASES[n]["top_k"] == 10]


def _model(debug_taps=False, max_batch=8):
    c = FULL_CASES[GREEDY[0]]
    return get_model(12, c["seed"], c["eos_gain"], debug_taps=debug_taps, max_new=608, max_prompt=400, max_text=256,
                     max_batch=max_batch)


def _row(name):
    c = FULL_CASES[name]
    a, t, text, pl, langs = case_inputs(c)
    us = None if c["useed"] is None else synth.uniforms(4096, 1, c["useed"])[:, 0]
    return c, dict(text=text[0], prompt=a[0], enroll=t.shape[-1], prompt_language=pl, text_language=langs), us


def _first_diff(out, gold):
    d = np.argwhere(out != gold)
    return None if len(d) == 0 else tuple(int(v) for v in d[0])


def _assert_codes(name, out, g):
    gold = g["codes"][0]
    assert out.shape == gold.shape, (name, out.shape, gold.shape)
    fd = _first_diff(out, gold)
    if fd is not None:
        t, q = fd
        marg = float(g["ar_margin"][t]) if q == 0 else float(g["nar_margin"][q - 1])
        raise AssertionError(f"{name}: first differing id at frame {t}, codebook {q}: got {out[t, q]}, reference {gold[t, q]}; "
                             f"reference decision margin there {marg:.3e}; {int((out != gold).sum())} ids differ")


@pytest.mark.parametrize("name", sorted(FULL_CASES))
def test_full_length_row_alone_matches_reference(name):
    """one utterance per call = exactly one reference VALLE.inference call (models/vallex.py:458-686), 600 frames."""
    c, row, us = _row(name)
    m = _model()
    out = m.inference_batch([row], top_k=c["top_k"], uniforms=None if us is None else us[:, None],
                            force_eos_at=c["force_eos_at"])[0]
    _assert_codes(name, out, golden(name))


@pytest.mark.parametrize("names", [GREEDY, TOPK], ids=["greedy_x3", "topk10_x3"])
def test_full_length_rows_batched_match_reference(names):
    """three full-length rows (en / zh / ja presets) in ONE inference_batch call: ~2900 packed rows -> DMA bf16x3 GEMM,
    attn_full_h2 at L ~ 860-1090, ragged dec_attn contexts; each row must still equal its own reference run."""
    m = _model()
    rows, cols = [], []
    for n in names:
        c, row, us = _row(n)
        rows.append(row)
        cols.append(us)
    c0 = FULL_CASES[names[0]]
    us = None if cols[0] is None else np.stack(cols, axis=1)
    outs = m.inference_batch(rows, top_k=c0["top_k"], uniforms=us, force_eos_at=c0["force_eos_at"])
    for n, o in zip(names, outs):
        _assert_codes(n, o, golden(n))


@pytest.mark.parametrize("name", ["nl12_full_en_greedy", "nl12_full_ja_topk10"])
def test_full_length_teacher_forced_logits(name):
    """feed the reference's own 600 tokens through the cached decode step: logits at every 50th step within AR_TOL of the
    reference's, and for the greedy case the arg-max reproduces the reference token at EVERY step."""
    c, row, _ = _row(name)
    g = golden(name)
    m = _model(debug_taps=True, max_batch=2)
    eng = m.engine
    eng.ar_prefill(m.make_batch([row]))
    codes0 = g["codes"][0, :, 0]
    worst = 0.0
    for t in range(len(codes0)):
        lg = eng.ar_logits()[0]
        if t % FULL_LOGIT_EVERY == 0:
            ref = g["ar_logits"][t // FULL_LOGIT_EVERY]
            worst = max(worst, float(np.abs(lg - ref).max()))
            np.testing.assert_allclose(lg, ref, atol=AR_TOL, rtol=0, err_msg=f"step {t}")
        if c["top_k"] == 1:
            assert int(np.argmax(lg)) == int(codes0[t]), f"step {t} (reference margin {g['ar_margin'][t]:.3e})"
        eng.ar_step(np.array([codes0[t]], np.int32))
    print(f"{name}: max |logit - reference| over the sampled steps = {worst:.2e}")


@pytest.mark.parametrize("name", ["nl12_full_en_greedy", "nl12_full_zh_topk10"])
def test_full_length_nar_logits_all_stages(name):
    """NAR stages on the reference's first codebook: all 7 stages' logits (first 16 generated rows) within NAR_TOL of the
    reference's, and codebooks 2..8 bit-exact, at Ltot ~ 983 (one row -> register-staged GEMM) ..."""
    c, row, _ = _row(name)
    g = golden(name)
    m = _model(debug_taps=True, max_batch=2)
    codes = m.engine.nar(m.make_batch([row]), [g["codes"][0, :, 0].astype(np.int32)])[0]
    T = g["codes"].shape[1]
    for st in range(7):
        lg = m.engine.read_tap(f"nar_logits{st}", T * 1024).reshape(T, 1024)
        np.testing.assert_allclose(lg[:16], g["nar_logits"][st], atol=NAR_TOL, rtol=0, err_msg=f"stage {st}")
    np.testing.assert_array_equal(codes, g["codes"][0])


def test_full_length_nar_logits_batched_dma_gemm():
    """... and with two rows in the call (packed rows >= 1024 -> gemm_bf16x3_dma_kernel): stage logits of row 0 against the
    reference (the tap holds the generated rows of all sequences back to back)."""
    names = ["nl12_full_en_greedy", "nl12_full_ja_greedy"]
    m = _model(debug_taps=True, max_batch=2)
    rows = [_row(n)[1] for n in names]
    gs = [golden(n) for n in names]
    codes = m.engine.nar(m.make_batch(rows), [g["codes"][0, :, 0].astype(np.int32) for g in gs])
    T = [g["codes"].shape[1] for g in gs]
    for st in range(7):
        lg = m.engine.read_tap(f"nar_logits{st}", sum(T) * 1024).reshape(sum(T), 1024)
        np.testing.assert_allclose(lg[:16], gs[0]["nar_logits"][st], atol=NAR_TOL, rtol=0, err_msg=f"stage {st} row 0")
        np.testing.assert_allclose(lg[T[0]:T[0] + 16], gs[1]["nar_logits"][st], atol=NAR_TOL, rtol=0, err_msg=f"stage {st} row 1")
    for cd, g in zip(codes, gs):
        np.testing.assert_array_equal(cd, g["codes"][0])


def test_bench_row0_matches_live_reference_golden():
    """round 6: row 0 of bench.py's workload (synthetic prompt, Tp = 221, S = 167, top-k 10 with the uniforms bench.py injects,
    600 frames) against the ids the LIVE reference's own VALLE.inference produced for it (tests/golden/bench_row0.npz, written by
    tools/cpu_reference.py) -- the fixture behind `parity.*.ids_equal_reference_golden` of the bench line"""
    import os
    import bench
    from oracle.make_golden import GOLD
    g = np.load(os.path.join(GOLD, "bench_row0.npz"))
    frames = int(g["frames"])
    r = bench.make_rows(0, 1)[0]
    us = np.random.default_rng(1234).random(frames + 1).astype(np.float32)
    m = _model()
    out = m.inference_batch([r], top_k=10, uniforms=us[:, None], force_eos_at=frames)[0]
    gold = g["codes"].astype(np.int64)
    assert out.shape == gold.shape and _first_diff(out, gold) is None, (out.shape, _first_diff(out, gold), float(g["ar_margin"].min()))
