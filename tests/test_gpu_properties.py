"""GPU, full 12-layer model at the BASELINE batch shape: size-independent properties that must hold where the CPU oracle
is too slow to run (32 rows x hundreds of frames):
  * run-to-run determinism (no atomics, fixed reduction orders): two identical calls give identical ids and audio;
  * batching invariance: row i of a ragged 32-row batch == the same row run alone (the reference can only run rows alone);
  * KV-cache consistency: the logits after k cached decode steps == the logits of a fresh prefill whose prompt already
    contains those k frames (the reference's cached and full-recompute paths agree to 7e-7);
  * stop rule: forced EOS / the 16*S cap give exactly the predicted lengths."""
import numpy as np
import pytest

from oracle import synth
from tests._util import get_model

pytestmark = pytest.mark.gpu


def _rows(n, seed0=900):
    rows = []
    rng = np.random.default_rng(seed0)
    for i in range(n):
        tp, sp = int(rng.integers(30, 90)), int(rng.integers(5, 20))
        a, t = synth.synth_prompt(tp, sp, seed=seed0 + i)
        text = np.concatenate([t[0], synth.synth_text(int(rng.integers(20, 40)), seed0 + i)])
        lang = ("en", "zh", "ja")[i % 3]
        rows.append(dict(text=text, prompt=a[0], enroll=sp, prompt_language=lang, text_language=lang))
    return rows


def test_full_model_batch_properties():
    m = get_model(12, 0, 0.0, vocos=True, max_new=128, max_prompt=128, max_text=64, max_batch=32)   # eos_gain 0: EOS never in the top-10
    rows = _rows(32)
    us = synth.uniforms(128, 32, 31)
    out1 = m.inference_batch(rows, top_k=10, uniforms=us, force_eos_at=48)
    out2 = m.inference_batch(rows, top_k=10, uniforms=us, force_eos_at=48)
    assert all(o.shape == (48, 8) for o in out1)
    for a, b in zip(out1, out2):                                   # determinism, all 8 codebooks
        np.testing.assert_array_equal(a, b)
    for i in (0, 7, 31):                                           # batching invariance (row alone, its own uniforms column)
        alone = m.inference_batch([rows[i]], top_k=10, uniforms=us[:, i:i + 1], force_eos_at=48)[0]
        np.testing.assert_array_equal(alone, out1[i])
    w1 = m.engine.vocos_decode(out1, 2)
    w2 = m.engine.vocos_decode(out2, 2)
    for a, b in zip(w1, w2):
        assert a.shape == (48 * 320,) and np.isfinite(a).all()
        np.testing.assert_array_equal(a, b)


def test_odd_batch_with_rows_ending_at_different_steps():
    """The fused decode attention runs two launch slots per workgroup: 19 rows leave the last workgroup with ONE row, and rows
    that stop at different steps (the 16*S cap with S = 3..8, models/vallex.py:577) leave workgroups with one or two finished
    halves while the others go on.  Every row must equal the same row run alone (context-split path), ids and length."""
    m = get_model(2, 1, 0.0, max_new=160, max_prompt=96, max_text=32, max_batch=32)      # eos_gain 0: only the cap stops a row
    rng = np.random.default_rng(5)
    rows = []
    for i in range(19):
        tp, sp, st = int(rng.integers(20, 80)), int(rng.integers(1, 3)), int(rng.integers(2, 7))
        a, t = synth.synth_prompt(tp, sp, seed=300 + i)
        rows.append(dict(text=np.concatenate([t[0], synth.synth_text(st, 300 + i)]), prompt=a[0], enroll=sp,
                         prompt_language="en", text_language=("en", "zh", "ja")[i % 3]))
    us = synth.uniforms(160, 19, 77)
    out = m.inference_batch(rows, top_k=10, uniforms=us, sync_every=4)
    lens = [o.shape[0] for o in out]
    assert lens == [16 * len(r["text"]) for r in rows] and len(set(lens)) > 3, lens
    for i in (0, 5, 9, 18):                                         # 18: the row that has a workgroup to itself
        alone = m.inference_batch([rows[i]], top_k=10, uniforms=us[:, i:i + 1])[0]
        np.testing.assert_array_equal(alone, out[i])


def test_contexts_sharing_the_gpu_on_cu_partitions():
    """vx_config.cu_mask: two contexts confined to disjoint halves of the CUs, driven from two host threads at the same time
    (bench.py --contexts 2), give exactly the ids of an unconfined context."""
    import threading
    from vallex_amd._capi import cu_partition
    kw = dict(max_new=64, max_prompt=96, max_text=64, max_batch=32)
    rows = _rows(32, seed0=1200)
    us = synth.uniforms(64, 32, 5)
    want = get_model(2, 1, 0.0, **kw).inference_batch(rows, top_k=10, uniforms=us, force_eos_at=40)
    masks = cu_partition(2)
    assert masks[0] & masks[1] == 0 and bin(masks[0] | masks[1]).count("1") == 256
    halves = [get_model(2, 1, 0.0, cu_mask=mk, **kw) for mk in masks]
    got = [None, None]

    def run(i):
        for _ in range(3):
            got[i] = halves[i].inference_batch(rows, top_k=10, uniforms=us, force_eos_at=40)

    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for g in got:
        for a, b in zip(want, g):
            np.testing.assert_array_equal(a, b)


def test_kv_cache_matches_fresh_prefill_full_model():
    m = get_model(12, 0, 0.0, vocos=True, max_new=128, max_prompt=128, max_text=64, max_batch=32)   # eos_gain 0: EOS never in the top-10
    eng = m.engine
    r = _rows(1, seed0=777)[0]
    k = 12
    forced = np.random.default_rng(1).integers(0, 1024, size=k)
    eng.ar_prefill(m.make_batch([r]))
    for t in range(k):
        eng.ar_step(np.array([forced[t]], np.int32))
    cached = eng.ar_logits()[0]
    # same state reached by a prefill whose audio prompt already holds the k frames (codebooks 1..7 are irrelevant to AR)
    ext = np.concatenate([r["prompt"], np.repeat(forced[:, None], 8, axis=1)], axis=0)
    eng.ar_prefill(m.make_batch([dict(r, prompt=ext)]))
    fresh = eng.ar_logits()[0]
    np.testing.assert_allclose(cached, fresh, atol=2e-4, rtol=0)   # two different kernels sets (decode vs full-sequence)
    assert int(np.argmax(cached)) == int(np.argmax(fresh))


def test_kv_cache_matches_fresh_prefill_at_long_context():
    """BASELINE config 5 regime (long-text mode: the previous chunk's ~560 frames are the prompt): after 500 cached decode steps on
    top of S = 200, Tp = 600 the context is ~1300; the logits must equal those of a fresh prefill whose prompt already holds the
    500 frames (two different kernel sets: decode vs full-sequence), and dec_attn walked contexts 800 .. 1300 on the way."""
    m = get_model(12, 0, 0.0, max_new=512, max_prompt=1152, max_text=256, max_batch=2)
    eng = m.engine
    a, t = synth.synth_prompt(600, 100, seed=4321)
    text = np.concatenate([t[0], synth.synth_text(100, 4321)])
    r = dict(text=text, prompt=a[0], enroll=100, prompt_language="en", text_language="en")
    k = 500
    forced = np.random.default_rng(7).integers(0, 1024, size=k)
    eng.ar_prefill(m.make_batch([r]))
    for i in range(k):
        eng.ar_step(np.array([forced[i]], np.int32))
    cached = eng.ar_logits()[0]
    ext = np.concatenate([r["prompt"], np.repeat(forced[:, None], 8, axis=1)], axis=0)         # Tp = 1100
    eng.ar_prefill(m.make_batch([dict(r, prompt=ext)]))
    fresh = eng.ar_logits()[0]
    np.testing.assert_allclose(cached, fresh, atol=3e-4, rtol=0)
    assert int(np.argmax(cached)) == int(np.argmax(fresh))


def test_stop_rules_lengths():
    m = get_model(2, 1, 1.0, max_new=320, max_prompt=400, max_text=256)
    a, t = synth.synth_prompt(10, 3, seed=2)
    text = np.concatenate([t[0], synth.synth_text(4, 2)])         # S = 7 -> cap 16*S = 112 frames (models/vallex.py:577)
    row = dict(text=text, prompt=a[0], enroll=3, prompt_language="en", text_language="en")
    assert m.inference_batch([row], top_k=1)[0].shape[0] == 16 * 7
    assert m.inference_batch([row], top_k=1, force_eos_at=17)[0].shape[0] == 17
    # the arena (max_new) cutting a row before the reference's rule would is reported, not silent
    small = get_model(2, 1, 1.0, max_new=64, max_prompt=400, max_text=256)
    with pytest.warns(RuntimeWarning, match="max_new"):
        assert small.inference_batch([row], top_k=1)[0].shape[0] == 64


@pytest.mark.gpu
def test_attention_kernels_agree():
    """f16x2 attention (product path, variant 20) and bf16x3 attention (VX_ATTN_X3=1, variant 10) against the exact-fp32 MFMA
    kernel on the same scratch data, ragged tail and prefix-LM mask included: fp32-class agreement (all accumulate in fp32; the
    splits are exact to 2^-22 / 2^-27).  A NaN here is how an fp16 head overflow would show (P is scaled by 2^14 relative to the
    running maximum, which therefore has to be the true maximum of BOTH lanes that share a query)."""
    import vallex_amd
    eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
    for batch, length in ((8, 77), (32, 988)):
        for causal in (False, True):
            for variant in (20, 10):
                _, diff = eng.bench_attn(batch, length, causal, variant, 1)
                assert 0.0 <= diff < 5e-6, (batch, length, causal, variant, diff)


@pytest.mark.gpu
def test_split_operand_gemm_kernels_agree_with_fp32():
    """The full-sequence GEMM kernels against the exact-fp32 MFMA kernel on the same scratch data (uniform [-1, 1), first and last
    256 rows, M = 31616 = the NAR row count of the bench): the two bf16x3 kernels compute the same sums in the same order, and
    f16x2 (the default) is as close to the fp32 kernel as bf16x3 is -- all differences are fp32-reassociation sized."""
    import vallex_amd
    eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
    for (N, K, tol) in ((1024, 1024, 5e-4), (1024, 4096, 2e-3)):
        d_x3, d_dma, d_h2 = (eng.bench_gemm(31616, N, K, k, 1)[1] for k in (1, 2, 6))
        assert abs(d_x3 - d_dma) <= 1e-6, (d_x3, d_dma)
        assert d_x3 < tol and d_h2 < tol, (N, K, d_x3, d_h2)
        assert d_h2 <= 1.5 * d_x3, (N, K, d_x3, d_h2)            # f16x2 is not further from fp32 than bf16x3 is
    # short and ragged row counts go through the same 256-row-tile kernel (edge rows clamped, never stored)
    for M in (1, 77, 255, 257, 983):
        assert eng.bench_gemm(M, 3072, 1024, 6, 1)[1] < 5e-4, M


@pytest.mark.gpu
def test_fp32_dma_gemm_kernels_equal_the_register_staged_kernel_bit_for_bit():
    """round 5: the LDS-DMA fp32 GEMM kernels (256 x 128, 128 x 128 and 256 x 256 tiles; the reference-arithmetic mode's projections) issue, per
    output element, the same v_mfma_f32_32x32x2_f32 sequence on the same operands as the register-staged kernel -- the harness compares
    the first and last 256 rows against that kernel: the difference must be EXACTLY zero, on the NAR shapes, on ragged row counts (edge
    rows clamped, never stored), on an N that does not fill its last column tile and on a single K tile; the product's own choice too"""
    import vallex_amd
    eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
    for (M, N, K) in ((31616, 3072, 1024), (12288, 1024, 4096), (2049, 1024, 1024), (983, 4096, 1024), (300, 1100, 32), (1, 128, 64)):
        for kernel in (4, 5, 14, 0):                          # LDS-DMA 256 x 128 / 128 x 128 / 256 x 256 forced, the product's choice
            us, diff = eng.bench_gemm(M, N, K, kernel, 1)
            assert diff == 0.0, (M, N, K, kernel, diff)


@pytest.mark.gpu
def test_f16x2_four_wave_gemm_equals_the_eight_wave_kernel():
    """round 5: the 4-wave (128 x 128 per wave) f16x2 GEMM accumulates every output element in the same order as the 8-wave kernel --
    measured against the SAME fp32 reference the harness reports the same maximum difference, to the last bit, on the NAR shapes and on
    ragged row counts (the product picks it for long row sets; kernel 15 forces it, 8 forces the 8-wave kernel)"""
    import vallex_amd
    eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
    for (M, N, K) in ((31616, 3072, 1024), (31616, 1024, 4096), (2049, 1024, 1024), (300, 256, 64)):
        d4, d8 = eng.bench_gemm(M, N, K, 15, 1)[1], eng.bench_gemm(M, N, K, 8, 1)[1]
        assert d4 == d8 and d4 < 2e-3, (M, N, K, d4, d8)


@pytest.mark.gpu
def test_f16x2_four_wave_gemm_epilogue_branches_equal_the_eight_wave_kernel_word_for_word():
    """round 6 (ADVICE): the four-wave kernel's rewritten epilogue -- bias staged through LDS, residual vectors prefetched one block
    ahead, resid_rows clamping, the out_planes split with its lane exchange -- against the eight-wave kernel on the same operand planes,
    reached DIRECTLY (vx_bench_gemm_epilogue), not through whatever tile the cost model picks at a model test's row count:
    mode 0 bias + ReLU + out_planes (linear1), 1 bias + residual through a scattered resid_rows map (trimmed out_proj), 2 bias + residual
    in place (out_proj / linear2); full and ragged row counts, K of two tiles and of 128 tiles.  Every word must be identical."""
    import vallex_amd
    eng = vallex_amd.Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
    for (M, N, K) in ((4096, 1024, 1024), (2049, 1024, 4096), (983, 4096, 1024), (257, 256, 64), (19200, 1024, 1024)):
        for mode in (0, 1, 2, 10, 11, 12):                     # + 10: the eight-wave template on 128 x 128 tiles (prefill, trimmed layers)
            bad, tot = eng.bench_gemm_epilogue(M, N, K, mode)
            assert bad == 0 and tot >= M * N, (M, N, K, mode, bad, tot)
