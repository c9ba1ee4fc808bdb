"""CPU: numerics of the f16x2 GEMM experiment (vall-e-x_amd/csrc/gemm_f16x2_dma.hip, VX_GEMM_H2=1): x = h + t/2048 with
h = fp16(x), t = fp16((x - h) * 2048); a.b ~= h.h + (h.t + t.h)/2048.
  * its representation error sits well inside the accumulation noise of an ordinary fp32 matmul;
  * with that error injected into every multi-row projection of the oracle (prefill + NAR, where the engine would use the
    kernel), the greedy / sampled token ids of the live-reference goldens do not change.
This is evidence for running the experiment on hardware, not a parity claim for the HIP kernel."""
import os

import numpy as np
import pytest
import torch

from oracle import synth
from oracle import vallex_oracle as VO
from oracle.make_golden import CASES, GOLD, case_inputs


def _split_np(x):
    h = x.astype(np.float16)
    t = ((x - h.astype(np.float32)) * 2048).astype(np.float16)
    return h.astype(np.float64), t.astype(np.float64)


def test_representation_error_is_below_fp32_accumulation_noise():
    rng = np.random.default_rng(0)
    K = 1024
    A = rng.standard_normal((128, K)).astype(np.float32)                     # LayerNorm-like activations
    W = (rng.uniform(-1, 1, (128, K)) / np.sqrt(K)).astype(np.float32)       # weights ~ 1/sqrt(K)
    exact = A.astype(np.float64) @ W.astype(np.float64).T
    ah, at = _split_np(A)
    wh, wt = _split_np(W)
    h2 = ah @ wh.T + (ah @ wt.T + at @ wh.T) / 2048
    f32 = (A @ W.T).astype(np.float64)
    err_h2, err_f32 = np.abs(h2 - exact), np.abs(f32 - exact)
    assert err_h2.max() < 5e-7 and err_h2.mean() < 1e-7
    assert err_h2.mean() < 0.5 * err_f32.mean()                              # inside the noise an fp32 matmul already has


@pytest.mark.parametrize("name", ["nl2_greedy_eos", "nl2_topk10"])
def test_token_ids_survive_f16x2_projections(name):
    orig = VO.F.linear

    def split(x):
        h = x.to(torch.float16)
        return h.double(), ((x - h.float()) * 2048).to(torch.float16).double()

    def linear_h2(inp, w, b=None):
        if inp.dim() >= 2 and inp.shape[0] >= 2 and w.shape[0] >= 256:      # multi-row projections, not the decode steps
            xh, xt = split(inp.float())
            wh, wt = split(w.float())
            y = (xh @ wh.T + (xh @ wt.T + xt @ wh.T) / 2048).float()
            return y + b if b is not None else y
        return orig(inp, w, b)

    c = CASES[name]
    orc = VO.VallexOracle(synth.vallex_state_dict(c["num_layers"], c["seed"], c["eos_gain"]), c["num_layers"])
    a, t, text, pl, langs = case_inputs(c)
    us = None if c["useed"] is None else synth.uniforms(4096, 1, c["useed"])[:, 0]
    VO.F.linear = linear_h2
    try:
        out = orc.inference(text, np.array([text.shape[-1]]), a, t.shape[-1], top_k=c["top_k"],
                            temperature=c.get("temperature", 1.0), prompt_language=pl, text_language=langs, uniforms=us,
                            force_eos_at=c["force_eos_at"])
    finally:
        VO.F.linear = orig
    np.testing.assert_array_equal(out, np.load(os.path.join(GOLD, name + ".npz"))["codes"])
