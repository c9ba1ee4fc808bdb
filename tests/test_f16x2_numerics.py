"""CPU: numerics of the f16x2 arithmetic of the full-sequence projections (vall-e-x_amd/csrc/gemm_f16x2.hip, the default since
round 2): X = x * 2^s, h = fp16(X), t = fp16(X - h) (activations s = 5, every weight tensor its own s from max |w|);
A.B ~= h.h + h.t + t.h in one fp32 accumulator, descaled by 2^-(sa + sw).
  * its representation error sits well inside the accumulation noise of an ordinary fp32 matmul;
  * with that error injected into every multi-row projection and into both contractions of the full-sequence attention of the
    oracle (prefill + NAR), the greedy / sampled token ids of the live-reference goldens do not change (all eight short goldens
    were checked once; two stay in the suite).
The parity claim for the HIP kernel itself is made by the GPU suite (every golden bit-exact through the C ABI); this file is the
model-level error analysis behind the choice."""
import os

import numpy as np
import pytest
import torch

from oracle import synth
from oracle import vallex_oracle as VO
from oracle.make_golden import CASES, GOLD, case_inputs


ACT_SHIFT = 5                                                                # vx_common.h: H2_ACT_SHIFT


def _w_shift(w):
    """engine.hip split_w: max |w| * 2^shift in [16384, 32768)"""
    mx = float(np.abs(np.asarray(w)).max())
    return int(np.clip(15 - np.frexp(mx)[1], 0, 24)) if mx > 0 else 24


def _split_np(x, shift):
    X = (x * np.float32(2.0 ** shift)).astype(np.float32)
    h = X.astype(np.float16)
    t = (X - h.astype(np.float32)).astype(np.float16)
    return h.astype(np.float64), t.astype(np.float64)


def test_representation_error_is_below_fp32_accumulation_noise():
    rng = np.random.default_rng(0)
    K = 1024
    A = rng.standard_normal((128, K)).astype(np.float32)                     # LayerNorm-like activations
    W = (rng.uniform(-1, 1, (128, K)) / np.sqrt(K)).astype(np.float32)       # weights ~ 1/sqrt(K)
    exact = A.astype(np.float64) @ W.astype(np.float64).T
    sw = _w_shift(W)
    ah, at = _split_np(A, ACT_SHIFT)
    wh, wt = _split_np(W, sw)
    assert np.abs(ah).max() < 65504 and 16384 <= np.abs(wh).max() <= 32768
    h2 = (ah @ wh.T + ah @ wt.T + at @ wh.T) * 2.0 ** -(ACT_SHIFT + sw)
    f32 = (A @ W.T).astype(np.float64)
    err_h2, err_f32 = np.abs(h2 - exact), np.abs(f32 - exact)
    assert err_h2.max() < 5e-7 and err_h2.mean() < 1e-7
    assert err_h2.mean() < 0.5 * err_f32.mean()                              # inside the noise an fp32 matmul already has


def _mha_h2(self, x, prefix, mask, past=None):
    """VallexOracle._mha with both contractions of FULL-SEQUENCE attention (T >= 2: prefill / NAR) on the f16x2 scheme of
    attn_full_h2.hip: Q/8 and K split at 2^5, P at 2^14 (relative to the row maximum), V at 2^5, tails at the heads' scale."""
    import math
    import torch.nn.functional as F

    def split(t, shift):
        X = t.float() * float(2.0 ** shift)
        h = X.to(torch.float16)
        assert torch.isfinite(h.float()).all()
        return h.double(), (X - h.float()).to(torch.float16).double()

    def mm(a, sa, b, sb):                                                # a @ b with f16x2 operands at 2^sa / 2^sb
        ah, at = split(a, sa)
        bh, bt = split(b, sb)
        return ((ah @ bh + ah @ bt + at @ bh) * 2.0 ** -(sa + sb)).float()

    T = x.shape[0]
    qkv = F.linear(x, self.w[prefix + ".in_proj_weight"], self.w[prefix + ".in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    hd = self.d // self.h
    q = q.view(T, self.h, hd).transpose(0, 1)
    k = k.view(T, self.h, hd).transpose(0, 1)
    v = v.view(T, self.h, hd).transpose(0, 1)
    if past is not None:
        k = torch.cat((past[0], k), dim=-2)
        v = torch.cat((past[1], v), dim=-2)
    full = T >= 2
    att = (mm(q * (1.0 / math.sqrt(hd)), 5, k.transpose(-2, -1), 5) if full else (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hd)))
    if mask is not None:
        att = att.masked_fill(mask, float("-inf"))
    if full:
        p = torch.exp(att - att.max(dim=-1, keepdim=True).values)       # unnormalised, <= 1: what the kernel splits (x 2^14)
        y = mm(p, 14, v, 5) / p.sum(dim=-1, keepdim=True)
    else:
        y = F.softmax(att, dim=-1) @ v
    y = y.transpose(0, 1).contiguous().view(T, self.d)
    y = F.linear(y, self.w[prefix + ".out_proj.weight"], self.w[prefix + ".out_proj.bias"])
    return y, (k, v)


@pytest.mark.parametrize("name", ["nl2_greedy_eos", "nl2_topk10"])
def test_token_ids_survive_f16x2_projections(name, monkeypatch):
    monkeypatch.setattr(VO.VallexOracle, "_mha", _mha_h2)                    # attention on f16x2 as well
    orig = VO.F.linear

    def split(x, shift):
        X = x * float(2.0 ** shift)
        h = X.to(torch.float16)
        return h.double(), (X - h.float()).to(torch.float16).double()

    def linear_h2(inp, w, b=None):
        if inp.dim() >= 2 and inp.shape[0] >= 2 and w.shape[0] >= 256:      # multi-row projections, not the decode steps
            sw = _w_shift(w.numpy())
            xh, xt = split(inp.float(), ACT_SHIFT)
            wh, wt = split(w.float(), sw)
            y = ((xh @ wh.T + xh @ wt.T + xt @ wh.T) * 2.0 ** -(ACT_SHIFT + sw)).float()
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


def test_operand_headroom_of_the_f16x2_range():
    """How far the operands of the f16x2 kernels are from the end of their fp16 range (|x| < 2047 at the activation scale 2^5),
    measured on the oracle over a 12-layer AR prefill + one full NAR stage at the BASELINE shape (983 rows): default-init weights,
    the trained-like weights (LayerNorm gains to 4, massive FFN channels, heavy tails) and, as the counter-example the engine's
    fp32 fallback exists for, weights rescaled out of range.  The real checkpoint is not available offline; this is the margin the
    two synthetic regimes leave."""
    from oracle.make_golden import FULL_CASES, TRAINED_CASES, case_state_dict
    out = {}
    for label, c in (("default-init", FULL_CASES["nl12_full_en_greedy"]), ("trained-like", TRAINED_CASES["nl12_trained_en_greedy"])):
        orc = VO.VallexOracle(case_state_dict(c), 12)
        orc.stats = {}
        a, t, text, pl, langs = case_inputs(c)
        with torch.no_grad():
            orc.ar_prefill(torch.from_numpy(text[0].astype(np.int64)), torch.from_numpy(a[0, :, 0].astype(np.int64)), t.shape[-1], pl, langs)
            orc._nar_stack(torch.randn(983, 1024) * 2.0, orc.w["nar_stage_embeddings.0.word_embeddings.weight"])
        out[label] = dict(orc.stats)
        assert max(orc.stats.values()) < 2047.0, (label, orc.stats)
    print("max |operand| seen by the f16x2 kernels (range ends at 2047):", {k: {n: round(v, 1) for n, v in d.items()} for k, d in out.items()})
    assert out["trained-like"]["ffn"] > 4 * out["default-init"]["ffn"]            # the massive channels are there ...
    assert max(out["trained-like"].values()) < 2047.0 / 2                          # ... with more than 2x headroom left
    # counter-example: the 2^12 rescaling of 64 FFN channels (same fp32 function) leaves the range -> the engine re-runs in fp32
    c = dict(CASES["nl2_topk10"], range_kind="ffn")
    orc = VO.VallexOracle(case_state_dict(c), 2)
    orc.stats = {}
    a, t, text, pl, langs = case_inputs(c)
    with torch.no_grad():
        orc.ar_prefill(torch.from_numpy(text[0].astype(np.int64)), torch.from_numpy(a[0, :, 0].astype(np.int64)), t.shape[-1], pl, langs)
    assert orc.stats["ffn"] > 2047.0
