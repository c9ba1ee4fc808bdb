"""Shared helpers for the parity tests (test infrastructure; the only place besides bench/smoke that touches oracle/)."""
import os

import numpy as np

import vallex_amd  # noqa: F401  (registers the package under an importable name)
from oracle import synth
from oracle.make_golden import CASES, GOLD, case_inputs
from vallex_amd.models.vallex import VALLE

_MODELS = {}


def have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def get_model(num_layers, seed, eos_gain=1.0, vocos=False, debug_taps=False, max_new=320, max_prompt=400, max_text=256,
              max_batch=32, use_graph=True, attn_gain=1.0, cu_mask=0):
    key = (num_layers, seed, eos_gain, vocos, debug_taps, max_new, max_prompt, max_text, max_batch, use_graph, attn_gain, cu_mask)
    if key not in _MODELS:
        if len(_MODELS) >= 3:                      # keep device memory bounded across the test session
            _MODELS.pop(next(iter(_MODELS))).__dict__.pop("_engine", None)
        m = VALLE(1024, 16, num_layers, norm_first=True, add_prenet=False, prefix_mode=1, share_embedding=True,
                  nar_scale_factor=1.0, prepend_bos=True, num_quantizers=8, engine_max_new=max_new,
                  engine_max_prompt=max_prompt, engine_max_text=max_text, engine_max_batch=max_batch,
                  engine_debug_taps=debug_taps, engine_use_graph=use_graph, engine_cu_mask=cu_mask)
        m.to("cuda:0").load_state_dict(synth.vallex_state_dict(num_layers, seed, eos_gain, attn_gain), strict=True)
        if vocos:
            m.load_vocos_state_dict(synth.vocos_state_dict(2))
        _MODELS[key] = m
    return _MODELS[key]


_SDS = {}


def case_model(c, arith="default", **kw):
    """engine for the weights of a make_golden case dict (trained-like / out-of-range variants included); `arith` selects the
    arithmetic of the full-sequence path (vx_config.arith).  The state-dict is cached per weight recipe: the 12-layer
    trained-like one takes ~20 s of CPU to build and is shared by the three arithmetic modes."""
    from oracle.make_golden import case_state_dict
    wkey = (c["num_layers"], c["seed"], c.get("eos_gain", 1.0), c.get("attn_gain", 1.0), bool(c.get("trained")), c.get("range_kind"), c.get("outlier"))
    opts = dict(vocos=False, debug_taps=False, max_new=320, max_prompt=400, max_text=256, max_batch=32, use_graph=True)
    opts.update(kw)
    key = ("case", wkey, arith) + tuple(sorted(opts.items()))
    if key not in _MODELS:
        if len(_MODELS) >= 3:
            _MODELS.pop(next(iter(_MODELS))).__dict__.pop("_engine", None)
        if wkey not in _SDS:
            if len(_SDS) >= 2:
                _SDS.pop(next(iter(_SDS)))
            _SDS[wkey] = case_state_dict(c)
        m = VALLE(1024, 16, c["num_layers"], norm_first=True, add_prenet=False, prefix_mode=1, share_embedding=True,
                  nar_scale_factor=1.0, prepend_bos=True, num_quantizers=8, engine_max_new=opts["max_new"],
                  engine_max_prompt=opts["max_prompt"], engine_max_text=opts["max_text"], engine_max_batch=opts["max_batch"],
                  engine_debug_taps=opts["debug_taps"], engine_use_graph=opts["use_graph"], engine_arith=arith)
        m.to("cuda:0").load_state_dict(_SDS[wkey], strict=True)
        _MODELS[key] = m
    return _MODELS[key]


def inputs_row(c, inputs=None):
    """(row dict, uniforms column or None) of a make_golden case (or of explicit (a, t, text, pl, langs) inputs)"""
    a, t, text, pl, langs = inputs if inputs is not None else case_inputs(c)
    row = dict(text=text[0], prompt=a[0], enroll=t.shape[-1], prompt_language=pl, text_language=langs)
    us = None if c["useed"] is None else synth.uniforms(4096, 1, c["useed"])[:, 0]
    return row, us


def assert_codes(name, out, g, key="codes"):
    gold = g[key][0]
    assert out.shape == gold.shape, (name, out.shape, gold.shape)
    d = np.argwhere(out != gold)
    if len(d):
        t, q = (int(v) for v in d[0])
        raise AssertionError(f"{name}: first differing id at frame {t}, codebook {q}: got {out[t, q]}, reference {gold[t, q]}; "
                             f"{len(d)} ids differ")


def teacher_forced_logit_error(m, row, g, every, codes_key="codes"):
    """feed the reference's own first-codebook ids through the cached decode step; returns (max |logit - reference| over the
    stored steps, number of steps whose arg-max differs from the reference's id -- only meaningful for greedy goldens)"""
    eng = m.engine
    eng.ar_prefill(m.make_batch([row]))
    codes0 = g[codes_key][0, :, 0]
    worst, flips = 0.0, 0
    for t in range(len(codes0)):
        lg = eng.ar_logits()[0]
        if t % every == 0 and t // every < g["ar_logits"].shape[0]:
            worst = max(worst, float(np.abs(lg - g["ar_logits"][t // every]).max()))
        flips += int(np.argmax(lg)) != int(codes0[t])
        eng.ar_step(np.array([codes0[t]], np.int32))
    return worst, flips


def nar_logit_error(m, row, g, codes_key="codes"):
    """NAR stages on the reference's first codebook (needs debug_taps): (codes (T, 8), max |logit - reference| per stage over the
    first 16 generated rows)"""
    codes = m.engine.nar(m.make_batch([row]), [g[codes_key][0, :, 0].astype(np.int32)])[0]
    T = g[codes_key].shape[1]
    errs = []
    for st in range(7):
        lg = m.engine.read_tap(f"nar_logits{st}", T * 1024).reshape(T, 1024)
        errs.append(float(np.abs(lg[:16] - g["nar_logits"][st]).max()))
    return codes, errs


def case_row(name):
    c = CASES[name]
    a, t, text, pl, langs = case_inputs(c)
    row = dict(text=text[0], prompt=a[0], enroll=t.shape[-1], prompt_language=pl, text_language=langs)
    nb = c.get("best_of", 1)
    us = None if c["useed"] is None else synth.uniforms(4096, nb, c["useed"])
    if us is not None and nb == 1:
        us = us[:, 0]
    return c, row, us


def golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"))
