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
