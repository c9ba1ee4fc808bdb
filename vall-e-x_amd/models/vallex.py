"""`VALLE` with the reference's constructor / load_state_dict / inference signatures (models/vallex.py:405-686),
implemented as a thin host shim over the C ABI (include/vallex_hip.h).  No torch modules, no CPU math: weights go
straight to the GPU engine, tokens come back.

Differences a caller can observe (all documented in DESIGN.md):
  * `inference` accepts the same arguments and returns the same LongTensor (1, T, 8), including `best_of` /
    `length_penalty` / `return_worst` (beams = rows of the micro-batch, selection on the device-accumulated log-probs);
  * extra, reference-less entry points for what BASELINE.json measures: `inference_batch` (distinct utterances in one
    call; the reference can only batch beams of ONE utterance, models/vallex.py:491,525-527) and the reproducibility
    hooks `uniforms=` / `force_eos_at=` (the reference needs monkey-patching for the same effect).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Union

import numpy as np

from .. import macros
from .._capi import ARITH, Batch, Engine

try:  # torch is plumbing only: the reference API hands tensors in and out
    import torch
except Exception:  # pragma: no cover
    torch = None

NUM_AUDIO_TOKENS = macros.NUM_AUDIO_TOKENS


def _np(a, dtype=None):
    if torch is not None and isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    a = np.asarray(a)
    return a.astype(dtype) if dtype is not None else a


def fresh_seed() -> int:
    """Seed of the device sampler for ONE call.  The reference samples from torch's global generator (torch.multinomial,
    models/vallex.py:850): repeated calls differ, `torch.manual_seed` makes a script reproducible.  Same contract here:
    every call that is not given an explicit `seed=` draws a new 62-bit seed from that generator."""
    if torch is not None:
        return int(torch.randint(0, 2 ** 62, (1,)).item())
    return int(np.random.default_rng().integers(0, 2 ** 62))


def vocos_expected_keys() -> List[str]:
    """The tensors of `charactr/vocos-encodec-24khz` the Vocos head computes with: codebook table, embed conv,
    AdaLayerNorm tables, 8 ConvNeXt blocks, final LayerNorm, ISTFT head projection."""
    k = ["feature_extractor.codebook_weights", "backbone.embed.weight", "backbone.embed.bias", "backbone.norm.scale.weight",
         "backbone.norm.shift.weight"]
    for i in range(8):
        k += [f"backbone.convnext.{i}." + s for s in ("dwconv.weight", "dwconv.bias", "norm.scale.weight", "norm.shift.weight",
                                                      "pwconv1.weight", "pwconv1.bias", "pwconv2.weight", "pwconv2.bias", "gamma")]
    return k + ["backbone.final_layer_norm.weight", "backbone.final_layer_norm.bias", "head.out.weight", "head.out.bias"]


def expected_keys(num_layers: int) -> List[str]:
    """State-dict layout of the reference (models/vallex.py:55-264,405-445)."""
    k = ["ar_text_embedding.word_embeddings.weight", "nar_text_embedding.word_embeddings.weight",
         "ar_audio_embedding.word_embeddings.weight", "ar_text_position.alpha", "ar_audio_position.alpha"]

    def layer(p, adaptive):
        out = [p + s for s in ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
                               "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight",
                               "linear2.bias")]
        for n in ("norm1", "norm2"):
            if adaptive:
                out += [p + n + s for s in (".project_layer.weight", ".project_layer.bias", ".norm.weight", ".norm.bias")]
            else:
                out += [p + n + ".weight", p + n + ".bias"]
        return out

    for i in range(num_layers):
        k += layer(f"ar_decoder.layers.{i}.", False)
    k += ["ar_decoder.norm.weight", "ar_decoder.norm.bias", "ar_predict_layer.weight"]
    k += [f"nar_audio_embeddings.{j}.word_embeddings.weight" for j in range(8)]
    k += ["nar_text_position.alpha", "nar_audio_position.alpha"]
    for i in range(num_layers):
        k += layer(f"nar_decoder.layers.{i}.", True)
    k += ["nar_decoder.norm.project_layer.weight", "nar_decoder.norm.project_layer.bias", "nar_decoder.norm.norm.weight",
          "nar_decoder.norm.norm.bias"]
    k += [f"nar_predict_layers.{j}.weight" for j in range(7)]
    k += [f"nar_stage_embeddings.{j}.word_embeddings.weight" for j in range(7)]
    k += ["ar_language_embedding.word_embeddings.weight", "nar_language_embedding.word_embeddings.weight"]
    return k


def sine_pe_table(rows: int, d: int = 1024) -> np.ndarray:
    """SinePositionalEmbedding.extend_pe (modules/embedding.py:75-91), built with the same torch fp32 ops on the host
    and uploaded -- never recomputed with device sin/cos."""
    if torch is not None:
        pe = torch.zeros(rows, d)
        position = torch.arange(0, rows, dtype=torch.float32).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        return pe.numpy()
    pos = np.arange(rows, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, d, 2, dtype=np.float32) * np.float32(-(math.log(10000.0) / d)))
    pe = np.zeros((rows, d), np.float32)
    pe[:, 0::2] = np.sin(pos * div)
    pe[:, 1::2] = np.cos(pos * div)
    return pe


class VALLE:
    """Drop-in for models.vallex.VALLE on the inference path."""

    def __init__(self, d_model: int, nhead: int, num_layers: int, norm_first: bool = True, add_prenet: bool = False,
                 prefix_mode: int = 0, share_embedding: bool = True, nar_scale_factor: float = 1.0, **kwargs):
        # the shipped checkpoint's configuration (utils/generation.py:67-78) is what the kernels implement
        if (d_model, nhead) != (1024, 16) or not norm_first or add_prenet or prefix_mode != 1 or nar_scale_factor != 1.0:
            raise NotImplementedError("the gfx950 engine implements the shipped VALL-E X configuration only: "
                                      "d_model=1024, nhead=16, norm_first, no prenet, prefix_mode=1")
        if kwargs.get("num_quantizers", 8) != 8 or not kwargs.get("prepend_bos", True):
            raise NotImplementedError("num_quantizers=8 and prepend_bos=True only")
        self.num_layers = num_layers
        self.language_ID = {"en": 0, "zh": 1, "ja": 2}          # models/vallex.py:439-443
        self._sd: Optional[Dict[str, np.ndarray]] = None
        self._vocos_sd: Optional[Dict[str, np.ndarray]] = None
        self._encodec_sd: Optional[Dict[str, np.ndarray]] = None
        self._engine: Optional[Engine] = None
        self._device_id = 0
        self.engine_opts = dict(max_batch=int(kwargs.get("engine_max_batch", 32)),
                                max_text=int(kwargs.get("engine_max_text", 512)),
                                max_prompt=int(kwargs.get("engine_max_prompt", 2048)),   # >= max_new: sliding-window carry-over
                                max_new=int(kwargs.get("engine_max_new", 2048)),
                                use_graph=bool(kwargs.get("engine_use_graph", True)),
                                debug_taps=bool(kwargs.get("engine_debug_taps", False)),
                                cu_mask=int(kwargs.get("engine_cu_mask", 0)),
                                arith=kwargs.get("engine_arith", "default"))

    # ---- nn.Module-ish surface used by the reference's callers --------------------------------------------------
    def to(self, device):
        s = str(device)
        if not s.startswith("cuda"):
            raise RuntimeError("vall-e-x_amd runs on MI355X only (device 'cuda[:N]'); there is no CPU path")
        self._device_id = int(s.split(":")[1]) if ":" in s else 0
        return self

    def eval(self):
        return self

    def load_state_dict(self, state_dict, strict: bool = True):
        """VALLE.load_state_dict(checkpoint["model"], strict=True) (utils/generation.py:79-83)."""
        want = expected_keys(self.num_layers)
        missing = [k for k in want if k not in state_dict]
        unexpected = [k for k in state_dict if k not in set(want)]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for VALLE: missing {missing[:4]}... "
                               f"unexpected {unexpected[:4]}...")
        self._sd = {k: _np(state_dict[k], np.float32) for k in want if k in state_dict}
        self._engine = None
        return self

    def load_vocos_state_dict(self, state_dict):
        """Weights of `Vocos.from_pretrained('charactr/vocos-encodec-24khz')` (utils/generation.py:89), vocos key names.  The
        published checkpoint also carries the whole EnCodec model under `feature_extractor.encodec.*` and the ISTFT window
        buffer; only the tensors the head computes with (`vocos_expected_keys`) are uploaded, the rest is ignored."""
        want = vocos_expected_keys()
        missing = [k for k in want if k not in state_dict]
        if missing:
            raise RuntimeError(f"Error(s) in loading the Vocos state_dict: missing {missing[:4]}{'...' if len(missing) > 4 else ''}")
        self._vocos_sd = {k: _np(state_dict[k], np.float32) for k in want}
        self._engine = None
        return self

    def load_encodec_state_dict(self, state_dict):
        """Weights of `EncodecModel.encodec_model_24khz()` (data/tokenizer.py:71-73): encodec-package, transformers-port or
        canonical key names.  The RVQ codebooks and the SEANet decoder are required; if the SEANet encoder is in the dict too,
        `AudioTokenizer.encode` (prompt enrolment) is available as well."""
        from ..data.tokenizer import canonical_encodec_state_dict
        self._encodec_sd = canonical_encodec_state_dict(state_dict)
        self._engine = None
        return self

    # ---- engine ------------------------------------------------------------------------------------------------
    @property
    def engine(self) -> Engine:
        if self._engine is None:
            if self._sd is None:
                raise RuntimeError("load_state_dict() first")
            o = self.engine_opts
            eng = Engine(self._device_id, self.num_layers, o["max_batch"], o["max_text"], o["max_prompt"], o["max_new"],
                         o["use_graph"], self._vocos_sd is not None, o["debug_taps"], self._encodec_sd is not None,
                         o["cu_mask"], ARITH[o["arith"]] if isinstance(o["arith"], str) else int(o["arith"]))
            for k, v in self._sd.items():
                eng.load_tensor(k, v)
            tmax = o["max_text"] + o["max_prompt"] + o["max_new"] + 16
            eng.load_tensor("pe_table", sine_pe_table(max(4000, tmax)))
            if self._vocos_sd is not None:
                for k, v in self._vocos_sd.items():
                    eng.load_tensor("vocos." + k, v)
            if self._encodec_sd is not None:
                for k, v in self._encodec_sd.items():
                    eng.load_tensor("encodec." + k, v)
            eng.finalize()
            self._engine = eng
        return self._engine

    def _lang_row(self, S: int, enroll: int, prompt_language, text_language) -> np.ndarray:
        out = np.empty(S, np.int32)
        out[:enroll] = self.language_ID[prompt_language]              # KeyError on unknown language, like the reference
        if isinstance(text_language, str):
            out[enroll:] = self.language_ID[text_language]
        else:
            ids = [self.language_ID[t] for t in text_language]
            if len(ids) != S - enroll:
                raise RuntimeError(f"text_language list has {len(ids)} entries for {S - enroll} text tokens")
            out[enroll:] = ids
        return out

    # ---- the reference entry point ---------------------------------------------------------------------------------
    def inference(self, x, x_lens, y, enroll_x_lens, top_k: int = -100, temperature: float = 1.0,
                  prompt_language: str = None, text_language: Union[str, List[str]] = None, best_of: int = 1,
                  length_penalty: float = 1.0, return_worst: bool = False, *, uniforms=None, force_eos_at=None,
                  seed: Optional[int] = None):
        xa, xl, ya = _np(x), _np(x_lens), _np(y)
        assert xa.ndim == 2, xa.shape                      # models/vallex.py:488-493
        assert xl.ndim == 1, xl.shape
        assert ya.ndim == 3, ya.shape
        assert ya.shape[0] == 1, ya.shape
        assert np.all(xl > 0)
        S = int(xl.max())
        row = dict(text=xa[0, :S], prompt=ya[0], enroll=int(_np(enroll_x_lens)), prompt_language=prompt_language,
                   text_language=text_language)
        u = None
        if uniforms is not None:
            u = np.asarray(uniforms, np.float32)
            u = u.reshape(-1, max(1, int(best_of))) if u.ndim == 1 else u
        codes = self.inference_batch([row], top_k=top_k, temperature=temperature, uniforms=u, force_eos_at=force_eos_at,
                                     seed=seed, best_of=best_of, length_penalty=length_penalty,
                                     return_worst=return_worst)[0]
        out = codes[None]
        return torch.from_numpy(out) if torch is not None else out

    def continual(self, x, x_lens, y):
        """`VALLE.continual` (models/vallex.py:688-787): NAR-only continuation.  The first half of `y` (1, T, 8) -- at most
        3 s = 225 frames -- is the acoustic prompt, the first codebook of the remaining frames is taken as given and
        codebooks 2..8 of those frames are predicted; the text gets NO language embedding on this path.
        Returns (1, T - prefix_len, 8) like the reference."""
        xa, xl, ya = _np(x), _np(x_lens), _np(y)
        assert xa.ndim == 2, xa.shape                      # models/vallex.py:706-712
        assert xl.ndim == 1, xl.shape
        assert ya.ndim == 3, ya.shape
        assert ya.shape[0] == 1, ya.shape
        assert np.all(xl > 0)
        S = int(xl.max())
        prefix_len = min(int(ya.shape[1] * 0.5), 3 * 75)   # :722
        text = _np(xa[0, :S], np.int32)
        batch = Batch([text], [np.full(S, -1, np.int32)], [_np(ya[0, :prefix_len], np.int32).reshape(-1, 8)])
        codes = self.engine.nar(batch, [_np(ya[0, prefix_len:, 0], np.int32)])[0]
        out = codes[None]
        return torch.from_numpy(out) if torch is not None else out

    def inference_batch(self, rows: Sequence[dict], top_k: int = -100, temperature: float = 1.0, uniforms=None,
                        force_eos_at=None, seed: Optional[int] = None, sync_every: int = 8, best_of: int = 1,
                        length_penalty: float = 1.0, return_worst: bool = False) -> List[np.ndarray]:
        """rows[i] = dict(text ids (S,), prompt codes (Tp,8), enroll, prompt_language, text_language).
        Row i equals `inference` run alone on that row.  Returns one (T_i, 8) int64 array per row."""
        if seed is None:
            seed = fresh_seed()
        texts, langs, prompts = [], [], []
        for r in rows:
            t = _np(r["text"], np.int32).reshape(-1)
            p = _np(r["prompt"], np.int32).reshape(-1, 8)
            texts.append(t)
            prompts.append(p)
            langs.append(self._lang_row(len(t), int(r["enroll"]), r["prompt_language"], r["text_language"]))
        return self.engine.infer(Batch(texts, langs, prompts), top_k=top_k, temperature=temperature, uniforms=uniforms,
                                 seed=seed, force_eos_at=force_eos_at, sync_every=sync_every, best_of=best_of,
                                 length_penalty=length_penalty, return_worst=return_worst)

    def make_batch(self, rows: Sequence[dict]) -> Batch:
        texts = [_np(r["text"], np.int32).reshape(-1) for r in rows]
        prompts = [_np(r["prompt"], np.int32).reshape(-1, 8) for r in rows]
        langs = [self._lang_row(len(t), int(r["enroll"]), r["prompt_language"], r["text_language"])
                 for t, r in zip(texts, rows)]
        return Batch(texts, langs, prompts)
