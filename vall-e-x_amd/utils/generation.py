"""Drop-in for the reference's `utils/generation.py` API on MI355X:

    preload_models()                                                     utils/generation.py:50-89
    generate_audio(text, prompt=None, language='auto', accent='no-accent')        :92-152
    generate_audio_from_long_text(text, prompt=None, language='auto', accent='no-accent', mode='sliding-window')  :155-276

Same signatures, same `.npz` prompt format (keys audio_tokens/text_tokens/lang_code), same return type
(np.float32 (n,)), same exceptions for the same misuse.  Bodies are host glue around the C ABI; AR+NAR and the Vocos
head run in libvallex_hip.so.  The CPU text front-end (G2P cleaners, langid, sentence splitting) is NOT part of the hot
path: it is pluggable through `text_tokenizer` / `sentence_splitter`, exactly like the reference's module
globals, and already-tokenised phoneme ids may be passed in place of `text`.
"""
from __future__ import annotations

import logging
import os
from typing import Callable, List, Optional, Sequence, Tuple, Union

import numpy as np

from ..macros import (NUM_LAYERS, NUM_HEAD, N_DIM, NUM_QUANTIZERS, PREFIX_MODE, SAMPLE_RATE, code2lang,  # noqa: F401
                      lang2token, langdropdown2token, token2lang)
from ..models.vallex import VALLE

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

# module globals, as in the reference (utils/generation.py:30-48)
device = "cuda:0"
model: Optional[VALLE] = None
vocos = None
text_tokenizer: Optional[Callable[[str], Tuple[List[int], List[str]]]] = None   # text -> (phone ids, per-id langs)
sentence_splitter: Optional[Callable[[str], List[str]]] = None
language_detector: Optional[Callable[[str], str]] = None
rng = None      # test hook: an object with .random() replaces the sliding-window coin; None = the reference's own draw,
#                `torch.rand(1) < 0.5` (utils/generation.py:264), which follows torch.manual_seed like the reference does

checkpoints_dir = "./checkpoints/"
model_checkpoint_name = "vallex-checkpoint.pt"
PRESET_DIRS = ["./presets/", "./customs/"]


class VocosHIP:
    """Stand-in for the pip `vocos` object the reference holds (utils/generation.py:89,148-150): the two calls the
    reference makes, fused on the GPU."""

    def __init__(self, valle: VALLE):
        self._m = valle

    def codes_to_features(self, codes):
        c = codes.detach().cpu().numpy() if torch is not None and isinstance(codes, torch.Tensor) else np.asarray(codes)
        assert c.ndim == 3 and c.shape[0] == NUM_QUANTIZERS, c.shape            # (8, B, T)
        return ("codes", np.ascontiguousarray(np.transpose(c, (1, 2, 0))).astype(np.int64))   # (B, T, 8)

    def decode(self, features, bandwidth_id=None):
        tag, codes = features
        assert tag == "codes"
        bid = 2 if bandwidth_id is None else int(np.asarray(
            bandwidth_id.detach().cpu().numpy() if torch is not None and isinstance(bandwidth_id, torch.Tensor)
            else bandwidth_id).reshape(-1)[0])
        audio = self._m.engine.vocos_decode([codes[i] for i in range(codes.shape[0])], bid)
        out = np.stack(audio) if len({a.shape[0] for a in audio}) == 1 else audio
        return torch.from_numpy(out) if torch is not None and isinstance(out, np.ndarray) else out


def _torch_load(path, allow_unsafe_pickle: Optional[bool] = None):
    """torch.load(path, map_location='cpu') as the reference does (utils/generation.py:79), under torch >= 2.6's
    weights_only=True: a checkpoint of plain tensors -- what `["model"]` is -- needs nothing else.  Published training
    checkpoints carry harmless bookkeeping objects next to "model"; the ones known from the reference's trainer
    (argparse.Namespace, pathlib paths, OrderedDict) are allow-listed for this one load.  Anything else is REFUSED: the
    permissive unpickler executes code from the file, so it runs only on an explicit opt-in --
    `allow_unsafe_pickle=True` (preload_models(..., allow_unsafe_pickle=True)) or VALLEX_ALLOW_UNSAFE_PICKLE=1 -- for a
    file whose origin the caller trusts, like the reference's torch 2.0 `torch.load` did for every file."""
    import argparse
    import collections
    import pathlib
    if allow_unsafe_pickle is None:
        allow_unsafe_pickle = os.environ.get("VALLEX_ALLOW_UNSAFE_PICKLE", "") == "1"
    if allow_unsafe_pickle:
        logging.warning(f"{path}: loading with weights_only=False (explicit opt-in): the file can execute arbitrary code")
        return torch.load(path, map_location="cpu", weights_only=False)
    if not hasattr(torch.serialization, "safe_globals"):
        raise RuntimeError(f"torch {torch.__version__} has no torch.serialization.safe_globals (needs torch >= 2.5): the allow-listed "
                           "weights-only load of a training checkpoint is not available; pass allow_unsafe_pickle=True for a file you trust")
    paths = [pathlib.PosixPath, pathlib.PurePosixPath, pathlib.Path]
    # Python >= 3.13 moved the classes to pathlib._local, so `cls.__module__` no longer matches the "pathlib.PosixPath" a checkpoint
    # written by an older interpreter names: register them under the historical name as well (torch accepts (obj, "qualified.name"))
    safe = [argparse.Namespace, collections.OrderedDict] + paths
    safe += [(cls, f"pathlib.{cls.__name__}") for cls in paths if cls.__module__ != "pathlib"]
    try:
        with torch.serialization.safe_globals(safe):
            return torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:                                   # pickle.UnpicklingError: weights-only refused a global
        if "weights_only" not in str(e) and "Weights only" not in str(e):
            raise
        raise RuntimeError(
            f"{path}: torch refused to load it with weights_only=True ({type(e).__name__}: it pickles objects outside the "
            f"allow-list). Loading such a file executes code from it. If you trust its origin, pass allow_unsafe_pickle=True "
            f"(preload_models / tools/verify_checkpoint.py --allow-unsafe-pickle) or set VALLEX_ALLOW_UNSAFE_PICKLE=1.") from e


def preload_models(checkpoint: Optional[str] = None, vocos_checkpoint: Optional[str] = None, state_dict=None,
                   vocos_state_dict=None, num_layers: int = NUM_LAYERS, allow_unsafe_pickle: Optional[bool] = None,
                   **engine_opts):
    """Build the engine and load weights.  With no arguments behaves like the reference (expects
    ./checkpoints/vallex-checkpoint.pt; there is no network here, so a missing file raises instead of downloading).
    Checkpoint files are read with torch's weights-only unpickler; allow_unsafe_pickle=True is the explicit opt-in to the
    permissive one (see _torch_load)."""
    global model, vocos
    if state_dict is None:
        path = checkpoint or os.path.join(checkpoints_dir, model_checkpoint_name)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found (the reference downloads it, utils/generation.py:53-65; "
                                    "no network here): pass checkpoint= or state_dict=")
        state_dict = _torch_load(path, allow_unsafe_pickle)["model"]                                  # :79-83
    m = VALLE(N_DIM, NUM_HEAD, num_layers, norm_first=True, add_prenet=False, prefix_mode=PREFIX_MODE,
              share_embedding=True, nar_scale_factor=1.0, prepend_bos=True, num_quantizers=NUM_QUANTIZERS,
              **{"engine_" + k: v for k, v in engine_opts.items()})
    m.to(device).load_state_dict(state_dict, strict=True)
    m.eval()
    if vocos_state_dict is None and vocos_checkpoint is not None:
        vocos_state_dict = _torch_load(vocos_checkpoint, allow_unsafe_pickle)
    if vocos_state_dict is not None:
        m.load_vocos_state_dict(vocos_state_dict)
    model = m
    vocos = VocosHIP(m) if vocos_state_dict is not None else None
    return None


def _load_prompt(prompt):
    """utils/generation.py:103-119: path, preset name or custom name -> (audio (1,Tp,8), text (1,Sp), lang str)."""
    prompt_path = prompt
    for d in [""] + PRESET_DIRS:
        cand = prompt if d == "" else d + prompt + ".npz"
        if os.path.exists(cand):
            prompt_path = cand
            break
    else:
        raise ValueError(f"Cannot find prompt {prompt}")
    data = np.load(prompt_path)
    return (np.asarray(data["audio_tokens"]).astype(np.int32), np.asarray(data["text_tokens"]).astype(np.int32),
            code2lang[int(data["lang_code"])])


def _tokenize(text, lang_token) -> Tuple[np.ndarray, Union[List[str], None]]:
    """utils/generation.py:127-132 (`text_tokenizer.tokenize(f"_{text}")` + collater).  A sequence of ints is taken
    as already-tokenised phoneme ids."""
    if not isinstance(text, str):
        ids = np.asarray(text, np.int32).reshape(-1)
        if ids.size == 0:
            raise ValueError("Empty text is given")                    # utils/g2p/__init__.py:23-24
        return ids, None
    if text_tokenizer is None:
        raise RuntimeError("no text front-end configured: set vallex_amd.utils.generation.text_tokenizer to a callable "
                           "text -> (phoneme ids, languages) (the reference's PhonemeBpeTokenizer), or pass phoneme ids")
    ids, langs = text_tokenizer(f"_{lang_token}{text}{lang_token}".strip())
    if len(ids) == 0:
        raise ValueError("Empty text is given")
    return np.asarray(ids, np.int32), list(langs)


def _carry_coin() -> bool:
    """utils/generation.py:264: `if torch.rand(1) < 0.5` -- drawn from torch's global generator, so a script that calls
    torch.manual_seed flips the same coins as it does with the reference."""
    if rng is not None:
        return rng.random() < 0.5
    if torch is not None:
        return bool(torch.rand(1) < 0.5)
    return float(np.random.random()) < 0.5


def _detect(text, language):
    if language != "auto":
        return language
    if not isinstance(text, str):
        raise ValueError("language='auto' needs a text string")
    if language_detector is None:
        raise RuntimeError("language='auto' needs a language detector (the reference uses langid, "
                           "utils/generation.py:96): set vallex_amd.utils.generation.language_detector")
    return language_detector(text)


def _infer_one(text, audio_prompts, text_prompts, lang_pr, language, accent, **kw):
    lang_token = lang2token[language]                                  # KeyError on unknown language (macros.py:8-13)
    lang = token2lang[lang_token]
    phone_tokens, langs = _tokenize(text, lang_token)
    enroll_x_lens = text_prompts.shape[-1]
    text_tokens = np.concatenate([text_prompts.reshape(-1), phone_tokens])[None]          # :133
    lens = np.array([text_tokens.shape[-1]], np.int32)
    if lang_pr is None:
        lang_pr = lang if lang != "mix" else "en"                      # :123
    lang_eff = lang if accent == "no-accent" else token2lang[langdropdown2token[accent]]   # :136
    text_language = (langs if langs is not None else lang_eff) if accent == "no-accent" else lang_eff
    if text_language == "mix":
        raise KeyError("mix")                                          # language_ID has no 'mix' (models/vallex.py:439-443)
    out = model.inference(text_tokens, lens, audio_prompts, enroll_x_lens=enroll_x_lens, top_k=-100, temperature=1,
                          prompt_language=lang_pr, text_language=text_language, **kw)
    return out, phone_tokens


def generate_audio(text, prompt=None, language="auto", accent="no-accent", **kw):
    if model is None or vocos is None:
        raise RuntimeError("call preload_models() first")
    if isinstance(text, str):
        text = text.replace("\n", "").strip(" ")
    language = _detect(text, language)
    if prompt is not None:
        audio_prompts, text_prompts, lang_pr = _load_prompt(prompt)
    else:
        audio_prompts = np.zeros([1, 0, NUM_QUANTIZERS], np.int32)     # :121-123
        text_prompts = np.zeros([1, 0], np.int32)
        lang_pr = None
    logging.info(f"synthesize text: {text}")
    encoded_frames, _ = _infer_one(text, audio_prompts, text_prompts, lang_pr, language, accent, **kw)
    frames = (encoded_frames.permute(2, 0, 1) if torch is not None and isinstance(encoded_frames, torch.Tensor)
              else np.transpose(np.asarray(encoded_frames), (2, 0, 1)))                                         # :148
    features = vocos.codes_to_features(frames)
    samples = vocos.decode(features, bandwidth_id=np.array([2]))
    s = samples.squeeze()
    return s.cpu().numpy() if torch is not None and isinstance(s, torch.Tensor) else np.asarray(s)


def generate_audio_batch(texts, prompts=None, language="auto", accent="no-accent", text_languages=None, **kw):
    """Batch form of `generate_audio` (no reference equivalent: the reference synthesises one utterance per call,
    utils/generation.py:92-152): utterance i of the batch equals `generate_audio(texts[i], prompts[i], language[i], accent)`.
    `texts`: strings (need the tokenizer hook) or phoneme-id arrays; `prompts`: one preset name / path / None per utterance (or
    one for all); `language`: one string or one per utterance; `text_languages`: optional per-utterance per-id language lists
    (what the tokenizer returned, `TextFrontendService`).  All utterances go through ONE `VALLE.inference_batch` call and ONE
    Vocos call.  Returns a list of float32 waveforms."""
    if model is None or vocos is None:
        raise RuntimeError("call preload_models() first")
    n = len(texts)
    prompts = list(prompts) if isinstance(prompts, (list, tuple)) else [prompts] * n
    langs_in = list(language) if isinstance(language, (list, tuple)) else [language] * n
    text_languages = list(text_languages) if text_languages is not None else [None] * n
    rows = []
    for text, prompt, lang_in, tl in zip(texts, prompts, langs_in, text_languages):
        if isinstance(text, str):
            text = text.replace("\n", "").strip(" ")
        lang_in = _detect(text, lang_in)
        if prompt is not None:
            audio_prompts, text_prompts, lang_pr = _load_prompt(prompt)
        else:
            audio_prompts = np.zeros([1, 0, NUM_QUANTIZERS], np.int32)
            text_prompts = np.zeros([1, 0], np.int32)
            lang_pr = None
        lang_token = lang2token[lang_in]
        lang = token2lang[lang_token]
        phone_tokens, langs = _tokenize(text, lang_token)
        if tl is not None:
            langs = list(tl)
        if lang_pr is None:
            lang_pr = lang if lang != "mix" else "en"
        lang_eff = lang if accent == "no-accent" else token2lang[langdropdown2token[accent]]
        text_language = (langs if langs is not None else lang_eff) if accent == "no-accent" else lang_eff
        if text_language == "mix":
            raise KeyError("mix")
        rows.append(dict(text=np.concatenate([text_prompts.reshape(-1), phone_tokens]), prompt=audio_prompts[0],
                         enroll=text_prompts.shape[-1], prompt_language=lang_pr, text_language=text_language))
    codes = model.inference_batch(rows, top_k=-100, temperature=1, **kw)
    return model.engine.vocos_decode(codes, 2)


def generate_audio_from_long_text(text, prompt=None, language="auto", accent="no-accent", mode="sliding-window", **kw):
    """utils/generation.py:155-276.  `text` may also be a list of sentences / id arrays (pre-split)."""
    if model is None or vocos is None:
        raise RuntimeError("call preload_models() first")
    if prompt is None or prompt == "":
        prompt = None
        mode = "sliding-window"                                        # :162-163: overrides ANY mode, also an unknown one
    if isinstance(text, str):
        if sentence_splitter is None:
            raise RuntimeError("no sentence splitter configured (the reference uses utils/sentence_cutter.py): set "
                               "vallex_amd.utils.generation.sentence_splitter or pass a list of sentences")
        sentences = sentence_splitter(text)
    else:
        sentences = list(text)
    if language == "auto":                                             # :166-167: langid.classify(text) on the WHOLE input
        language = _detect(text if isinstance(text, str) else (sentences[0] if sentences else ""), language)
    if prompt is not None:
        audio_prompts, text_prompts, lang_pr = _load_prompt(prompt)
    else:
        audio_prompts = np.zeros([1, 0, NUM_QUANTIZERS], np.int32)
        text_prompts = np.zeros([1, 0], np.int32)
        lang_pr = None
    # the reference only looks at `mode` here, after the override, the sentence split, the language detection and the prompt
    # lookup (its if / elif / else ends in the raise, utils/generation.py:197,229,275-276) -- same order of errors
    if mode not in ("fixed-prompt", "sliding-window"):
        raise ValueError(f"No such mode {mode}")                       # :276
    original = (audio_prompts, text_prompts)
    chunks = []
    for sent in sentences:
        if isinstance(sent, str):
            sent = sent.replace("\n", "").strip(" ")
            if sent == "":
                continue                                               # :198-199,236-237
        frames, phone_tokens = _infer_one(sent, audio_prompts, text_prompts, lang_pr, language, accent, **kw)
        f = frames.numpy() if torch is not None and isinstance(frames, torch.Tensor) else np.asarray(frames)
        chunks.append(f)
        if mode == "sliding-window":
            if _carry_coin():                                          # torch.rand(1) < 0.5  (:264)
                # encoded_frames[:, :, -NUM_QUANTIZERS:] slices the codebook axis, i.e. keeps ALL frames (:265)
                audio_prompts = f[:, :, -NUM_QUANTIZERS:].astype(np.int32)
                text_prompts = np.asarray(phone_tokens, np.int32)[None]             # text_tokens[:, enroll_x_lens:] (:266)
            else:
                audio_prompts, text_prompts = original                 # :268-269
    if not chunks:
        return np.zeros(0, np.float32)
    complete = np.concatenate(chunks, axis=1)                          # (1, sum T, 8)
    features = vocos.codes_to_features(np.transpose(complete, (2, 0, 1)))
    samples = vocos.decode(features, bandwidth_id=np.array([2]))
    s = samples.squeeze()
    return s.cpu().numpy() if torch is not None and isinstance(s, torch.Tensor) else np.asarray(s)
