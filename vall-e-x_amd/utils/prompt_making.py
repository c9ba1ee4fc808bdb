"""Prompt enrolment on MI355X: mirror of the reference's `utils/prompt_making.make_prompt` (utils/prompt_making.py:57-84) and
`data/tokenizer.tokenize_audio` (data/tokenizer.py:99-111) -- waveform in, `.npz` voice preset out, in the reference's wire format
(`audio_tokens (1, T, 8)`, `text_tokens (1, S)`, `lang_code`), so the file drops into `generate_audio(prompt=...)` of either
implementation.  The audio side (EnCodec SEANet encoder + RVQ, 6 kbps) runs in libvallex_hip.so (`vx_encodec_encode`).

What is NOT here (third-party CPU code outside the hot path): Whisper transcription (pass `transcript=`), the G2P /
BPE tokenizer and langid (same pluggable hooks as `utils.generation`).  Audio at another sample rate is resampled to 24 kHz like the
reference does (data/tokenizer.py:105 `convert_audio` = `torchaudio.transforms.Resample(sr, 24000)`): `resample_sinc_hann` below restates
torchaudio's published default (Hann-windowed sinc, lowpass_filter_width 6, rolloff 0.99) -- torchaudio is not installed here, so this
one piece is NOT pinned to the package; assign `resampler = lambda wav, sr, target_sr: ...` to plug in the real one.  Stereo is averaged
like the reference does (utils/prompt_making.py:63-64).
"""
from __future__ import annotations

import logging
import math
import os
from typing import Callable, Optional, Tuple, Union

import numpy as np

from ..data.tokenizer import AudioTokenizer
from ..macros import lang2code, lang2token
from . import generation as G

codec: Optional[AudioTokenizer] = None        # module global like the reference's (utils/prompt_making.py:27)
CUSTOMS_DIR = "./customs/"
resampler: Optional[Callable[[np.ndarray, int, int], np.ndarray]] = None      # (wav (C, L), sr, target_sr) -> (C, L'); None = built-in


def resample_sinc_hann(wav: np.ndarray, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
                       rolloff: float = 0.99) -> np.ndarray:
    """Band-limited resampling of (C, L) fp32 audio: the algorithm torchaudio documents for `transforms.Resample` with its defaults
    (resampling_method "sinc_interp_hann").  With o, n = the two rates over their gcd: n phase kernels of 2*width + o taps,
    k_i[j] = sinc(t) * cos^2(pi t / (2 lpw)) * base / o with t = clamp((-i / n + (j - width) / o) * base, +-lpw), base =
    min(o, n) * rolloff, width = ceil(lpw * o / base) (built in fp64, applied in fp32); the zero-padded signal is correlated at
    stride o, the n phases interleave to the output, cut to ceil(n * L / o) samples."""
    import torch
    import torch.nn.functional as F
    wav = np.asarray(wav, np.float32)
    g = math.gcd(int(orig_freq), int(new_freq))
    o, n = int(orig_freq) // g, int(new_freq) // g
    if o == n:
        return wav
    base = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base)
    idx = torch.arange(-width, width + o, dtype=torch.float64)[None, None] / o
    # torchaudio's _get_sinc_resample_kernel with dtype=None: the phase term is an int64 arange divided in fp32 and promoted to
    # fp64 when idx is added -- reproduced so that the fp32 taps are the package's bit for bit
    t = ((torch.arange(0, -n, -1).to(torch.float32) / n).to(torch.float64)[:, None, None] + idx) * base
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = (torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / o)).to(torch.float32)    # (n, 1, 2*width + o)
    x = torch.from_numpy(np.ascontiguousarray(wav.reshape(-1, wav.shape[-1])))
    length = x.shape[-1]
    x = F.pad(x, (width, width + o))
    y = F.conv1d(x[:, None], kernels, stride=o).transpose(1, 2).reshape(x.shape[0], -1)
    y = y[:, : math.ceil(n * length / o)]
    return y.reshape(wav.shape[:-1] + (y.shape[-1],)).numpy()


def _load_wav(path: str) -> Tuple[np.ndarray, int]:
    from scipy.io import wavfile                     # PCM / float WAV; the reference uses torchaudio.load
    sr, data = wavfile.read(path)
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    elif data.dtype.kind == "u":
        data = (data.astype(np.float32) - 128.0) / 128.0
    data = np.asarray(data, np.float32)
    wav = data[None, :] if data.ndim == 1 else data.T            # (channels, samples) like torchaudio
    return wav, int(sr)


def tokenize_audio(tokenizer: AudioTokenizer, audio: Union[str, Tuple[np.ndarray, int]]):
    """data/tokenizer.py:99-111: (path | (wav (C, L), sr)) -> `tokenizer.encode(wav[None])` = [(codes (1, 8, T), None)]."""
    wav, sr = _load_wav(audio) if isinstance(audio, str) else audio
    wav = np.asarray(wav.detach().cpu().numpy() if hasattr(wav, "detach") else wav, np.float32)
    if wav.ndim == 1:
        wav = wav[None]
    if wav.shape[0] > 1:                                          # convert_audio(..., target_channels=1): channel mean, then resample
        wav = wav.mean(0, keepdims=True)
    if sr != tokenizer.sample_rate:
        wav = np.asarray((resampler or resample_sinc_hann)(wav, int(sr), tokenizer.sample_rate), np.float32)
    return tokenizer.encode(wav[None])                            # (1, 1, L)


def make_transcript(name, wav, sr, transcript: Optional[str] = None):
    """utils/prompt_making.py:87-120 without Whisper: the transcript must be given; language from the detector hook.
    Like the reference (:91-92, `wav /= wav.abs().max()` on the caller's FloatTensor) a waveform whose peak exceeds 1 is
    normalised IN PLACE, so the audio that `tokenize_audio` encodes afterwards is the normalised one."""
    if isinstance(wav, np.ndarray) and wav.size and float(np.abs(wav).max()) > 1:
        wav /= np.abs(wav).max()
    if transcript is None or transcript == "":
        raise RuntimeError("no transcript given: the reference transcribes with Whisper here (utils/prompt_making.py:99-110), which "
                           "is outside this package; pass transcript=")
    if G.language_detector is None:
        raise RuntimeError("set vallex_amd.utils.generation.language_detector (the reference uses langid.classify, "
                           "utils/prompt_making.py:113)")
    lang = G.language_detector(transcript)
    lang_token = lang2token[lang]
    return lang_token + transcript + lang_token, lang


def make_prompt(name: str, audio_prompt_path: Union[str, Tuple[np.ndarray, int]], transcript: Optional[str] = None,
                save_dir: Optional[str] = None) -> str:
    """utils/prompt_making.py:57-84.  `audio_prompt_path`: a WAV path or `(wav (C, L), sr)`.  Returns the path of the written .npz."""
    global codec
    if G.model is None:
        raise RuntimeError("call preload_models() first (with EnCodec weights: model.load_encodec_state_dict)")
    if codec is None:
        codec = AudioTokenizer(device=G.device, valle=G.model)
    wav_pr, sr = _load_wav(audio_prompt_path) if isinstance(audio_prompt_path, str) else audio_prompt_path
    wav_pr = np.array(wav_pr, np.float32)                         # own copy: make_transcript may normalise it in place
    if wav_pr.ndim == 1:
        wav_pr = wav_pr[None]
    if wav_pr.shape[-1] / sr > 15:                                # :60-61
        raise ValueError(f"Prompt too long, expect length below 15 seconds, got {wav_pr.shape[-1] / sr} seconds.")
    if wav_pr.shape[0] == 2:                                      # :62-63
        wav_pr = wav_pr.mean(0, keepdims=True)
    text_pr, lang_pr = make_transcript(name, wav_pr, sr, transcript)
    encoded_frames = tokenize_audio(codec, (wav_pr, sr))          # :67
    codes = encoded_frames[0][0]
    codes = codes.numpy() if hasattr(codes, "numpy") else np.asarray(codes)
    audio_tokens = np.transpose(codes, (0, 2, 1))                 # (1, T, 8)   (:68)
    if G.text_tokenizer is None:
        raise RuntimeError("set vallex_amd.utils.generation.text_tokenizer (the reference's PhonemeBpeTokenizer.tokenize)")
    phonemes, _langs = G.text_tokenizer(f"{text_pr}".strip())     # :71
    text_tokens = np.asarray(phonemes, np.int64)[None]            # text_collater([phonemes])  (:72-76)
    d = save_dir if save_dir is not None else CUSTOMS_DIR
    os.makedirs(d, exist_ok=True)
    save_path = os.path.join(d, f"{name}.npz")
    np.savez(save_path, audio_tokens=audio_tokens, text_tokens=text_tokens, lang_code=lang2code[lang_pr])      # :81-82
    logging.info(f"Successful. Prompt saved to {save_path}")
    return save_path
