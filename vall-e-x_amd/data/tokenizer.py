"""Mirror of the reference's `data/tokenizer.py` (`AudioTokenizer`, :63-96) on MI355X.

`AudioTokenizer.decode(frames)` = `EncodecModel.encodec_model_24khz().decode(frames)` -- the EnCodec SEANet decoder, the
reference's legacy vocoder beside Vocos (README.md:29-30); `AudioTokenizer.encode(wav)` = `.encode(wav)` at 6 kbps -- the
SEANet encoder + residual VQ used for prompt enrolment (data/tokenizer.py:92-111).  The arithmetic lives in the pip package
`encodec`; here it runs in libvallex_hip.so (`vx_encodec_decode`, `vx_encodec_encode`).

Weights: pass the `encodec` package's state-dict (`decoder.model.N...`, weight_g / weight_v or plain weight after
`remove_weight_norm`, data/tokenizer.py:33-60), the `transformers` port's (`decoder.layers.N...parametrizations...`), or
the canonical folded names of the C ABI -- `canonical_encodec_state_dict` normalises all three.
"""
from __future__ import annotations

import re
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def _np(a) -> np.ndarray:
    if torch is not None and isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.asarray(a, dtype=np.float32)


def _fold(g: np.ndarray, v: np.ndarray) -> np.ndarray:
    """torch weight_norm (dim=0): w = g * v / ||v||, the norm taken over every dim but 0."""
    n = np.sqrt((v.astype(np.float64) ** 2).reshape(v.shape[0], -1).sum(1)).reshape((-1,) + (1,) * (v.ndim - 1))
    return (g.astype(np.float64).reshape(n.shape) * v.astype(np.float64) / n).astype(np.float32)


_SUB = {"block.1": "block1", "block.3": "block3", "block1": "block1", "block3": "block3", "shortcut": "shortcut"}


def canonical_encodec_state_dict(sd: Dict[str, Any]) -> Dict[str, np.ndarray]:
    """-> {'quantizer.{q}.embed', '{decoder|encoder}.{i}.weight|bias', '....{i}.block1|block3|shortcut.weight|bias',
    'decoder.1.lstm.*', 'encoder.13.lstm.*'} with weight-norm folded, for the first 8 codebooks, the SEANet decoder and
    (when present) the SEANet encoder."""
    out: Dict[str, np.ndarray] = {}
    pend: Dict[str, Dict[str, np.ndarray]] = {}
    for k, v in sd.items():
        m = re.match(r"quantizer\.(?:vq\.)?layers\.(\d+)\.(?:_codebook|codebook)\.embed$", k) or \
            re.match(r"quantizer\.(\d+)\.embed$", k)
        if m:
            if int(m.group(1)) < 8:
                out[f"quantizer.{int(m.group(1))}.embed"] = _np(v)
            continue
        m = re.match(r"(decoder|encoder)\.(?:model\.|layers\.)?(\d+)\.(.*)$", k)
        if not m:
            continue
        side, i, rest = m.group(1), int(m.group(2)), m.group(3)
        if rest.startswith("lstm."):
            out[f"{side}.{i}.{rest}"] = _np(v)
            continue
        sub = ""
        for a, b in _SUB.items():
            if rest.startswith(a + "."):
                sub, rest = b + ".", rest[len(a) + 1:]
                break
        rest = re.sub(r"^(conv\.conv\.|convtr\.convtr\.|conv\.)", "", rest)
        base = f"{side}.{i}.{sub}"
        if rest in ("weight", "bias"):
            out[base + rest] = _np(v)
        elif rest in ("weight_g", "parametrizations.weight.original0"):
            pend.setdefault(base, {})["g"] = _np(v)
        elif rest in ("weight_v", "parametrizations.weight.original1"):
            pend.setdefault(base, {})["v"] = _np(v)
    for base, gv in pend.items():
        if "g" in gv and "v" in gv:                      # a lone half (stray / partial dict) is ignored
            out[base + "weight"] = _fold(gv["g"], gv["v"])
    return out


class AudioTokenizer:
    """EnCodec audio (decode side).  Mirrors data/tokenizer.py:63-96: `.sample_rate`, `.channels`, `.decode(frames)`."""

    sample_rate = 24000
    channels = 1

    def __init__(self, device: Any = None, valle=None):
        self._device = device
        self._m = valle                     # a vallex_amd VALLE holding the engine with EnCodec weights loaded

    @property
    def device(self):
        return self._device

    def encode(self, wav):
        """wav (B, 1, L) or (B, L) mono 24 kHz fp32 -> [(codes (B, 8, T) int64, None)] like `codec.encode` at 6 kbps
        (data/tokenizer.py:92-93); T = ceil(L / 320).  All rows of one call have the same length, as a tensor does."""
        w = wav.detach().cpu().numpy() if torch is not None and isinstance(wav, torch.Tensor) else np.asarray(wav)
        w = np.asarray(w, np.float32)
        if w.ndim == 3:
            assert w.shape[1] == 1, w.shape              # mono (convert_audio(..., target_channels=1), data/tokenizer.py:103)
            w = w[:, 0]
        assert w.ndim == 2, w.shape
        codes = self._m.engine.encodec_encode([w[i] for i in range(w.shape[0])])
        out = np.ascontiguousarray(np.transpose(np.stack(codes), (0, 2, 1))).astype(np.int64)      # (B, 8, T)
        return [(torch.from_numpy(out) if torch is not None else out, None)]

    def decode(self, frames):
        """frames: [(codes (B, 8, T), scale=None)] as produced by encodec; returns (B, 1, 320*T)."""
        assert len(frames) == 1, "encodec_model_24khz is not chunked: exactly one frame"
        codes, scale = frames[0]
        assert scale is None
        c = codes.detach().cpu().numpy() if torch is not None and isinstance(codes, torch.Tensor) else np.asarray(codes)
        assert c.ndim == 3 and c.shape[1] == 8, c.shape
        bt8 = np.ascontiguousarray(np.transpose(c, (0, 2, 1))).astype(np.int64)
        audio = self._m.engine.encodec_decode([bt8[i] for i in range(bt8.shape[0])])
        out = np.stack(audio)[:, None, :]
        return torch.from_numpy(out) if torch is not None else out
