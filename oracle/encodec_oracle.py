"""CPU oracle for the EnCodec 24 kHz SEANet **decoder** (the "1-D ConvTranspose stack" of
BASELINE.json's north_star; reference call site `AudioTokenizer.decode` -> `codec.decode(frames)`,
data/tokenizer.py:95-96, legacy alternative to the Vocos head, README.md:29-30).

TEST INFRASTRUCTURE ONLY (see oracle/README.md).

The arithmetic lives in the pip package `encodec` (unpinned, requirements.txt:7), absent offline.  It is restated here
functionally in torch and **pinned** against the faithful port that IS installed in the build container,
`transformers.models.encodec.modeling_encodec` (oracle/make_golden_encodec.py loads the same synthetic weights into
`transformers.EncodecModel` and commits its output to tests/golden/encodec_*.npz).

Pipeline (EncodecModel._decode_frame): codes (B, 8, T) -> sum_q embed_q[code] (B,128,T) -> Conv1d(128,512,k7, causal
reflect pad) -> 2-layer LSTM(512) + skip -> 4 x [ELU, ConvTranspose1d(C, C/2, k=2r, stride r, trim r on the right),
ResnetBlock(C/2): shortcut_1x1(x) + conv1x1(ELU(conv_k3(ELU(x))))], r = 8,5,4,2 -> ELU -> Conv1d(32,1,k7) -> (B,1,320T).
Weight-norm is folded (w = g * v / ||v||, norm over all dims but 0) before use.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

RATIOS = (8, 5, 4, 2)
HID = 128
NF = 32
NQ = 8
BINS = 1024


def encodec_state_dict(seed: int = 3) -> "OrderedDict[str, np.ndarray]":
    """Synthetic decoder + RVQ weights, canonical (weight-norm folded) names used by the engine:
       quantizer.{q}.embed (1024,128); decoder.{i}.weight/.bias; decoder.{i}.block1|block3|shortcut.weight/.bias;
       decoder.1.lstm.{weight_ih,weight_hh,bias_ih,bias_hh}_l{0,1}."""
    rng = np.random.default_rng(seed)

    def u(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return rng.uniform(-b, b, size=shape).astype(np.float32)

    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for q in range(NQ):
        sd[f"quantizer.{q}.embed"] = (rng.standard_normal((BINS, HID)) * 0.5).astype(np.float32)
    c = NF * 16                                                   # 512
    sd["decoder.0.weight"] = u((c, HID, 7), HID * 7)
    sd["decoder.0.bias"] = u((c,), HID * 7)
    for l in range(2):
        sd[f"decoder.1.lstm.weight_ih_l{l}"] = u((4 * c, c), c)
        sd[f"decoder.1.lstm.weight_hh_l{l}"] = u((4 * c, c), c)
        sd[f"decoder.1.lstm.bias_ih_l{l}"] = u((4 * c,), c)
        sd[f"decoder.1.lstm.bias_hh_l{l}"] = u((4 * c,), c)
    idx = 3
    for r in RATIOS:
        sd[f"decoder.{idx}.weight"] = u((c, c // 2, 2 * r), c * 2)       # ConvTranspose1d (Cin, Cout, k)
        sd[f"decoder.{idx}.bias"] = u((c // 2,), c * 2)
        d = c // 2
        p = f"decoder.{idx + 1}."
        sd[p + "block1.weight"] = u((d // 2, d, 3), d * 3)
        sd[p + "block1.bias"] = u((d // 2,), d * 3)
        sd[p + "block3.weight"] = u((d, d // 2, 1), d // 2)
        sd[p + "block3.bias"] = u((d,), d // 2)
        sd[p + "shortcut.weight"] = u((d, d, 1), d)
        sd[p + "shortcut.bias"] = u((d,), d)
        c = d
        idx += 3
    sd["decoder.15.weight"] = u((1, NF, 7), NF * 7)
    sd["decoder.15.bias"] = u((1,), NF * 7)
    return sd


def encodec_encoder_state_dict(seed: int = 4) -> "OrderedDict[str, np.ndarray]":
    """Synthetic SEANet ENCODER weights (weight-norm folded), canonical names:
       encoder.{i}.weight/.bias for the plain convs (i = 0, 3, 6, 9, 12, 15), encoder.{i}.block1|block3|shortcut.weight/.bias
       for the residual blocks (i = 1, 4, 7, 10), encoder.13.lstm.{weight_ih,weight_hh,bias_ih,bias_hh}_l{0,1}.
    The RVQ codebooks are the decoder dict's quantizer.{q}.embed."""
    rng = np.random.default_rng(seed)

    def u(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return rng.uniform(-b, b, size=shape).astype(np.float32)

    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    c = NF
    sd["encoder.0.weight"] = u((c, 1, 7), 7)
    sd["encoder.0.bias"] = u((c,), 7)
    idx = 1
    for r in reversed(RATIOS):                                    # 2, 4, 5, 8
        p = f"encoder.{idx}."
        sd[p + "block1.weight"] = u((c // 2, c, 3), c * 3)
        sd[p + "block1.bias"] = u((c // 2,), c * 3)
        sd[p + "block3.weight"] = u((c, c // 2, 1), c // 2)
        sd[p + "block3.bias"] = u((c,), c // 2)
        sd[p + "shortcut.weight"] = u((c, c, 1), c)
        sd[p + "shortcut.bias"] = u((c,), c)
        sd[f"encoder.{idx + 2}.weight"] = u((2 * c, c, 2 * r), c * 2 * r)      # Conv1d(c, 2c, k = 2r, stride r)
        sd[f"encoder.{idx + 2}.bias"] = u((2 * c,), c * 2 * r)
        c *= 2
        idx += 3
    for l in range(2):                                            # encoder.13: 2-layer LSTM(512) + skip
        sd[f"encoder.13.lstm.weight_ih_l{l}"] = u((4 * c, c), c)
        sd[f"encoder.13.lstm.weight_hh_l{l}"] = u((4 * c, c), c)
        sd[f"encoder.13.lstm.bias_ih_l{l}"] = u((4 * c,), c)
        sd[f"encoder.13.lstm.bias_hh_l{l}"] = u((4 * c,), c)
    sd["encoder.15.weight"] = u((HID, c, 7), c * 7)
    sd["encoder.15.bias"] = u((HID,), c * 7)
    return sd


def _pad_causal_reflect(x: torch.Tensor, pad: int) -> torch.Tensor:
    """EncodecConv1d._pad1d(mode='reflect') for (pad, 0) with stride 1: reflect on the left; if the sequence is not
    longer than the pad, zeros are appended first and cut off again afterwards."""
    if pad == 0:
        return x
    length = x.shape[-1]
    extra = 0
    if length <= pad:
        extra = pad - length + 1
        x = F.pad(x, (0, extra))
    y = F.pad(x, (pad, 0), mode="reflect")
    return y[..., : y.shape[-1] - extra]


class EncodecDecoderOracle:
    def __init__(self, state_dict: Dict[str, np.ndarray]):
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state_dict.items()}

    def _conv(self, x, name, k):
        return F.conv1d(_pad_causal_reflect(x, k - 1), self.w[name + ".weight"], self.w[name + ".bias"])

    def _lstm(self, x):                                           # EncodecLSTM: (B,C,T) -> lstm over T, + skip
        h = x.permute(2, 0, 1)
        inp = h
        for l in range(2):
            wi, wh = self.w[f"decoder.1.lstm.weight_ih_l{l}"], self.w[f"decoder.1.lstm.weight_hh_l{l}"]
            bi, bh = self.w[f"decoder.1.lstm.bias_ih_l{l}"], self.w[f"decoder.1.lstm.bias_hh_l{l}"]
            B, C = inp.shape[1], inp.shape[2]
            hs = torch.zeros(B, C)
            cs = torch.zeros(B, C)
            outs = []
            for t in range(inp.shape[0]):
                g = F.linear(inp[t], wi, bi) + F.linear(hs, wh, bh)
                i, f, gg, o = g.chunk(4, dim=-1)                  # torch.nn.LSTM gate order
                cs = torch.sigmoid(f) * cs + torch.sigmoid(i) * torch.tanh(gg)
                hs = torch.sigmoid(o) * torch.tanh(cs)
                outs.append(hs)
            inp = torch.stack(outs)
        return (inp + h).permute(1, 2, 0)

    def decode(self, codes_bt8: np.ndarray) -> np.ndarray:
        """codes (B, T, 8) int -> audio (B, 320*T) fp32."""
        codes = torch.from_numpy(np.asarray(codes_bt8).astype(np.int64))
        B, T, _ = codes.shape
        emb = torch.full((), 0.0)
        for q in range(NQ):                                       # EncodecResidualVectorQuantizer.decode
            emb = emb + F.embedding(codes[:, :, q], self.w[f"quantizer.{q}.embed"]).permute(0, 2, 1)
        x = self._conv(emb, "decoder.0", 7)
        x = self._lstm(x)
        idx = 3
        for r in RATIOS:
            x = F.elu(x)
            x = F.conv_transpose1d(x, self.w[f"decoder.{idx}.weight"], self.w[f"decoder.{idx}.bias"], stride=r)
            x = x[..., : x.shape[-1] - r]                         # causal: trim k - stride = r on the right
            p = f"decoder.{idx + 1}."
            h = self._conv(F.elu(x), p + "block1", 3)
            h = self._conv(F.elu(h), p + "block3", 1)
            x = self._conv(x, p + "shortcut", 1) + h
            idx += 3
        x = self._conv(F.elu(x), "decoder.15", 7)
        return x[:, 0].numpy()


def _pad1d_reflect(x: torch.Tensor, left: int, right: int) -> torch.Tensor:
    """EncodecConv1d._pad1d(mode='reflect'): inputs not longer than the larger pad are zero-extended first, the
    extension is cut off again afterwards."""
    length = x.shape[-1]
    max_pad = max(left, right)
    extra = 0
    if length <= max_pad:
        extra = max_pad - length + 1
        x = F.pad(x, (0, extra))
    y = F.pad(x, (left, right), mode="reflect")
    return y[..., : y.shape[-1] - extra]


class EncodecEncoderOracle:
    """EnCodec 24 kHz SEANet ENCODER + residual vector quantiser at 6 kbps (8 codebooks): the prompt-enrolment path
    `tokenize_audio` -> `AudioTokenizer.encode` -> `codec.encode(wav)` (data/tokenizer.py:92-111, utils/prompt_making.py:57-84;
    SURVEY.md section 8f rank 3).  TEST INFRASTRUCTURE ONLY; restated from the pip package's algorithm and pinned against
    `transformers.EncodecModel.encode` (oracle/make_golden_encodec.py -> tests/golden/encodec_enc_*.npz).

    wav (B, L) -> Conv1d(1,32,k7) -> 4 x [ResnetBlock(C), ELU, Conv1d(C, 2C, k=2r, stride r)], r = 2,4,5,8 -> LSTM + skip -> ELU
    -> Conv1d(512,128,k7) -> (B,128,T), T = ceil(L/320); every conv is causal: left pad k - stride (reflect) plus the right
    pad that completes the last stride frame.  RVQ: for q in 0..7: idx_q = argmax_c -(|r|^2 - 2 r.e_c + |e_c|^2);
    r -= e[idx_q]."""

    def __init__(self, enc_state: Dict[str, np.ndarray], dec_state: Dict[str, np.ndarray]):
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in enc_state.items()}
        self.codebooks = [torch.from_numpy(np.ascontiguousarray(dec_state[f"quantizer.{q}.embed"])) for q in range(NQ)]

    def _conv(self, x, name, stride=1):
        w, b = self.w[name + ".weight"], self.w[name + ".bias"]
        k = w.shape[-1]
        pad_total = k - stride
        length = x.shape[-1]
        n_frames = math.ceil((length - k + pad_total) / stride + 1) - 1          # EncodecConv1d._get_extra_padding_for_conv1d
        extra = n_frames * stride + k - pad_total - length
        return F.conv1d(_pad1d_reflect(x, pad_total, extra), w, b, stride=stride)

    def _lstm(self, x):
        h = x.permute(2, 0, 1)
        inp = h
        for l in range(2):
            wi, wh = self.w[f"encoder.13.lstm.weight_ih_l{l}"], self.w[f"encoder.13.lstm.weight_hh_l{l}"]
            bi, bh = self.w[f"encoder.13.lstm.bias_ih_l{l}"], self.w[f"encoder.13.lstm.bias_hh_l{l}"]
            B, C = inp.shape[1], inp.shape[2]
            hs, cs, outs = torch.zeros(B, C), torch.zeros(B, C), []
            for t in range(inp.shape[0]):
                g = F.linear(inp[t], wi, bi) + F.linear(hs, wh, bh)
                i, f, gg, o = g.chunk(4, dim=-1)
                cs = torch.sigmoid(f) * cs + torch.sigmoid(i) * torch.tanh(gg)
                hs = torch.sigmoid(o) * torch.tanh(cs)
                outs.append(hs)
            inp = torch.stack(outs)
        return (inp + h).permute(1, 2, 0)

    def embeddings(self, wav: np.ndarray) -> torch.Tensor:
        """wav (B, L) fp32 -> encoder output (B, 128, T)."""
        x = torch.from_numpy(np.asarray(wav, np.float32))[:, None, :]
        x = self._conv(x, "encoder.0")
        idx = 1
        for r in reversed(RATIOS):
            p = f"encoder.{idx}."
            h = self._conv(F.elu(x), p + "block1")
            h = self._conv(F.elu(h), p + "block3")
            x = self._conv(x, p + "shortcut") + h
            x = self._conv(F.elu(x), f"encoder.{idx + 2}", stride=r)
            idx += 3
        x = self._lstm(x)
        return self._conv(F.elu(x), "encoder.15")

    def quantize(self, emb: torch.Tensor) -> np.ndarray:
        """(B, 128, T) -> codes (B, T, 8) int64 (EncodecResidualVectorQuantizer.encode at 6 kbps)."""
        residual = emb
        out = []
        for q in range(NQ):
            e = self.codebooks[q]
            hs = residual.permute(0, 2, 1).reshape(-1, HID)
            et = e.t()
            dist = -(hs.pow(2).sum(1, keepdim=True) - 2 * hs @ et + et.pow(2).sum(0, keepdim=True))
            ind = dist.max(dim=-1).indices.view(residual.shape[0], residual.shape[2])
            residual = residual - F.embedding(ind, e).permute(0, 2, 1)
            out.append(ind)
        return torch.stack(out, dim=-1).numpy()

    def encode(self, wav: np.ndarray) -> np.ndarray:
        return self.quantize(self.embeddings(wav))
