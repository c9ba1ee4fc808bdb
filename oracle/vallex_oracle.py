"""CPU oracle: a functional restatement of the VALL-E X inference hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this file; the product (vall-e-x_amd/) never
does, and fails loudly if its HIP library is missing.

It restates, op by op on torch-CPU fp32 tensors, what the reference computes on
the path BASELINE.json names (all citations relative to /root/reference):

  * `VALLE.inference`                      models/vallex.py:458-686
  * `TransformerEncoder(.infer)/Layer`     modules/transformer.py:265-373,402-473
  * `LayerNorm` / `AdaptiveLayerNorm`      modules/transformer.py:57-74,93-108
  * `multi_head_attention_forward`         modules/activation.py:114-167
  * `TokenEmbedding` / `SinePositionalEmbedding`  modules/embedding.py:43-47,68-97
  * `topk_sampling` / `top_k_top_p_filtering`     models/vallex.py:791-853
  * Vocos `codes_to_features` + `decode`   pip `vocos` (unpinned, requirements.txt:23;
    call sites utils/generation.py:148-150) -- arithmetic recalled from the
    package, NOT present in /root/reference.

Pinning status
  * AR + NAR: pinned against the live reference run in the build container
    (oracle/make_golden.py imports /root/reference, loads the same synthetic
    state-dict with strict=True, and commits tokens/logits to tests/golden/).
    The reference itself ships no golden vectors or tests for this path
    (SURVEY.md §4), so the live reference IS the pin.
  * Vocos head: **parity unpinned to the pip package** -- the `vocos`/`encodec`
    packages and their weights are absent offline.  Partial pins that do exist
    (tests/test_vocos_oracle_pieces.py): `VocosOracle.head` == the installed
    `transformers` port of the package's head code (`Xcodec2ISTFTHead`: Linear ->
    exp/clip -> polar -> vocos.spectral_ops.ISTFT(padding="same"), every sample,
    edges included); the backbone and `codes_to_features` == the same weights in
    torch.nn modules wired as recalled; the residual branch of
    a ConvNeXt block == transformers' `Qwen3OmniMoeConvNeXtBlock` (a third-party
    1-D ConvNeXt block of the same lineage).  Still recalled only: the order
    embed conv -> AdaLayerNorm -> blocks -> final LayerNorm, and the block norm
    being the id-conditioned AdaLayerNorm.

The oracle keeps a real KV cache and only embeds the newest token per step; the
reference re-embeds all of `y` and rebuilds the mask every step
(models/vallex.py:529-549) but consumes only the last row, so results agree
(checked by tests/test_oracle_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch
import torch.nn.functional as F

from . import synth

LANG_ID = {"en": 0, "zh": 1, "ja": 2}          # models/vallex.py:439-443
LN_EPS = 1e-5                                   # modules/transformer.py:197


def sine_pe(n: int, d: int = synth.D_MODEL) -> torch.Tensor:
    """modules/embedding.py:75-91 -- exactly the reference's host construction."""
    pe = torch.zeros(n, d)
    position = torch.arange(0, n, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def inverse_cdf_sample(probs: torch.Tensor, u: float) -> int:
    """Deterministic stand-in for torch.multinomial(probs, 1) (models/vallex.py:850):
    first index whose running fp32 sum exceeds u * total.  Used with injected
    uniforms so sampling is reproducible on any device."""
    p = probs.to(torch.float32).reshape(-1)
    c = torch.cumsum(p, 0)
    t = np.float32(u) * c[-1].item()
    idx = int(torch.searchsorted(c, torch.tensor(t, dtype=torch.float32), right=True).item())
    nz = torch.nonzero(p > 0).reshape(-1)
    return min(max(idx, int(nz[0])), int(nz[-1]))


class VallexOracle:
    def __init__(self, state_dict: Dict[str, np.ndarray], num_layers: int):
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state_dict.items()}
        self.nl = num_layers
        self.d = synth.D_MODEL
        self.h = synth.N_HEAD
        self.pe = sine_pe(4000)
        # optional operand statistics (a dict): running max |.| of what the f16x2 kernels of the HIP engine see as GEMM / attention
        # operands on the FULL-SEQUENCE paths -- "ln" (LayerNorm outputs), "q8" (q / 8), "k", "v", "att" (attention output),
        # "ffn" (ReLU'd hidden activations).  Their fp16 range at the activation scale 2^5 ends at 2047.
        self.stats = None

    def _stat(self, key, t):
        if self.stats is not None and t.shape[0] > 1:                  # multi-row calls only: the cached decode step is exact fp32
            self.stats[key] = max(self.stats.get(key, 0.0), float(t.abs().max()))

    # ---- building blocks -------------------------------------------------
    def _pe(self, n):
        if self.pe.shape[0] < n:                       # modules/embedding.py:68-74
            self.pe = sine_pe(n)
        return self.pe

    def _ln(self, x, prefix):                          # modules/transformer.py:57-74
        return F.layer_norm(x, (self.d,), self.w[prefix + ".weight"], self.w[prefix + ".bias"], LN_EPS)

    def _adaln(self, x, prefix, stage_emb):            # modules/transformer.py:93-108
        wb = F.linear(stage_emb, self.w[prefix + ".project_layer.weight"], self.w[prefix + ".project_layer.bias"])
        weight, bias = torch.split(wb, self.d, dim=-1)
        return weight * self._ln(x, prefix + ".norm") + bias

    def _mha(self, x, prefix, mask: Optional[torch.Tensor], past=None):
        """modules/activation.py:142-167.  x (T,d); mask bool (T,ctx) True=hidden; past (k,v) (H,P,64)."""
        T = x.shape[0]
        qkv = F.linear(x, self.w[prefix + ".in_proj_weight"], self.w[prefix + ".in_proj_bias"])
        q, k, v = qkv.chunk(3, dim=-1)
        self._stat("ln", x); self._stat("q8", q * 0.125); self._stat("k", k); self._stat("v", v)
        hd = self.d // self.h
        q = q.view(T, self.h, hd).transpose(0, 1)
        k = k.view(T, self.h, hd).transpose(0, 1)
        v = v.view(T, self.h, hd).transpose(0, 1)
        if past is not None:
            k = torch.cat((past[0], k), dim=-2)
            v = torch.cat((past[1], v), dim=-2)
        att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hd))
        if mask is not None:
            att = att.masked_fill(mask, float("-inf"))
        att = F.softmax(att, dim=-1)
        y = (att @ v).transpose(0, 1).contiguous().view(T, self.d)
        self._stat("att", y)
        y = F.linear(y, self.w[prefix + ".out_proj.weight"], self.w[prefix + ".out_proj.bias"])
        return y, (k, v)

    def _ffn(self, x, prefix):                         # modules/transformer.py:371-373 (ReLU :187)
        h = F.relu(F.linear(x, self.w[prefix + "linear1.weight"], self.w[prefix + "linear1.bias"]))
        self._stat("ln", x); self._stat("ffn", h)
        return F.linear(h, self.w[prefix + "linear2.weight"], self.w[prefix + "linear2.bias"])

    def _ar_stack(self, x, mask, past, taps=None):
        """TransformerEncoder.infer modules/transformer.py:447-473, layer :337-347."""
        new = []
        for i in range(self.nl):
            p = f"ar_decoder.layers.{i}."
            a, kv = self._mha(self._ln(x, p + "norm1"), p + "self_attn", mask, None if past is None else past[i])
            x = x + a
            x = x + self._ffn(self._ln(x, p + "norm2"), p)
            new.append(kv)
            if taps is not None:
                taps.setdefault("ar_layer_out", []).append(x.clone())
        return self._ln(x, "ar_decoder.norm"), new

    def _nar_stack(self, x, stage_emb, taps=None):
        """TransformerEncoder.forward modules/transformer.py:436-445, layer :296-302 (no mask)."""
        for i in range(self.nl):
            p = f"nar_decoder.layers.{i}."
            a, _ = self._mha(self._adaln(x, p + "norm1", stage_emb), p + "self_attn", None)
            x = x + a
            x = x + self._ffn(self._adaln(x, p + "norm2", stage_emb), p)
            if taps is not None:
                taps.setdefault("nar_layer_out", []).append(x.clone())
        return self._adaln(x, "nar_decoder.norm", stage_emb)

    def _text_embed(self, which, text, enroll, prompt_language, text_language):
        """models/vallex.py:497-507 (AR) / :622-632 (NAR).  prompt_language = text_language = None: no language embedding
        (VALLE.continual, models/vallex.py:716-719, 727-729)."""
        x = self.w[f"{which}_text_embedding.word_embeddings.weight"][text].clone()
        if prompt_language is not None or text_language is not None:
            lang = self.w[f"{which}_language_embedding.word_embeddings.weight"]
            x[:enroll] += lang[LANG_ID[prompt_language]]
            if isinstance(text_language, str):
                x[enroll:] += lang[LANG_ID[text_language]]
            else:
                ids = torch.tensor([LANG_ID[t] for t in text_language], dtype=torch.long)
                x[enroll:] += lang[ids]
        alpha = self.w[f"{which}_text_position.alpha"]
        return x * 1.0 + alpha * self._pe(x.shape[0])[: x.shape[0]]

    def text_language_ids(self, S, enroll, prompt_language, text_language) -> np.ndarray:
        """Per-token model language ids (what the C-ABI batch descriptor carries)."""
        out = np.empty(S, np.int32)
        out[:enroll] = LANG_ID[prompt_language]
        if isinstance(text_language, str):
            out[enroll:] = LANG_ID[text_language]
        else:
            out[enroll:] = [LANG_ID[t] for t in text_language]
        return out

    # ---- sampling (models/vallex.py:791-853) ------------------------------
    @staticmethod
    def sample(logits: torch.Tensor, top_k: int, temperature: float, u: Optional[float]):
        logits = logits.clone().reshape(1, -1)
        if temperature != 1.0:
            logits = logits / temperature
        if top_k > 0:
            k = min(max(top_k, 1), logits.size(-1))
            thr = torch.topk(logits, k)[0][..., -1, None]
            logits[logits < thr] = -float("inf")          # ties kept (:808)
        probs = F.softmax(logits, dim=-1)
        if u is None:
            nz = torch.nonzero(probs.reshape(-1) > 0).reshape(-1)
            if nz.numel() == 1:
                tok = int(nz[0])
            else:
                tok = int(torch.multinomial(probs, 1).item())
        else:
            tok = inverse_cdf_sample(probs, u)
        logp = F.log_softmax(logits.float(), dim=-1)[0, tok].item()
        return tok, logp

    # ---- AR ---------------------------------------------------------------
    def ar_prefill(self, text, prompt_codes0, enroll, prompt_language, text_language, taps=None):
        """First `ar_decoder.infer` call (models/vallex.py:528-562): returns
        (last-row hidden after final LN, kv list, S)."""
        x = self._text_embed("ar", text, enroll, prompt_language, text_language)
        S = x.shape[0]
        y = torch.cat([torch.tensor([synth.BOS_ID]), prompt_codes0])          # :515-517
        y_emb = self.w["ar_audio_embedding.word_embeddings.weight"][y]
        y_pos = y_emb + self.w["ar_audio_position.alpha"] * self._pe(len(y))[: len(y)]
        xy = torch.cat([x, y_pos], 0)
        L = xy.shape[0]
        mask = torch.zeros(L, L, dtype=torch.bool)                            # :535-549
        mask[:S, S:] = True
        mask[S:, S:] = torch.triu(torch.ones(L - S, L - S, dtype=torch.bool), diagonal=1)
        if taps is not None:
            taps["ar_prefill_in"] = xy.clone()
        out, kv = self._ar_stack(xy, mask, None, taps)
        if taps is not None:
            taps["ar_prefill_out"] = out.clone()
        return out[-1], kv, S

    def ar_step(self, token: int, pos: int, kv):
        """One cached decode step: embed newest token at audio position `pos`
        (BOS is position 0), run the 12 layers on one row."""
        e = self.w["ar_audio_embedding.word_embeddings.weight"][token]
        xy = (e + self.w["ar_audio_position.alpha"] * self._pe(pos + 1)[pos]).unsqueeze(0)
        out, kv = self._ar_stack(xy, None, kv)
        return out[-1], kv

    def ar_logits(self, hidden):
        return F.linear(hidden, self.w["ar_predict_layer.weight"])           # :568

    def ar_generate(self, text, prompt_codes0, enroll, prompt_language, text_language, top_k=-100,
                    temperature=1.0, uniforms: Optional[Sequence[float]] = None,
                    force_eos_at: Optional[int] = None, taps=None, return_logp: bool = False):
        """models/vallex.py:528-598 for best_of=1.  `force_eos_at=n` forces EOS as the
        (n+1)-th sample (the reference-side equivalent is the topk_sampling hook in
        oracle/make_golden.py) so synthetic weights can emulate an 8 s utterance."""
        h, kv, S = self.ar_prefill(text, prompt_codes0, enroll, prompt_language, text_language, taps)
        Tp = len(prompt_codes0)
        y_len = Tp + 1
        gen: List[int] = []
        step = 0
        sum_logp = np.float32(0.0)
        while True:
            logits = self.ar_logits(h)
            if taps is not None:
                taps.setdefault("ar_logits", []).append(logits.clone())
            u = None if uniforms is None else float(uniforms[step])
            tok, lp = self.sample(logits, top_k, temperature, u)
            sum_logp = np.float32(sum_logp + np.float32(lp))                 # :572 (beam not finished yet)
            if force_eos_at is not None and step >= force_eos_at:
                tok = synth.EOS_ID
            if tok == synth.EOS_ID or (y_len - Tp) > S * 16:               # :575-578
                # :579-582 raises SyntaxError only if prompts.shape[1] == y.shape[1]; with
                # prepend_bos=True (utils/generation.py:76) y always has the extra BOS, so
                # the reference returns an EMPTY (1,0,8) result instead of raising.
                break
            gen.append(tok)
            y_len += 1
            h, kv = self.ar_step(tok, y_len - 1, kv)
            step += 1
        return (gen, float(sum_logp)) if return_logp else gen

    # ---- NAR (models/vallex.py:600-686, prefix_mode 1) ----------------------
    def nar_generate(self, text, prompts, codes0: Sequence[int], enroll, prompt_language, text_language,
                     taps=None, stage_logits: Optional[list] = None) -> np.ndarray:
        Tp = prompts.shape[0]
        T = len(codes0)
        y = torch.cat([prompts[:, 0], torch.tensor(list(codes0), dtype=torch.long)])
        emb = [self.w[f"nar_audio_embeddings.{j}.word_embeddings.weight"] for j in range(synth.NUM_QUANTIZERS)]
        y_emb = emb[0][y].clone()                                            # :605-607
        x = self._text_embed("nar", text, enroll, prompt_language, text_language)
        S = x.shape[0]
        for j in range(1, synth.NUM_QUANTIZERS):                             # :659-662
            y_emb[:Tp] += emb[j][prompts[:, j]]
        codes = [torch.tensor(list(codes0), dtype=torch.long)]
        alpha = self.w["nar_audio_position.alpha"]
        for i in range(synth.NUM_QUANTIZERS - 1):                            # :664-683
            y_pos = y_emb + alpha * self._pe(Tp + T)[: Tp + T]
            xy = torch.cat([x, y_pos], 0)
            stage = self.w[f"nar_stage_embeddings.{i}.word_embeddings.weight"]
            dec = self._nar_stack(xy, stage, taps if i == 0 else None)
            logits = F.linear(dec[S + Tp:], self.w[f"nar_predict_layers.{i}.weight"])
            if taps is not None:
                taps.setdefault("nar_logits", []).append(logits.clone())
            if stage_logits is not None:                                     # bench.py's parity block: the decision margins
                stage_logits.append(logits)
            samples = torch.argmax(logits, dim=-1)
            codes.append(samples)
            if i < synth.NUM_QUANTIZERS - 2:
                y_emb[Tp:] += emb[i + 1][samples]
        return torch.stack(codes, dim=-1).numpy()                            # (T, 8)

    # ---- VALLE.inference (models/vallex.py:458-686) -------------------------
    def inference(self, x, x_lens, y, enroll_x_lens, top_k=-100, temperature=1.0, prompt_language=None,
                  text_language=None, uniforms=None, force_eos_at=None, taps=None, best_of=1, length_penalty=1.0,
                  return_worst=False) -> np.ndarray:
        x = np.asarray(x); y = np.asarray(y)
        assert x.ndim == 2 and y.ndim == 3 and y.shape[0] == 1            # :488-493
        text = torch.from_numpy(x[0].astype(np.int64))
        prompts = torch.from_numpy(y[0].astype(np.int64))
        if best_of > 1:
            # beams never interact (models/vallex.py:525-598 runs them as batch rows): N independent samplings with
            # their own uniform column, then the :583-594 selection on sum(logp) / len^penalty, len = #non-EOS of y
            us = np.asarray(uniforms, np.float32).reshape(-1, best_of)
            beams = [self.ar_generate(text, prompts[:, 0], int(enroll_x_lens), prompt_language, text_language, top_k,
                                      temperature, us[:, i], force_eos_at, None, return_logp=True) for i in range(best_of)]
            lengths = torch.tensor([1 + prompts.shape[0] + len(g) for g, _ in beams])
            avg = torch.tensor([lp for _, lp in beams], dtype=torch.float32) / lengths ** length_penalty
            pick = int(torch.argmin(avg)) if return_worst else int(torch.argmax(avg))
            gen = beams[pick][0]
            if taps is not None:
                taps["beam_avg_logprobs"] = avg.numpy()
        else:
            gen = self.ar_generate(text, prompts[:, 0], int(enroll_x_lens), prompt_language, text_language,
                                   top_k, temperature, uniforms, force_eos_at, taps)
        codes = self.nar_generate(text, prompts, gen, int(enroll_x_lens), prompt_language, text_language, taps)
        return codes[None]                                                   # (1, T, 8) int64

    # ---- VALLE.continual (models/vallex.py:688-787, prefix_mode 1 branch :760-784) -------------------------
    def continual(self, x, x_lens, y) -> np.ndarray:
        """NAR-only continuation: the first half of `y` (at most 3 s = 225 frames) is the acoustic prompt, the first
        codebook of the rest is taken as given, codebooks 2..8 of the rest are predicted.  No language embedding is
        added to the text (unlike `inference`).  Returns (1, T - prefix_len, 8)."""
        x = np.asarray(x); y = np.asarray(y)
        assert x.ndim == 2 and y.ndim == 3 and y.shape[0] == 1            # :706-709
        assert np.all(np.asarray(x_lens) > 0)
        text = torch.from_numpy(x[0].astype(np.int64))
        yy = torch.from_numpy(y[0].astype(np.int64))
        prefix_len = min(int(yy.shape[0] * 0.5), 3 * 75)                  # :722
        codes = self.nar_generate(text, yy[:prefix_len], yy[prefix_len:, 0].tolist(), 0, None, None)
        return codes[None]


# ---------------------------------------------------------------------------
# Vocos head (pip `vocos`) -- parity unpinned to the package, partial pins in the module doc.
# ---------------------------------------------------------------------------
class VocosOracle:
    def __init__(self, state_dict: Dict[str, np.ndarray]):
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state_dict.items()}

    def codes_to_features(self, codes: torch.Tensor) -> torch.Tensor:
        """codes (8,B,T) int -> (B,128,T).  vocos/pretrained.py codes_to_features."""
        nq = codes.shape[0]
        offsets = torch.arange(0, 1024 * nq, 1024)
        idx = codes.long() + offsets.view(-1, 1, 1)
        feat = F.embedding(idx, self.w["feature_extractor.codebook_weights"]).sum(dim=0)
        return feat.transpose(1, 2)

    def _adaln(self, x, prefix, bid):
        c = synth.VOCOS_DIM
        x = F.layer_norm(x, (c,), eps=1e-6)
        return x * self.w[prefix + ".scale.weight"][bid] + self.w[prefix + ".shift.weight"][bid]

    def convnext_branch(self, x: torch.Tensor, i: int, bid: int) -> torch.Tensor:
        """the residual branch of ConvNeXt block i on x (B, C, T): depthwise conv k7 (centred) -> AdaLayerNorm -> Linear -> GELU (erf)
        -> Linear -> gamma.  Operation order pinned against a third-party 1-D ConvNeXt block (tests/test_vocos_oracle_pieces.py)."""
        p = f"backbone.convnext.{i}."
        c = x.shape[1]
        y = F.conv1d(x, self.w[p + "dwconv.weight"], self.w[p + "dwconv.bias"], padding=3, groups=c)
        y = self._adaln(y.transpose(1, 2), p + "norm", bid)
        y = F.linear(y, self.w[p + "pwconv1.weight"], self.w[p + "pwconv1.bias"])
        y = F.gelu(y)
        y = F.linear(y, self.w[p + "pwconv2.weight"], self.w[p + "pwconv2.bias"])
        return (self.w[p + "gamma"] * y).transpose(1, 2)

    def backbone(self, feat: torch.Tensor, bid: int) -> torch.Tensor:
        c = synth.VOCOS_DIM
        x = F.conv1d(feat, self.w["backbone.embed.weight"], self.w["backbone.embed.bias"], padding=3)
        x = self._adaln(x.transpose(1, 2), "backbone.norm", bid).transpose(1, 2)
        for i in range(synth.VOCOS_LAYERS):
            x = x + self.convnext_branch(x, i, bid)
        return F.layer_norm(x.transpose(1, 2), (c,), self.w["backbone.final_layer_norm.weight"],
                            self.w["backbone.final_layer_norm.bias"], 1e-6)

    def head(self, x: torch.Tensor) -> torch.Tensor:
        n_fft, hop = synth.VOCOS_NFFT, synth.VOCOS_HOP
        o = F.linear(x, self.w["head.out.weight"], self.w["head.out.bias"]).transpose(1, 2)
        mag, p = o.chunk(2, dim=1)
        mag = torch.clip(torch.exp(mag), max=1e2)
        S = mag * (torch.cos(p) + 1j * torch.sin(p))
        pad = (n_fft - hop) // 2
        window = torch.hann_window(n_fft)
        B, _, T = S.shape
        ifft = torch.fft.irfft(S, n_fft, dim=1, norm="backward") * window[None, :, None]
        out_size = (T - 1) * hop + n_fft
        yy = F.fold(ifft, output_size=(1, out_size), kernel_size=(1, n_fft), stride=(1, hop))[:, 0, 0, pad:-pad]
        wsq = window.square().expand(1, T, -1).transpose(1, 2)
        env = F.fold(wsq, output_size=(1, out_size), kernel_size=(1, n_fft), stride=(1, hop)).squeeze()[pad:-pad]
        return yy / env

    def decode_codes(self, codes_t8: np.ndarray, bandwidth_id: int = 2) -> np.ndarray:
        """codes (B,T,8) -> audio (B, 320*T) fp32.  utils/generation.py:148-152."""
        frames = torch.from_numpy(np.asarray(codes_t8).astype(np.int64)).permute(2, 0, 1)
        feat = self.codes_to_features(frames)
        x = self.backbone(feat, bandwidth_id)
        return self.head(x).numpy()
