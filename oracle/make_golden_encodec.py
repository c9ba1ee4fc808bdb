"""Pin oracle/encodec_oracle.py against the installed port of EnCodec (`transformers.EncodecModel`, build container only).

    python -m oracle.make_golden_encodec

Loads oracle.encodec_oracle.encodec_state_dict() into a default-config EncodecModel (= encodec_24khz: hidden 128, filters
32, ratios 8/5/4/2, k7, 2-layer LSTM, causal, weight-norm, 1024x128 codebooks) -- weight-norm is set with g = ||v|| so the
effective weight equals the folded one -- and stores `EncodecModel.decode` outputs for a few code tensors in
tests/golden/encodec_*.npz.
"""
import os

import numpy as np
import torch

from .encodec_oracle import (NQ, RATIOS, EncodecDecoderOracle, EncodecEncoderOracle, encodec_encoder_state_dict,
                             encodec_state_dict)

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
CASES = {"encodec_T37": (2, 37, 11), "encodec_T5": (1, 5, 12)}       # name -> (B, T, seed); T=5 exercises the short-pad path


def case_codes(name):
    B, T, seed = CASES[name]
    return np.random.default_rng(seed).integers(0, 1024, size=(B, T, NQ), dtype=np.int64)


# encoder cases: name -> (B, samples, seed); 7777 is not a multiple of the 320-sample hop (exercises the right padding)
ENC_CASES = {"encodec_enc_12000": (2, 12000, 21), "encodec_enc_7777": (1, 7777, 22)}


def case_wav(name):
    B, L, seed = ENC_CASES[name]
    rng = np.random.default_rng(seed)
    t = np.arange(L, dtype=np.float32) / 24000.0
    f0 = rng.uniform(90.0, 400.0, size=(B, 1)).astype(np.float32)
    wav = 0.3 * np.sin(2 * np.pi * f0 * t[None]) + 0.1 * rng.standard_normal((B, L)).astype(np.float32)
    return wav.astype(np.float32)


def load_into_transformers(sd, enc=None):
    from transformers import EncodecConfig, EncodecModel
    m = EncodecModel(EncodecConfig()).eval()
    tsd = m.state_dict()

    def put_wn(prefix, w):
        v = torch.from_numpy(w)
        tsd[prefix + ".parametrizations.weight.original1"] = v.clone()
        tsd[prefix + ".parametrizations.weight.original0"] = v.flatten(1).norm(dim=1).view(-1, 1, 1)   # g = ||v|| -> w = v

    for q in range(NQ):
        tsd[f"quantizer.layers.{q}.codebook.embed"] = torch.from_numpy(sd[f"quantizer.{q}.embed"])
    put_wn("decoder.layers.0.conv", sd["decoder.0.weight"])
    tsd["decoder.layers.0.conv.bias"] = torch.from_numpy(sd["decoder.0.bias"])
    for l in range(2):
        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            tsd[f"decoder.layers.1.lstm.{n}_l{l}"] = torch.from_numpy(sd[f"decoder.1.lstm.{n}_l{l}"])
    idx = 3
    for _ in RATIOS:
        put_wn(f"decoder.layers.{idx}.conv", sd[f"decoder.{idx}.weight"])
        tsd[f"decoder.layers.{idx}.conv.bias"] = torch.from_numpy(sd[f"decoder.{idx}.bias"])
        for mine, theirs in (("block1", "block.1"), ("block3", "block.3"), ("shortcut", "shortcut")):
            put_wn(f"decoder.layers.{idx + 1}.{theirs}.conv", sd[f"decoder.{idx + 1}.{mine}.weight"])
            tsd[f"decoder.layers.{idx + 1}.{theirs}.conv.bias"] = torch.from_numpy(sd[f"decoder.{idx + 1}.{mine}.bias"])
        idx += 3
    put_wn("decoder.layers.15.conv", sd["decoder.15.weight"])
    tsd["decoder.layers.15.conv.bias"] = torch.from_numpy(sd["decoder.15.bias"])
    if enc is not None:
        for i in (0, 3, 6, 9, 12, 15):
            put_wn(f"encoder.layers.{i}.conv", enc[f"encoder.{i}.weight"])
            tsd[f"encoder.layers.{i}.conv.bias"] = torch.from_numpy(enc[f"encoder.{i}.bias"])
        for i in (1, 4, 7, 10):
            for mine, theirs in (("block1", "block.1"), ("block3", "block.3"), ("shortcut", "shortcut")):
                put_wn(f"encoder.layers.{i}.{theirs}.conv", enc[f"encoder.{i}.{mine}.weight"])
                tsd[f"encoder.layers.{i}.{theirs}.conv.bias"] = torch.from_numpy(enc[f"encoder.{i}.{mine}.bias"])
        for l in range(2):
            for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                tsd[f"encoder.layers.13.lstm.{n}_l{l}"] = torch.from_numpy(enc[f"encoder.13.lstm.{n}_l{l}"])
    m.load_state_dict(tsd, strict=True)
    return m


def main():
    sd = encodec_state_dict(3)
    m = load_into_transformers(sd)
    orc = EncodecDecoderOracle(sd)
    for name in CASES:
        codes = case_codes(name)
        with torch.no_grad():
            ref = m.decode(torch.from_numpy(codes).permute(0, 2, 1)[None], [None])[0][:, 0].numpy()
        mine = orc.decode(codes)
        print(name, ref.shape, float(np.abs(ref).max()), "oracle max |err|", float(np.abs(ref - mine).max()))
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), audio=ref.astype(np.float32))


def main_encoder():
    sd, enc = encodec_state_dict(3), encodec_encoder_state_dict(4)
    m = load_into_transformers(sd, enc)
    orc = EncodecEncoderOracle(enc, sd)
    for name in ENC_CASES:
        wav = case_wav(name)
        with torch.no_grad():
            ref = m.encode(torch.from_numpy(wav)[:, None, :], bandwidth=6.0)[0][0].permute(0, 2, 1).numpy()   # (B, T, 8)
            emb = m.encoder(torch.from_numpy(wav)[:, None, :]).numpy()
        mine = orc.encode(wav)
        print(name, ref.shape, "codes equal:", bool((ref == mine).all()), "embedding max |err|",
              float(np.abs(emb - orc.embeddings(wav).numpy()).max()))
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), codes=ref.astype(np.int64), emb=emb[:, :, :4].astype(np.float32))


if __name__ == "__main__":
    main_encoder()
    main()
