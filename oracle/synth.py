"""Seeded synthetic weights + inputs for the VALL-E X hot path.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): imported by tests/, bench.py's
cpu_baseline leg, __graft_entry__.smoke() and oracle/make_golden.py.  The
product path (vall-e-x_amd/) never imports this module.

Why synthetic: neither `vallex-checkpoint.pt` nor the Vocos weights exist
offline (reference downloads them at run time, utils/generation.py:53-65,89).
Shapes/keys follow the reference state-dict exactly (SURVEY.md §A.4;
models/vallex.py:55-264,405-445) so the same dict goes through
`VALLE.load_state_dict(strict=True)` in the reference *and* through
`vx_load_tensor` in the HIP engine.

Everything is generated with numpy PCG64 (platform independent), in a fixed
key order, so the GPU box regenerates bit-identical tensors without shipping
1.5 GB of fixtures.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

D_MODEL = 1024
N_HEAD = 16
D_FF = 4096
NUM_TEXT_TOKENS = 2048          # models/macros.py:2
NUM_AUDIO_TOKENS = 1024         # models/macros.py:5
NUM_QUANTIZERS = 8              # macros.py:5
EOS_ID = NUM_AUDIO_TOKENS       # models/vallex.py:573
BOS_ID = NUM_AUDIO_TOKENS + 1   # models/vallex.py:517

# Vocos charactr/vocos-encodec-24khz
VOCOS_DIM = 384
VOCOS_IDIM = 1152
VOCOS_LAYERS = 8
VOCOS_INCH = 128
VOCOS_NFFT = 1280
VOCOS_HOP = 320
VOCOS_NADA = 4
VOCOS_CODEBOOK_ROWS = 16384


def _uniform(rng, shape, bound):
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def _normal(rng, shape, std=1.0):
    return (rng.standard_normal(size=shape) * std).astype(np.float32)


def vallex_state_dict(num_layers: int = 12, seed: int = 0, eos_gain: float = 1.0,
                      attn_gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """The 374-key (for 12 layers) fp32 state-dict, key order = reference order.

    eos_gain > 1 scales the EOS row of ar_predict_layer so greedy decoding
    terminates naturally (random weights otherwise always run to the 16*S cap,
    SURVEY.md §8c).
    attn_gain > 1 scales the q and k rows of every in_proj (weights and biases): with the default init the attention
    scores have std ~0.5, i.e. a nearly uniform softmax that hides errors in K/V and in the score arithmetic; gain 3
    gives scores of std ~4.5 and peaky, trained-looking attention ("sharp" golden cases).  The RNG stream, and with it
    every other tensor, is the same for any gain.
    """
    rng = np.random.default_rng(seed)
    d, f = D_MODEL, D_FF
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    sd["ar_text_embedding.word_embeddings.weight"] = _normal(rng, (NUM_TEXT_TOKENS, d))
    sd["nar_text_embedding.word_embeddings.weight"] = _normal(rng, (NUM_TEXT_TOKENS, d))
    sd["ar_audio_embedding.word_embeddings.weight"] = _normal(rng, (NUM_AUDIO_TOKENS + 2, d))
    sd["ar_text_position.alpha"] = np.array([0.9], np.float32)
    sd["ar_audio_position.alpha"] = np.array([1.1], np.float32)

    def layer(prefix, adaptive):
        xav = math.sqrt(6.0 / (d + 3 * d))
        sd[prefix + "self_attn.in_proj_weight"] = _uniform(rng, (3 * d, d), xav)
        sd[prefix + "self_attn.in_proj_bias"] = _uniform(rng, (3 * d,), 0.02)
        if attn_gain != 1.0:
            sd[prefix + "self_attn.in_proj_weight"][: 2 * d] *= np.float32(attn_gain)
            sd[prefix + "self_attn.in_proj_bias"][: 2 * d] *= np.float32(attn_gain)
        sd[prefix + "self_attn.out_proj.weight"] = _uniform(rng, (d, d), 1 / math.sqrt(d))
        sd[prefix + "self_attn.out_proj.bias"] = _uniform(rng, (d,), 0.02)
        sd[prefix + "linear1.weight"] = _uniform(rng, (f, d), 1 / math.sqrt(d))
        sd[prefix + "linear1.bias"] = _uniform(rng, (f,), 1 / math.sqrt(d))
        sd[prefix + "linear2.weight"] = _uniform(rng, (d, f), 1 / math.sqrt(f))
        sd[prefix + "linear2.bias"] = _uniform(rng, (d,), 1 / math.sqrt(f))
        for n in ("norm1", "norm2"):
            if adaptive:
                sd[prefix + n + ".project_layer.weight"] = _uniform(rng, (2 * d, d), 1 / math.sqrt(d))
                # bias near (1, 0) so the adaptive scale is O(1) like a trained model
                b = _uniform(rng, (2 * d,), 0.05)
                b[:d] += 1.0
                sd[prefix + n + ".project_layer.bias"] = b
                sd[prefix + n + ".norm.weight"] = (1.0 + _uniform(rng, (d,), 0.1)).astype(np.float32)
                sd[prefix + n + ".norm.bias"] = _uniform(rng, (d,), 0.05)
            else:
                sd[prefix + n + ".weight"] = (1.0 + _uniform(rng, (d,), 0.1)).astype(np.float32)
                sd[prefix + n + ".bias"] = _uniform(rng, (d,), 0.05)

    for i in range(num_layers):
        layer(f"ar_decoder.layers.{i}.", False)
    sd["ar_decoder.norm.weight"] = (1.0 + _uniform(rng, (d,), 0.1)).astype(np.float32)
    sd["ar_decoder.norm.bias"] = _uniform(rng, (d,), 0.05)
    w = _uniform(rng, (NUM_AUDIO_TOKENS + 1, d), 1 / math.sqrt(d))
    w[EOS_ID] *= eos_gain
    sd["ar_predict_layer.weight"] = w
    sd["nar_audio_embeddings.0.word_embeddings.weight"] = _normal(rng, (NUM_AUDIO_TOKENS + 1, d))
    for j in range(1, NUM_QUANTIZERS):
        # rows double as predict-layer weights (tying, models/vallex.py:261-264):
        # keep them at Linear scale so NAR logits stay O(1).
        sd[f"nar_audio_embeddings.{j}.word_embeddings.weight"] = _normal(rng, (NUM_AUDIO_TOKENS, d), 0.5)
    sd["nar_text_position.alpha"] = np.array([1.0], np.float32)
    sd["nar_audio_position.alpha"] = np.array([1.0], np.float32)
    for i in range(num_layers):
        layer(f"nar_decoder.layers.{i}.", True)
    sd["nar_decoder.norm.project_layer.weight"] = _uniform(rng, (2 * d, d), 1 / math.sqrt(d))
    b = _uniform(rng, (2 * d,), 0.05)
    b[:d] += 1.0
    sd["nar_decoder.norm.project_layer.bias"] = b
    sd["nar_decoder.norm.norm.weight"] = (1.0 + _uniform(rng, (d,), 0.1)).astype(np.float32)
    sd["nar_decoder.norm.norm.bias"] = _uniform(rng, (d,), 0.05)
    for j in range(NUM_QUANTIZERS - 1):
        if j <= NUM_QUANTIZERS - 3:      # tied to nar_audio_embeddings[j+2]
            sd[f"nar_predict_layers.{j}.weight"] = sd[f"nar_audio_embeddings.{j + 2}.word_embeddings.weight"]
        else:
            sd[f"nar_predict_layers.{j}.weight"] = _normal(rng, (NUM_AUDIO_TOKENS, d), 0.5)
    for j in range(NUM_QUANTIZERS - 1):
        sd[f"nar_stage_embeddings.{j}.word_embeddings.weight"] = _normal(rng, (1, d))
    sd["ar_language_embedding.word_embeddings.weight"] = _normal(rng, (3, d))
    sd["nar_language_embedding.word_embeddings.weight"] = _normal(rng, (3, d))
    return sd


def _layer_prefixes(num_layers):
    return [f"{w}_decoder.layers.{i}." for w in ("ar", "nar") for i in range(num_layers)]


def trained_like_state_dict(num_layers: int = 12, seed: int = 0, logit_gain: float = 6.0,
                            emb_offset: float = 2.0, massive_bias: float = 4.0, ln_hi: float = 4.0,
                            eos_gain: float = 0.0) -> "OrderedDict[str, np.ndarray]":
    """A state-dict with the STATISTICS of a trained checkpoint (utils/generation.py:79-83 loads one; it is not available
    offline), derived from `vallex_state_dict(num_layers, seed, eos_gain=0)` by a second, independent RNG stream -- the base
    stream and with it every other fixture stays what it was.  What the default init lacks and this adds:
      * heavy-tailed weights: every projection matrix is multiplied element-wise by a log-normal factor (sigma 0.45, rms kept)
        and 1 element in 4096 by a further 8 (outlier weights);
      * LayerNorm gains ~ U(0.5, 4) and biases ~ U(-0.5, 0.5) on every norm (plain and adaptive): Q/K/V and the FFN inputs are
        2-3x larger, the attention scores ~6x (peaky softmax);
      * massive FFN channels: three hidden units per layer with 24x the input weights and a positive bias (mostly on,
        activations of a few hundred), read out with 0.25x weights;
      * massive residual dimensions: two embedding columns carry a constant offset in every token table;
      * decisive AR logits: ar_predict_layer x logit_gain (logit std ~9, median top-2 gap ~2 >> fp32 noise).
    (Stronger settings -- offset 12, bias 20, gain 16 -- collapse greedy decoding into 2-cycles of ~10 distinct ids; these keep
    the statistics and some variety.)
    The EOS row stays zero by default (eos_gain 0), so runs end at the forced step only; eos_gain > 0 gives the EOS logit the
    statistics of the other 1024 (sampled runs then end by themselves, at a different step per beam)."""
    sd = vallex_state_dict(num_layers, seed, eos_gain=eos_gain)
    rng = np.random.default_rng(770_000 + seed)
    d = D_MODEL
    keep_rms = np.float32(math.exp(-0.45 * 0.45))            # E[exp(2 * 0.45 z)] ** -0.5
    for p in _layer_prefixes(num_layers):
        for name in ("self_attn.in_proj_weight", "self_attn.out_proj.weight", "linear1.weight", "linear2.weight"):
            w = sd[p + name]
            w *= np.exp(0.45 * rng.standard_normal(w.shape)).astype(np.float32) * keep_rms
            out = rng.random(w.shape) < (1.0 / 4096.0)
            w[out] *= np.float32(8.0)
        ch = rng.choice(D_FF, size=3, replace=False)
        sd[p + "linear1.weight"][ch] *= np.float32(24.0)
        sd[p + "linear1.bias"][ch] = np.abs(sd[p + "linear1.bias"][ch]) * np.float32(24.0) + np.float32(massive_bias)
        sd[p + "linear2.weight"][:, ch] *= np.float32(0.25)
        adaptive = p.startswith("nar")
        for n in ("norm1", "norm2"):
            q = p + n + (".norm" if adaptive else "")
            sd[q + ".weight"][:] = rng.uniform(0.5, ln_hi, size=d).astype(np.float32)
            sd[q + ".bias"][:] = rng.uniform(-0.5, 0.5, size=d).astype(np.float32)
    for q in ("ar_decoder.norm", "nar_decoder.norm.norm"):
        sd[q + ".weight"][:] = rng.uniform(0.5, ln_hi, size=d).astype(np.float32)
        sd[q + ".bias"][:] = rng.uniform(-0.5, 0.5, size=d).astype(np.float32)
    cols = rng.choice(d, size=2, replace=False)
    seen = set()
    for k, v in sd.items():
        if "embedding" in k and "stage" not in k and v.ndim == 2 and v.shape[1] == d and id(v) not in seen:
            seen.add(id(v))                                   # tied tables (nar_predict_layers.j) are the same array object
            v[:, cols] += np.float32(emb_offset)
    sd["ar_predict_layer.weight"] *= np.float32(logit_gain)
    return sd


def out_of_range_state_dict(sd: "OrderedDict[str, np.ndarray]", num_layers: int, kind: str, gain_log2: int = 12,
                            stacks=("ar", "nar")) -> "OrderedDict[str, np.ndarray]":
    """The SAME function as `sd`, bit for bit in fp32, with operands that leave the fp16 range of the f16x2 kernels
    (|x| >= 2047): power-of-two rescalings that cancel exactly --
      'ffn': the first 64 hidden units of every layer: linear1 rows and biases x 2^g, the matching linear2 columns x 2^-g
             (relu(2^g a) 2^-g w == relu(a) w exactly: power-of-two scaling commutes with every fp32 rounding) -> hidden
             activations of ~5000 (modules/transformer.py:371-373 puts no bound on them);
      'v':   head 0: the V rows of in_proj (weights and biases) x 2^g, the out_proj columns of that head x 2^-g -> |v| ~ 3000;
      'k':   head 1: the K rows x 2^g, the Q rows x 2^-g (q.k unchanged) -> |k| ~ 3000.
    A correct engine therefore returns exactly the ids of the unscaled model; the reference does (oracle/make_golden.py
    checks it on the live reference)."""
    out = OrderedDict()
    copied = {}
    for k, v in sd.items():                                   # keep the tying (same array object under two keys)
        copied.setdefault(id(v), v.copy())
        out[k] = copied[id(v)]
    up, dn = np.float32(2.0 ** gain_log2), np.float32(2.0 ** -gain_log2)
    d = D_MODEL
    for p in [q for q in _layer_prefixes(num_layers) if q.split("_")[0] in stacks]:
        if kind == "ffn":
            out[p + "linear1.weight"][:64] *= up
            out[p + "linear1.bias"][:64] *= up
            out[p + "linear2.weight"][:, :64] *= dn
        elif kind == "v":
            out[p + "self_attn.in_proj_weight"][2 * d: 2 * d + 64] *= up
            out[p + "self_attn.in_proj_bias"][2 * d: 2 * d + 64] *= up
            out[p + "self_attn.out_proj.weight"][:, :64] *= dn
        elif kind == "k":
            out[p + "self_attn.in_proj_weight"][d + 64: d + 128] *= up
            out[p + "self_attn.in_proj_bias"][d + 64: d + 128] *= up
            out[p + "self_attn.in_proj_weight"][64:128] *= dn
            out[p + "self_attn.in_proj_bias"][64:128] *= dn
        else:
            raise ValueError(kind)
    return out


def outlier_state_dict(sd: "OrderedDict[str, np.ndarray]", num_layers: int, factor: float = 1000.0) -> "OrderedDict[str, np.ndarray]":
    """A DIFFERENT function than `sd` (its golden comes from the live reference run on these weights): ONE element of every
    transformer projection weight -- in_proj, out_proj, linear1, linear2 of both stacks -- is set to `factor` x the init bound
    of its tensor (31.25 for K = 1024, 15.6 for K = 4096 at factor 1000).  The f16x2 kernels scale each weight tensor by a
    power of two chosen from its own max |w|: with the maximum set by one outlier all the ordinary
    weights sit 10 bits lower in the fp16 head + tail pair -- the case where the format loses PRECISION (tails drifting towards
    fp16 subnormals) rather than range.  The range guard is not expected to fire."""
    out = OrderedDict()
    copied = {}
    for k, v in sd.items():                                   # keep the tying (same array object under two keys)
        copied.setdefault(id(v), v.copy())
        out[k] = copied[id(v)]
    for i, p in enumerate(_layer_prefixes(num_layers)):
        for j, name in enumerate(("self_attn.in_proj_weight", "self_attn.out_proj.weight", "linear1.weight", "linear2.weight")):
            w = out[p + name]
            n, k = (7 + 13 * i + 5 * j) % w.shape[0], (11 + 17 * i + 3 * j) % w.shape[1]
            w[n, k] = np.float32(factor / np.sqrt(w.shape[1])) * (np.float32(-1.0) if (i + j) & 1 else np.float32(1.0))
    return out


def vocos_state_dict(seed: int = 2) -> "OrderedDict[str, np.ndarray]":
    """Synthetic weights in the key layout of `charactr/vocos-encodec-24khz`
    (recalled from the pip `vocos` package, SURVEY.md §A.5 -- parity unpinned)."""
    rng = np.random.default_rng(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    c, ic, h = VOCOS_DIM, VOCOS_INCH, VOCOS_IDIM
    sd["feature_extractor.codebook_weights"] = _normal(rng, (VOCOS_CODEBOOK_ROWS, ic), 0.3)
    sd["backbone.embed.weight"] = _uniform(rng, (c, ic, 7), 1 / math.sqrt(ic * 7))
    sd["backbone.embed.bias"] = _uniform(rng, (c,), 1 / math.sqrt(ic * 7))
    sd["backbone.norm.scale.weight"] = (1.0 + _uniform(rng, (VOCOS_NADA, c), 0.1)).astype(np.float32)
    sd["backbone.norm.shift.weight"] = _uniform(rng, (VOCOS_NADA, c), 0.1)
    for i in range(VOCOS_LAYERS):
        p = f"backbone.convnext.{i}."
        sd[p + "dwconv.weight"] = _uniform(rng, (c, 1, 7), 1 / math.sqrt(7))
        sd[p + "dwconv.bias"] = _uniform(rng, (c,), 1 / math.sqrt(7))
        sd[p + "norm.scale.weight"] = (1.0 + _uniform(rng, (VOCOS_NADA, c), 0.1)).astype(np.float32)
        sd[p + "norm.shift.weight"] = _uniform(rng, (VOCOS_NADA, c), 0.1)
        sd[p + "pwconv1.weight"] = _uniform(rng, (h, c), 1 / math.sqrt(c))
        sd[p + "pwconv1.bias"] = _uniform(rng, (h,), 1 / math.sqrt(c))
        sd[p + "pwconv2.weight"] = _uniform(rng, (c, h), 1 / math.sqrt(h))
        sd[p + "pwconv2.bias"] = _uniform(rng, (c,), 1 / math.sqrt(h))
        sd[p + "gamma"] = _uniform(rng, (c,), 0.3)
    sd["backbone.final_layer_norm.weight"] = (1.0 + _uniform(rng, (c,), 0.1)).astype(np.float32)
    sd["backbone.final_layer_norm.bias"] = _uniform(rng, (c,), 0.05)
    # keep log-magnitudes moderate so exp() stays well inside the clip at 100
    sd["head.out.weight"] = _uniform(rng, (VOCOS_NFFT + 2, c), 0.5 / math.sqrt(c))
    sd["head.out.bias"] = _uniform(rng, (VOCOS_NFFT + 2,), 0.1)
    return sd


def synth_text(n: int, seed: int) -> np.ndarray:
    """n phoneme ids in 5..69 (bpe_69.json symbol range)."""
    return np.random.default_rng(1000 + seed).integers(5, 70, size=(n,), dtype=np.int64)


def synth_prompt(tp: int, sp: int, seed: int):
    """(audio_tokens (1,tp,8) int64 in 0..1023, text_tokens (1,sp) int64 in 5..69)."""
    rng = np.random.default_rng(2000 + seed)
    a = rng.integers(0, NUM_AUDIO_TOKENS, size=(1, tp, NUM_QUANTIZERS), dtype=np.int64)
    t = rng.integers(5, 70, size=(1, sp), dtype=np.int64)
    return a, t


def uniforms(n_steps: int, batch: int, seed: int = 1234) -> np.ndarray:
    """Injected sampling uniforms in [0,1), shape (n_steps, batch) float32."""
    return np.random.default_rng(seed).random(size=(n_steps, batch), dtype=np.float32)
