"""CPU: host logic of the reference-API mirror (signatures, wire formats, error behaviour) -- everything that must
happen before the first GPU call."""
import inspect
import os

import numpy as np
import pytest

import vallex_amd
from oracle import synth
from oracle.vallex_oracle import sine_pe
from vallex_amd.models.vallex import VALLE, expected_keys, sine_pe_table, vocos_expected_keys
from vallex_amd.utils import generation as G

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _valle(nl=2):
    return VALLE(1024, 16, nl, norm_first=True, add_prenet=False, prefix_mode=1, share_embedding=True,
                 nar_scale_factor=1.0, prepend_bos=True, num_quantizers=8)


def test_signatures_match_reference():
    # models/vallex.py:458-471
    p = list(inspect.signature(VALLE.inference).parameters)
    assert p[:12] == ["self", "x", "x_lens", "y", "enroll_x_lens", "top_k", "temperature", "prompt_language",
                      "text_language", "best_of", "length_penalty", "return_worst"]
    d = inspect.signature(VALLE.inference).parameters
    assert d["top_k"].default == -100 and d["temperature"].default == 1.0 and d["best_of"].default == 1
    # utils/generation.py:92,155
    assert list(inspect.signature(G.generate_audio).parameters)[:4] == ["text", "prompt", "language", "accent"]
    g = inspect.signature(G.generate_audio_from_long_text).parameters
    assert list(g)[:5] == ["text", "prompt", "language", "accent", "mode"] and g["mode"].default == "sliding-window"
    assert G.SAMPLE_RATE == vallex_amd.SAMPLE_RATE == 24000


def test_state_dict_wire_format_strict():
    assert len(expected_keys(12)) == 374
    sd = synth.vallex_state_dict(2, 0)
    m = _valle(2)
    m.load_state_dict(sd, strict=True)
    bad = dict(sd)
    bad.pop("ar_predict_layer.weight")
    with pytest.raises(RuntimeError, match="missing"):
        _valle(2).load_state_dict(bad, strict=True)
    bad = dict(sd)
    bad["extra.weight"] = np.zeros(1, np.float32)
    with pytest.raises(RuntimeError, match="unexpected"):
        _valle(2).load_state_dict(bad, strict=True)
    with pytest.raises(NotImplementedError):
        VALLE(512, 8, 2, prefix_mode=1)


def test_inference_asserts_like_reference():
    m = _valle(2)
    x = np.ones((1, 5), np.int32)
    y = np.zeros((1, 4, 8), np.int32)
    with pytest.raises(AssertionError):
        m.inference(x[0], np.array([5]), y, 2, prompt_language="en", text_language="en")        # x.ndim
    with pytest.raises(AssertionError):
        m.inference(x, np.array([5]), y[0], 2, prompt_language="en", text_language="en")        # y.ndim
    with pytest.raises(AssertionError):
        m.inference(x, np.array([5]), np.zeros((2, 4, 8), np.int32), 2, prompt_language="en", text_language="en")
    with pytest.raises(AssertionError):
        m.inference(x, np.array([0]), y, 2, prompt_language="en", text_language="en")           # x_lens > 0


def test_continual_host_logic():
    """VALLE.continual (models/vallex.py:688-787): signature, asserts, and the batch it hands to the engine (prefix = first
    half of y capped at 225 frames, first codebook of the rest passed through, language id -1 = no language embedding)."""
    assert list(inspect.signature(VALLE.continual).parameters) == ["self", "x", "x_lens", "y"]
    m = _valle(2)
    x = np.arange(5, 12, dtype=np.int32)[None]
    y = np.random.default_rng(0).integers(0, 1024, size=(1, 61, 8))
    with pytest.raises(AssertionError):
        m.continual(x[0], np.array([7]), y)
    with pytest.raises(AssertionError):
        m.continual(x, np.array([7]), y[0])
    with pytest.raises(AssertionError):
        m.continual(x, np.array([0]), y)

    seen = {}

    class FakeEngine:                                   # stands in for the HIP engine: records what vx_nar would get
        def nar(self, batch, codes0):
            seen["batch"], seen["codes0"] = batch, codes0
            return [np.zeros((len(codes0[0]), 8), np.int64)]

    m.__dict__["_engine"] = FakeEngine()
    out = m.continual(x, np.array([7]), y)
    assert tuple(out.shape) == (1, 31, 8)
    b = seen["batch"]
    assert b.n == 1 and b.text_lens[0] == 7 and b.prompt_lens[0] == 30
    assert (b.text_lang[0, :7] == -1).all() and (b.text_ids[0, :7] == x[0]).all()
    assert (b.prompt_codes[0, :30] == y[0, :30]).all() and (seen["codes0"][0] == y[0, 30:, 0]).all()
    y_long = np.zeros((1, 470, 8), np.int64)
    m.continual(x, np.array([7]), y_long)
    assert seen["batch"].prompt_lens[0] == 225 and len(seen["codes0"][0]) == 245


def test_language_rows_and_model_ids():
    m = _valle(2)
    assert m.language_ID == {"en": 0, "zh": 1, "ja": 2}                     # models/vallex.py:439-443
    assert G.code2lang == {0: "zh", 1: "ja", 2: "en"}                      # macros.py:15-19 (different table!)
    np.testing.assert_array_equal(m._lang_row(5, 2, "ja", "en"), [2, 2, 0, 0, 0])
    np.testing.assert_array_equal(m._lang_row(5, 2, "zh", ["en", "zh", "ja"]), [1, 1, 0, 1, 2])
    with pytest.raises(KeyError):
        m._lang_row(3, 1, "fr", "en")


def test_pe_table_is_the_reference_table():
    np.testing.assert_array_equal(sine_pe_table(300), sine_pe(300).numpy())


def test_generation_errors_before_any_gpu_work(tmp_path, monkeypatch):
    monkeypatch.setattr(G, "model", None)
    monkeypatch.setattr(G, "vocos", None)
    with pytest.raises(RuntimeError, match="preload_models"):
        G.generate_audio([5, 6, 7], language="en")
    with pytest.raises(FileNotFoundError):
        G.preload_models(checkpoint=str(tmp_path / "nope.pt"))
    monkeypatch.setattr(G, "model", object())
    monkeypatch.setattr(G, "vocos", object())
    with pytest.raises(ValueError, match="Cannot find prompt"):
        G.generate_audio([5, 6, 7], prompt="no_such_preset", language="en")
    # the reference looks at `mode` only after the prompt lookup (utils/generation.py:169-176 before :197/:229/:276)
    with pytest.raises(ValueError, match="Cannot find prompt"):
        G.generate_audio_from_long_text([[5, 6]], prompt="x", language="en", mode="bogus")
    with pytest.raises(ValueError, match="No such mode"):
        G.generate_audio_from_long_text([[5, 6]], prompt=os.path.join(GOLD, "presets", "librispeech_1.npz"), language="en",
                                        mode="bogus")
    with pytest.raises(KeyError):
        G.generate_audio([5, 6, 7], language="fr")
    with pytest.raises(ValueError, match="Empty text"):
        G.generate_audio([], language="en")
    with pytest.raises(RuntimeError, match="language detector"):
        G.generate_audio("hello", language="auto")
    with pytest.raises(RuntimeError, match="text front-end"):
        G.generate_audio("hello", language="en")


def test_npz_prompt_wire_format():
    a, t, lang = G._load_prompt(os.path.join(GOLD, "presets", "librispeech_1.npz"))
    assert a.shape == (1, 225, 8) and t.shape == (1, 58) and lang == "en" and a.dtype == np.int32
    monkey = G.PRESET_DIRS
    try:
        G.PRESET_DIRS = [os.path.join(GOLD, "presets") + os.sep]
        a, t, lang = G._load_prompt("cafe")                               # preset name lookup (utils/generation.py:103-110)
        assert lang == "ja"
    finally:
        G.PRESET_DIRS = monkey


class _FakeModel:
    """records what generate_audio* hand to VALLE.inference; returns `frames` frames of zeros"""

    def __init__(self, frames=3):
        self.calls, self.frames = [], frames

    def inference(self, x, x_lens, y, enroll_x_lens=0, **kw):
        self.calls.append(dict(x=np.array(x), y=np.array(y), enroll=enroll_x_lens, **kw))
        return np.zeros((1, self.frames, 8), np.int64) + len(self.calls)


class _FakeVocos:
    def codes_to_features(self, frames):
        return np.asarray(frames)

    def decode(self, features, bandwidth_id=None):
        return np.zeros((1, 320 * features.shape[-1]), np.float32)


def test_long_text_detects_language_on_the_whole_text_and_flips_the_reference_coin(monkeypatch):
    """utils/generation.py:166-167 classifies the WHOLE text once (not sentence by sentence); :264 draws the sliding-window
    coin with torch.rand(1), so torch.manual_seed reproduces which chunks carry their prompt over"""
    import torch
    fm = _FakeModel()
    seen = []
    monkeypatch.setattr(G, "model", fm)
    monkeypatch.setattr(G, "vocos", _FakeVocos())
    monkeypatch.setattr(G, "rng", None)
    monkeypatch.setattr(G, "language_detector", lambda t: (seen.append(t), "en")[1])
    monkeypatch.setattr(G, "sentence_splitter", lambda t: t.split("|"))
    monkeypatch.setattr(G, "text_tokenizer", lambda t: ([7 + (len(t) % 5)] * 4, ["en"] * 4))
    text = "first sentence|second one|third"
    torch.manual_seed(1234)
    expect = [bool(torch.rand(1) < 0.5) for _ in range(3)]               # the reference's draws, in its order
    torch.manual_seed(1234)
    wav = G.generate_audio_from_long_text(text, prompt=None, language="auto", mode="sliding-window", seed=5)
    assert seen == [text]                                                # one detection, on the whole input
    assert wav.shape == (3 * 3 * 320,) and len(fm.calls) == 3
    # chunk k+1 is prompted by chunk k's frames exactly where the coin said "carry" (else back to the original = empty prompt)
    for k in (1, 2):
        carried = fm.calls[k]["y"].shape[1] == fm.frames
        assert carried == expect[k - 1], (k, carried, expect)
        assert fm.calls[k]["enroll"] == (4 if carried else 0)


def test_vocos_state_dict_keeps_only_the_head_tensors():
    """the published vocos checkpoint also holds the whole EnCodec model and the ISTFT window: ignored; a missing head tensor
    fails at load time, not at the first decode"""
    sd = synth.vocos_state_dict(2)
    assert list(sd) == vocos_expected_keys()                              # the synthetic dict IS the expected layout
    extra = dict(sd)
    extra["feature_extractor.encodec.decoder.model.0.conv.conv.weight_v"] = np.zeros((4, 4, 7), np.float32)
    extra["feature_extractor.encodec.quantizer.vq.layers.0._codebook.inited"] = np.ones(1, bool)
    extra["head.istft.window"] = np.hanning(1280).astype(np.float32)
    m = _valle(2).load_vocos_state_dict(extra)
    assert list(m._vocos_sd) == vocos_expected_keys() and all(v.dtype == np.float32 for v in m._vocos_sd.values())
    bad = dict(sd)
    bad.pop("head.out.bias")
    with pytest.raises(RuntimeError, match="missing"):
        _valle(2).load_vocos_state_dict(bad)


def test_make_transcript_normalises_a_clipping_waveform_in_place():
    """utils/prompt_making.py:91-92: `if wav.abs().max() > 1: wav /= wav.abs().max()` mutates the tensor make_prompt goes on to
    tokenize -- the mirror must hand the SAME (normalised) samples to the EnCodec encoder."""
    import numpy as np
    from vallex_amd.utils import generation as G
    from vallex_amd.utils import prompt_making as PM
    G.language_detector = lambda text: "en"
    try:
        wav = np.array([[0.5, -3.0, 1.5, 0.0]], np.float32)
        text, lang = PM.make_transcript("x", wav, 24000, "hello")
        assert (text, lang) == ("[EN]hello[EN]", "en")
        np.testing.assert_array_equal(wav, np.array([[0.5, -3.0, 1.5, 0.0]], np.float32) / np.float32(3.0))
        quiet = np.array([[0.5, -0.9]], np.float32)
        PM.make_transcript("x", quiet, 24000, "hello")
        np.testing.assert_array_equal(quiet, np.array([[0.5, -0.9]], np.float32))          # peaks <= 1 stay untouched
    finally:
        G.language_detector = None


def test_long_text_without_prompt_accepts_any_mode_like_the_reference(monkeypatch):
    """utils/generation.py:161-163: no prompt => mode = 'sliding-window' BEFORE anything looks at it, so mode='x' works there"""
    fm = _FakeModel()
    monkeypatch.setattr(G, "model", fm)
    monkeypatch.setattr(G, "vocos", _FakeVocos())
    for prompt in (None, ""):
        wav = G.generate_audio_from_long_text([[5, 6, 7], [8, 9]], prompt=prompt, language="en", mode="x")
        assert wav.shape == (320 * 6,)
    assert len(fm.calls) == 4


class _NotOnTheAllowList:
    """a picklable class the weights-only unpickler does not know (stands for anything a hostile checkpoint could carry)"""


def test_preload_models_reads_the_checkpoint_file_like_the_reference(tmp_path, monkeypatch):
    """utils/generation.py:50-89 with no arguments: ./checkpoints/vallex-checkpoint.pt -> torch.load(...)["model"] ->
    load_state_dict(strict=True).  Goes through the FILE (torch.save of {"model": state_dict, ...} as the published checkpoint
    is laid out), under torch's weights_only default, and hands all 374 tensors of the 12-layer layout to the engine."""
    import torch
    from oracle import synth
    from vallex_amd.models.vallex import expected_keys
    nl = 12
    rng = np.random.default_rng(0)
    sd = {}
    for k, v in synth.vallex_state_dict(1, 0).items():                    # shapes of one layer, replicated: cheap to build
        sd[k] = v
    full = {}
    for k in expected_keys(nl):
        src = k
        for i in range(1, nl):
            src = src.replace(f"layers.{i}.", "layers.0.")
        full[k] = torch.from_numpy(np.ascontiguousarray(sd[src]))
    assert len(full) == 374
    (tmp_path / "checkpoints").mkdir()
    torch.save({"model": full, "optimizer": {"state": {}, "param_groups": [{"lr": 0.1}]}, "epoch": 3},
               tmp_path / "checkpoints" / "vallex-checkpoint.pt")
    monkeypatch.chdir(tmp_path)
    got = {}

    class RecordingVALLE:
        def __init__(self, *a, **kw):
            got["ctor"] = (a, kw)

        def to(self, dev):
            got["device"] = dev
            return self

        def load_state_dict(self, state_dict, strict=True):
            got["keys"], got["strict"] = list(state_dict), strict
            got["dtypes"] = {str(v.dtype) for v in state_dict.values()}
            return self

        def eval(self):
            return self

    monkeypatch.setattr(G, "VALLE", RecordingVALLE)
    monkeypatch.setattr(G, "model", None)
    monkeypatch.setattr(G, "vocos", None)
    G.preload_models()                                                    # no arguments, like README.md:156-158
    assert got["strict"] is True and sorted(got["keys"]) == sorted(expected_keys(12)) and got["dtypes"] == {"torch.float32"}
    assert got["ctor"][0][:3] == (1024, 16, 12)
    assert G.model is not None and G.vocos is None                       # no Vocos file given: generate_audio would refuse
    # a checkpoint with the trainer's bookkeeping next to "model" (argparse.Namespace, a path): allow-listed, still weights-only
    import argparse
    import pathlib
    torch.save({"model": full, "args": argparse.Namespace(lr=1e-4), "exp_dir": pathlib.PosixPath("exp/valle")},
               tmp_path / "checkpoints" / "vallex-checkpoint.pt")
    got.clear()
    G.preload_models()
    assert len(got["keys"]) == 374
    # any OTHER pickled class is refused (the permissive unpickler runs code from the file) unless the caller opts in
    torch.save({"model": full, "sampler": _NotOnTheAllowList()}, tmp_path / "checkpoints" / "vallex-checkpoint.pt")
    got.clear()
    monkeypatch.delenv("VALLEX_ALLOW_UNSAFE_PICKLE", raising=False)
    with pytest.raises(RuntimeError, match="allow_unsafe_pickle"):
        G.preload_models()
    assert not got
    G.preload_models(allow_unsafe_pickle=True)
    assert len(got["keys"]) == 374
    got.clear()
    monkeypatch.setenv("VALLEX_ALLOW_UNSAFE_PICKLE", "1")
    G.preload_models()
    assert len(got["keys"]) == 374
    monkeypatch.delenv("VALLEX_ALLOW_UNSAFE_PICKLE")
    # the real VALLE mirror accepts the file's dict strictly (and rejects one with a missing key)
    from vallex_amd.models.vallex import VALLE as RealVALLE
    m = RealVALLE(1024, 16, 12, norm_first=True, add_prenet=False, prefix_mode=1, share_embedding=True, nar_scale_factor=1.0,
                  prepend_bos=True, num_quantizers=8)
    m.load_state_dict(torch.load(tmp_path / "checkpoints" / "vallex-checkpoint.pt", weights_only=False)["model"], strict=True)
    assert len(m._sd) == 374 and all(v.dtype == np.float32 for v in m._sd.values())
    bad = dict(full)
    bad.pop("ar_predict_layer.weight")
    with pytest.raises(RuntimeError, match="missing"):
        m.load_state_dict(bad, strict=True)
