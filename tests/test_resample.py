"""CPU: the built-in resampler of the prompt-enrolment path (vallex_amd.utils.prompt_making.resample_sinc_hann), a restatement of
torchaudio's documented default `Resample` (data/tokenizer.py:105 via encodec's convert_audio).  torchaudio is not installed, so these
are PROPERTY tests of a band-limited resampler, not a pin to the package: lengths, identity, DC, in-band sines, out-of-band rejection,
the polyphase structure (integer-ratio outputs at the original instants), and the plug-in hook."""
import math

import numpy as np
import pytest

import vallex_amd  # noqa: F401
from vallex_amd.utils import prompt_making as PM


def _sine(freq, sr, n, phase=0.3):
    return np.sin(2 * np.pi * freq * np.arange(n) / sr + phase).astype(np.float32)[None]


@pytest.mark.parametrize("sr", [8000, 16000, 22050, 44100, 48000])
def test_length_dc_and_inband_sine(sr):
    n = sr // 4 + 17
    out = PM.resample_sinc_hann(np.ones((1, n), np.float32), sr, 24000)
    assert out.shape == (1, math.ceil(24000 * n / sr)) and out.dtype == np.float32
    edge = 200
    np.testing.assert_allclose(out[0, edge:-edge], 1.0, atol=2e-3)                     # a constant stays a constant (away from the zero padding)
    f = 440.0
    got = PM.resample_sinc_hann(_sine(f, sr, n), sr, 24000)
    want = _sine(f, 24000, got.shape[1])
    np.testing.assert_allclose(got[0, edge:-edge], want[0, edge:-edge], atol=3e-3)     # same continuous-time signal, new grid


def test_identity_and_two_channels():
    x = np.random.default_rng(0).standard_normal((2, 1000)).astype(np.float32)
    assert np.array_equal(PM.resample_sinc_hann(x, 24000, 24000), x)
    y = PM.resample_sinc_hann(x, 48000, 24000)
    assert y.shape == (2, 500)
    np.testing.assert_allclose(y[1], PM.resample_sinc_hann(x[1:], 48000, 24000)[0], atol=2e-6)  # channels are independent (conv batching aside)


def test_downsampling_removes_what_the_new_rate_cannot_hold():
    n = 48000 // 2
    hi = PM.resample_sinc_hann(_sine(18000.0, 48000, n), 48000, 24000)                  # above the new Nyquist (12 kHz)
    assert float(np.abs(hi[0, 200:-200]).max()) < 2e-2
    lo = PM.resample_sinc_hann(_sine(3000.0, 48000, n), 48000, 24000)
    np.testing.assert_allclose(lo[0, 200:-200], _sine(3000.0, 24000, lo.shape[1])[0, 200:-200], atol=3e-3)


def test_integer_upsampling_keeps_the_original_samples():
    """12 kHz -> 24 kHz: every second output instant is an input instant; a band-limited input comes back there"""
    x = (0.5 * _sine(500.0, 12000, 3000) + 0.25 * _sine(2100.0, 12000, 3000, 1.1)).astype(np.float32)
    y = PM.resample_sinc_hann(x, 12000, 24000)
    assert y.shape == (1, 6000)
    np.testing.assert_allclose(y[0, 0::2][100:-100], x[0, 100:-100], atol=2e-3)


def test_hook_replaces_the_builtin():
    calls = []

    class Tok:
        sample_rate = 24000

        def encode(self, w):
            calls.append(("encode", w.shape))
            return [(np.zeros((1, 8, 1), np.int64), None)]

    PM.resampler = lambda wav, sr, target: (calls.append(("hook", sr, target)), np.zeros((1, 24000), np.float32))[1]
    try:
        PM.tokenize_audio(Tok(), (np.zeros((2, 16000), np.float32), 16000))
    finally:
        PM.resampler = None
    assert calls == [("hook", 16000, 24000), ("encode", (1, 1, 24000))]
    PM.tokenize_audio(Tok(), (np.zeros((1, 16000), np.float32), 16000))               # built-in: 16000 samples -> 24000
    assert calls[-1] == ("encode", (1, 1, 24000))
