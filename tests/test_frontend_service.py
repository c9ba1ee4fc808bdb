"""CPU: the host-side batch text front-end service: the reference's tokenizer contract
(utils/g2p/__init__.py:15-25) run for whole batches on a thread pool, overlapped with the consumer of the previous batch."""
import threading
import time

import numpy as np
import pytest

import vallex_amd  # noqa: F401
from vallex_amd.utils.frontend import TextFrontendService, synthesize_stream


def fake_tokenizer(text):
    """stands in for PhonemeBpeTokenizer.tokenize: one id per character of the cleaned text, language from the wrapper token"""
    assert text.startswith("_")
    body = text[1:]
    lang = {"[EN]": "en", "[ZH]": "zh", "[JA]": "ja"}[body[:4]]
    inner = body[4:-4]
    time.sleep(0.02)
    ids = [5 + (ord(ch) % 60) for ch in inner]
    return ids, [lang] * len(ids)


def test_batch_tokenisation_keeps_order_and_contract():
    with TextFrontendService(fake_tokenizer, workers=8) as svc:
        texts = [f"utterance number {i}\n" for i in range(40)]
        t0 = time.perf_counter()
        toks = svc.tokenize_batch(texts, ["en", "zh", "ja", "en"] * 10)
        dt = time.perf_counter() - t0
        assert dt < 40 * 0.02 * 0.6                               # ran on the pool, not serially
        for i, t in enumerate(toks):
            want = [5 + (ord(ch) % 60) for ch in f"utterance number {i}"]      # newline and spaces stripped like the reference
            assert t.ids.dtype == np.int32 and t.ids.tolist() == want
            assert t.langs == [("en", "zh", "ja", "en")[i % 4]] * len(want)
        with pytest.raises(ValueError, match="Empty text"):
            svc.tokenize_batch(["fine", "   "], ["en", "en"])
        with pytest.raises(KeyError):
            svc.tokenize_batch(["x"], ["klingon"])


def test_stream_overlaps_frontend_with_the_consumer():
    """while the consumer (the GPU call in production) works on batch k, batch k+1 is being tokenised"""
    active = {"tok": 0, "overlap": 0}
    lock = threading.Lock()

    def tok(text):
        with lock:
            active["tok"] += 1
        out = fake_tokenizer(text)
        with lock:
            active["tok"] -= 1
        return out

    def consumer(reqs, toks):
        time.sleep(0.05)
        with lock:
            active["overlap"] += active["tok"] > 0
        return [np.zeros(len(t.ids), np.float32) for t in toks]

    reqs = [dict(text=f"request {i}", language="en") for i in range(70)]
    with TextFrontendService(tok, workers=2) as svc:
        outs = list(synthesize_stream(reqs, svc, batch_size=32, synthesize=consumer))
    assert [len(o) for o in outs] == [32, 32, 6]
    assert outs[2][5].shape == (len("request 69"),)
    assert active["overlap"] >= 1
