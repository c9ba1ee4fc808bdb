"""CPU: the text front-end glue (vall-e-x_amd/utils/g2p.py: tag segmentation, conversion order, punctuation rule, per-character
language labels, BPE ids) against the LIVE reference tokenizer's outputs (tests/golden/g2p_frontend.json, made by
oracle/make_golden_frontend.py with the reference's three language modules replaced by the stand-in converters used here)."""
import json
import os

import pytest

import vallex_amd  # noqa: F401
from oracle.make_golden_frontend import build_char_tokenizer, stand_in_converters
from vallex_amd.utils.g2p import PhonemeBpeTokenizer, clean_tagged_text, tagged_segments

GOLD = os.path.join(os.path.dirname(__file__), "golden")
_G = json.load(open(os.path.join(GOLD, "g2p_frontend.json"), encoding="utf-8"))
CASES, SYMBOLS = _G["cases"], _G["symbols"]


@pytest.fixture(scope="module")
def tok(tmp_path_factory):
    # an equivalent of the reference's utils/g2p/bpe_69.json, re-created from its symbol table (checked against the reference's
    # own file by the generator)
    path = build_char_tokenizer(SYMBOLS, str(tmp_path_factory.mktemp("bpe") / "bpe_69.json"))
    return PhonemeBpeTokenizer(path, stand_in_converters())


@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_tokenize_equals_live_reference(tok, case):
    if "error" in case:
        with pytest.raises(ValueError, match="Empty text is given"):          # utils/g2p/__init__.py:23-24
            tok.tokenize(case["text"].strip())
        return
    ids, langs = tok.tokenize(case["text"].strip())
    assert list(ids) == case["ids"] and list(langs) == case["langs"]
    assert len(ids) == len(langs) and all(0 <= i < 70 for i in ids)


def test_segments_labels_and_missing_converter():
    assert tagged_segments("x[EN]a[EN]y[ZH]b[ZH]") == ["[EN]a[EN]", "[ZH]b[ZH]"]
    ph, langs = clean_tagged_text("[JA]a[JA][EN]b c[EN]", {"ja": str.upper, "en": lambda s: s})
    assert ph == "A.b c." and langs == ["ja"] * 2 + ["en"] * 4
    with pytest.raises(RuntimeError, match="converter"):
        clean_tagged_text("[ZH]x[ZH]", {"en": str})


def test_plugs_into_generation_hook_and_frontend_service(tok):
    """the reference's call shape: generate_audio hands `_[EN]text[EN]` to text_tokenizer (utils/generation.py:126-128)"""
    from vallex_amd.utils.frontend import TextFrontendService
    with TextFrontendService(tok.tokenize, workers=2) as svc:
        out = svc.tokenize_batch(["hello world", "你好"], ["en", "zh"])
    assert [t.language for t in out] == ["en", "zh"]
    assert list(out[0].ids) == CASES[0]["ids"] and out[0].langs == CASES[0]["langs"]
    assert set(out[1].langs) == {"zh"}


def test_mix_language_code_switching_reaches_the_model_as_per_id_languages(tok, monkeypatch):
    """README.md:217-222 of the reference: language='mix' with user-written tags -> lang_token '' (macros.py:8-13), prompt
    language 'en' when there is no prompt (utils/generation.py:123), text_language = the tokenizer's per-id list (:146)."""
    import numpy as np
    from vallex_amd.utils import generation as G
    seen = {}

    class FakeModel:
        def inference(self, x, x_lens, y, enroll_x_lens=0, **kw):
            seen.update(x=np.array(x), enroll=enroll_x_lens, **kw)
            return np.zeros((1, 2, 8), np.int64)

    class FakeVocos:
        def codes_to_features(self, frames):
            return np.asarray(frames)

        def decode(self, features, bandwidth_id=None):
            return np.zeros((1, 320 * features.shape[-1]), np.float32)

    monkeypatch.setattr(G, "model", FakeModel())
    monkeypatch.setattr(G, "vocos", FakeVocos())
    monkeypatch.setattr(G, "text_tokenizer", tok.tokenize)
    wav = G.generate_audio("[ZH]你好[ZH][EN]yes[EN]", language="mix", seed=3)
    assert wav.shape == (640,)
    ids, langs = tok.tokenize("[ZH]你好[ZH][EN]yes[EN]")
    assert list(seen["x"][0]) == list(ids) and seen["enroll"] == 0
    assert seen["prompt_language"] == "en" and seen["text_language"] == langs and set(langs) == {"zh", "en"}
    assert seen["top_k"] == -100 and seen["temperature"] == 1                      # utils/generation.py:141-142
