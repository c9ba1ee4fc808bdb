"""CPU: the text front-end glue (vall-e-x_amd/utils/g2p.py: tag segmentation, conversion order, punctuation rule, per-character
language labels, BPE ids) against the LIVE reference tokenizer's outputs (tests/golden/g2p_frontend.json, made by
oracle/make_golden_frontend.py with the reference's three language modules replaced by the stand-in converters used here)."""
import json
import os

import pytest

import vallex_amd  # noqa: F401
from oracle.make_golden_frontend import stand_in_converters
from vallex_amd.utils.g2p import PhonemeBpeTokenizer, clean_tagged_text, tagged_segments

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = json.load(open(os.path.join(GOLD, "g2p_frontend.json"), encoding="utf-8"))


@pytest.fixture(scope="module")
def tok():
    return PhonemeBpeTokenizer(os.path.join(GOLD, "bpe_69.json"), stand_in_converters())


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
