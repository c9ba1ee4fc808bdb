"""Text front-end glue of the reference (`utils/g2p/__init__.py:11-25`, `utils/g2p/cleaners.py:25-61`), without its G2P rules.

The reference turns `"[EN]hello[EN][ZH]你好[ZH]"` into phoneme ids in three steps: (1) cut the text into language-tagged
segments, (2) run each segment through that language's text -> IPA converter (`english_to_ipa2`, `chinese_to_ipa`,
`japanese_to_ipa2`: rule code over third-party packages -- eng_to_ipa, pypinyin, jieba, cn2an, pyopenjtalk -- none of which is
part of the hot path or installed here), (3) BPE-encode the IPA string with `bpe_69.json` and label every id with the language
of its segment.  Steps (1) and (3) and the punctuation rule between them are the DATA FORMAT the model consumes (ids + per-id
languages, `models/vallex.py:499-505`); they are mirrored here so that a caller only has to plug in the three converters:

    from utils.g2p.english import english_to_ipa2          # the reference's own modules, or any other G2P
    tok = PhonemeBpeTokenizer("utils/g2p/bpe_69.json", {"en": english_to_ipa2, "zh": chinese_to_ipa, "ja": japanese_to_ipa2})
    vallex_amd.utils.generation.text_tokenizer = tok.tokenize          # or TextFrontendService(tok.tokenize)

Pinned against the live reference tokenizer (its language modules stubbed with the same stand-in converters) by
`oracle/make_golden_frontend.py` -> `tests/golden/g2p_frontend.json`, `tests/test_g2p_frontend.py`.
"""
from __future__ import annotations

import re
from typing import Callable, Dict, List, Sequence, Tuple

TAGS = (("en", "[EN]"), ("zh", "[ZH]"), ("ja", "[JA]"))          # label priority of a segment: EN, then ZH, then JA
_SUB_ORDER = ("zh", "ja", "en")                                  # order in which a segment's inner spans are converted
_SPAN = {lang: re.compile(re.escape(tag) + r"(.*?)" + re.escape(tag)) for lang, tag in TAGS}
_CLOSERS = ".,!?-…~"                                             # a segment that ends in none of these gets a full stop


def tagged_segments(text: str) -> List[str]:
    """Every `[XX]...[XX]` span of every language (non-greedy, found per language over the whole text), ordered by where it
    starts -- the reference's segmentation, including its behaviour on nested tags (an inner span is listed again)."""
    found = [m for lang, _ in TAGS for m in _SPAN[lang].finditer(text)]
    found.sort(key=lambda m: m.start())
    return [text[m.start():m.end()] for m in found]


def segment_to_phonemes(segment: str, converters: Dict[str, Callable[[str], str]]) -> str:
    """One tagged segment -> its phoneme string: inner spans replaced by `converter(inner) + " "`, trailing white space dropped,
    a full stop appended unless the segment already ends in punctuation."""
    out = segment
    for lang in _SUB_ORDER:
        tag = dict(TAGS)[lang]
        if tag in out:
            conv = converters.get(lang)
            if conv is None:
                raise RuntimeError(f"no text->IPA converter configured for language '{lang}' (the reference uses "
                                   "utils/g2p/{english,mandarin,japanese}.py): pass converters={'" + lang + "': fn}")
            out = _SPAN[lang].sub(lambda m, conv=conv: conv(m.group(1)) + " ", out)
    out = out.rstrip()
    if out and out[-1] not in _CLOSERS:
        out += "."
    return out


def clean_tagged_text(text: str, converters: Dict[str, Callable[[str], str]]) -> Tuple[str, List[str]]:
    """`cje_cleaners`: (phoneme string, one language label per CHARACTER of it).  Text outside any tag pair is dropped."""
    phonemes, langs = "", []
    for seg in tagged_segments(text):
        ph = segment_to_phonemes(seg, converters)
        lang = next(lg for lg, tag in TAGS if tag in seg)
        phonemes += ph
        langs += [lang] * len(ph)
    return phonemes, langs


class PhonemeBpeTokenizer:
    """Same constructor default, method and return value as the reference class; `converters` maps 'en' / 'zh' / 'ja' to a
    text -> IPA function."""

    def __init__(self, tokenizer_path: str = "./utils/g2p/bpe_1024.json", converters: Dict[str, Callable[[str], str]] = None):
        from tokenizers import Tokenizer                           # the reference's own dependency (requirements.txt)
        self.tokenizer = Tokenizer.from_file(tokenizer_path)
        self.converters = dict(converters or {})

    def tokenize(self, text: str) -> Tuple[Sequence[int], List[str]]:
        phonemes, langs = clean_tagged_text(text, self.converters)
        ids = self.tokenizer.encode(phonemes.replace(" ", "_")).ids
        assert len(ids) == len(langs), (len(ids), len(langs))      # one id per phoneme character (character-level BPE)
        if not len(ids):
            raise ValueError("Empty text is given")
        return ids, langs
