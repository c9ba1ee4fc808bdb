"""Host-side batch text front-end service.

The reference tokenises ONE utterance per `generate_audio` call on the caller's thread, right before the model call
(`utils/generation.py:127-132` -> `PhonemeBpeTokenizer.tokenize`, `utils/g2p/__init__.py:15-25`: cleaners/G2P, BPE, per-id
languages); with a GPU that turns 32 x 8 s of audio out in under a second that serial CPU stage becomes the bottleneck.
This module keeps the reference's tokenizer contract -- a callable `text -> (phoneme ids, per-id languages)` that raises
`ValueError("Empty text is given")` on empty input -- and runs it for whole batches on a thread pool, so the front-end work of
batch k+1 overlaps the GPU work of batch k (`libvallex_hip.so` calls release the GIL).

    svc = TextFrontendService(PhonemeBpeTokenizer("utils/g2p/bpe_69.json").tokenize, workers=16)
    for wavs in synthesize_stream(requests, svc, batch_size=32): ...

The G2P rules themselves (cleaners, langid, jieba, ...) are third-party CPU code outside the hot path: they are plugged in, not
re-implemented.
"""
from __future__ import annotations

import os
from concurrent.futures import Future, ThreadPoolExecutor
from typing import Callable, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from ..macros import lang2token, token2lang

Tokenizer = Callable[[str], Tuple[Sequence[int], Sequence[str]]]


class TokenizedText:
    """What `generate_audio` builds per utterance before the model call: ids + per-id language strings."""

    __slots__ = ("ids", "langs", "language")

    def __init__(self, ids, langs, language):
        self.ids = np.asarray(ids, np.int32).reshape(-1)
        self.langs = list(langs) if langs is not None else None
        self.language = language


class TextFrontendService:
    def __init__(self, tokenizer: Tokenizer, workers: Optional[int] = None):
        if tokenizer is None:
            raise RuntimeError("no text front-end configured: pass the reference's PhonemeBpeTokenizer(...).tokenize")
        self.tokenizer = tokenizer
        self.workers = workers or min(32, (os.cpu_count() or 4))
        self._pool = ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="vx-frontend")

    def _one(self, text: str, language: str) -> TokenizedText:
        lang_token = lang2token[language]                         # KeyError on an unknown language (macros.py:8-13)
        text = text.replace("\n", "").strip(" ")                  # utils/generation.py:94
        ids, langs = self.tokenizer(f"_{lang_token}{text}{lang_token}".strip())     # :126-128
        if len(ids) == 0:
            raise ValueError("Empty text is given")               # utils/g2p/__init__.py:23-24
        if len(langs) != len(ids):
            raise AssertionError("tokenizer returned %d ids for %d languages" % (len(ids), len(langs)))
        return TokenizedText(ids, langs, token2lang[lang_token])

    def submit(self, texts: Sequence[str], languages: Sequence[str]) -> List[Future]:
        """Start tokenising a batch; returns one future per utterance (order preserved)."""
        if len(texts) != len(languages):
            raise ValueError("texts and languages must have the same length")
        return [self._pool.submit(self._one, t, l) for t, l in zip(texts, languages)]

    def tokenize_batch(self, texts: Sequence[str], languages: Sequence[str]) -> List[TokenizedText]:
        """Blocking form; the first failing utterance raises its exception (like the reference would for that call)."""
        return [f.result() for f in self.submit(texts, languages)]

    def close(self):
        self._pool.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def synthesize_stream(requests: Iterable[dict], service: TextFrontendService, batch_size: int = 32,
                      synthesize: Optional[Callable[[List[dict], List[TokenizedText]], List[np.ndarray]]] = None
                      ) -> Iterator[List[np.ndarray]]:
    """Pipeline: requests (dicts with `text`, `language`, optional `prompt`) are cut into batches; while the GPU runs batch k
    (`synthesize`, default `generation.generate_audio_batch` on the already tokenised batch) the pool tokenises batch k+1.
    Yields one list of waveforms per batch, in request order."""
    if synthesize is None:
        from . import generation as G

        def synthesize(reqs, toks):
            return G.generate_audio_batch([t.ids for t in toks], prompts=[r.get("prompt") for r in reqs],
                                          language=[t.language for t in toks], text_languages=[t.langs for t in toks])

    def batches():
        cur = []
        for r in requests:
            cur.append(r)
            if len(cur) == batch_size:
                yield cur
                cur = []
        if cur:
            yield cur

    pending = None                       # (requests, futures) of the batch whose tokenisation is in flight
    for reqs in batches():
        futs = service.submit([r["text"] for r in reqs], [r["language"] for r in reqs])
        if pending is not None:
            p_reqs, p_futs = pending
            yield synthesize(p_reqs, [f.result() for f in p_futs])          # GPU on batch k, pool on batch k+1
        pending = (reqs, futs)
    if pending is not None:
        p_reqs, p_futs = pending
        yield synthesize(p_reqs, [f.result() for f in p_futs])
