"""GPU: the batch-32 decode path against the live reference.  With 32 rows in a call the AR step runs its batch-32 kernel
configuration (one context split, out_proj fused into dec_attn, 16 head slabs reduced by the LayerNorm prologue) -- a different
code path from the 1-3 row calls of the other golden tests (context-split dec_attn + combine + separate out_proj GEMM).  Three of
the 32 rows are full-length golden rows (600 frames, 12 layers): they must reproduce the reference's ids bit for bit in this
batch, too."""
import numpy as np
import pytest

from oracle import synth
from oracle.make_golden import FULL_CASES, case_inputs
from tests._util import get_model, golden

pytestmark = pytest.mark.gpu


def _golden_row(name):
    c = FULL_CASES[name]
    a, t, text, pl, langs = case_inputs(c)
    us = None if c["useed"] is None else synth.uniforms(4096, 1, c["useed"])[:, 0]
    return dict(text=text[0], prompt=a[0], enroll=t.shape[-1], prompt_language=pl, text_language=langs), us


@pytest.mark.parametrize("kind", ["greedy", "topk10"])
def test_golden_rows_inside_a_32_row_batch(kind):
    names = [n for n in sorted(FULL_CASES) if n.endswith(kind)]
    c0 = FULL_CASES[names[0]]
    m = get_model(12, c0["seed"], c0["eos_gain"], max_new=608, max_prompt=400, max_text=256, max_batch=32)
    rows, cols, where = [], [], {}
    rng = np.random.default_rng(77)
    for i in range(32):
        if i in (3, 17, 30):
            n = names[len(where)]
            r, us = _golden_row(n)
            where[i] = n
        else:
            tp, sp = int(rng.integers(150, 300)), int(rng.integers(20, 80))
            a, t = synth.synth_prompt(tp, sp, seed=4000 + i)
            lang = ("en", "zh", "ja")[i % 3]
            r = dict(text=np.concatenate([t[0], synth.synth_text(100, 4000 + i)]), prompt=a[0], enroll=sp, prompt_language=lang,
                     text_language=lang)
            us = synth.uniforms(4096, 1, 9000 + i)[:, 0]
        rows.append(r)
        cols.append(us if us is not None else np.zeros(4096, np.float32))
    us = None if c0["useed"] is None else np.stack(cols, axis=1)
    outs = m.inference_batch(rows, top_k=c0["top_k"], uniforms=us, force_eos_at=600)
    assert all(o.shape == (600, 8) for o in outs)
    for i, n in where.items():
        gold = golden(n)["codes"][0]
        d = np.argwhere(outs[i] != gold)
        assert len(d) == 0, f"{n} as row {i} of 32: first differing id at frame {d[0][0]}, codebook {d[0][1]}; {len(d)} differ"
