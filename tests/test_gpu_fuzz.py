"""GPU: seeded random ragged batches against the CPU oracle, row by row.

Batch sizes 1 .. 16 walk through the decode chains of the engine (<= 4 rows: reduce + LayerNorm / combine folded into the
consuming GEMM; 5 .. 7 rows: context-split dec_attn + combine + separate out_proj; 8 .. 16 rows: out_proj fused into the context-split
dec_attn with 4 / 3 / 2 splits and the combine behind W_o -- contexts this short leave some splits EMPTY; the 32-row chain has its
own golden test);
prompt lengths 0 .. 90, text lengths 1 .. 30, three languages, top-k = 10 with injected uniforms and an EOS-friendly weight set
(eos_gain 2.5), so the rows of a batch END AT DIFFERENT STEPS -- finished rows keep riding along as dead columns.  Every row must
equal the oracle run on that row alone (= one reference VALLE.inference call)."""
import numpy as np
import pytest

from oracle import synth
from oracle.vallex_oracle import VallexOracle
from tests._util import get_model

pytestmark = pytest.mark.gpu

NL, SEED, EOS_GAIN, CAP = 2, 12, 2.5, 36


@pytest.mark.parametrize("batch,trial", [(1, 0), (2, 1), (3, 2), (4, 3), (5, 4), (7, 5), (9, 6), (4, 7), (1, 8), (8, 9), (10, 10), (12, 11), (16, 12)])
def test_random_ragged_batch_rows_equal_the_oracle(batch, trial):
    rng = np.random.default_rng(9000 + trial)
    m = get_model(NL, SEED, EOS_GAIN, max_new=64, max_prompt=128, max_text=64, max_batch=16)
    orc = VallexOracle(synth.vallex_state_dict(NL, SEED, EOS_GAIN), NL)
    rows, cols = [], []
    for i in range(batch):
        tp = int(rng.choice([0, 1, 2, int(rng.integers(3, 91))]))
        sp = 0 if tp == 0 else int(rng.integers(1, 13))
        nt = int(rng.integers(1, 19))
        a, t = synth.synth_prompt(tp, sp, seed=int(rng.integers(1, 1 << 30)))
        txt = np.concatenate([t[0], synth.synth_text(nt, int(rng.integers(1, 1 << 30)))])
        lang = ("en", "zh", "ja")[int(rng.integers(0, 3))]
        rows.append(dict(text=txt, prompt=a[0], enroll=sp, prompt_language=lang, text_language=("en", "zh", "ja")[int(rng.integers(0, 3))]))
        cols.append(synth.uniforms(4096, 1, int(rng.integers(1, 1 << 30)))[:, 0])
    outs = m.inference_batch(rows, top_k=10, uniforms=np.stack(cols, axis=1), force_eos_at=CAP)
    lens = []
    for i, (r, u) in enumerate(zip(rows, cols)):
        ref = orc.inference(r["text"][None], np.array([len(r["text"])]), r["prompt"][None], r["enroll"], top_k=10,
                            prompt_language=r["prompt_language"], text_language=r["text_language"], uniforms=u, force_eos_at=CAP)[0]
        assert outs[i].shape == ref.shape, (trial, i, outs[i].shape, ref.shape)
        np.testing.assert_array_equal(outs[i], ref, err_msg=f"trial {trial} row {i} of {batch}")
        lens.append(ref.shape[0])
    print(f"trial {trial}: batch {batch}, generated lengths {lens}")
