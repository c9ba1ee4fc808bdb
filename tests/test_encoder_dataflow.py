"""CPU: numpy emulation of the DATA FLOW of vx_encodec_encode (vall-e-x_amd/csrc/engine.hip + encodec.hip) -- channels-last
activations, the weight re-layouts done at vx_finalize_weights, im2col with the causal reflect rule, the strided conv as a
GEMM over OVERLAPPING rows of a padded copy (lda = r*C, K = 2*r*C), and the residual-VQ select -- checked against the encoder
oracle.  It pins the index arithmetic of the HIP path where no GPU is available; the kernels themselves are covered by the
GPU test in test_encodec.py."""
import numpy as np

from oracle.encodec_oracle import EncodecEncoderOracle, encodec_encoder_state_dict, encodec_state_dict
from oracle.make_golden_encodec import case_wav


def _elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0))).astype(np.float32)


def _im2col_mode0(x, k, elu):
    """im2col_seq_kernel mode 0: out[t][tap*C + c] = f(x[t + tap - (k-1)]), negative index j -> x[-j] (0 if -j >= T)."""
    T, C = x.shape
    out = np.zeros((T, k * C), np.float32)
    for tap in range(k):
        j = np.arange(T) + tap - (k - 1)
        neg = j < 0
        jj = np.where(neg, -j, j)
        ok = jj < T
        v = np.where(ok[:, None], x[np.minimum(jj, T - 1)], 0.0)
        out[:, tap * C:(tap + 1) * C] = _elu(v) if elu else v
    return out


def _pad_elu(x, r, n_out):
    """enc_pad_elu_kernel: rows = (n_out + 1) r, left = r, reflect on both sides, short inputs zero-extended."""
    L, C = x.shape
    rows, extra = (n_out + 1) * r, n_out * r - L
    max_pad = max(r, extra)
    Le = L + (max_pad - L + 1) if L <= max_pad else L
    out = np.zeros((rows, C), np.float32)
    for j in range(rows):
        u = j - r
        if u < 0:
            u = -u
        elif u >= Le:
            u = 2 * (Le - 1) - u
        if 0 <= u < L:
            out[j] = _elu(x[u])
    return out


def emulate(enc, dec, wav):
    L = wav.shape[0]
    # first conv: out[t][c] = b[c] + sum_tap w[c][tap] * wav[reflect(t + tap - 6)]
    w0, b0 = enc["encoder.0.weight"][:, 0], enc["encoder.0.bias"]
    xs = np.zeros((L, 7), np.float32)
    for tap in range(7):
        j = np.arange(L) + tap - 6
        jj = np.where(j < 0, -j, j)
        xs[:, tap] = np.where(jj < L, wav[np.minimum(jj, L - 1)], 0.0)
    x = xs @ w0.T + b0
    C = 32
    for s4, r in enumerate((2, 4, 5, 8)):
        pR, pD = f"encoder.{1 + 3 * s4}.", f"encoder.{3 + 3 * s4}."
        Lc = x.shape[0]
        n_out = -(-Lc // r)
        sc = x @ enc[pR + "shortcut.weight"][:, :, 0].T + enc[pR + "shortcut.bias"]
        w1 = enc[pR + "block1.weight"]                                   # (C/2, C, 3) -> [C/2][tap*C + c]
        w1r = np.transpose(w1, (0, 2, 1)).reshape(C // 2, 3 * C)
        h = _elu(_im2col_mode0(x, 3, True) @ w1r.T + enc[pR + "block1.bias"])
        out = h @ enc[pR + "block3.weight"][:, :, 0].T + enc[pR + "block3.bias"] + sc
        pad = _pad_elu(out, r, n_out)                                    # [(n_out+1) r][C]
        wd = enc[pD + "weight"]                                          # (2C, C, 2r) -> [2C][tap*C + c]
        wdr = np.transpose(wd, (0, 2, 1)).reshape(2 * C, 2 * r * C)
        flat = pad.reshape(-1)
        A = np.stack([flat[t * r * C: t * r * C + 2 * r * C] for t in range(n_out)])     # overlapping rows, lda = r*C
        x = (A @ wdr.T + enc[pD + "bias"]).astype(np.float32)
        C *= 2
    # LSTM + skip
    T = x.shape[0]
    inp = x
    for l in range(2):
        wi, wh = enc[f"encoder.13.lstm.weight_ih_l{l}"], enc[f"encoder.13.lstm.weight_hh_l{l}"]
        b = enc[f"encoder.13.lstm.bias_ih_l{l}"] + enc[f"encoder.13.lstm.bias_hh_l{l}"]
        xg = inp @ wi.T + b
        hs, cs, ys = np.zeros(512, np.float32), np.zeros(512, np.float32), []
        for t in range(T):
            g = xg[t] + wh @ hs
            i, f, gg, o = g[:512], g[512:1024], g[1024:1536], g[1536:]
            sig = lambda v: 1.0 / (1.0 + np.exp(-v))
            cs = sig(f) * cs + sig(i) * np.tanh(gg)
            hs = (sig(o) * np.tanh(cs)).astype(np.float32)
            ys.append(hs + (x[t] if l == 1 else 0))
        inp = np.stack(ys).astype(np.float32) if l == 1 else np.stack([y for y in ys]).astype(np.float32)
    w15 = np.transpose(enc["encoder.15.weight"], (0, 2, 1)).reshape(128, 7 * 512)      # [128][tap*512 + c]
    emb = _im2col_mode0(inp, 7, True) @ w15.T + enc["encoder.15.bias"]
    # residual VQ
    resid = emb.astype(np.float32).copy()
    codes = np.zeros((T, 8), np.int64)
    for q in range(8):
        E = dec[f"quantizer.{q}.embed"]
        e2 = (E * E).sum(1)
        s = resid @ E.T
        a = (resid * resid).sum(1, keepdims=True)
        d = -((a - 2.0 * s) + e2[None])
        codes[:, q] = d.argmax(1)
        resid = resid - E[codes[:, q]]
    return emb, codes


def test_hip_encoder_dataflow_reproduces_the_oracle():
    dec, enc = encodec_state_dict(3), encodec_encoder_state_dict(4)
    orc = EncodecEncoderOracle(enc, dec)
    for name, row in (("encodec_enc_7777", 0), ("encodec_enc_12000", 1)):
        wav = case_wav(name)[row][:4001]                                 # 4001 samples: every stage length is odd / ragged
        emb, codes = emulate(enc, dec, wav)
        ref = orc.embeddings(wav[None])
        assert emb.shape == (ref.shape[2], 128) and emb.shape[0] == -(-len(wav) // 320)
        np.testing.assert_allclose(emb, ref[0].numpy().T, atol=2e-5, rtol=0)
        np.testing.assert_array_equal(codes, orc.quantize(ref)[0])


def test_hip_encoder_dataflow_short_input():
    """inputs shorter than the pads exercise the zero-extend-then-reflect rule of EncodecConv1d._pad1d"""
    dec, enc = encodec_state_dict(3), encodec_encoder_state_dict(4)
    orc = EncodecEncoderOracle(enc, dec)
    for n in (3, 7, 33):
        wav = case_wav("encodec_enc_7777")[0][:n]
        emb, _ = emulate(enc, dec, wav)
        np.testing.assert_allclose(emb, orc.embeddings(wav[None])[0].numpy().T, atol=2e-5, rtol=0)
