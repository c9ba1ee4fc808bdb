// Vocoder drivers: the Vocos head (vocos.codes_to_features + vocos.decode, utils/generation.py:148-150) and the EnCodec 24 kHz
// SEANet decoder / encoder + RVQ (AudioTokenizer.decode / .encode, data/tokenizer.py:92-96); kernels in vocos.hip, encodec.hip,
// gemm_f32.hip and the skinny MFMA GEMM of decode.hip (LSTM recurrence).
#include "engine_ctx.h"

extern "C" {

int vx_vocos_decode(vx_ctx* c, const int64_t* codes, int32_t codes_stride, const int32_t* lens, int32_t batch,
                    int32_t bandwidth_id, float* audio, int64_t audio_stride) {
  if (!c || !codes || !lens || !audio) return VX_EINVAL;
  if (!c->finalized || !c->has_vocos) FAIL(VX_ESTATE, "Vocos weights not loaded");
  if (batch <= 0) FAIL(VX_EINVAL, "bad batch");
  if (bandwidth_id < 0 || bandwidth_id > 3) FAIL(VX_EINVAL, "bandwidth_id must be 0..3");
  HIPCHK(hipSetDevice(c->dev));
  const int C = 384, H = 1152, NBP = 1408, KP = 1312, NF = 1280;
  // The reference decodes any total length in one call (utils/generation.py:148-150, :271-273 for a whole long text).  The
  // arena holds `cap` frames, so the rows are cut into JOBS: a row that fits is one job; a longer row is cut into windows
  // whose centre [a, b) is decoded together with HALO frames of real context on each side.  Every op of the head is local in
  // time (9 convolutions of 7 taps = 27 frames of reach, per-frame LayerNorm / GEMMs, ISTFT overlap of 3 frames), so the
  // centre samples are the same floating-point operations in the same order as in a single full-length pass: bit-identical.
  constexpr int HALO = 32;
  const long cap = c->v_rows_cap;
  struct Job { int row, a, b, lo, hi; };
  std::vector<Job> jobs;
  for (int i = 0; i < batch; ++i) {
    const int T = lens[i];
    if (T < 0 || T > codes_stride) FAIL(VX_EINVAL, "row %d: bad length", i);
    if ((long)T * 320 > audio_stride) FAIL(VX_EINVAL, "audio_stride too small");
    for (long t = 0; t < (long)T * N_Q; ++t) {
      const int64_t v = codes[(long)i * codes_stride * N_Q + t];
      if (v < 0 || v >= AUDIO_VOCAB) FAIL(VX_EINVAL, "code out of range");
    }
    if (T == 0) continue;
    if (T <= cap) { jobs.push_back({i, 0, T, 0, T}); continue; }
    const int Wc = (int)cap - 2 * HALO;
    for (int a = 0; a < T; a += Wc) {
      const int bb = std::min(T, a + Wc);
      jobs.push_back({i, a, bb, std::max(0, a - HALO), std::min(T, bb + HALO)});
    }
  }
  hipStream_t st = c->stream;
  const std::string P = "vocos.backbone.";
  size_t j0 = 0;
  while (j0 < jobs.size()) {
    size_t j1 = j0;
    long R = 0;
    while (j1 < jobs.size() && R + (jobs[j1].hi - jobs[j1].lo) <= cap) { R += jobs[j1].hi - jobs[j1].lo; ++j1; }
    const int nj = (int)(j1 - j0);
    std::vector<int> seq_off(nj), seq_len(nj), row_t, row_len, cd;
    row_t.reserve(R); row_len.reserve(R); cd.reserve(R * N_Q);
    long off = 0;
    int maxT = 0;
    for (int j = 0; j < nj; ++j) {
      const Job& jb = jobs[j0 + j];
      const int T = jb.hi - jb.lo;
      seq_off[j] = (int)off; seq_len[j] = T; maxT = std::max(maxT, T);
      for (int t = 0; t < T; ++t) {
        row_t.push_back(t); row_len.push_back(T);
        for (int q = 0; q < N_Q; ++q) cd.push_back((int)codes[((long)jb.row * codes_stride + jb.lo + t) * N_Q + q]);
      }
      off += T;
    }
    MetaBuilder mb(c);
    const long o_off = mb.add(seq_off), o_len = mb.add(seq_len), o_rt = mb.add(row_t), o_rl = mb.add(row_len), o_cd = mb.add(cd);
    if (int e = upload_meta(c)) return e;
    launch_codebook_sum(mb.dev(o_cd), W(c, "vocos.feature_extractor.codebook_weights"), c->vfeat, (int)R, st);
    launch_im2col7(c->vfeat, 128, mb.dev(o_rt), mb.dev(o_rl), c->vcol, (int)R, st);
    gemm(c, c->vcol, 896, c->vc_embed_w, 896, W(c, P + "embed.bias"), nullptr, 0, nullptr, c->vx0, C, R, C, 896, ACT_NONE);
    launch_layernorm(c->vx0, C, c->vx0, C, (int)R, C, 1e-6f, nullptr, nullptr, W(c, P + "norm.scale.weight") + bandwidth_id * C,
                     W(c, P + "norm.shift.weight") + bandwidth_id * C, st);
    for (int i = 0; i < 8; ++i) {
      const std::string p = P + "convnext." + std::to_string(i) + ".";
      launch_dwconv7(c->vx0, W(c, p + "dwconv.weight"), W(c, p + "dwconv.bias"), mb.dev(o_rt), mb.dev(o_rl), c->vx1, (int)R, C, st);
      launch_layernorm(c->vx1, C, c->vx1, C, (int)R, C, 1e-6f, nullptr, nullptr, W(c, p + "norm.scale.weight") + bandwidth_id * C,
                       W(c, p + "norm.shift.weight") + bandwidth_id * C, st);
      gemm(c, c->vx1, C, W(c, p + "pwconv1.weight"), C, W(c, p + "pwconv1.bias"), nullptr, 0, nullptr, c->vhid, H, R, H, C, ACT_GELU);
      gemm(c, c->vhid, H, W(c, p + "pwconv2.weight"), H, W(c, p + "pwconv2.bias"), c->vx0, C, W(c, p + "gamma"), c->vx0, C, R, C, H,
           ACT_NONE);
    }
    launch_layernorm(c->vx0, C, c->vx1, C, (int)R, C, 1e-6f, W(c, P + "final_layer_norm.weight"), W(c, P + "final_layer_norm.bias"),
                     nullptr, nullptr, st);
    gemm(c, c->vx1, C, c->vc_head_w, C, c->vc_head_b, nullptr, 0, nullptr, c->vo, NBP, R, NBP, C, ACT_NONE);
    launch_istft_prep(c->vo, NBP, c->vreim, KP, (int)R, st);
    gemm(c, c->vreim, KP, c->vc_dft, KP, nullptr, nullptr, 0, nullptr, c->vframes, NF, R, NF, KP, ACT_NONE);
    // audio of job j lands packed at sample offset seq_off[j] * 320 (audio_stride 0 = packed)
    launch_overlap_add(c->vframes, NF, mb.dev(o_off), mb.dev(o_len), c->vc_win2, c->vaudio, 0, nj, maxT, st);
    for (int j = 0; j < nj; ++j) {
      const Job& jb = jobs[j0 + j];
      D2H(audio + (long)jb.row * audio_stride + (long)jb.a * 320, c->vaudio + ((long)seq_off[j] + (jb.a - jb.lo)) * 320, (size_t)(jb.b - jb.a) * 320 * sizeof(float));
    }
    SYNC();
    HIPCHK(hipGetLastError());
    j0 = j1;
  }
  return VX_OK;
}

// replaces: AudioTokenizer.decode -> codec.decode([(codes, None)]) (data/tokenizer.py:95-96): EnCodec 24 kHz SEANet decoder
int vx_encodec_decode(vx_ctx* c, const int64_t* codes, int32_t codes_stride, const int32_t* lens, int32_t batch,
                      float* audio, int64_t audio_stride) {
  if (!c || !codes || !lens || !audio) return VX_EINVAL;
  if (!c->finalized || !c->has_encodec) FAIL(VX_ESTATE, "EnCodec decoder weights not loaded");
  if (batch <= 0) FAIL(VX_EINVAL, "bad batch");
  HIPCHK(hipSetDevice(c->dev));
  hipStream_t st = c->stream;
  const int ratios[4] = {8, 5, 4, 2};
  for (int r0 = 0; r0 < batch; r0 += c->mbr) {
    const int nb = std::min(c->mbr, batch - r0);
    std::vector<int> seq_off(nb), seq_len(nb), cd;
    long F = 0;
    int maxT = 0;
    for (int i = 0; i < nb; ++i) {
      const int T = lens[r0 + i];
      if (T < 0 || T > c->cfg.max_new || T > codes_stride) FAIL(VX_EINVAL, "row %d: bad length", r0 + i);
      if ((long)T * 320 > audio_stride) FAIL(VX_EINVAL, "audio_stride too small");
      seq_off[i] = (int)F; seq_len[i] = T; maxT = std::max(maxT, T);
      for (int t = 0; t < T; ++t)
        for (int q = 0; q < N_Q; ++q) {
          const int64_t v = codes[((long)(r0 + i) * codes_stride + t) * N_Q + q];
          if (v < 0 || v >= AUDIO_VOCAB) FAIL(VX_EINVAL, "code out of range");
          cd.push_back((int)v);
        }
      F += T;
    }
    if (F == 0) continue;
    if (F > c->ec_frames_cap) FAIL(VX_EINVAL, "too many frames");
    MetaBuilder mb(c);
    const long o_off = mb.add(seq_off), o_len = mb.add(seq_len), o_cd = mb.add(cd);
    if (int e = upload_meta(c)) return e;
    const int* d_off = mb.dev(o_off);
    const int* d_len = mb.dev(o_len);
    // RVQ decode + first conv
    launch_codebook_sum(mb.dev(o_cd), c->ec_codebook, c->ec_e0, (int)F, st);
    launch_im2col_seq(c->ec_e0, 128, 7, 0, 0, d_off, d_len, 1, c->ec_col, 896, nb, maxT, st);
    gemm(c, c->ec_col, 896, c->ec_w0, 896, W(c, "encodec.decoder.0.bias"), nullptr, 0, nullptr, c->ec_x0, 512, F, 512, 896,
         ACT_NONE);
    // 2-layer LSTM + skip: input projections as one GEMM per layer, the recurrence on the skinny MFMA GEMM
    const float* lin = c->ec_x0;
    for (int l = 0; l < 2; ++l) {
      const std::string sfx = "_l" + std::to_string(l);
      gemm(c, lin, 512, W(c, "encodec.decoder.1.lstm.weight_ih" + sfx), 512, c->ec_lstm_b[l], nullptr, 0, nullptr, c->ec_xg,
           2048, F, 2048, 512, ACT_NONE);
      HIPCHK(hipMemsetAsync(c->ec_hp, 0, (size_t)MB * 512 * sizeof(float), st));
      HIPCHK(hipMemsetAsync(c->ec_c, 0, (size_t)MB * 512 * sizeof(float), st));
      float* yout = l == 0 ? c->ec_y1 : c->ec_y2;
      {
        ProfScope ps(c, 5);                        // the recurrence of one layer: maxT dependent (GEMV-like GEMM | cell) pairs
        if (c->prof_on) c->prof[5].bytes += (double)maxT;        // "bytes" of class 5 = recurrence steps
        for (int t = 0; t < maxT; ++t) {
          launch_skinny_gemm(c->ec_whh_p[l], c->ec_hp, c->ec_pg, 2048, 512, 2, st);
          launch_lstm_cell(c->ec_pg, 2, c->ec_xg, d_off, d_len, t, c->ec_c, c->ec_hp, yout, l == 1 ? c->ec_x0 : nullptr, nb, st);
        }
      }
      lin = yout;
    }
    // 4 x [ELU, ConvTranspose1d, ResnetBlock]
    const float* cur = c->ec_y2;
    int C = 512;
    long R = 1;
    for (int s4 = 0; s4 < 4; ++s4) {
      const int r = ratios[s4], O = C / 2;
      launch_im2col_seq(cur, C, 2, 1, 1, d_off, d_len, (int)R, c->ec_col, 2 * C, nb, (long)maxT * R, st);
      gemm(c, c->ec_col, 2 * C, c->ec_wT[s4], 2 * C, c->ec_bT[s4], nullptr, 0, nullptr, c->ec_a, r * O, F * R, r * O, 2 * C,
           ACT_NONE);
      R *= r;
      C = O;
      const std::string pR = "encodec.decoder." + std::to_string(4 + 3 * s4);
      const long M = F * R;
      const int ldh = std::max(C / 2, 32);
      gemm(c, c->ec_a, C, W(c, pR + ".shortcut.weight"), C, W(c, pR + ".shortcut.bias"), nullptr, 0, nullptr, c->ec_sc, C, M, C,
           C, ACT_NONE);
      launch_im2col_seq(c->ec_a, C, 3, 0, 1, d_off, d_len, (int)R, c->ec_col, 3 * C, nb, (long)maxT * R, st);
      if (ldh != C / 2) HIPCHK(hipMemsetAsync(c->ec_h, 0, (size_t)M * ldh * sizeof(float), st));
      gemm(c, c->ec_col, 3 * C, c->ec_w1[s4], 3 * C, W(c, pR + ".block1.bias"), nullptr, 0, nullptr, c->ec_h, ldh, M, C / 2,
           3 * C, ACT_ELU);
      gemm(c, c->ec_h, ldh, c->ec_w3[s4], ldh, W(c, pR + ".block3.bias"), c->ec_sc, C, nullptr, c->ec_out, C, M, C, ldh,
           ACT_NONE);
      cur = c->ec_out;
    }
    const long astride = (long)c->cfg.max_new * 320;
    launch_final_conv(cur, W(c, "encodec.decoder.15.weight"), W(c, "encodec.decoder.15.bias"), d_off, d_len, (int)R,
                      c->ec_audio, astride, nb, (long)maxT * R, st);
    for (int i = 0; i < nb; ++i)
      D2H(audio + (long)(r0 + i) * audio_stride, c->ec_audio + (long)i * astride, (size_t)seq_len[i] * 320 * sizeof(float));
    SYNC();
    HIPCHK(hipGetLastError());
  }
  return VX_OK;
}

// replaces: AudioTokenizer.encode -> codec.encode(wav) (data/tokenizer.py:92-111, called by tokenize_audio for prompt
// enrolment, utils/prompt_making.py:57-84): EnCodec 24 kHz SEANet encoder + residual VQ at 6 kbps (8 codebooks).
// wav [batch][wav_stride] fp32 mono 24 kHz, lens [batch] samples -> codes [batch][codes_stride][8], out_lens = ceil(len / 320).
int vx_encodec_encode(vx_ctx* c, const float* wav, int64_t wav_stride, const int32_t* lens, int32_t batch,
                      int64_t* codes, int32_t codes_stride, int32_t* out_lens) {
  if (!c || !wav || !lens || !codes || !out_lens) return VX_EINVAL;
  if (!c->finalized || !c->has_encodec_enc) FAIL(VX_ESTATE, "EnCodec encoder weights not loaded");
  if (batch <= 0) FAIL(VX_EINVAL, "bad batch");
  HIPCHK(hipSetDevice(c->dev));
  hipStream_t st = c->stream;
  const int ratios[4] = {2, 4, 5, 8};
  const long sample_cap = std::min<long>((long)c->cfg.max_new * 320, c->ec_frames_cap * 320);
  for (int r0 = 0; r0 < batch; r0 += c->mbr) {
    const int nb = std::min(c->mbr, batch - r0);
    // stage lengths per sequence: L -> ceil(L/2) -> ceil(/4) -> ceil(/5) -> ceil(/8) = frames
    std::vector<int> seq_off(nb), seq_len(nb), one_off, one_len;
    std::vector<std::array<long, 5>> Ls(nb);
    long F = 0;
    int maxT = 0;
    for (int i = 0; i < nb; ++i) {
      const long L = lens[r0 + i];
      if (L <= 0 || L > sample_cap || L > wav_stride) FAIL(VX_EINVAL, "row %d: bad length %ld (cap %ld samples)", r0 + i, L, sample_cap);
      Ls[i][0] = L;
      for (int s4 = 0; s4 < 4; ++s4) Ls[i][s4 + 1] = (Ls[i][s4] + ratios[s4] - 1) / ratios[s4];
      const int T = (int)Ls[i][4];
      if (T > codes_stride) FAIL(VX_EINVAL, "codes_stride too small");
      seq_off[i] = (int)F; seq_len[i] = T; maxT = std::max(maxT, T);
      F += T;
      for (int s4 = 0; s4 < 4; ++s4) { one_off.push_back(0); one_len.push_back((int)Ls[i][s4]); }   // resblock im2col of stage s4
    }
    if (F > c->ec_frames_cap) FAIL(VX_EINVAL, "too many frames");
    MetaBuilder mb(c);
    const long o_off = mb.add(seq_off), o_len = mb.add(seq_len), o_1off = mb.add(one_off), o_1len = mb.add(one_len);
    if (int e = upload_meta(c)) return e;
    const int* d_off = mb.dev(o_off);
    const int* d_len = mb.dev(o_len);
    // ---- convolutional stack, one sequence at a time (prompts are few and long; the arena is reused) ----
    for (int i = 0; i < nb; ++i) {
      H2D(c->ec_audio, wav + (long)(r0 + i) * wav_stride, (size_t)Ls[i][0] * sizeof(float));
      launch_enc_first_conv(c->ec_audio, Ls[i][0], W(c, "encodec.encoder.0.weight"), W(c, "encodec.encoder.0.bias"), c->ec_a, st);
      int C = 32;
      for (int s4 = 0; s4 < 4; ++s4) {
        const int r = ratios[s4];
        const long Lc = Ls[i][s4], n_out = Ls[i][s4 + 1];
        const std::string pR = "encodec.encoder." + std::to_string(1 + 3 * s4), pD = "encodec.encoder." + std::to_string(3 + 3 * s4);
        const int ldh = std::max(C / 2, 32);
        // ResnetBlock: shortcut(x) + conv_k1(ELU(conv_k3(ELU(x))))
        gemm(c, c->ec_a, C, W(c, pR + ".shortcut.weight"), C, W(c, pR + ".shortcut.bias"), nullptr, 0, nullptr, c->ec_sc, C, Lc, C,
             C, ACT_NONE);
        launch_im2col_seq(c->ec_a, C, 3, 0, 1, mb.dev(o_1off) + i * 4 + s4, mb.dev(o_1len) + i * 4 + s4, 1, c->ec_col, 3 * C, 1, Lc, st);
        if (ldh != C / 2) HIPCHK(hipMemsetAsync(c->ec_h, 0, (size_t)Lc * ldh * sizeof(float), st));
        gemm(c, c->ec_col, 3 * C, c->en_w1[s4], 3 * C, W(c, pR + ".block1.bias"), nullptr, 0, nullptr, c->ec_h, ldh, Lc, C / 2, 3 * C,
             ACT_ELU);
        gemm(c, c->ec_h, ldh, c->en_w3[s4], ldh, W(c, pR + ".block3.bias"), c->ec_sc, C, nullptr, c->ec_out, C, Lc, C, ldh, ACT_NONE);
        // ELU + Conv1d(C, 2C, k = 2r, stride r), causal: left pad r, right pad to a whole frame (both reflect); the window
        // of output frame t' is rows [t' r, (t' + 2) r) of the padded copy -> a GEMM with overlapping A rows (lda = r C)
        const long rows = (n_out + 1) * r, extra = n_out * r - Lc;
        const long max_pad = std::max<long>(r, extra);
        const long Le = Lc <= max_pad ? Lc + (max_pad - Lc + 1) : Lc;         // EncodecConv1d._pad1d: short inputs are zero-extended
        launch_enc_pad_elu(c->ec_out, Lc, Le, C, r, rows, c->ec_col, st);
        float* dst = s4 < 3 ? c->ec_a : c->ec_x0 + (size_t)seq_off[i] * 512;
        gemm(c, c->ec_col, r * C, c->en_wd[s4], 2 * r * C, W(c, pD + ".bias"), nullptr, 0, nullptr, dst, 2 * C, n_out, 2 * C, 2 * r * C,
             ACT_NONE);
        C *= 2;
      }
    }
    // ---- 2-layer LSTM + skip on the packed frames (all sequences in lock-step, as in the decoder) ----
    const float* lin = c->ec_x0;
    for (int l = 0; l < 2; ++l) {
      const std::string sfx = "_l" + std::to_string(l);
      gemm(c, lin, 512, W(c, "encodec.encoder.13.lstm.weight_ih" + sfx), 512, c->en_lstm_b[l], nullptr, 0, nullptr, c->ec_xg, 2048, F,
           2048, 512, ACT_NONE);
      HIPCHK(hipMemsetAsync(c->ec_hp, 0, (size_t)MB * 512 * sizeof(float), st));
      HIPCHK(hipMemsetAsync(c->ec_c, 0, (size_t)MB * 512 * sizeof(float), st));
      float* yout = l == 0 ? c->ec_y1 : c->ec_y2;
      for (int t = 0; t < maxT; ++t) {
        launch_skinny_gemm(c->en_whh_p[l], c->ec_hp, c->ec_pg, 2048, 512, 2, st);
        launch_lstm_cell(c->ec_pg, 2, c->ec_xg, d_off, d_len, t, c->ec_c, c->ec_hp, yout, l == 1 ? c->ec_x0 : nullptr, nb, st);
      }
      lin = yout;
    }
    // ---- ELU + Conv1d(512, 128, k7) -> embeddings [F][128] ----
    launch_im2col_seq(c->ec_y2, 512, 7, 0, 1, d_off, d_len, 1, c->ec_col, 3584, nb, maxT, st);
    gemm(c, c->ec_col, 3584, c->en_w15, 3584, W(c, "encodec.encoder.15.bias"), nullptr, 0, nullptr, c->ec_e0, 128, F, 128, 3584, ACT_NONE);
    // ---- residual VQ, 8 codebooks: scores = r . E_q^T, argmax of -(|r|^2 - 2 s + |e|^2), r -= E_q[code] ----
    for (int q = 0; q < N_Q; ++q) {
      const float* Eq = c->ec_codebook + (size_t)q * 1024 * 128;
      gemm(c, c->ec_e0, 128, Eq, 128, nullptr, nullptr, 0, nullptr, c->en_scores, 1024, F, 1024, 128, ACT_NONE);
      launch_rvq_select(c->ec_e0, c->en_scores, c->en_e2 + (size_t)q * 1024, Eq, c->en_codes, q, F, st);
    }
    std::vector<long long> hc((size_t)F * 8);
    D2H(hc.data(), c->en_codes, hc.size() * sizeof(long long));
    SYNC();
    HIPCHK(hipGetLastError());
    for (int i = 0; i < nb; ++i) {
      out_lens[r0 + i] = seq_len[i];
      for (int t = 0; t < seq_len[i]; ++t)
        for (int q = 0; q < N_Q; ++q)
          codes[((long)(r0 + i) * codes_stride + t) * N_Q + q] = (int64_t)hc[((size_t)seq_off[i] + t) * 8 + q];
    }
  }
  return VX_OK;
}

}  // extern "C"
