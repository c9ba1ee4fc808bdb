// Weight ingest: the reference state-dict (374 keys for 12 layers) arrives tensor by tensor through
// vx_load_tensor; vx_finalize_weights checks presence and shapes (load_state_dict(strict=True), utils/generation.py:79-83),
// allocates the arenas and derives every device image the kernels read: f16x2 / bf16x3 operand planes, packed decode images,
// the positional table (modules/embedding.py:75-91), the AdaLN stage projections (modules/transformer.py:96-100), the Vocos
// head's matrices (Vocos.from_pretrained, utils/generation.py:89) and the EnCodec decoder / encoder images.
#include "engine_ctx.h"

namespace {

int need(vx_ctx* c, const std::string& name, std::initializer_list<int64_t> shape) {
  auto it = c->w.find(name);
  if (it == c->w.end()) FAIL(VX_ENOTFOUND, "missing tensor '%s'", name.c_str());
  if (it->second.shape != std::vector<int64_t>(shape)) FAIL(VX_EINVAL, "tensor '%s' has the wrong shape", name.c_str());
  return VX_OK;
}

int pack(vx_ctx* c, const float* Wt, int N, int K, int Npad, float** out) {
  if (int e = dev_alloc(c, out, (size_t)Npad * K, false)) return e;
  launch_pack_weight(Wt, N, K, *out, Npad, c->stream);
  return VX_OK;
}

}  // namespace

extern "C" {

int vx_load_tensor(vx_ctx* c, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
  if (!c) return VX_EINVAL;
  if (!name || !data || !shape || ndim < 0 || ndim > 4) FAIL(VX_EINVAL, "bad tensor argument");
  if (c->finalized) FAIL(VX_ESTATE, "weights already finalized");
  HIPCHK(hipSetDevice(c->dev));
  Tensor t;
  t.n = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); t.n *= (size_t)shape[i]; }
  auto it = c->w.find(name);
  if (it != c->w.end()) { (void)hipFree(it->second.d); c->w.erase(it); }
  void* q = nullptr;
  HIPCHK(hipMalloc(&q, std::max<size_t>(t.n, 1) * sizeof(float)));
  t.d = reinterpret_cast<float*>(q);
  H2D(t.d, data, t.n * sizeof(float));
  c->w[name] = t;
  return VX_OK;
}

int vx_finalize_weights(vx_ctx* c) {
  if (!c) return VX_EINVAL;
  if (c->finalized) return VX_OK;
  HIPCHK(hipSetDevice(c->dev));
  const int NL = c->NL, d = D_MODEL, f = D_FF;
  // ---- presence + shape of the reference state-dict ----
  int e;
#define NEED(...) if ((e = need(c, __VA_ARGS__))) return e
  NEED("ar_text_embedding.word_embeddings.weight", {2048, d});
  NEED("nar_text_embedding.word_embeddings.weight", {2048, d});
  NEED("ar_audio_embedding.word_embeddings.weight", {AUDIO_VOCAB + 2, d});
  NEED("ar_language_embedding.word_embeddings.weight", {3, d});
  NEED("nar_language_embedding.word_embeddings.weight", {3, d});
  for (const char* a : {"ar_text_position.alpha", "ar_audio_position.alpha", "nar_text_position.alpha",
                        "nar_audio_position.alpha"})
    NEED(a, {1});
  NEED("ar_decoder.norm.weight", {d});
  NEED("ar_decoder.norm.bias", {d});
  NEED("ar_predict_layer.weight", {AR_LOGITS, d});
  NEED("nar_audio_embeddings.0.word_embeddings.weight", {AUDIO_VOCAB + 1, d});
  for (int j = 1; j < N_Q; ++j) NEED("nar_audio_embeddings." + std::to_string(j) + ".word_embeddings.weight", {AUDIO_VOCAB, d});
  for (int j = 0; j < N_Q - 1; ++j) {
    NEED("nar_predict_layers." + std::to_string(j) + ".weight", {AUDIO_VOCAB, d});
    NEED("nar_stage_embeddings." + std::to_string(j) + ".word_embeddings.weight", {1, d});
  }
  NEED("nar_decoder.norm.project_layer.weight", {2 * d, d});
  NEED("nar_decoder.norm.project_layer.bias", {2 * d});
  NEED("nar_decoder.norm.norm.weight", {d});
  NEED("nar_decoder.norm.norm.bias", {d});
  c->ar.resize(NL);
  c->nar.resize(NL);
  for (int which = 0; which < 2; ++which)
    for (int l = 0; l < NL; ++l) {
      const std::string p = std::string(which ? "nar" : "ar") + "_decoder.layers." + std::to_string(l) + ".";
      NEED(p + "self_attn.in_proj_weight", {3 * d, d});
      NEED(p + "self_attn.in_proj_bias", {3 * d});
      NEED(p + "self_attn.out_proj.weight", {d, d});
      NEED(p + "self_attn.out_proj.bias", {d});
      NEED(p + "linear1.weight", {f, d});
      NEED(p + "linear1.bias", {f});
      NEED(p + "linear2.weight", {d, f});
      NEED(p + "linear2.bias", {d});
      const std::string n1 = which ? p + "norm1.norm." : p + "norm1.", n2 = which ? p + "norm2.norm." : p + "norm2.";
      NEED(n1 + "weight", {d});
      NEED(n1 + "bias", {d});
      NEED(n2 + "weight", {d});
      NEED(n2 + "bias", {d});
      if (which) {
        NEED(p + "norm1.project_layer.weight", {2 * d, d});
        NEED(p + "norm1.project_layer.bias", {2 * d});
        NEED(p + "norm2.project_layer.weight", {2 * d, d});
        NEED(p + "norm2.project_layer.bias", {2 * d});
      }
      LayerW& L = which ? c->nar[l] : c->ar[l];
      L.in_w = W(c, p + "self_attn.in_proj_weight"); L.in_b = W(c, p + "self_attn.in_proj_bias");
      L.out_w = W(c, p + "self_attn.out_proj.weight"); L.out_b = W(c, p + "self_attn.out_proj.bias");
      L.l1_w = W(c, p + "linear1.weight"); L.l1_b = W(c, p + "linear1.bias");
      L.l2_w = W(c, p + "linear2.weight"); L.l2_b = W(c, p + "linear2.bias");
      L.n1_w = W(c, n1 + "weight"); L.n1_b = W(c, n1 + "bias");
      L.n2_w = W(c, n2 + "weight"); L.n2_b = W(c, n2 + "bias");
    }
#undef NEED

  // ---- geometry + arenas ----
  c->mbr = std::min(c->cfg.max_batch, MB);
  c->Tmax = c->cfg.max_text + 1 + c->cfg.max_prompt + c->cfg.max_new + 1;
  c->Mmax = (long)c->mbr * (c->cfg.max_text + c->cfg.max_prompt + c->cfg.max_new + 1);
  c->gen_stride = c->cfg.max_new;
  const long M = c->Mmax + 128;
  if ((e = dev_alloc(c, &c->fx, (size_t)M * d))) return e;
  // kernel selection (read once per context): the defaults are the measured best
  if (c->cfg.arith == VX_ARITH_F16X2) { c->gemm_mode = 0; c->attn_x3 = true; c->attn_h2 = true; }
  else if (c->cfg.arith == VX_ARITH_BF16X3) { c->gemm_mode = 1; c->attn_x3 = true; c->attn_h2 = false; }
  else if (c->cfg.arith == VX_ARITH_F32) { c->gemm_mode = 2; c->attn_x3 = false; }
  else if (c->cfg.arith != VX_ARITH_DEFAULT) FAIL(VX_EINVAL, "vx_config.arith must be 0..3");
  if (c->cfg.arith == VX_ARITH_DEFAULT) {      // the environment only speaks when the caller did not choose
  if (const char* ev = getenv("VX_GEMM_X3")) if (ev[0] == '1') c->gemm_mode = 1;
  if (const char* ev = getenv("VX_GEMM_F32")) if (ev[0] == '1') c->gemm_mode = 2;
  if (const char* ev = getenv("VX_ATTN_F32")) c->attn_x3 = !(ev[0] == '1');
  if (const char* ev = getenv("VX_ATTN_X3")) c->attn_h2 = !(ev[0] == '1');
  }
  if ((e = dev_alloc(c, &c->fxn, (size_t)M * d))) return e;
  if ((e = dev_alloc(c, &c->fqkv, (size_t)M * 3 * d))) return e;
  // in f16x2 mode the attention output and the FFN hidden activations only ever exist as operand planes (fa3 / fa3b)
  if (!(c->gemm_mode == 0 && c->attn_x3) && (e = dev_alloc(c, &c->fatt, (size_t)M * d))) return e;
  if (c->gemm_mode != 0 && (e = dev_alloc(c, &c->fffn, (size_t)M * f))) return e;
  // the fp32 activations of the range-guard fallback: allocated here, not inside the first request that falls back (a
  // synchronising hipMalloc of hundreds of MB in the timed path that could fail after finalize succeeded)
  if (range_guarded(c) && (e = ensure_f32_buffers(c))) return e;
  if ((e = dev_alloc(c, &c->fyemb, (size_t)M * d))) return e;
  if ((e = dev_alloc(c, &c->flogits, (size_t)((long)c->mbr * c->cfg.max_new + 128) * AUDIO_VOCAB))) return e;
  c->imeta_cap = std::max(M * 24, (long)c->cfg.max_batch * c->cfg.max_new * 12) + 65536;
  if ((e = dev_alloc(c, &c->imeta, (size_t)c->imeta_cap))) return e;
  const size_t cache = (size_t)NL * c->mbr * N_HEAD * c->Tmax * D_HEAD;
  // zero-initialised once: the fused dec_attn requests rows 0 .. DEC_ATTN_TILE - 1 of a (slot, head) stream before it knows the
  // context length (decode.hip) and, for a slot past the batch, the stream of slot 0 -- the values are discarded (the tile is
  // requested again, clamped, when the context is shorter), but they must be DEFINED memory: no tool flags an uninitialised read
  // and no stale NaN pattern of a previous owner of the pages can ever meet an arithmetic instruction.  Needs Tmax >= DEC_ATTN_TILE
  // (guard below; decode.hip static_asserts the tile size against it).
  if ((e = dev_alloc(c, &c->kc, cache, true))) return e;
  if ((e = dev_alloc(c, &c->vc, cache, true))) return e;
  if ((e = dev_alloc(c, &c->dh, (size_t)MB * d))) return e;
  if ((e = dev_alloc(c, &c->dh2, (size_t)MB * d))) return e;
  if (const char* ev = getenv("VX_SB_FUSE")) c->sb_fuse = !(ev[0] == '0');
  if ((e = dev_alloc(c, &c->xp, (size_t)MB * d))) return e;
  if ((e = dev_alloc(c, &c->xp_att, (size_t)MB * d))) return e;
  if ((e = dev_alloc(c, &c->xp4, (size_t)2 * MB * f))) return e;   // linear1's two split-K slabs, packed image
  if (const char* ev = getenv("VX_QKV_BALANCED")) c->qkv_bal = !(ev[0] == '0');
  static_assert(SK_QKV_BAL_Q == 8 && SK_QKV_BAL_KV == SK_QKV, "skinny_qkv_bal_kernel writes 8 q slabs and SK_QKV k / v slabs into p_qkv");
  if ((e = dev_alloc(c, &c->p_qkv, (size_t)std::max(SK_QKV, SK_QKV_BAL_Q) * MB * 3 * d))) return e;
  if ((e = dev_alloc(c, &c->p_o, (size_t)std::max(SK_OUT, SK_L2) * MB * d))) return e;
  if ((e = dev_alloc(c, &c->p_oh, (size_t)4 * N_HEAD * MB * d))) return e;      // per-head slabs of the fused out_proj; x up to 4 context splits (8 .. 16 rows)
  if (const char* ev = getenv("VX_FUSE_OUT")) c->fuse_out = !(ev[0] == '0');
  // the fused dec_attn requests the first DEC_ATTN_TILE rows of every (slot, head) stream before it knows the context length
  if (c->Tmax < DEC_ATTN_TILE) c->fuse_out = false;
  if (const char* ev = getenv("VX_BALANCE_ROWS")) c->balance_rows = !(ev[0] == '0');
  if (const char* ev = getenv("VX_NAR_TRIM")) c->nar_trim = ev[0] == '1';
  if (const char* ev = getenv("VX_GRAPH_MULTI")) c->graph_multi = !(ev[0] == '0');
  if (const char* ev = getenv("VX_SB_QKV")) c->sb_qkv_rows = atoi(ev);
  if (const char* ev = getenv("VX_SB_QKV_NSPLIT")) c->sb_qkv_nsplit = atoi(ev);
  if (const char* ev = getenv("VX_FUSE_SPLIT")) c->fuse_split = atoi(ev);
  if (const char* ev = getenv("VX_ATT_NSPLIT")) c->att_nsplit_force = std::max(0, std::min(16, atoi(ev)));
  if ((e = dev_alloc(c, &c->p_logits, (size_t)SK_PRED * MB * PRED_NPAD))) return e;
  if ((e = dev_alloc(c, &c->part_o, (size_t)MB * N_HEAD * 16 * D_HEAD))) return e;
  if ((e = dev_alloc(c, &c->part_ml, (size_t)MB * N_HEAD * 16 * 2))) return e;
  if ((e = dev_alloc(c, &c->qk_new, (size_t)MB * N_HEAD * 2 * D_HEAD))) return e;
  if ((e = dev_alloc(c, &c->d_logits, (size_t)MB * AR_LOGITS))) return e;
  if ((e = dev_alloc(c, &c->sum_logp, (size_t)MB))) return e;
  c->uniforms_cap = (long)(c->cfg.max_new + 2) * MB;
  if ((e = dev_alloc(c, &c->d_uniforms, (size_t)c->uniforms_cap))) return e;
  for (int** p : {&c->cur_tok, &c->cur_pos, &c->ctx_len, &c->n_gen, &c->active, &c->text_len, &c->force_tok, &c->n_active,
                  &c->slot_of})
    if ((e = dev_alloc(c, p, MB))) return e;
  if ((e = dev_alloc(c, &c->slot_meta, 4 * MB))) return e;
  if ((e = dev_alloc(c, &c->gen, (size_t)MB * c->gen_stride))) return e;

  // ---- positional table, built on the host exactly like modules/embedding.py:75-91 (fp32 ops in the same order) ----
  {
    c->pe_rows = std::max(4000, c->Tmax + 8);
    std::vector<float> pe((size_t)c->pe_rows * d);
    std::vector<float> div(d / 2);
    const float k = -(float)(log(10000.0) / d);            // python float math.log(10000.0)/d, then cast in the product
    for (int i = 0; i < d / 2; ++i) div[i] = expf((float)(2 * i) * k);
    for (int p = 0; p < c->pe_rows; ++p)
      for (int i = 0; i < d / 2; ++i) {
        const float a = (float)p * div[i];
        pe[(size_t)p * d + 2 * i] = sinf(a);
        pe[(size_t)p * d + 2 * i + 1] = cosf(a);
      }
    if ((e = dev_alloc(c, &c->pe, pe.size(), false))) return e;
    H2D(c->pe, pe.data(), pe.size() * sizeof(float));
  }
  // a caller-supplied table (built with torch on the host) overrides ours bit for bit
  if (const float* user_pe = W(c, "pe_table")) {
    const Tensor& t = c->w["pe_table"];
    if (t.shape.size() == 2 && t.shape[1] == d && t.shape[0] >= c->Tmax) { c->pe = const_cast<float*>(user_pe); c->pe_rows = (int)t.shape[0]; }
  }

  // ---- 16-bit operand planes of every transformer projection used on the full-sequence paths ----
  if ((e = dev_alloc(c, &c->range_flag, 1))) return e;
  if ((e = dev_alloc(c, &c->seed_dev, 1))) return e;
  if (c->gemm_mode != 2) {
    const int P = c->gemm_mode == 0 ? 2 : 3;
    unsigned* d_max = nullptr;
    if (c->gemm_mode == 0 && (e = dev_alloc(c, reinterpret_cast<int**>(&d_max), 1))) return e;
    auto split_w = [&](const float* Wt, int N, int K, unsigned short** out) -> int {
      if (int e2 = dev_alloc(c, out, (size_t)P * h2_plane(N, K, H2_TILE_W), false)) return e2;
      if (c->gemm_mode == 0) {
        // f16x2: the tensor's own power-of-two scale, max |w| * 2^shift in [16384, 32768) (vx_common.h); shift in [0, 24]
        unsigned bits = 0;
        HIPCHK(hipMemsetAsync(d_max, 0, sizeof(unsigned), c->stream));
        launch_absmax(Wt, (long)N * K, d_max, c->stream);
        D2H(&bits, d_max, sizeof(unsigned));
        SYNC();
        float mx;
        memcpy(&mx, &bits, sizeof mx);
        int shift = 24;
        if (mx > 0.f && isfinite(mx)) { int ex; (void)frexpf(mx, &ex); shift = std::max(0, std::min(24, 15 - ex)); }
        c->w_shift[*out] = shift;
        launch_split2h(Wt, K, N, K, nullptr, *out, h2_plane(N, K, H2_TILE_W), H2_TILE_W, c->range_flag, ldexpf(1.0f, shift), c->stream);
      } else launch_split3(Wt, K, N, K, nullptr, *out, (long)N * K, c->stream);
      return VX_OK;
    };
    for (int which = 0; which < 2; ++which)
      for (int l = 0; l < NL; ++l) {
        LayerW& L = which ? c->nar[l] : c->ar[l];
        if ((e = split_w(L.in_w, 3 * d, d, &L.in_w3))) return e;
        if ((e = split_w(L.out_w, d, d, &L.out_w3))) return e;
        if ((e = split_w(L.l1_w, f, d, &L.l1_w3))) return e;
        if ((e = split_w(L.l2_w, d, f, &L.l2_w3))) return e;
      }
    for (int j = 0; j < N_Q - 1; ++j)
      if ((e = split_w(W(c, "nar_predict_layers." + std::to_string(j) + ".weight"), AUDIO_VOCAB, d, &c->pred_w3[j]))) return e;
    if ((e = dev_alloc(c, &c->fa3, (size_t)P * (c->Mmax + 256) * f))) return e;       // zeroed: the pad rows of a last tile are read
    if (c->gemm_mode == 0 && (e = dev_alloc(c, &c->fa3b, (size_t)2 * (c->Mmax + 256) * f))) return e;
  }

  // ---- packed decode images of the AR stack ----
  for (int l = 0; l < NL; ++l) {
    LayerW& L = c->ar[l];
    if ((e = pack(c, L.in_w, 3 * d, d, 3 * d, &L.in_wp))) return e;
    if ((e = pack(c, L.out_w, d, d, d, &L.out_wp))) return e;
    if ((e = dev_alloc(c, &L.out_wh, (size_t)d * d, false))) return e;
    launch_pack_wo_heads(L.out_w, L.out_wh, c->stream);
    if ((e = dev_alloc(c, &L.l1_wp, (size_t)f * d, false))) return e;       // 16-row tile image (fused linear1)
    launch_pack_weight16(L.l1_w, f, d, L.l1_wp, c->stream);
    if ((e = pack(c, L.l2_w, d, f, d, &L.l2_wp))) return e;
  }
  if ((e = pack(c, W(c, "ar_predict_layer.weight"), AR_LOGITS, d, PRED_NPAD, &c->pred_wp))) return e;

  // ---- AdaLN projections of the 7 stage embeddings (modules/transformer.py:96-100), input independent ----
  {
    const int nnorm = 2 * NL + 1;
    if ((e = dev_alloc(c, &c->ada, (size_t)(N_Q - 1) * nnorm * 2 * d, false))) return e;
    for (int st = 0; st < N_Q - 1; ++st) {
      const float* emb = W(c, "nar_stage_embeddings." + std::to_string(st) + ".word_embeddings.weight");
      for (int n = 0; n < nnorm; ++n) {
        std::string p;
        if (n == 2 * NL) p = "nar_decoder.norm.project_layer.";
        else p = "nar_decoder.layers." + std::to_string(n / 2) + (n % 2 ? ".norm2" : ".norm1") + ".project_layer.";
        launch_gemv(W(c, p + "weight"), emb, W(c, p + "bias"), c->ada + ((size_t)st * nnorm + n) * 2 * d, 2 * d, d,
                    c->stream);
      }
    }
  }
  {
    std::vector<const float*> tabs(N_Q);
    for (int j = 0; j < N_Q; ++j) tabs[j] = W(c, "nar_audio_embeddings." + std::to_string(j) + ".word_embeddings.weight");
    float** tmp = nullptr;
    if ((e = dev_alloc(c, &tmp, N_Q, false))) return e;
    H2D((void*)tmp, tabs.data(), N_Q * sizeof(float*));
    c->nar_tabs_dev = const_cast<const float**>(tmp);
  }

  // ---- Vocos head (optional) ----
  if (c->cfg.with_vocos && c->w.count("vocos.head.out.weight")) {
    const int C = 384, H = 1152, NB = 1282, NBP = 1408, KP = 1312, NF = 1280;
#define NEED(...) if ((e = need(c, __VA_ARGS__))) return e
    NEED("vocos.feature_extractor.codebook_weights", {16384, 128});
    NEED("vocos.backbone.embed.weight", {C, 128, 7});
    NEED("vocos.backbone.embed.bias", {C});
    NEED("vocos.backbone.norm.scale.weight", {4, C});
    NEED("vocos.backbone.norm.shift.weight", {4, C});
    for (int i = 0; i < 8; ++i) {
      const std::string p = "vocos.backbone.convnext." + std::to_string(i) + ".";
      NEED(p + "dwconv.weight", {C, 1, 7});
      NEED(p + "dwconv.bias", {C});
      NEED(p + "norm.scale.weight", {4, C});
      NEED(p + "norm.shift.weight", {4, C});
      NEED(p + "pwconv1.weight", {H, C});
      NEED(p + "pwconv1.bias", {H});
      NEED(p + "pwconv2.weight", {C, H});
      NEED(p + "pwconv2.bias", {C});
      NEED(p + "gamma", {C});
    }
    NEED("vocos.backbone.final_layer_norm.weight", {C});
    NEED("vocos.backbone.final_layer_norm.bias", {C});
    NEED("vocos.head.out.weight", {NB, C});
    NEED("vocos.head.out.bias", {NB});
#undef NEED
    // embed conv weight (384,128,7) -> [384][tap*128 + c] to match the im2col rows
    {
      std::vector<float> w((size_t)C * 128 * 7), w2((size_t)C * 896);
      D2H(w.data(), W(c, "vocos.backbone.embed.weight"), w.size() * sizeof(float)); SYNC();
      for (int o = 0; o < C; ++o)
        for (int ch = 0; ch < 128; ++ch)
          for (int tap = 0; tap < 7; ++tap) w2[(size_t)o * 896 + tap * 128 + ch] = w[((size_t)o * 128 + ch) * 7 + tap];
      if ((e = dev_alloc(c, &c->vc_embed_w, w2.size(), false))) return e;
      H2D(c->vc_embed_w, w2.data(), w2.size() * sizeof(float));
    }
    // head weight/bias padded 1282 -> 1408 rows (GEMM N multiple of 128)
    if ((e = dev_alloc(c, &c->vc_head_w, (size_t)NBP * C))) return e;
    if ((e = dev_alloc(c, &c->vc_head_b, NBP))) return e;
    SYNC();
    HIPCHK(hipMemcpyAsync(c->vc_head_w, W(c, "vocos.head.out.weight"), (size_t)NB * C * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->vc_head_b, W(c, "vocos.head.out.bias"), (size_t)NB * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    // inverse real DFT (irfft n=1280, norm="backward") with the hann window folded in, as a [1280][1312] matrix:
    // frame[n] = win[n]/N * ( re0 + (-1)^n re_{N/2} + 2 sum_{k=1}^{N/2-1} re_k cos(2 pi k n/N) - im_k sin(2 pi k n/N) )
    {
      std::vector<float> dft((size_t)NF * KP, 0.f), win2(NF);
      const double PI = 3.14159265358979323846;
      for (int n = 0; n < NF; ++n) {
        const double wn = 0.5 - 0.5 * cos(2.0 * PI * n / NF);             // torch.hann_window(periodic=True)
        win2[n] = (float)((double)(float)wn * (double)(float)wn);
        const double sc = (double)(float)wn / NF;
        float* row = &dft[(size_t)n * KP];
        row[0] = (float)sc;
        row[640] = (float)(sc * ((n & 1) ? -1.0 : 1.0));
        for (int k = 1; k < 640; ++k) {
          const double ang = 2.0 * PI * (double)((long)k * n % NF) / NF;
          row[k] = (float)(2.0 * sc * cos(ang));
          row[641 + k] = (float)(-2.0 * sc * sin(ang));
        }
      }
      if ((e = dev_alloc(c, &c->vc_dft, dft.size(), false))) return e;
      if ((e = dev_alloc(c, &c->vc_win2, win2.size(), false))) return e;
      H2D(c->vc_dft, dft.data(), dft.size() * sizeof(float));
      H2D(c->vc_win2, win2.data(), win2.size() * sizeof(float));
    }
    c->v_rows_cap = std::max<long>((long)c->cfg.max_batch * c->cfg.max_new, 512);   // frames per decode pass (longer inputs: windows)
    const long R = c->v_rows_cap + 128;
    if ((e = dev_alloc(c, &c->vfeat, (size_t)R * 128))) return e;
    if ((e = dev_alloc(c, &c->vcol, (size_t)R * 896))) return e;
    if ((e = dev_alloc(c, &c->vx0, (size_t)R * C))) return e;
    if ((e = dev_alloc(c, &c->vx1, (size_t)R * C))) return e;
    if ((e = dev_alloc(c, &c->vhid, (size_t)R * H))) return e;
    if ((e = dev_alloc(c, &c->vo, (size_t)R * NBP))) return e;
    if ((e = dev_alloc(c, &c->vreim, (size_t)R * KP))) return e;
    if ((e = dev_alloc(c, &c->vframes, (size_t)R * NF))) return e;
    if ((e = dev_alloc(c, &c->vaudio, (size_t)R * 320))) return e;
    c->has_vocos = true;
  }
  // ---- EnCodec SEANet decoder (optional; data/tokenizer.py:95-96 path) ----
  if (c->cfg.with_encodec && c->w.count("encodec.decoder.0.weight")) {
    const int ratios[4] = {8, 5, 4, 2};
#define NEED(...) if ((e = need(c, __VA_ARGS__))) return e
    for (int q = 0; q < N_Q; ++q) NEED("encodec.quantizer." + std::to_string(q) + ".embed", {1024, 128});
    NEED("encodec.decoder.0.weight", {512, 128, 7});
    NEED("encodec.decoder.0.bias", {512});
    for (int l = 0; l < 2; ++l) {
      const std::string sfx = "_l" + std::to_string(l);
      NEED("encodec.decoder.1.lstm.weight_ih" + sfx, {2048, 512});
      NEED("encodec.decoder.1.lstm.weight_hh" + sfx, {2048, 512});
      NEED("encodec.decoder.1.lstm.bias_ih" + sfx, {2048});
      NEED("encodec.decoder.1.lstm.bias_hh" + sfx, {2048});
    }
    {
      int C = 512;
      for (int st = 0; st < 4; ++st) {
        const int r = ratios[st], O = C / 2;
        const std::string pT = "encodec.decoder." + std::to_string(3 + 3 * st), pR = "encodec.decoder." + std::to_string(4 + 3 * st);
        NEED(pT + ".weight", {C, O, 2 * r});
        NEED(pT + ".bias", {O});
        NEED(pR + ".block1.weight", {O / 2, O, 3});
        NEED(pR + ".block1.bias", {O / 2});
        NEED(pR + ".block3.weight", {O, O / 2, 1});
        NEED(pR + ".block3.bias", {O});
        NEED(pR + ".shortcut.weight", {O, O, 1});
        NEED(pR + ".shortcut.bias", {O});
        C = O;
      }
    }
    NEED("encodec.decoder.15.weight", {1, 32, 7});
    NEED("encodec.decoder.15.bias", {1});
#undef NEED
    auto fetch = [&](const std::string& name, std::vector<float>& host) -> int {
      const Tensor& t = c->w[name];
      host.resize(t.n);
      D2H(host.data(), t.d, t.n * sizeof(float)); SYNC();
      return VX_OK;
    };
    auto upload = [&](const std::vector<float>& host, float** dev) -> int {
      if (int e2 = dev_alloc(c, dev, host.size(), false)) return e2;
      H2D(*dev, host.data(), host.size() * sizeof(float));
      return VX_OK;
    };
    std::vector<float> w, w2, b, b2;
    // RVQ codebooks, concatenated [8*1024][128]
    if ((e = dev_alloc(c, &c->ec_codebook, (size_t)N_Q * 1024 * 128, false))) return e;
    for (int q = 0; q < N_Q; ++q)
      HIPCHK(hipMemcpyAsync(c->ec_codebook + (size_t)q * 1024 * 128, W(c, "encodec.quantizer." + std::to_string(q) + ".embed"), (size_t)1024 * 128 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    // first conv (512,128,7) -> [512][tap*128 + c]
    if ((e = fetch("encodec.decoder.0.weight", w))) return e;
    w2.assign((size_t)512 * 896, 0.f);
    for (int o = 0; o < 512; ++o)
      for (int ch = 0; ch < 128; ++ch)
        for (int tap = 0; tap < 7; ++tap) w2[(size_t)o * 896 + tap * 128 + ch] = w[((size_t)o * 128 + ch) * 7 + tap];
    if ((e = upload(w2, &c->ec_w0))) return e;
    for (int l = 0; l < 2; ++l) {
      const std::string sfx = "_l" + std::to_string(l);
      if ((e = fetch("encodec.decoder.1.lstm.bias_ih" + sfx, b))) return e;
      if ((e = fetch("encodec.decoder.1.lstm.bias_hh" + sfx, b2))) return e;
      for (size_t i = 0; i < b.size(); ++i) b[i] += b2[i];
      if ((e = upload(b, &c->ec_lstm_b[l]))) return e;
      if ((e = pack(c, W(c, "encodec.decoder.1.lstm.weight_hh" + sfx), 2048, 512, 2048, &c->ec_whh_p[l]))) return e;
    }
    {
      int C = 512;
      for (int st = 0; st < 4; ++st) {
        const int r = ratios[st], O = C / 2, K = 2 * r;
        const std::string pT = "encodec.decoder." + std::to_string(3 + 3 * st), pR = "encodec.decoder." + std::to_string(4 + 3 * st);
        // ConvTranspose1d weight (C, O, 2r) -> [(ph*O + o)][tap*C + c] = w[c][o][ph + tap*r]
        if ((e = fetch(pT + ".weight", w))) return e;
        w2.assign((size_t)r * O * 2 * C, 0.f);
        for (int ph = 0; ph < r; ++ph)
          for (int o = 0; o < O; ++o)
            for (int tap = 0; tap < 2; ++tap)
              for (int ch = 0; ch < C; ++ch)
                w2[((size_t)ph * O + o) * (2 * C) + tap * C + ch] = w[((size_t)ch * O + o) * K + ph + tap * r];
        if ((e = upload(w2, &c->ec_wT[st]))) return e;
        if ((e = fetch(pT + ".bias", b))) return e;
        b2.resize((size_t)r * O);
        for (int ph = 0; ph < r; ++ph)
          for (int o = 0; o < O; ++o) b2[(size_t)ph * O + o] = b[o];
        if ((e = upload(b2, &c->ec_bT[st]))) return e;
        // resblock conv k3 (O/2, O, 3) -> [O/2][tap*O + c]
        if ((e = fetch(pR + ".block1.weight", w))) return e;
        w2.assign((size_t)(O / 2) * 3 * O, 0.f);
        for (int o = 0; o < O / 2; ++o)
          for (int ch = 0; ch < O; ++ch)
            for (int tap = 0; tap < 3; ++tap) w2[(size_t)o * 3 * O + tap * O + ch] = w[((size_t)o * O + ch) * 3 + tap];
        if ((e = upload(w2, &c->ec_w1[st]))) return e;
        // resblock conv k1 (O, O/2, 1) -> [O][ldh], ldh = max(O/2, 32) (K of the GEMM must be a multiple of 32)
        const int ldh = std::max(O / 2, 32);
        if ((e = fetch(pR + ".block3.weight", w))) return e;
        w2.assign((size_t)O * ldh, 0.f);
        for (int o = 0; o < O; ++o)
          for (int ch = 0; ch < O / 2; ++ch) w2[(size_t)o * ldh + ch] = w[(size_t)o * (O / 2) + ch];
        if ((e = upload(w2, &c->ec_w3[st]))) return e;
        C = O;
      }
    }
    c->ec_frames_cap = (long)c->mbr * c->cfg.max_new;
    const size_t Fc = (size_t)c->ec_frames_cap + 8;
    if ((e = dev_alloc(c, &c->ec_e0, Fc * 128))) return e;
    if ((e = dev_alloc(c, &c->ec_x0, Fc * 512))) return e;
    if ((e = dev_alloc(c, &c->ec_y1, Fc * 512))) return e;
    if ((e = dev_alloc(c, &c->ec_y2, Fc * 512))) return e;
    if ((e = dev_alloc(c, &c->ec_xg, Fc * 2048))) return e;
    if ((e = dev_alloc(c, &c->ec_col, Fc * 30720))) return e;
    if ((e = dev_alloc(c, &c->ec_a, Fc * 10240))) return e;
    if ((e = dev_alloc(c, &c->ec_sc, Fc * 10240))) return e;
    if ((e = dev_alloc(c, &c->ec_out, Fc * 10240))) return e;
    if ((e = dev_alloc(c, &c->ec_h, Fc * 10240))) return e;
    if ((e = dev_alloc(c, &c->ec_audio, (size_t)c->mbr * c->cfg.max_new * 320))) return e;
    if ((e = dev_alloc(c, &c->ec_hp, (size_t)MB * 512))) return e;
    if ((e = dev_alloc(c, &c->ec_c, (size_t)MB * 512))) return e;
    if ((e = dev_alloc(c, &c->ec_pg, (size_t)2 * MB * 2048))) return e;
    c->has_encodec = true;
    // ---- encoder + RVQ encode (optional: needs the "encodec.encoder.*" tensors; data/tokenizer.py:92-111 path) ----
    if (c->w.count("encodec.encoder.0.weight")) {
      const int eratios[4] = {2, 4, 5, 8};
#define NEED(...) if ((e = need(c, __VA_ARGS__))) return e
      NEED("encodec.encoder.0.weight", {32, 1, 7});
      NEED("encodec.encoder.0.bias", {32});
      {
        int C = 32;
        for (int st = 0; st < 4; ++st) {
          const int r = eratios[st];
          const std::string pR = "encodec.encoder." + std::to_string(1 + 3 * st), pD = "encodec.encoder." + std::to_string(3 + 3 * st);
          NEED(pR + ".block1.weight", {C / 2, C, 3});
          NEED(pR + ".block1.bias", {C / 2});
          NEED(pR + ".block3.weight", {C, C / 2, 1});
          NEED(pR + ".block3.bias", {C});
          NEED(pR + ".shortcut.weight", {C, C, 1});
          NEED(pR + ".shortcut.bias", {C});
          NEED(pD + ".weight", {2 * C, C, 2 * r});
          NEED(pD + ".bias", {2 * C});
          C *= 2;
        }
      }
      for (int l = 0; l < 2; ++l) {
        const std::string sfx = "_l" + std::to_string(l);
        NEED("encodec.encoder.13.lstm.weight_ih" + sfx, {2048, 512});
        NEED("encodec.encoder.13.lstm.weight_hh" + sfx, {2048, 512});
        NEED("encodec.encoder.13.lstm.bias_ih" + sfx, {2048});
        NEED("encodec.encoder.13.lstm.bias_hh" + sfx, {2048});
      }
      NEED("encodec.encoder.15.weight", {128, 512, 7});
      NEED("encodec.encoder.15.bias", {128});
#undef NEED
      {
        int C = 32;
        for (int st = 0; st < 4; ++st) {
          const int r = eratios[st], K = 2 * r;
          const std::string pR = "encodec.encoder." + std::to_string(1 + 3 * st), pD = "encodec.encoder." + std::to_string(3 + 3 * st);
          // resblock conv k3 (C/2, C, 3) -> [C/2][tap*C + c]
          if ((e = fetch(pR + ".block1.weight", w))) return e;
          w2.assign((size_t)(C / 2) * 3 * C, 0.f);
          for (int o = 0; o < C / 2; ++o)
            for (int ch = 0; ch < C; ++ch)
              for (int tap = 0; tap < 3; ++tap) w2[(size_t)o * 3 * C + tap * C + ch] = w[((size_t)o * C + ch) * 3 + tap];
          if ((e = upload(w2, &c->en_w1[st]))) return e;
          // resblock conv k1 (C, C/2, 1) -> [C][ldh], ldh = max(C/2, 32)
          const int ldh = std::max(C / 2, 32);
          if ((e = fetch(pR + ".block3.weight", w))) return e;
          w2.assign((size_t)C * ldh, 0.f);
          for (int o = 0; o < C; ++o)
            for (int ch = 0; ch < C / 2; ++ch) w2[(size_t)o * ldh + ch] = w[(size_t)o * (C / 2) + ch];
          if ((e = upload(w2, &c->en_w3[st]))) return e;
          // strided conv (2C, C, 2r) -> [2C][tap*C + c]: the window of an output frame is 2r consecutive channels-last rows
          if ((e = fetch(pD + ".weight", w))) return e;
          w2.assign((size_t)2 * C * K * C, 0.f);
          for (int o = 0; o < 2 * C; ++o)
            for (int ch = 0; ch < C; ++ch)
              for (int tap = 0; tap < K; ++tap) w2[(size_t)o * K * C + tap * C + ch] = w[((size_t)o * C + ch) * K + tap];
          if ((e = upload(w2, &c->en_wd[st]))) return e;
          C *= 2;
        }
      }
      for (int l = 0; l < 2; ++l) {
        const std::string sfx = "_l" + std::to_string(l);
        if ((e = fetch("encodec.encoder.13.lstm.bias_ih" + sfx, b))) return e;
        if ((e = fetch("encodec.encoder.13.lstm.bias_hh" + sfx, b2))) return e;
        for (size_t i = 0; i < b.size(); ++i) b[i] += b2[i];
        if ((e = upload(b, &c->en_lstm_b[l]))) return e;
        if ((e = pack(c, W(c, "encodec.encoder.13.lstm.weight_hh" + sfx), 2048, 512, 2048, &c->en_whh_p[l]))) return e;
      }
      // last conv (128, 512, 7) -> [128][tap*512 + c]
      if ((e = fetch("encodec.encoder.15.weight", w))) return e;
      w2.assign((size_t)128 * 3584, 0.f);
      for (int o = 0; o < 128; ++o)
        for (int ch = 0; ch < 512; ++ch)
          for (int tap = 0; tap < 7; ++tap) w2[(size_t)o * 3584 + tap * 512 + ch] = w[((size_t)o * 512 + ch) * 7 + tap];
      if ((e = upload(w2, &c->en_w15))) return e;
      // |e_c|^2 of every codeword (EncodecEuclideanCodebook.quantize: embed.pow(2).sum(0))
      b.assign((size_t)N_Q * 1024, 0.f);
      for (int q = 0; q < N_Q; ++q) {
        if ((e = fetch("encodec.quantizer." + std::to_string(q) + ".embed", w))) return e;
        for (int cw = 0; cw < 1024; ++cw) {
          float acc = 0.f;
          for (int k = 0; k < 128; ++k) acc += w[(size_t)cw * 128 + k] * w[(size_t)cw * 128 + k];
          b[(size_t)q * 1024 + cw] = acc;
        }
      }
      if ((e = upload(b, &c->en_e2))) return e;
      if ((e = dev_alloc(c, &c->en_scores, Fc * 1024))) return e;
      {
        void* qp = nullptr;
        HIPCHK(hipMalloc(&qp, Fc * 8 * sizeof(long long)));
        c->allocs.push_back(qp);
        c->en_codes = reinterpret_cast<long long*>(qp);
      }
      c->has_encodec_enc = true;
    }
  }
  SYNC();
  HIPCHK(hipGetLastError());
  {
    bool raised = false;          // weights are scaled from their own maximum: only a non-finite weight can raise the flag here
    if ((e = take_range_flag(c, &raised))) return e;
    if (raised) FAIL(VX_EINVAL, "vx_finalize_weights: a projection weight is not finite (NaN / inf in the state-dict)");
  }
  c->finalized = true;
  return VX_OK;
}

}  // extern "C"
