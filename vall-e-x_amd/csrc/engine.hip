// Engine: context, weight ingest, arenas, launch sequences and the C ABI (include/vallex_hip.h).
// Host-side counterpart of VALLE.inference (models/vallex.py:458-686) + the Vocos call of
// utils/generation.py:148-150; every hot op is a hand-written gfx950 kernel from the sibling .hip files.
#include "engine_ctx.h"

namespace {
// the message of a failed vx_create (no context to hang it on): per thread, contexts are created from several host threads
// (bench.py --contexts) and vx_last_error(NULL) must not read a string another thread is assigning
thread_local std::string g_create_err;
}  // namespace

namespace vxe {

const float* W(vx_ctx* c, const std::string& name) {
  auto it = c->w.find(name);
  return it == c->w.end() ? nullptr : it->second.d;
}

// ---- pinned transfer ring (engine_ctx.h: PinRing) ----------------------------------------------------------
static int ring_slot(vx_ctx* c, size_t n, char** slot) {
  PinRing& r = c->ring;
  if (!r.base) FAIL(VX_ESTATE, "the context has no pinned transfer ring");
  const size_t a = (n + 255) & ~(size_t)255;
  if (r.head + a > r.cap) if (int e = xfer_sync(c)) return e;
  *slot = r.base + r.head;
  r.head += a;
  return VX_OK;
}

int xfer_h2d(vx_ctx* c, void* dst_dev, const void* src_host, size_t bytes) {
  const char* s = static_cast<const char*>(src_host);
  char* d = static_cast<char*>(dst_dev);
  while (bytes) {
    const size_t n = std::min(bytes, XFER_CHUNK);
    char* slot = nullptr;
    if (int e = ring_slot(c, n, &slot)) return e;
    memcpy(slot, s, n);
    HIPCHK(hipMemcpyAsync(d, slot, n, hipMemcpyHostToDevice, c->stream));
    s += n; d += n; bytes -= n;
  }
  return VX_OK;
}

int xfer_d2h(vx_ctx* c, void* dst_host, const void* src_dev, size_t bytes) {
  char* d = static_cast<char*>(dst_host);
  const char* s = static_cast<const char*>(src_dev);
  while (bytes) {
    const size_t n = std::min(bytes, XFER_CHUNK);
    char* slot = nullptr;
    if (int e = ring_slot(c, n, &slot)) return e;
    HIPCHK(hipMemcpyAsync(slot, s, n, hipMemcpyDeviceToHost, c->stream));
    c->ring.pend.push_back({d, slot, n});
    s += n; d += n; bytes -= n;
  }
  return VX_OK;
}

int xfer_sync(vx_ctx* c) {
  const hipError_t e = hipStreamSynchronize(c->stream);
  PinRing& r = c->ring;
  if (e == hipSuccess) {
    size_t total = 0;
    for (const PinRing::Pend& p : r.pend) total += p.n;
    if (total < ((size_t)4 << 20)) {
      for (const PinRing::Pend& p : r.pend) memcpy(p.dst, p.src, p.n);
    } else {
      // a large delivery (the audio of a batch: 24.6 MB for 32 x 8 s) is cut into ~1 MiB pieces and copied by four host threads: one
      // thread moves ~10 GB/s out of the ring, i.e. 2.5 ms behind a Vocos pass whose GEMMs take 4 ms
      struct Piece { char* dst; const char* src; size_t n; };
      std::vector<Piece> pieces;
      for (const PinRing::Pend& p : r.pend)
        for (size_t o = 0; o < p.n; o += (size_t)1 << 20)
          pieces.push_back({static_cast<char*>(p.dst) + o, p.src + o, std::min((size_t)1 << 20, p.n - o)});
      constexpr int NT = 4;
      std::thread th[NT - 1];
      auto work = [&pieces](int k) { for (size_t i = k; i < pieces.size(); i += NT) memcpy(pieces[i].dst, pieces[i].src, pieces[i].n); };
      bool started[NT - 1] = {};
      for (int k = 1; k < NT; ++k) {
        try { th[k - 1] = std::thread(work, k); started[k - 1] = true; }
        catch (...) {}                                  // no thread to be had: this share is copied here, below (nothing crosses the C ABI)
      }
      work(0);
      for (int k = 1; k < NT; ++k) {
        if (started[k - 1]) th[k - 1].join();
        else work(k);
      }
    }
  }
  r.pend.clear();
  r.head = 0;
  HIPCHK(e);
  return VX_OK;
}

// ---- int metadata upload ---------------------------------------------------------------------------------
int upload_meta(vx_ctx* c) {
  if ((long)c->hmeta.size() > c->imeta_cap) FAIL(VX_EINVAL, "row metadata overflow (%zu > %ld)", c->hmeta.size(), c->imeta_cap);
  H2D(c->imeta, c->hmeta.data(), c->hmeta.size() * sizeof(int));      // staged: the host vector is free for the next call
  return VX_OK;
}

int tap_store(vx_ctx* c, const std::string& name, const float* src, size_t n) {
  if (!c->cfg.debug_taps) return VX_OK;
  Tensor& t = c->taps[name];
  if (t.n < n) {
    void* q = nullptr;
    HIPCHK(hipMalloc(&q, n * sizeof(float)));
    c->allocs.push_back(q);
    t.d = reinterpret_cast<float*>(q);
  }
  t.n = n;
  HIPCHK(hipMemcpyAsync(t.d, src, n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  return VX_OK;
}

// ---- dense helpers -----------------------------------------------------------------------------------------
void gemm(vx_ctx* c, const float* A, int lda, const float* Wt, int ldw, const float* bias, const float* resid, int ldr,
          const float* colscale, float* C, int ldc, long M, int N, int K, int act, const int* gather, int cls, const int* resid_rows) {
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = Wt; g.ldw = ldw; g.bias = bias; g.resid = resid; g.ldr = ldr; g.colscale = colscale;
  g.C = C; g.ldc = ldc; g.M = (int)M; g.N = N; g.K = K; g.act = act; g.row_gather = gather; g.resid_rows = resid_rows;
  ProfScope ps(c, cls);               // class 2 = transformer projections (proj), 4 = the fp32 GEMMs of Vocos / EnCodec
  if (c->prof_on) c->prof[cls].bytes += 2.0 * (double)M * N * K;      // flops for this class
  launch_gemm_f32(g, c->stream);
}

// transformer projection (F.linear of modules/activation.py:144,166, modules/transformer.py:371-373, models/vallex.py:677)
// f16x2 mode extras: `a_pre` = A operand already in plane form (skip the split pass); `out_pl` = write the result as the next
// GEMM's A planes (K = N) instead of fp32 rows (C may then be null).
void proj(vx_ctx* c, const float* A, int lda, const float* Wf, const unsigned short* W3, const float* bias,
          const float* resid, int ldr, float* C, int ldc, long M, int N, int K, int act, const int* gather = nullptr,
          const unsigned short* a_pre = nullptr, unsigned short* out_pl = nullptr, const int* resid_rows = nullptr) {
  if (c->gemm_mode == 2 || !W3) {
    gemm(c, A, lda, Wf, K, bias, resid, ldr, nullptr, C, ldc, M, N, K, act, gather, 2, resid_rows);
    return;
  }
  const long a_plane = c->gemm_mode == 0 ? h2_plane(M, K, H2_TILE_A) : (long)M * K;
  if (!a_pre) {
    if (c->gemm_mode == 0) launch_split2h(A, lda, M, K, gather, c->fa3, a_plane, H2_TILE_A, c->range_flag, H2_ACT_SCALE, c->stream);
    else launch_split3(A, lda, M, K, gather, c->fa3, a_plane, c->stream);
  }
  GemmX3Args g{};
  g.A = a_pre ? a_pre : c->fa3; g.a_plane = a_plane; g.W = W3; g.w_plane = c->gemm_mode == 0 ? h2_plane(N, K, H2_TILE_W) : (long)N * K;
  g.bias = bias; g.resid = resid; g.ldr = ldr;
  g.colscale = nullptr; g.C = C; g.ldc = ldc; g.M = (int)M; g.N = N; g.K = K; g.act = act;
  g.out_planes = out_pl; g.out_plane = out_pl ? h2_plane(M, N, H2_TILE_A) : 0; g.range_flag = c->range_flag;
  g.resid_rows = resid_rows;                     // f16x2 kernel only (the one mode that trims rows)
  if (c->gemm_mode == 0) g.descale = ldexpf(1.0f, -(H2_ACT_SHIFT + c->w_shift.at(W3)));
  ProfScope ps(c, 2);
  if (c->prof_on) c->prof[2].bytes += 2.0 * (double)M * N * K;
  if (c->gemm_mode == 0) launch_gemm_f16x2(g, c->stream);
  else if (M >= 1024) launch_gemm_bf16x3_dma(g, c->stream);        // 256-row tiles: not for short row sets
  else launch_gemm_bf16x3(g, c->stream);
}

// Row trimming of the LAST decoder layer of a NAR stage (f16x2 mode, and since round 6 the reference-arithmetic fp32 mode): only the generated frames of every sequence reach a predict
// layer (models/vallex.py:672-679), so behind the K / V projection -- which attention needs for ALL rows -- the layer only has to
// produce those rows: attention queries, out_proj, norm2 and the FFN run on the Mc = sum T_b compacted rows.  Every op of the
// block treats rows independently, so each kept row goes through exactly the arithmetic it would see untrimmed: same ids, same
// logits, bit for bit.  The compacted residual stream lives in c->fxn (f16x2 mode: unused otherwise) or in the QKV buffer (fp32 mode).
struct Trim {
  long Mc;               // kept rows
  const int* q_first;    // [batch] first kept sequence-local row (S + Tp)
  const int* c_off;      // [batch] first compacted row of the sequence
  const int* rows;       // [Mc] packed row of every compacted row (residual gather)
  double attn_flops;     // 4 * T_b * L_b * 1024 summed: the queries that are still computed
};

// one pre-norm block on packed rows (modules/transformer.py:296-302 / :337-347) -- shared by AR prefill and NAR
int full_layer(vx_ctx* c, const LayerW& L, long M, const int* seq_off, const int* seq_len, const int* prefix_len,
               int batch, int max_len, const float* ada1, const float* ada2, float* kcl, float* vcl,
               const int* row_b, const int* row_t, double attn_flops, const Trim* tr = nullptr) {
  // f16x2 mode: every producer of a GEMM operand (the two LayerNorms, the attention, linear1's epilogue) writes the operand
  // planes itself -- no fp32 round trip of the normalised / attended / hidden activations and no split pass on any edge.
  const bool pl = c->gemm_mode == 0;
  const long pl1024 = h2_plane(M, D_MODEL, H2_TILE_A);
  launch_layernorm(c->fx, D_MODEL, pl ? nullptr : c->fxn, D_MODEL, (int)M, D_MODEL, LN_EPS, L.n1_w, L.n1_b, ada1,
                   ada1 ? ada1 + D_MODEL : nullptr, c->stream, pl ? c->fa3 : nullptr, pl1024, c->range_flag);
  proj(c, c->fxn, D_MODEL, L.in_w, L.in_w3, L.in_b, nullptr, 0, c->fqkv, 3 * D_MODEL, M, 3 * D_MODEL, D_MODEL, ACT_NONE, nullptr,
       pl ? c->fa3 : nullptr);
  if (kcl) launch_kv_scatter(c->fqkv, row_b, row_t, (int)M, kcl, vcl, c->Tmax, c->stream);
  const bool att_pl = pl && c->attn_x3;
  if (tr && !pl) {
    // reference arithmetic (fp32 projections + fp32 attention, round 6): the same trimming on fp32 rows.  The compacted residual
    // stream lives in the QKV buffer, which is free once the attention has read it (stream order); fatt / fxn / fffn hold the
    // compacted attention output, normalised rows and hidden activations.  gemm_f32 reads the residual of out_proj through the row
    // map (GemmArgs.resid_rows); every kept row goes through the arithmetic it would see untrimmed.
    float* xc = c->fqkv;
    {
      ProfScope ps(c, 3);
      if (c->prof_on) c->prof[3].bytes += tr->attn_flops;
      launch_attn_full(c->fqkv, c->fatt, seq_off, seq_len, prefix_len, batch, max_len, c->stream, tr->q_first, tr->c_off);
    }
    proj(c, c->fatt, D_MODEL, L.out_w, L.out_w3, L.out_b, c->fx, D_MODEL, xc, D_MODEL, tr->Mc, D_MODEL, D_MODEL, ACT_NONE, nullptr, nullptr,
         nullptr, tr->rows);
    launch_layernorm(xc, D_MODEL, c->fxn, D_MODEL, (int)tr->Mc, D_MODEL, LN_EPS, L.n2_w, L.n2_b, ada2, ada2 ? ada2 + D_MODEL : nullptr,
                     c->stream);
    proj(c, c->fxn, D_MODEL, L.l1_w, L.l1_w3, L.l1_b, nullptr, 0, c->fffn, D_FF, tr->Mc, D_FF, D_MODEL, ACT_RELU);
    proj(c, c->fffn, D_FF, L.l2_w, L.l2_w3, L.l2_b, xc, D_MODEL, xc, D_MODEL, tr->Mc, D_MODEL, D_FF, ACT_NONE);
    return VX_OK;
  }
  if (tr) {                                        // f16x2 projections + f16x2 attention
    const long plc = h2_plane(tr->Mc, D_MODEL, H2_TILE_A);
    {
      ProfScope ps(c, 3);
      if (c->prof_on) c->prof[3].bytes += tr->attn_flops;
      launch_attn_full_h2(c->fqkv, nullptr, seq_off, seq_len, prefix_len, batch, max_len, c->stream, c->fa3, plc, c->range_flag, -1,
                          tr->q_first, tr->c_off);
    }
    // x' = x[kept rows] + out_proj(attention): residual read through the row map, result compacted in fxn
    proj(c, nullptr, D_MODEL, L.out_w, L.out_w3, L.out_b, c->fx, D_MODEL, c->fxn, D_MODEL, tr->Mc, D_MODEL, D_MODEL, ACT_NONE, nullptr,
         c->fa3, nullptr, tr->rows);
    launch_layernorm(c->fxn, D_MODEL, nullptr, D_MODEL, (int)tr->Mc, D_MODEL, LN_EPS, L.n2_w, L.n2_b, ada2, ada2 ? ada2 + D_MODEL : nullptr,
                     c->stream, c->fa3, plc, c->range_flag);
    proj(c, nullptr, D_MODEL, L.l1_w, L.l1_w3, L.l1_b, nullptr, 0, nullptr, D_FF, tr->Mc, D_FF, D_MODEL, ACT_RELU, nullptr, c->fa3, c->fa3b);
    proj(c, nullptr, D_FF, L.l2_w, L.l2_w3, L.l2_b, c->fxn, D_MODEL, c->fxn, D_MODEL, tr->Mc, D_MODEL, D_FF, ACT_NONE, nullptr, c->fa3b);
    return VX_OK;
  }
  {
    ProfScope ps(c, 3);
    if (c->prof_on) c->prof[3].bytes += attn_flops;
    if (c->attn_x3 && c->attn_h2)
      launch_attn_full_h2(c->fqkv, att_pl ? nullptr : c->fatt, seq_off, seq_len, prefix_len, batch, max_len, c->stream,
                          att_pl ? c->fa3 : nullptr, pl1024, c->range_flag);
    else if (c->attn_x3)
      launch_attn_full_x3(c->fqkv, att_pl ? nullptr : c->fatt, seq_off, seq_len, prefix_len, batch, max_len, 0, c->stream,
                          att_pl ? c->fa3 : nullptr, pl1024);
    else launch_attn_full(c->fqkv, c->fatt, seq_off, seq_len, prefix_len, batch, max_len, c->stream);
  }
  proj(c, c->fatt, D_MODEL, L.out_w, L.out_w3, L.out_b, c->fx, D_MODEL, c->fx, D_MODEL, M, D_MODEL, D_MODEL, ACT_NONE, nullptr,
       att_pl ? c->fa3 : nullptr);
  launch_layernorm(c->fx, D_MODEL, pl ? nullptr : c->fxn, D_MODEL, (int)M, D_MODEL, LN_EPS, L.n2_w, L.n2_b, ada2,
                   ada2 ? ada2 + D_MODEL : nullptr, c->stream, pl ? c->fa3 : nullptr, pl1024, c->range_flag);
  if (pl) {
    // linear1's epilogue writes relu(x W1^T + b1) directly as linear2's A planes
    proj(c, nullptr, D_MODEL, L.l1_w, L.l1_w3, L.l1_b, nullptr, 0, nullptr, D_FF, M, D_FF, D_MODEL, ACT_RELU, nullptr, c->fa3, c->fa3b);
    proj(c, nullptr, D_FF, L.l2_w, L.l2_w3, L.l2_b, c->fx, D_MODEL, c->fx, D_MODEL, M, D_MODEL, D_FF, ACT_NONE, nullptr, c->fa3b);
  } else {
    proj(c, c->fxn, D_MODEL, L.l1_w, L.l1_w3, L.l1_b, nullptr, 0, c->fffn, D_FF, M, D_FF, D_MODEL, ACT_RELU);
    proj(c, c->fffn, D_FF, L.l2_w, L.l2_w3, L.l2_b, c->fx, D_MODEL, c->fx, D_MODEL, M, D_MODEL, D_FF, ACT_NONE);
  }
  return VX_OK;
}

// ---- f16x2 range guard ---------------------------------------------------------------------------------------------
// The f16x2 kernels need every scaled operand to fit fp16: |LayerNorm / attention / ReLU'd FFN activation| < 2047, |q|/8, |k|,
// |v| < 2047 (vx_common.h).  The reference puts no bound on any of them (fp32 throughout: modules/transformer.py:371-373,
// modules/activation.py:144-166), and trained transformers are known for massive activations.  The producers of the operand
// planes raise `range_flag`; it is read at a host sync the phase has anyway (no extra round trip in the common case) and a
// raised flag RE-RUNS THE PHASE on the exact-fp32 kernels (gemm_f32 / attn_full: the reference's own arithmetic, fp32 weights
// are resident, the two extra activation buffers are allocated on first use) -- the call never fails and never returns
// non-finite garbage because of the reduced-operand format.  vx_last_fallbacks() reports how often that happened.
bool range_guarded(const vx_ctx* c) { return c->gemm_mode == 0 || (c->attn_x3 && c->attn_h2); }


int ensure_f32_buffers(vx_ctx* c) {
  const long M = c->Mmax + 128;
  if (!c->fatt) if (int e = dev_alloc(c, &c->fatt, (size_t)M * D_MODEL)) return e;
  if (!c->fffn) if (int e = dev_alloc(c, &c->fffn, (size_t)M * D_FF)) return e;
  return VX_OK;
}

// synchronous read-and-clear (paths without a sync of their own: vx_finalize_weights, the vx_ar_prefill test seam)
int take_range_flag(vx_ctx* c, bool* raised) {
  *raised = false;
  if (!range_guarded(c)) return VX_OK;
  int flag = 0;
  D2H(&flag, c->range_flag, sizeof(int));
  SYNC();
  if (flag) HIPCHK(hipMemsetAsync(c->range_flag, 0, sizeof(int), c->stream));
  *raised = flag != 0;
  return VX_OK;
}

int check_batch(vx_ctx* c, const vx_batch* b, int max_rows) {
  if (!c->finalized) FAIL(VX_ESTATE, "weights not finalized");
  if (!b) FAIL(VX_EINVAL, "null batch");
  if (b->struct_size != sizeof(vx_batch))
    FAIL(VX_EINVAL, "vx_batch.struct_size is %u, this library expects %zu (ABI version %d)", b->struct_size, sizeof(vx_batch), VX_ABI_VERSION);
  if (b->batch <= 0 || b->batch > max_rows) FAIL(VX_EINVAL, "batch must be in 1..%d", max_rows);
  for (int i = 0; i < b->batch; ++i) {
    const int S = b->text_lens[i], Tp = b->prompt_lens[i];
    if (S <= 0) FAIL(VX_EINVAL, "x_lens must be > 0 (models/vallex.py:493)");       // assert torch.all(x_lens > 0)
    if (S > c->cfg.max_text || S > b->text_stride) FAIL(VX_EINVAL, "row %d: text length %d exceeds max_text", i, S);
    if (Tp < 0 || Tp > c->cfg.max_prompt || Tp > b->prompt_stride)
      FAIL(VX_EINVAL, "row %d: prompt length %d exceeds max_prompt", i, Tp);
    for (int s = 0; s < S; ++s) {
      const int id = b->text_ids[(long)i * b->text_stride + s], lg = b->text_lang[(long)i * b->text_stride + s];
      if (id < 0 || id >= 2048) FAIL(VX_EINVAL, "row %d: text id %d out of range", i, id);
      if (lg < -1 || lg > 2) FAIL(VX_EINVAL, "row %d: language id %d out of range", i, lg);   // -1: no language embedding
    }
    for (int t = 0; t < Tp * N_Q; ++t) {
      const int v = b->prompt_codes[(long)i * b->prompt_stride * N_Q + t];
      if (v < 0 || v >= AUDIO_VOCAB) FAIL(VX_EINVAL, "row %d: prompt code %d out of range", i, v);
    }
  }
  return VX_OK;
}

// ---- AR prefill (models/vallex.py:497-562, first ar_decoder.infer call) ------------------------------------
// beams > 1 (nb must be 1): best_of.  The reference repeats the prompt N times and runs N identical prefills
// (models/vallex.py:525-527); here the ONE row is prefilled once and its KV cache, residual row and logits are copied to
// N decode rows, which then sample independently.
int ar_prefill(vx_ctx* c, const vx_batch* b, int r0, int nb, int beams = 1) {
  const int NL = c->NL;
  const int nrows = beams > 1 ? beams : nb;          // decode rows after this call
  std::vector<int> seq_off(nb), seq_len(nb), S_(nb), dst_t, id_t, lang_t, pos_t, dst_a, id_a, pos_a, row_b, row_t;
  long M = 0;
  int max_len = 0;
  for (int i = 0; i < nb; ++i) {
    const int r = r0 + i, S = b->text_lens[r], Tp = b->prompt_lens[r];
    seq_off[i] = (int)M; seq_len[i] = S + 1 + Tp; S_[i] = S;
    max_len = std::max(max_len, seq_len[i]);
    for (int s = 0; s < S; ++s) {
      dst_t.push_back((int)M + s);
      id_t.push_back(b->text_ids[(long)r * b->text_stride + s]);
      lang_t.push_back(b->text_lang[(long)r * b->text_stride + s]);
      pos_t.push_back(s);
    }
    for (int t = 0; t <= Tp; ++t) {                       // BOS then prompt codebook 0 (models/vallex.py:515-517)
      dst_a.push_back((int)M + S + t);
      id_a.push_back(t == 0 ? BOS_ID : b->prompt_codes[((long)r * b->prompt_stride + (t - 1)) * N_Q]);
      pos_a.push_back(t);
    }
    for (int t = 0; t < seq_len[i]; ++t) { row_b.push_back(i); row_t.push_back(t); }
    M += seq_len[i];
  }
  if (M > c->Mmax) FAIL(VX_EINVAL, "prefill rows %ld exceed arena %ld", M, c->Mmax);
  c->h_L = seq_len;
  MetaBuilder mb(c);
  const long o_off = mb.add(seq_off), o_len = mb.add(seq_len), o_S = mb.add(S_), o_dt = mb.add(dst_t), o_it = mb.add(id_t),
             o_lt = mb.add(lang_t), o_pt = mb.add(pos_t), o_da = mb.add(dst_a), o_ia = mb.add(id_a), o_pa = mb.add(pos_a),
             o_rt = mb.add(row_t);
  // decode state
  std::vector<int> st_pos(nrows), st_ctx(nrows), st_zero(nrows, 0), st_one(nrows, 1), st_S(nrows);
  for (int i = 0; i < nrows; ++i) {
    const int j = beams > 1 ? 0 : i;
    st_pos[i] = b->prompt_lens[r0 + j]; st_ctx[i] = seq_len[j]; st_S[i] = S_[j];
  }
  // dec_attn launch order (decode.hip): rows by context length, the longest ceil(nb/2) first (descending), then the rest
  // ascending, so launch slots y and y + nb/2 -- which share a CU or a workgroup -- hold a long and a short context.  The
  // order of the contexts never changes during generation (every active row grows by one per step).
  std::vector<int> by_len(nrows), st_ord(nrows);
  for (int i = 0; i < nrows; ++i) by_len[i] = i;
  // (<= SB_ROWS rows: batch order -- the fused small-batch attention relies on slot == row, and there is nothing to balance)
  const bool balance = c->balance_rows && nrows > SB_ROWS;
  if (balance) std::stable_sort(by_len.begin(), by_len.end(), [&](int a, int b2) { return st_ctx[a] > st_ctx[b2]; });
  if (!balance) st_ord = by_len;                                  // batch order (also VX_BALANCE_ROWS=0)
  else {
    const int first = (nrows + 1) / 2;
    for (int y = 0; y < first; ++y) st_ord[y] = by_len[y];
    for (int y = first; y < nrows; ++y) st_ord[y] = by_len[nrows - 1 - (y - first)];
  }
  std::vector<int> st_meta(4 * nrows, 0), st_slot(nrows);
  for (int y = 0; y < nrows; ++y) {
    st_meta[4 * y] = st_ord[y]; st_meta[4 * y + 1] = st_ctx[st_ord[y]]; st_meta[4 * y + 2] = 1;
    st_slot[st_ord[y]] = y;
  }
  // The KV arena is indexed by LAUNCH SLOT, not by batch row: dec_attn knows the address of its K / V stream from its block id
  // alone and requests the first tile before the slot record (row, context, active) has arrived (decode.hip).  The prefill
  // scatters every sequence's K / V into the arena row of its slot; best_of: the one prefilled row has slot 0 (equal contexts
  // keep batch order) and beam_kv_broadcast copies arena row 0 to the arena rows 1 .. beams-1 = the other beams' slots.
  for (int& rb : row_b) rb = st_slot[rb];
  const long o_rb = mb.add(row_b);
  const long o_sp = mb.add(st_pos), o_sc = mb.add(st_ctx), o_z = mb.add(st_zero), o_1 = mb.add(st_one), o_sS = mb.add(st_S),
             o_meta = mb.add(st_meta), o_slot = mb.add(st_slot);
  // Row trimming of the LAST prefill layer (round 6; struct Trim): the decode only continues from the last row of every sequence
  // (models/vallex.py:568: the logits of the newest position), so behind the K / V projection -- the cache needs every row -- the last
  // layer runs its attention queries, out_proj and FFN on nb rows instead of M.
  std::vector<int> tq_first(nb), tc_off(nb), t_rows(nb);
  double trim_flops = 0;
  for (int i = 0; i < nb; ++i) {
    tq_first[i] = seq_len[i] - 1; tc_off[i] = i; t_rows[i] = seq_off[i] + seq_len[i] - 1;
    trim_flops += 4.0 * seq_len[i] * D_MODEL;
  }
  const long o_tq = mb.add(tq_first), o_tc = mb.add(tc_off), o_tr = mb.add(t_rows);
  if (int e = upload_meta(c)) return e;
  const size_t ib = nrows * sizeof(int);
  HIPCHK(hipMemcpyAsync(c->cur_pos, mb.dev(o_sp), ib, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->ctx_len, mb.dev(o_sc), ib, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->n_gen, mb.dev(o_z), ib, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->cur_tok, mb.dev(o_z), ib, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->active, mb.dev(o_1), ib, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->text_len, mb.dev(o_sS), ib, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->slot_meta, mb.dev(o_meta), 4 * ib, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->slot_of, mb.dev(o_slot), ib, hipMemcpyDeviceToDevice, c->stream));
  H2D(c->n_active, &nrows, sizeof(int));
  c->cur_batch = nrows;
  // enough (row, head, split) 8-wave workgroups to put >= 2 on every CU; one split (no combine launch) from 32 rows up
  c->nsplit = std::max(1, std::min(16, 512 / (nrows * N_HEAD)));
  // 5 .. 16 rows: ONE 8-wave workgroup per CU (256 / (rows x 16) splits, at least 2) -- round 6 sweep at 5 / 8 / 12 / 16 rows x contexts
  // ~500 / ~1300 (profiles/r06_sb_sweep2.txt): 5 rows 6 -> 3 splits -3.5 %, 8 rows 4 -> 2 splits -3 % / -0.6 % of the AR phase, 12 and 16 rows
  // stay at 2; one split (the fused out_proj) only pays from 17 rows up
  if (nrows > SB_ROWS && nrows <= 16) c->nsplit = std::max(2, 256 / (nrows * N_HEAD));
  // 8 .. 16 rows (round 6, profiles/r06_fuse_split_sweep.txt): out_proj folded into dec_attn WITH context splits -- the combine moves behind the
  // head's W_o slice (decode.hip dec_attn_kernel<true, *, true>, dec_reduce_ln_split_kernel), two launches per layer fewer.  The fused
  // kernel runs two rows per 16-wave workgroup: as many splits (<= 4) as keep 16 heads x row pairs x splits inside ONE round of the 256 CUs
  // (8 rows: 4, 9 .. 10: 3, 11 .. 16: 2; AR phase -2.7 % at 8 rows / context ~1000, -7 % at 10 rows, -5.5 % at 12, -4.6 % at 16).  5 .. 7 rows:
  // too few workgroups either way, the unfused chain stays (5 rows: 297 vs 299 ms, 6 rows: 302 vs 302).
  c->split_fused = false;
  if (c->fuse_out && c->fuse_split == 1 && nrows >= 8 && nrows <= 16) {
    const int ns = std::min(4, 256 / (N_HEAD * ((nrows + 1) / 2)));
    if (ns >= 2) { c->nsplit = ns; c->split_fused = true; }
  }
  if (c->att_nsplit_force > 0) {                                      // measurement switches (tools/gpu_call.sh sb_sweep, tools/fuse_split_sweep.sh)
    c->nsplit = c->att_nsplit_force;
    c->split_fused = c->fuse_out && c->fuse_split != 0 && nrows > SB_ROWS && c->nsplit >= 2 && c->nsplit <= 4;
  }
  // the small-batch chain compiles the split counts in (decode.hip): taken only for a combination that is instantiated
  c->sb_chain = false;
  if (c->sb_fuse && nrows <= SB_ROWS) {
    const int ns = nrows <= 2 ? 16 : 8;
    if (sb_chain_supported(SK_L2, SK_OUT, ns, nrows)) { c->nsplit = ns; c->sb_chain = true; }
    // ... with norm1 + QKV inside the attention launch for the smallest batches (VX_SB_QKV=<max rows>; its split count is tunable)
    c->sb_qkv = false;
    if (c->sb_chain && nrows <= c->sb_qkv_rows) {
      // the fused kernel holds one 8-wave workgroup per CU: 16 heads x rows x splits workgroups must fit the 256 CUs in ONE round,
      // and every workgroup of a head re-reads the head's q slice through L2 -- few splits win (profiles/r04_sb_qkv_ab.log)
      // one row: 16 splits = 256 workgroups (round 6 sweep, profiles/r06_sb_sweep.txt: equal to 8 at contexts ~400 / ~800, -2 % of the AR
      // phase at ~1300 -- the context share of a workgroup halves while the q slice it re-reads through L2 stays)
      const int fit[5] = {0, 16, 8, 4, 4};
      const int ns2 = c->sb_qkv_nsplit > 0 ? c->sb_qkv_nsplit : fit[nrows];
      // dec_attn_qkv_kernel addresses the KV arena and its partials by batch ROW and returns when slot_meta[slot].row != row
      // (decode.hip): it is only selected while the launch order is the identity -- checked here, not assumed from `balance` above
      bool identity = true;
      for (int y = 0; y < nrows; ++y) identity = identity && st_ord[y] == y;
      if (identity && sb_qkv_chain_supported(SK_L2, SK_OUT, ns2, nrows)) { c->nsplit = ns2; c->sb_qkv = true; }
    }
  }

  launch_embed_rows(c->fx, mb.dev(o_dt), W(c, "ar_text_embedding.word_embeddings.weight"), mb.dev(o_it),
                    W(c, "ar_language_embedding.word_embeddings.weight"), mb.dev(o_lt), W(c, "ar_text_position.alpha"),
                    c->pe, mb.dev(o_pt), (int)dst_t.size(), c->stream);
  launch_embed_rows(c->fx, mb.dev(o_da), W(c, "ar_audio_embedding.word_embeddings.weight"), mb.dev(o_ia), nullptr,
                    nullptr, W(c, "ar_audio_position.alpha"), c->pe, mb.dev(o_pa), (int)dst_a.size(), c->stream);
  if (int e = tap_store(c, "ar_prefill_in", c->fx, (size_t)M * D_MODEL)) return e;

  double attn_flops = 0;
  for (int i = 0; i < nb; ++i) attn_flops += 4.0 * seq_len[i] * (double)seq_len[i] * D_MODEL;
  const size_t cache_layer = (size_t)c->mbr * N_HEAD * c->Tmax * D_HEAD;
  const bool trim_h2 = c->gemm_mode == 0 && c->attn_x3 && c->attn_h2, trim_f32 = c->gemm_mode == 2 && !c->attn_x3;
  const bool trim = c->nar_trim && (trim_h2 || trim_f32) && !c->cfg.debug_taps;      // the same switch as the NAR stages (VX_NAR_TRIM)
  const Trim tr{nb, mb.dev(o_tq), mb.dev(o_tc), mb.dev(o_tr), trim_flops};
  for (int l = 0; l < NL; ++l) {
    if (int e = full_layer(c, c->ar[l], M, mb.dev(o_off), mb.dev(o_len), mb.dev(o_S), nb, max_len, nullptr, nullptr,
                           c->kc + l * cache_layer, c->vc + l * cache_layer, mb.dev(o_rb), mb.dev(o_rt), attn_flops,
                           (trim && l == NL - 1) ? &tr : nullptr))
      return e;
    if (c->cfg.debug_taps) {
      char nm[64];
      snprintf(nm, sizeof nm, "ar_layer_out.%d", l);
      if (int e = tap_store(c, nm, c->fx, (size_t)M * D_MODEL)) return e;
    }
  }
  // last row of every sequence -> decode residual stream h[b]  (trimmed: the last layer left exactly those rows, compacted, in the
  // buffer full_layer uses for the compacted residual stream: fxn in f16x2 mode, the QKV buffer in fp32 mode)
  if (trim)
    HIPCHK(hipMemcpyAsync(c->dh, trim_h2 ? c->fxn : c->fqkv, (size_t)nb * D_MODEL * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  else
    for (int i = 0; i < nb; ++i)
      HIPCHK(hipMemcpyAsync(c->dh + (size_t)i * D_MODEL, c->fx + ((size_t)seq_off[i] + seq_len[i] - 1) * D_MODEL,
                            D_MODEL * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  // final norm + ar_predict_layer on those rows (models/vallex.py:568)
  launch_dec_reduce_ln_pack(nullptr, 0, D_MODEL, nullptr, c->dh, nullptr, W(c, "ar_decoder.norm.weight"),
                            W(c, "ar_decoder.norm.bias"), c->xp, nb, c->stream);
  launch_skinny_gemm(c->pred_wp, c->xp, c->p_logits, PRED_NPAD, D_MODEL, SK_PRED, c->stream);
  if (beams > 1) {
    launch_beam_kv_broadcast(c->kc, c->vc, (long)cache_layer, NL, c->Tmax, seq_len[0], beams, c->stream);
    for (int i = 1; i < beams; ++i) {
      HIPCHK(hipMemcpyAsync(c->dh + (size_t)i * D_MODEL, c->dh, D_MODEL * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
      for (int ks = 0; ks < SK_PRED; ++ks)
        HIPCHK(hipMemcpyAsync(c->p_logits + ((size_t)ks * MB + i) * PRED_NPAD, c->p_logits + (size_t)ks * MB * PRED_NPAD,
                              PRED_NPAD * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    }
    c->h_L.assign(beams, seq_len[0]);
  }
  return VX_OK;
}

SampleArgs make_sample_args(vx_ctx* c, const vx_sampling* s, int commit, float* logits_out) {
  SampleArgs a{};
  a.partial = c->p_logits; a.splitk = SK_PRED; a.npad = PRED_NPAD;
  a.top_k = s ? s->top_k : 1;
  a.temperature = s ? s->temperature : 1.0f;
  a.uniforms = (s && s->uniforms) ? c->d_uniforms : nullptr;
  a.uniforms_stride = c->cur_batch;
  a.seed_dev = c->seed_dev;
  a.force_eos_at = s ? s->force_eos_at : -1;
  a.commit = commit;
  a.cur_tok = c->cur_tok; a.cur_pos = c->cur_pos; a.ctx_len = c->ctx_len; a.n_gen = c->n_gen; a.active = c->active;
  a.n_active = c->n_active; a.slot_meta = c->slot_meta; a.slot_of = c->slot_of;
  a.text_len = c->text_len; a.gen = c->gen; a.gen_stride = c->gen_stride; a.logits_out = logits_out;
  a.sum_logp = (s && s->best_of > 1) ? c->sum_logp : nullptr;
  a.batch = c->cur_batch;
  if (commit) {      // the sampler also embeds the committed token and applies norm1 of layer 0 (start of the next step)
    a.emb_tab = W(c, "ar_audio_embedding.word_embeddings.weight"); a.emb_alpha = W(c, "ar_audio_position.alpha");
    a.pe = c->pe; a.ln_g = c->ar[0].n1_w; a.ln_b = c->ar[0].n1_b; a.emb_h = c->dh; a.emb_xp = c->xp;
    a.wt = c->sb_chain ? 0 : 1;
  }
  return a;
}

// ---- one cached decode step (models/vallex.py:552-571 with kv_cache set) --------------------------------------
void ar_step_launches(vx_ctx* c, const SampleArgs* sa) {
  const int nb = c->cur_batch, NL = c->NL;
  hipStream_t st = c->stream;
  const size_t cache_layer = (size_t)c->mbr * N_HEAD * c->Tmax * D_HEAD;
  // sampling mode: the previous dec_sample already embedded the token and applied norm1 of layer 0; the teacher-forced
  // mode (sa == null, vx_ar_step) has to do it here
  if (!sa)
    launch_dec_embed_ln_pack(c->cur_tok, c->cur_pos, W(c, "ar_audio_embedding.word_embeddings.weight"),
                             W(c, "ar_audio_position.alpha"), c->pe, c->dh, c->ar[0].n1_w, c->ar[0].n1_b, c->xp, nb, st);
  if (c->sb_chain) {
    // small batch (BASELINE config 2: one utterance): 5 launches per layer -- QKV | dec_attn partials | out_proj | linear1 | linear2 --
    // every reduce + residual + LayerNorm and the context-split combine run in the prologue of the GEMM that consumes them
    // (decode.hip: skinny_gemm_sb_kernel, skinny16_sb_kernel).  The residual stream alternates between dh and dh2.
    float *hr = c->dh, *hw = c->dh2;            // the sampler / embed kernel left h in dh
    for (int l = 0; l < NL; ++l) {
      const LayerW& L = c->ar[l];
      if (c->sb_qkv) {
        // 4 launches per layer: norm1 + QKV run inside the split attention launch (decode.hip: dec_attn_qkv_kernel), which hands
        // nsplit + 1 partials per (row, head) to the out_proj prologue
        {
          ProfScope ps(c, 0);
          LAUNCH(launch_dec_attn_qkv(L.in_w, L.in_b, c->kc + l * cache_layer, c->vc + l * cache_layer, c->Tmax, c->slot_meta, c->part_o,
                                     c->part_ml, c->qk_new, c->nsplit, nb, l ? c->p_o : nullptr, l ? SK_L2 : 0,
                                     l ? c->ar[l - 1].l2_b : nullptr, hr, hw, L.n1_w, L.n1_b, c->xp, st));
          if (l) std::swap(hr, hw);
        }
        {
          ProfScope ps(c, 1);
          LAUNCH(launch_skinny_gemm_sb_combine(L.out_wp, c->p_o, D_MODEL, SK_OUT, c->part_o, c->part_ml, c->nsplit + 1, nb, st, c->qk_new));
        }
      } else {
      {
        ProfScope ps(c, 1);
        if (l == 0) launch_skinny_gemm(L.in_wp, c->xp, c->p_qkv, 3 * D_MODEL, D_MODEL, SK_QKV, st);      // xp = norm1(h) from the sampler
        else {
          LAUNCH(launch_skinny_gemm_sb_ln(L.in_wp, c->p_qkv, 3 * D_MODEL, SK_QKV, c->p_o, SK_L2, c->ar[l - 1].l2_b, hr, hw, L.n1_w, L.n1_b, nb, st));
          std::swap(hr, hw);
        }
      }
      {
        ProfScope ps(c, 0);
        LAUNCH(launch_dec_attn(c->p_qkv, SK_QKV, L.in_b, c->kc + l * cache_layer, c->vc + l * cache_layer, c->Tmax, c->slot_meta,
                        c->xp_att, c->part_o, c->part_ml, c->nsplit, nb, nullptr, c->p_oh, st));
      }
      { ProfScope ps(c, 1); LAUNCH(launch_skinny_gemm_sb_combine(L.out_wp, c->p_o, D_MODEL, SK_OUT, c->part_o, c->part_ml, c->nsplit, nb, st)); }
      }
      {
        ProfScope ps(c, 1);
        LAUNCH(launch_skinny16_sb_ln(L.l1_wp, L.l1_b, c->xp4, D_FF, c->p_o, SK_OUT, L.out_b, hr, hw, L.n2_w, L.n2_b, nb, st));
        std::swap(hr, hw);
      }
      { ProfScope ps(c, 1); launch_skinny_gemm(L.l2_wp, c->xp4, c->p_o, D_MODEL, D_FF, SK_L2, st); }
    }
    {
      ProfScope ps(c, 1);
      LAUNCH(launch_skinny_gemm_sb_ln(c->pred_wp, c->p_logits, PRED_NPAD, SK_PRED, c->p_o, SK_L2, c->ar[NL - 1].l2_b, hr, nullptr,
                               W(c, "ar_decoder.norm.weight"), W(c, "ar_decoder.norm.bias"), nb, st));
    }
    if (sa) LAUNCH(launch_dec_sample(*sa, st));
    return;
  }
  for (int l = 0; l < NL; ++l) {
    const LayerW& L = c->ar[l];
    {
      ProfScope ps(c, 1);
      if (c->qkv_bal) launch_skinny_qkv_balanced(L.in_wp, c->xp, c->p_qkv, st);
      else launch_skinny_gemm(L.in_wp, c->xp, c->p_qkv, 3 * D_MODEL, D_MODEL, SK_QKV, st, true);
    }
    // out_proj inside dec_attn: one context split (17 .. 32 rows), or 2 .. 4 splits with the combine moved behind W_o (8 .. 16 rows, round 6)
    const bool fused = c->fuse_out && (c->nsplit == 1 || c->split_fused);
    {
      ProfScope ps(c, 0);
      LAUNCH(launch_dec_attn(c->p_qkv, c->qkv_bal ? SK_QKV_BALANCED : SK_QKV, L.in_b, c->kc + l * cache_layer, c->vc + l * cache_layer, c->Tmax, c->slot_meta,
                      c->xp_att, c->part_o, c->part_ml, c->nsplit, nb, fused ? L.out_wh : nullptr, c->p_oh, st));
    }
    if (fused && c->nsplit > 1) {
      LAUNCH(launch_dec_reduce_ln_split(c->p_oh, c->part_ml, c->nsplit, L.out_b, c->dh, c->dh, L.n2_w, L.n2_b, c->xp, nb, st));
    } else if (fused) {
      launch_dec_reduce_ln_pack(c->p_oh, N_HEAD, D_MODEL, L.out_b, c->dh, c->dh, L.n2_w, L.n2_b, c->xp, nb, st);
    } else {
      if (c->nsplit > 1) launch_dec_attn_combine(c->part_o, c->part_ml, c->nsplit, c->active, c->xp_att, nb, st);
      { ProfScope ps(c, 1); launch_skinny_gemm(L.out_wp, c->xp_att, c->p_o, D_MODEL, D_MODEL, SK_OUT, st, true); }
      launch_dec_reduce_ln_pack(c->p_o, SK_OUT, D_MODEL, L.out_b, c->dh, c->dh, L.n2_w, L.n2_b, c->xp, nb, st);
    }
    { ProfScope ps(c, 1); launch_skinny16_relu_pack(L.l1_wp, c->xp, L.l1_b, c->xp4, D_FF, D_MODEL, st); }
    { ProfScope ps(c, 1); launch_skinny_gemm(L.l2_wp, c->xp4, c->p_o, D_MODEL, D_FF, SK_L2, st, true); }
    const float* ng = (l + 1 < NL) ? c->ar[l + 1].n1_w : W(c, "ar_decoder.norm.weight");
    const float* nbp = (l + 1 < NL) ? c->ar[l + 1].n1_b : W(c, "ar_decoder.norm.bias");
    launch_dec_reduce_ln_pack(c->p_o, SK_L2, D_MODEL, L.l2_b, c->dh, c->dh, ng, nbp, c->xp, nb, st);
  }
  { ProfScope ps(c, 1); launch_skinny_gemm(c->pred_wp, c->xp, c->p_logits, PRED_NPAD, D_MODEL, SK_PRED, st, true); }
  if (sa) LAUNCH(launch_dec_sample(*sa, st));
}

// the decode kernels compile these split counts in (decode.hip); retuning one without instantiating it must not reach a GPU
static_assert(SK_QKV == 4, "dec_attn_kernel<*, 4> sums four QKV slabs in its prologue");
static_assert(SK_OUT == 4, "skinny16_sb_kernel<4> / dec_reduce_ln_pack<4> are the instantiated out_proj consumers");
static_assert(SK_L2 == 8 || SK_L2 == 4, "skinny_gemm_sb_kernel<0, 8 | 4> are the instantiated linear2 consumers");
static_assert(SK_PRED == 4 || SK_PRED == 2 || SK_PRED == 1, "dec_sample_kernel<4 | 2 | 1>");

int launch_status(vx_ctx* c) {
  if (!c->launch_fail) return VX_OK;
  const char* what = c->launch_fail;
  c->launch_fail = nullptr;
  FAIL(VX_EINVAL, "kernel configuration not compiled into this library: %s", what);
}

// nsteps consecutive decode steps.  Graph mode keeps TWO instantiated graphs per signature: one step, and GRAPH_STEPS steps in one
// graph -- between two hipGraphLaunch calls the GPU idles for ~9 us (kernel traces, profiles/r04_gaps_*.csv: the gap behind every
// dec_sample launch), which a multi-step graph pays once per GRAPH_STEPS steps.  All step state (positions, lengths, flags, the
// sampler's counters) lives on the device, so a replay is position independent.
constexpr int GRAPH_STEPS = 4;
int ar_step_run(vx_ctx* c, const SampleArgs* sa, const std::string& sig, int nsteps = 1) {
  if (!c->cfg.use_graph || c->prof_on == 1 || !sa) {
    for (int i = 0; i < nsteps; ++i) ar_step_launches(c, sa);
    HIPCHK(hipGetLastError());
    return launch_status(c);
  }
  if (c->graph_sig != sig) {
    for (hipGraphExec_t* ge : {&c->graph_exec, &c->graph_exec_n})
      if (*ge) { (void)hipGraphExecDestroy(*ge); *ge = nullptr; }
    c->graph_sig = sig;
  }
  hipGraphExec_t& ge = nsteps == 1 ? c->graph_exec : c->graph_exec_n;
  if (!ge) {
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nsteps; ++i) ar_step_launches(c, sa);
    HIPCHK(hipStreamEndCapture(c->stream, &g));
    if (int e = launch_status(c)) { (void)hipGraphDestroy(g); return e; }      // an incomplete step must never be replayed
    HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
  }
  HIPCHK(hipGraphLaunch(ge, c->stream));
  return VX_OK;
}

// ---- sticky fp32 fallback bookkeeping (engine_ctx.h) -----------------------------------------------------------
// does this phase go to the fp32 kernels directly?  In sticky mode yes, except every FB_STICKY_PROBE_EVERY-th phase (a probe on f16x2)
static bool fb_direct(vx_ctx* c, bool sticky, int& age) {
  if (!sticky || !range_guarded(c)) return false;
  if (++age >= FB_STICKY_PROBE_EVERY) { age = 0; return false; }
  return true;
}
// outcome of a phase that RAN on f16x2: a raise counts towards sticky mode, a clean pass resets the count and leaves sticky mode
static void fb_outcome(vx_ctx* c, bool raised, int& raises, bool& sticky, int& age) {
  if (!range_guarded(c)) return;
  if (!raised) { raises = 0; sticky = false; age = 0; return; }
  if (++raises >= FB_STICKY_AFTER && !sticky) { sticky = true; age = 0; ++c->sticky_engaged; }
}

// ---- AR generation for one micro-batch -----------------------------------------------------------------------
int ar_generate(vx_ctx* c, const vx_batch* b, const vx_sampling* s, int r0, int nb, std::vector<int>& n_gen,
                std::vector<int>& gen, int beams = 1) {
  const int nb_rows = nb;                            // rows of the caller's batch that this micro-batch prefills
  // a context whose prefills keep leaving the fp16 range runs them on the exact-fp32 kernels straight away (sticky fallback)
  const bool direct_f32 = fb_direct(c, c->sticky_prefill_f32, c->sticky_prefill_age);
  if (direct_f32) {
    ++c->st_fb_prefill; ++c->fb_total;
    if (int e = ensure_f32_buffers(c)) return e;
    F32Scope f32(c);
    if (int e = ar_prefill(c, b, r0, nb_rows, beams)) return e;
  } else if (int e = ar_prefill(c, b, r0, nb_rows, beams)) return e;
  const int ub = beams > 1 ? beams : b->batch;       // columns of the caller's uniforms: [steps][batch] or [steps][best_of]
  if (beams > 1) nb = beams;
  if (s->uniforms) {
    // slice [steps][batch] -> [steps][nb] for this micro-batch
    // only the first gen_stride + 1 draws can ever be consumed (one per generated frame + the terminating sample)
    const long steps = std::min<long>(s->uniforms_steps, c->gen_stride + 1);
    if (steps * nb > c->uniforms_cap) FAIL(VX_EINVAL, "too many uniforms (%ld steps)", steps);
    std::vector<float> u((size_t)steps * nb);
    for (long t = 0; t < steps; ++t)
      for (int i = 0; i < nb; ++i) u[t * nb + i] = s->uniforms[t * ub + r0 + i];
    H2D(c->d_uniforms, u.data(), u.size() * sizeof(float));
    SYNC();
  }
  // the seed of the counter-based sampler lives in a device word: a new seed per call (the reference's contract, every call
  // draws from torch's generator) does not change the captured step graph
  const unsigned long long seed = s->seed;
  H2D(c->seed_dev, &seed, sizeof seed);
  SampleArgs sa = make_sample_args(c, s, 1, nullptr);
  std::vector<int> act(nb);
  bool any = true, raised = false;
  // first token from the prefill logits; the host sync that tells whether anything is still active also brings the range
  // flag of the prefill back (f16x2 guard, see F32Scope)
  auto first_sample = [&]() -> int {
    int flag = 0;
    HIPCHK(hipMemsetAsync(c->sum_logp, 0, MB * sizeof(float), c->stream));
    LAUNCH(launch_dec_sample(sa, c->stream));
    if (int e = launch_status(c)) return e;
    D2H(act.data(), c->active, nb * sizeof(int));
    if (range_guarded(c) && !direct_f32) D2H(&flag, c->range_flag, sizeof(int));
    SYNC();
    raised = flag != 0;
    any = std::any_of(act.begin(), act.end(), [](int v) { return v != 0; });
    return VX_OK;
  };
  if (int e = first_sample()) return e;
  if (!direct_f32) fb_outcome(c, raised, c->fb_prefill_raises, c->sticky_prefill_f32, c->sticky_prefill_age);
  if (raised) {
    // an operand of the prefill left the fp16 range: the K/V cache, the residual row and the logits are not to be trusted.
    // Re-run the prefill on the exact-fp32 kernels (it resets the decode state) and sample the first token again.
    HIPCHK(hipMemsetAsync(c->range_flag, 0, sizeof(int), c->stream));
    ++c->st_fb_prefill; ++c->fb_total;
    if (int e = ensure_f32_buffers(c)) return e;
    {
      F32Scope f32(c);
      if (int e = ar_prefill(c, b, r0, nb_rows, beams)) return e;
      if (int e = first_sample()) return e;
    }
  }
  char sig[160];
  snprintf(sig, sizeof sig, "b%d ns%d c%d%d%d k%d t%a u%d f%d l%d", nb, c->nsplit, (int)c->sb_chain, (int)c->sb_qkv, (int)c->split_fused, sa.top_k, sa.temperature,
           sa.uniforms != nullptr, sa.force_eos_at, sa.sum_logp != nullptr);
  const int sync_every = s->sync_every > 0 ? s->sync_every : 8;
  // with a forced EOS every row is inactive after force_eos_at steps: do not run on to the next host poll
  const int hard_cap = s->force_eos_at >= 0 ? std::min(c->gen_stride + 2, s->force_eos_at) : c->gen_stride + 2;
  int steps = 0;
  // GRAPH_STEPS steps per graph launch while that neither crosses a host poll nor the cap (nsteps is 1 or GRAPH_STEPS: two graphs)
  const int gs = (c->graph_multi && sync_every % GRAPH_STEPS == 0) ? GRAPH_STEPS : 1;
  while (any && steps < hard_cap) {
    const int n = (steps % gs == 0 && steps + gs <= hard_cap) ? gs : 1;
    if (int e = ar_step_run(c, &sa, sig, n)) return e;
    steps += n;
    if (steps % sync_every == 0) {
      D2H(act.data(), c->active, nb * sizeof(int));
      SYNC();
      any = std::any_of(act.begin(), act.end(), [](int v) { return v != 0; });
    }
  }
  n_gen.resize(nb);
  gen.resize((size_t)nb * c->gen_stride);
  D2H(n_gen.data(), c->n_gen, nb * sizeof(int));
  D2H(gen.data(), c->gen, gen.size() * sizeof(int));
  SYNC();
  c->st_steps += steps;
  if (c->prof_on) {
    // algorithmic KV bytes: every decode step of an active row reads ctx rows of K and V in all layers
    for (int i = 0; i < nb; ++i)
      for (int t = 1; t <= n_gen[i]; ++t)
        c->prof[0].bytes += (double)c->NL * ((double)(c->h_L[i] + t) * 2.0 * D_MODEL * 4.0);
    c->prof[1].bytes += (double)steps * ((double)c->NL * 12.0 * D_MODEL * D_MODEL + (double)AR_LOGITS * D_MODEL) * 4.0;
  }
  return VX_OK;
}

// ---- NAR: 7 stages (models/vallex.py:600-686, prefix_mode 1) ----------------------------------------------------
constexpr int VX_RETRY_F32 = 1;      // internal: the phase raised the f16x2 range flag, run it again on the fp32 kernels
int nar_generate_once(vx_ctx* c, const vx_batch* b, int r0, int nb, const std::vector<int>& T, const int* codes0,
                      long codes0_stride, std::vector<int>& out_codes /* [7][sumT] */, long& sumT_out) {
  const int NL = c->NL;
  std::vector<int> seq_off(nb), seq_len(nb), dst_t, id_t, lang_t, pos_t, ycodes, ynj, ydst, ypos, gen_rows, gen_y, q_first(nb), c_off(nb);
  long M = 0, Y = 0, sumT = 0;
  int max_len = 0;
  double trim_attn_flops = 0;
  {      // the host builds ~0.3 M ints of row metadata while the GPU waits between the AR and the NAR phase: no reallocation on the way
    size_t nS = 0, nY = 0, nT = 0;
    for (int i = 0; i < nb; ++i) { nS += b->text_lens[r0 + i]; nY += b->prompt_lens[r0 + i] + T[i]; nT += T[i]; }
    for (auto* v : {&dst_t, &id_t, &lang_t, &pos_t}) v->reserve(nS);
    for (auto* v : {&ynj, &ydst, &ypos}) v->reserve(nY);
    ycodes.reserve(nY * N_Q);
    gen_rows.reserve(nT); gen_y.reserve(nT);
  }
  for (int i = 0; i < nb; ++i) {
    const int r = r0 + i, S = b->text_lens[r], Tp = b->prompt_lens[r];
    seq_off[i] = (int)M; seq_len[i] = S + Tp + T[i];
    q_first[i] = S + Tp; c_off[i] = (int)sumT;
    trim_attn_flops += 4.0 * T[i] * (double)seq_len[i] * D_MODEL;
    max_len = std::max(max_len, seq_len[i]);
    for (int s = 0; s < S; ++s) {
      dst_t.push_back((int)M + s);
      id_t.push_back(b->text_ids[(long)r * b->text_stride + s]);
      lang_t.push_back(b->text_lang[(long)r * b->text_stride + s]);
      pos_t.push_back(s);
    }
    for (int t = 0; t < Tp + T[i]; ++t) {
      if (t < Tp) {
        for (int j = 0; j < N_Q; ++j) ycodes.push_back(b->prompt_codes[((long)r * b->prompt_stride + t) * N_Q + j]);
        ynj.push_back(N_Q);
      } else {
        ycodes.push_back(codes0[(long)i * codes0_stride + (t - Tp)]);
        for (int j = 1; j < N_Q; ++j) ycodes.push_back(0);
        ynj.push_back(1);
        gen_rows.push_back((int)M + S + t);
        gen_y.push_back((int)Y + t);
      }
      ydst.push_back((int)M + S + t);
      ypos.push_back(t);
    }
    M += seq_len[i]; Y += Tp + T[i]; sumT += T[i];
  }
  sumT_out = sumT;
  out_codes.assign((size_t)(N_Q - 1) * sumT, 0);
  if (sumT == 0) return VX_OK;
  if (M > c->Mmax) FAIL(VX_EINVAL, "NAR rows %ld exceed arena %ld", M, c->Mmax);
  MetaBuilder mb(c);
  const long o_off = mb.add(seq_off), o_len = mb.add(seq_len), o_dt = mb.add(dst_t), o_it = mb.add(id_t),
             o_lt = mb.add(lang_t), o_pt = mb.add(pos_t), o_yc = mb.add(ycodes), o_nj = mb.add(ynj), o_yd = mb.add(ydst),
             o_yp = mb.add(ypos), o_gr = mb.add(gen_rows), o_gy = mb.add(gen_y), o_qf = mb.add(q_first), o_co = mb.add(c_off);
  std::vector<int> zeros((size_t)(N_Q - 1) * sumT, 0);
  const long o_samples = mb.add(zeros);
  if (int e = upload_meta(c)) return e;

  launch_nar_yemb_init(c->fyemb, c->nar_tabs_dev, mb.dev(o_yc), mb.dev(o_nj), (int)Y, c->stream);
  double attn_flops = 0;
  for (int i = 0; i < nb; ++i) attn_flops += 4.0 * seq_len[i] * (double)seq_len[i] * D_MODEL;
  const int nnorm = 2 * NL + 1;
  for (int st = 0; st < N_Q - 1; ++st) {
    launch_embed_rows(c->fx, mb.dev(o_dt), W(c, "nar_text_embedding.word_embeddings.weight"), mb.dev(o_it),
                      W(c, "nar_language_embedding.word_embeddings.weight"), mb.dev(o_lt),
                      W(c, "nar_text_position.alpha"), c->pe, mb.dev(o_pt), (int)dst_t.size(), c->stream);
    launch_add_pe_scatter(c->fx, mb.dev(o_yd), c->fyemb, W(c, "nar_audio_position.alpha"), c->pe, mb.dev(o_yp), (int)Y,
                          c->stream);
    const float* ada = c->ada + (size_t)st * nnorm * 2 * D_MODEL;
    // the last layer only has to produce the generated rows (struct Trim); the per-layer taps want every row
    // f16x2 projections + f16x2 attention, or the reference arithmetic (fp32 projections + fp32 attention); the mixed modes of the
    // VX_GEMM_* / VX_ATTN_* switches and bf16x3 run every row
    const bool trim_h2 = c->gemm_mode == 0 && c->attn_x3 && c->attn_h2, trim_f32 = c->gemm_mode == 2 && !c->attn_x3;
    const bool trim = c->nar_trim && (trim_h2 || trim_f32) && !c->cfg.debug_taps;
    const Trim tr{sumT, mb.dev(o_qf), mb.dev(o_co), mb.dev(o_gr), trim_attn_flops};
    for (int l = 0; l < NL; ++l) {
      if (int e = full_layer(c, c->nar[l], M, mb.dev(o_off), mb.dev(o_len), nullptr, nb, max_len,
                             ada + (size_t)(2 * l) * 2 * D_MODEL, ada + (size_t)(2 * l + 1) * 2 * D_MODEL, nullptr, nullptr,
                             nullptr, nullptr, attn_flops, (trim && l == NL - 1) ? &tr : nullptr))
        return e;
      if (c->cfg.debug_taps && st == 0) {
        char nm[64];
        snprintf(nm, sizeof nm, "nar_layer_out.%d", l);
        if (int e = tap_store(c, nm, c->fx, (size_t)M * D_MODEL)) return e;
      }
    }
    const float* adaf = ada + (size_t)(2 * NL) * 2 * D_MODEL;
    char nm[64];
    snprintf(nm, sizeof nm, "nar_predict_layers.%d.weight", st);
    if (trim && trim_f32) {
      // the compacted residual stream (in the QKV buffer, full_layer) holds exactly the rows the predict layer reads: no gather
      launch_layernorm(c->fqkv, D_MODEL, c->fxn, D_MODEL, (int)sumT, D_MODEL, LN_EPS, W(c, "nar_decoder.norm.norm.weight"),
                       W(c, "nar_decoder.norm.norm.bias"), adaf, adaf + D_MODEL, c->stream);
      proj(c, c->fxn, D_MODEL, W(c, nm), c->pred_w3[st], nullptr, nullptr, 0, c->flogits, AUDIO_VOCAB, sumT, AUDIO_VOCAB, D_MODEL, ACT_NONE);
    } else if (trim) {
      // the compacted residual stream (fxn) holds exactly the rows the predict layer reads: the final norm writes the GEMM's
      // operand planes itself (bit-identical to a split of its fp32 result), no gather, no split pass
      launch_layernorm(c->fxn, D_MODEL, nullptr, D_MODEL, (int)sumT, D_MODEL, LN_EPS, W(c, "nar_decoder.norm.norm.weight"),
                       W(c, "nar_decoder.norm.norm.bias"), adaf, adaf + D_MODEL, c->stream, c->fa3, h2_plane(sumT, D_MODEL, H2_TILE_A),
                       c->range_flag);
      proj(c, nullptr, D_MODEL, W(c, nm), c->pred_w3[st], nullptr, nullptr, 0, c->flogits, AUDIO_VOCAB, sumT, AUDIO_VOCAB, D_MODEL,
           ACT_NONE, nullptr, c->fa3);
    } else {
      launch_layernorm(c->fx, D_MODEL, c->fxn, D_MODEL, (int)M, D_MODEL, LN_EPS, W(c, "nar_decoder.norm.norm.weight"),
                       W(c, "nar_decoder.norm.norm.bias"), adaf, adaf + D_MODEL, c->stream);
      proj(c, c->fxn, D_MODEL, W(c, nm), c->pred_w3[st], nullptr, nullptr, 0, c->flogits, AUDIO_VOCAB, sumT, AUDIO_VOCAB, D_MODEL,
           ACT_NONE, mb.dev(o_gr));
    }
    if (c->cfg.debug_taps) {                             // "nar_logits0" .. "nar_logits6": every stage's logits of the generated rows
      snprintf(nm, sizeof nm, "nar_logits%d", st);
      if (int e = tap_store(c, nm, c->flogits, (size_t)sumT * AUDIO_VOCAB)) return e;
    }
    int* samples = c->imeta + o_samples + (long)st * sumT;
    launch_argmax_rows(c->flogits, AUDIO_VOCAB, (int)sumT, AUDIO_VOCAB, samples, c->stream);
    if (st == 0 && c->fb_nar_raises > 0 && range_guarded(c)) {
      // this context has left the fp16 range in a NAR phase before: look at the flag behind stage 0 already, so that an
      // out-of-range model does not pay six more f16x2 stages before the fp32 re-run (costs one host sync; never in the common case)
      int early = 0;
      D2H(&early, c->range_flag, sizeof(int));
      SYNC();
      if (early) {
        HIPCHK(hipMemsetAsync(c->range_flag, 0, sizeof(int), c->stream));
        return VX_RETRY_F32;
      }
    }
    if (st < N_Q - 2) {
      snprintf(nm, sizeof nm, "nar_audio_embeddings.%d.word_embeddings.weight", st + 1);
      launch_embed_accum(c->fyemb, mb.dev(o_gy), W(c, nm), samples, (int)sumT, c->stream);
    }
  }
  D2H(out_codes.data(), c->imeta + o_samples, out_codes.size() * sizeof(int));
  int flag = 0;                        // the range flag rides on the sync that brings the ids back
  if (range_guarded(c)) D2H(&flag, c->range_flag, sizeof(int));
  SYNC();
  if (flag) {
    HIPCHK(hipMemsetAsync(c->range_flag, 0, sizeof(int), c->stream));
    return VX_RETRY_F32;
  }
  return VX_OK;
}

int nar_generate(vx_ctx* c, const vx_batch* b, int r0, int nb, const std::vector<int>& T, const int* codes0,
                 long codes0_stride, std::vector<int>& out_codes /* [7][sumT] */, long& sumT_out) {
  int e = VX_RETRY_F32;
  const bool direct_f32 = fb_direct(c, c->sticky_nar_f32, c->sticky_nar_age);          // sticky fallback (engine_ctx.h)
  if (!direct_f32) {
    e = nar_generate_once(c, b, r0, nb, T, codes0, codes0_stride, out_codes, sumT_out);
    if (e == VX_OK || e == VX_RETRY_F32) fb_outcome(c, e == VX_RETRY_F32, c->fb_nar_raises, c->sticky_nar_f32, c->sticky_nar_age);
    if (e != VX_RETRY_F32) return e;
  }
  // an operand of one of the 7 stages left the fp16 range: the whole phase (again) on the exact-fp32 kernels (F32Scope)
  ++c->st_fb_nar; ++c->fb_total;
  if ((e = ensure_f32_buffers(c))) return e;
  F32Scope f32(c);
  return nar_generate_once(c, b, r0, nb, T, codes0, codes0_stride, out_codes, sumT_out);
}

}  // namespace vxe

// =================================================================================================================
// C ABI
// =================================================================================================================
extern "C" {

const char* vx_last_error(const vx_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int32_t vx_abi_version(void) { return VX_ABI_VERSION; }

int vx_create(int device_id, const vx_config* cfg, vx_ctx** out) {
  if (!cfg || !out) { g_create_err = "null argument"; return VX_EINVAL; }
  if (cfg->struct_size != sizeof(vx_config)) {
    g_create_err = "vx_config.struct_size is " + std::to_string(cfg->struct_size) + ", this library expects " +
                   std::to_string(sizeof(vx_config)) + " (ABI version " + std::to_string(VX_ABI_VERSION) + ")";
    return VX_EINVAL;
  }
  if (cfg->num_layers <= 0 || cfg->max_batch <= 0 || cfg->max_text <= 0 || cfg->max_prompt < 0 || cfg->max_new <= 0) {
    g_create_err = "invalid vx_config";
    return VX_EINVAL;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_err = std::string("no HIP device available: ") + hipGetErrorString(e);
    return VX_EHIP;
  }
  if (device_id < 0 || device_id >= ndev) { g_create_err = "device_id out of range"; return VX_EINVAL; }
  vx_ctx* c = new vx_ctx();
  c->cfg = *cfg;
  c->dev = device_id;
  c->NL = cfg->num_layers;
  auto fail = [&](hipError_t err, const char* what) {
    g_create_err = std::string(what) + ": " + hipGetErrorString(err);
    delete c;
    return VX_EHIP;
  };
  if ((e = hipSetDevice(device_id)) != hipSuccess) return fail(e, "hipSetDevice");
  bool masked = false;
  for (uint32_t w : cfg->cu_mask) masked |= w != 0;
  if (masked) {      // a context that shares the GPU with others: its stream only dispatches to the CUs of cfg->cu_mask
    if ((e = hipExtStreamCreateWithCUMask(&c->stream, 8, cfg->cu_mask)) != hipSuccess)
      return fail(e, "hipExtStreamCreateWithCUMask");
  } else if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
  for (auto& ev : c->ev_t)
    if ((e = hipEventCreate(&ev)) != hipSuccess) {
      for (auto& e2 : c->ev_t) if (e2) (void)hipEventDestroy(e2);
      (void)hipStreamDestroy(c->stream);
      return fail(e, "hipEventCreate");
    }
  // the pinned transfer ring (engine_ctx.h): 64 MiB unless VX_PIN_MB says otherwise (>= 2 chunks)
  {
    size_t mb = 64;
    if (const char* ev = getenv("VX_PIN_MB")) mb = (size_t)std::max(16, std::min(1024, atoi(ev)));
    void* q = nullptr;
    if ((e = hipHostMalloc(&q, mb << 20, hipHostMallocDefault)) != hipSuccess) {
      for (auto& e2 : c->ev_t) if (e2) (void)hipEventDestroy(e2);
      (void)hipStreamDestroy(c->stream);
      return fail(e, "hipHostMalloc (pinned transfer ring)");
    }
    c->ring.base = static_cast<char*>(q);
    c->ring.cap = mb << 20;
  }
  *out = c;
  return VX_OK;
}

void vx_destroy(vx_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->dev);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
  if (c->graph_exec_n) (void)hipGraphExecDestroy(c->graph_exec_n);
  for (auto& p : c->prof) for (auto ev : p.ev) (void)hipEventDestroy(ev);
  for (auto ev : c->ev_t) if (ev) (void)hipEventDestroy(ev);
  for (void* p : c->allocs) (void)hipFree(p);
  for (auto& kv : c->w) (void)hipFree(kv.second.d);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->ring.base) (void)hipHostFree(c->ring.base);
  delete c;
}

int vx_synchronize(vx_ctx* c) {
  if (!c) return VX_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  SYNC();
  return VX_OK;
}

int vx_ar_prefill(vx_ctx* c, const vx_batch* b) {
  if (!c) return VX_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  if (int e = check_batch(c, b, c->mbr)) return e;
  c->st_fb_prefill = c->st_fb_nar = 0;
  bool raised = fb_direct(c, c->sticky_prefill_f32, c->sticky_prefill_age);
  if (!raised) {
    if (int e = ar_prefill(c, b, 0, b->batch)) return e;
    if (int e = take_range_flag(c, &raised)) return e;       // syncs when the mode is guarded
    fb_outcome(c, raised, c->fb_prefill_raises, c->sticky_prefill_f32, c->sticky_prefill_age);
  }
  if (raised) {
    ++c->st_fb_prefill; ++c->fb_total;
    if (int e = ensure_f32_buffers(c)) return e;
    F32Scope f32(c);
    if (int e = ar_prefill(c, b, 0, b->batch)) return e;
  }
  SYNC();                 // the seam returns with the prefill complete in every mode
  HIPCHK(hipGetLastError());
  return VX_OK;
}

int vx_ar_logits(vx_ctx* c, float* out) {
  if (!c || !out) return VX_EINVAL;
  if (c->cur_batch <= 0) FAIL(VX_ESTATE, "no prefill has run");
  HIPCHK(hipSetDevice(c->dev));
  SampleArgs sa = make_sample_args(c, nullptr, 0, c->d_logits);
  LAUNCH(launch_dec_sample(sa, c->stream));
  if (int e = launch_status(c)) return e;
  D2H(out, c->d_logits, (size_t)c->cur_batch * AR_LOGITS * sizeof(float));
  SYNC();
  return VX_OK;
}

int vx_ar_step(vx_ctx* c, const int32_t* tokens) {
  if (!c || !tokens) return VX_EINVAL;
  if (c->cur_batch <= 0) FAIL(VX_ESTATE, "no prefill has run");
  HIPCHK(hipSetDevice(c->dev));
  H2D(c->force_tok, tokens, c->cur_batch * sizeof(int));
  launch_dec_force_token(c->force_tok, c->cur_tok, c->cur_pos, c->ctx_len, c->n_gen, c->gen, c->gen_stride, c->active,
                         c->cur_batch, c->slot_meta, c->slot_of, c->stream);
  if (int e = ar_step_run(c, nullptr, "")) return e;
  SYNC();
  HIPCHK(hipGetLastError());
  return VX_OK;
}

int vx_nar(vx_ctx* c, const vx_batch* b, const int32_t* codes0, int32_t codes0_stride, const int32_t* lens,
           int64_t* out_codes, int32_t out_stride) {
  if (!c || !codes0 || !lens || !out_codes) return VX_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  if (int e = check_batch(c, b, c->mbr)) return e;
  std::vector<int> T(lens, lens + b->batch);
  for (int i = 0; i < b->batch; ++i) {
    if (T[i] < 0 || T[i] > c->cfg.max_new || T[i] > out_stride || T[i] > codes0_stride) FAIL(VX_EINVAL, "row %d: bad length %d", i, T[i]);
    for (int t = 0; t < T[i]; ++t) {                    // indexes nar_audio_embeddings.0 (1025 rows) on the device
      const int v = codes0[(long)i * codes0_stride + t];
      if (v < 0 || v > AUDIO_VOCAB) FAIL(VX_EINVAL, "row %d: first-codebook id %d out of range (0..1024)", i, v);
    }
  }
  std::vector<int> oc;
  long sumT = 0;
  c->st_fb_prefill = c->st_fb_nar = 0;
  if (int e = nar_generate(c, b, 0, b->batch, T, codes0, codes0_stride, oc, sumT)) return e;
  long off = 0;
  for (int i = 0; i < b->batch; ++i) {
    for (int t = 0; t < T[i]; ++t) {
      int64_t* o = out_codes + ((long)i * out_stride + t) * N_Q;
      o[0] = codes0[(long)i * codes0_stride + t];
      for (int st = 0; st < N_Q - 1; ++st) o[st + 1] = oc[(size_t)st * sumT + off + t];
    }
    off += T[i];
  }
  return VX_OK;
}

int vx_infer(vx_ctx* c, const vx_batch* b, const vx_sampling* s, int64_t* out_codes, int32_t out_stride,
             int32_t* out_lens) {
  if (!c || !s || !out_codes || !out_lens) return VX_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  if (s->struct_size != sizeof(vx_sampling))
    FAIL(VX_EINVAL, "vx_sampling.struct_size is %u, this library expects %zu (ABI version %d)", s->struct_size, sizeof(vx_sampling), VX_ABI_VERSION);
  if (int e = check_batch(c, b, c->cfg.max_batch)) return e;
  if (!(s->temperature > 0.f)) FAIL(VX_EINVAL, "temperature must be > 0");
  c->st_steps = 0; c->st_frames = 0; c->st_ar_ms = 0; c->st_nar_ms = 0; c->st_truncated = 0;
  c->st_fb_prefill = c->st_fb_nar = 0;
  // a row that fills the arena although neither EOS, the reference's 16*S cap nor a forced EOS ended it was cut short
  auto cut_by_arena = [&](int n, int S) {
    return n >= c->gen_stride && c->gen_stride < 16 * S && !(s->force_eos_at >= 0 && s->force_eos_at <= c->gen_stride);
  };
  if (s->best_of > 1) {
    // best-of-N beams of ONE utterance (models/vallex.py:491,525-527): the row is replicated N times, every beam
    // samples independently, beams that emit EOS stop; selection on sum(logp) / len^penalty (:583-594), then the NAR
    // stages run on the chosen beam only (:600).
    if (b->batch != 1) FAIL(VX_EINVAL, "best_of > 1 needs batch == 1 (models/vallex.py:491)");
    const int N = s->best_of;
    if (N > c->mbr) FAIL(VX_EINVAL, "best_of %d exceeds the micro-batch (%d)", N, c->mbr);
    const int Tp = b->prompt_lens[0];
    std::vector<int> n_gen, gen, oc;
    if (int e = ar_generate(c, b, s, 0, 1, n_gen, gen, N)) return e;     // ONE prefill, N decode rows
    std::vector<float> slp(N);
    D2H(slp.data(), c->sum_logp, N * sizeof(float)); SYNC();
    int best = 0, worst = 0;
    double bv = 0, wv = 0;
    for (int i = 0; i < N; ++i) {
      const double len = 1.0 + Tp + n_gen[i];                        // torch.sum(y != EOS): BOS + prompt + frames
      const double v = (double)(float)((float)slp[i] / powf((float)len, s->length_penalty));
      if (i == 0 || v > bv) { bv = v; best = i; }
      if (i == 0 || v < wv) { wv = v; worst = i; }
    }
    const int pick = s->return_worst ? worst : best;
    if (n_gen[pick] > out_stride) FAIL(VX_EINVAL, "out_stride %d too small for %d frames", out_stride, n_gen[pick]);
    long sumT = 0;
    std::vector<int> T1(1, n_gen[pick]);
    if (int e = nar_generate(c, b, 0, 1, T1, gen.data() + (size_t)pick * c->gen_stride, c->gen_stride, oc, sumT)) return e;
    out_lens[0] = n_gen[pick];
    c->st_frames = n_gen[pick];
    c->st_truncated = cut_by_arena(n_gen[pick], b->text_lens[0]) ? 1 : 0;
    for (int t = 0; t < n_gen[pick]; ++t) {
      int64_t* o = out_codes + (long)t * N_Q;
      o[0] = gen[(size_t)pick * c->gen_stride + t];
      for (int st = 0; st < N_Q - 1; ++st) o[st + 1] = oc[(size_t)st * sumT + t];
    }
    return VX_OK;
  }
  hipEvent_t e0 = c->ev_t[0], e1 = c->ev_t[1], e2 = c->ev_t[2];      // owned by the context: nothing to leak on an early return
  for (int r0 = 0; r0 < b->batch; r0 += c->mbr) {
    const int nb = std::min(c->mbr, b->batch - r0);
    std::vector<int> n_gen, gen, oc;
    HIPCHK(hipEventRecord(e0, c->stream));
    if (int e = ar_generate(c, b, s, r0, nb, n_gen, gen)) return e;
    HIPCHK(hipEventRecord(e1, c->stream));
    for (int i = 0; i < nb; ++i)
      if (n_gen[i] > out_stride) FAIL(VX_EINVAL, "out_stride %d too small for %d frames", out_stride, n_gen[i]);
    long sumT = 0;
    if (int e = nar_generate(c, b, r0, nb, n_gen, gen.data(), c->gen_stride, oc, sumT)) return e;
    HIPCHK(hipEventRecord(e2, c->stream));
    HIPCHK(hipEventSynchronize(e2));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1)); c->st_ar_ms += ms;
    HIPCHK(hipEventElapsedTime(&ms, e1, e2)); c->st_nar_ms += ms;
    long off = 0;
    for (int i = 0; i < nb; ++i) {
      out_lens[r0 + i] = n_gen[i];
      c->st_frames += n_gen[i];
      if (cut_by_arena(n_gen[i], b->text_lens[r0 + i])) ++c->st_truncated;
      for (int t = 0; t < n_gen[i]; ++t) {
        int64_t* o = out_codes + ((long)(r0 + i) * out_stride + t) * N_Q;
        o[0] = gen[(size_t)i * c->gen_stride + t];
        for (int st = 0; st < N_Q - 1; ++st) o[st + 1] = oc[(size_t)st * sumT + off + t];
      }
      off += n_gen[i];
    }
  }
  return VX_OK;
}

int64_t vx_read_tap(vx_ctx* c, const char* name, float* dst, int64_t max_floats) {
  if (!c || !name || !dst) return VX_EINVAL;
  auto it = c->taps.find(name);
  if (it == c->taps.end()) {
    // not a tap: a tensor as vx_load_tensor stored it (read-back check of an upload, tools/verify_checkpoint.py)
    it = c->w.find(name);
    if (it == c->w.end()) FAIL(VX_ENOTFOUND, "no tap or loaded tensor '%s'", name);
  }
  const int64_t n = std::min<int64_t>((int64_t)it->second.n, max_floats);
  HIPCHK(hipSetDevice(c->dev));
  SYNC();
  D2H(dst, it->second.d, n * sizeof(float)); SYNC();
  return n;
}

int vx_last_fallbacks(vx_ctx* c, int32_t* prefill_phases, int32_t* nar_phases, int64_t* lifetime) {
  if (!c) return VX_EINVAL;
  if (prefill_phases) *prefill_phases = c->st_fb_prefill;
  if (nar_phases) *nar_phases = c->st_fb_nar;
  if (lifetime) *lifetime = c->fb_total;
  return VX_OK;
}

int vx_fallback_state(vx_ctx* c, int32_t* sticky_prefill, int32_t* sticky_nar, int64_t* times_engaged) {
  if (!c) return VX_EINVAL;
  if (sticky_prefill) *sticky_prefill = c->sticky_prefill_f32;
  if (sticky_nar) *sticky_nar = c->sticky_nar_f32;
  if (times_engaged) *times_engaged = c->sticky_engaged;
  return VX_OK;
}

int vx_fallback_reset(vx_ctx* c) {
  if (!c) return VX_EINVAL;
  c->fb_prefill_raises = c->fb_nar_raises = 0;
  c->sticky_prefill_f32 = c->sticky_nar_f32 = false;
  c->sticky_prefill_age = c->sticky_nar_age = 0;
  return VX_OK;
}

int vx_arith_mode(vx_ctx* c, int32_t* gemm_mode, int32_t* attn_mode) {
  if (!c) return VX_EINVAL;
  if (!c->finalized) FAIL(VX_ESTATE, "weights not finalized");
  if (gemm_mode) *gemm_mode = c->gemm_mode;
  if (attn_mode) *attn_mode = !c->attn_x3 ? 2 : (c->attn_h2 ? 0 : 1);
  return VX_OK;
}

int vx_last_truncated(vx_ctx* c, int32_t* rows) {
  if (!c || !rows) return VX_EINVAL;
  *rows = c->st_truncated;
  return VX_OK;
}

int vx_last_stats(vx_ctx* c, int64_t* ar_steps, int64_t* frames, double* ar_ms, double* nar_ms) {
  if (!c) return VX_EINVAL;
  if (ar_steps) *ar_steps = c->st_steps;
  if (frames) *frames = c->st_frames;
  if (ar_ms) *ar_ms = c->st_ar_ms;
  if (nar_ms) *nar_ms = c->st_nar_ms;
  return VX_OK;
}

}  // extern "C"
