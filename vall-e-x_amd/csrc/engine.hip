// Engine: context, weight ingest, arenas, launch sequences and the C ABI (include/vallex_hip.h).
// Host-side counterpart of VALLE.inference (models/vallex.py:458-686) + the Vocos call of
// utils/generation.py:148-150; every hot op is a hand-written gfx950 kernel from the sibling .hip files.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <map>
#include <string>
#include <vector>

#include "../../include/vallex_hip.h"
#include "vx_common.h"

using namespace vx;

namespace {

std::string g_create_err;

struct Tensor {
  float* d = nullptr;
  std::vector<int64_t> shape;
  size_t n = 0;
};

struct LayerW {
  const float *in_w, *in_b, *out_w, *out_b, *l1_w, *l1_b, *l2_w, *l2_b, *n1_w, *n1_b, *n2_w, *n2_b;
  float *in_wp = nullptr, *out_wp = nullptr, *l1_wp = nullptr, *l2_wp = nullptr;   // packed decode images (AR only)
  float* out_wh = nullptr;                                                         // head-major W_o (fused out_proj in dec_attn)
  unsigned short *in_w3 = nullptr, *out_w3 = nullptr, *l1_w3 = nullptr, *l2_w3 = nullptr;   // 3 bf16 planes [3][N][K]
};

struct ProfClass {
  std::vector<hipEvent_t> ev;   // pairs
  size_t used = 0;
  double bytes = 0;
};

constexpr int SK_QKV = 4, SK_OUT = 4, SK_L2 = 8, SK_PRED = 4;
constexpr int PRED_NPAD = 1056;

}  // namespace

struct vx_ctx {
  vx_config cfg{};
  int dev = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_t[3] = {nullptr, nullptr, nullptr};   // AR / NAR phase timing of vx_infer (created once, vx_create)
  std::string err;
  std::map<std::string, Tensor> w;
  bool finalized = false;
  std::vector<void*> allocs;

  // derived weights
  int NL = 0;
  float* pe = nullptr;
  int pe_rows = 0;
  std::vector<LayerW> ar, nar;
  float* ada = nullptr;            // [7][2NL+1][2048]
  float* pred_wp = nullptr;        // packed ar_predict_layer
  const float** nar_tabs_dev = nullptr;
  bool has_vocos = false;
  float *vc_embed_w = nullptr, *vc_head_w = nullptr, *vc_head_b = nullptr, *vc_dft = nullptr, *vc_win2 = nullptr;

  // geometry
  int mbr = 0;                     // rows per micro-batch (<= 32)
  int Tmax = 0;                    // KV rows per (row, head)
  long Mmax = 0;                   // packed rows of a micro-batch on the full-sequence paths

  // arithmetic of the transformer projections of prefill / NAR: 0 = f16x2 (default; gemm_f16x2.hip), 1 = bf16x3
  // (VX_GEMM_X3=1; gemm_bf16x3*.hip), 2 = exact fp32 MFMA (VX_GEMM_F32=1; gemm_f32.hip).  All three keep every golden's ids.
  int gemm_mode = 0;
  bool attn_x3 = true;                        // 16-bit-plane attention (h2 or x3); VX_ATTN_F32=1 keeps the fp32 MFMA kernel
  bool attn_h2 = true;                        // f16x2 attention (attn_full_h2.hip); VX_ATTN_X3=1: bf16x3 (attn_full_x3.hip)
  int* range_flag = nullptr;       // device flag: an operand of an f16x2 GEMM / attention did not fit fp16 (read at the phase's
                                   // existing host sync; a raised flag re-runs the phase on the exact-fp32 kernels)
  unsigned long long* seed_dev = nullptr;   // seed of the counter-based sampler (device word: not part of the captured graph)
  int st_fb_prefill = 0, st_fb_nar = 0;     // phases of the last call that were re-run in fp32 (vx_last_fallbacks)
  long fb_total = 0;                        // ... since the context was created
  unsigned short* fa3b = nullptr;  // second plane buffer: linear1 writes linear2's A planes straight from its epilogue (f16x2 mode)
  unsigned short* fa3 = nullptr;   // activation planes [2 or 3][M][K<=4096]
  unsigned short* pred_w3[N_Q - 1] = {};
  // full-sequence arena
  float *fx = nullptr, *fxn = nullptr, *fqkv = nullptr, *fatt = nullptr, *fffn = nullptr, *fyemb = nullptr,
        *flogits = nullptr;
  int* imeta = nullptr;            // device int scratch for row metadata
  long imeta_cap = 0;
  std::vector<int> hmeta;          // host staging for imeta

  // decode arena
  float *kc = nullptr, *vc = nullptr;      // [NL][mbr*16][Tmax][64]
  float *dh = nullptr, *dh2 = nullptr, *xp = nullptr, *xp_att = nullptr, *xp4 = nullptr;
  bool sb_fuse = true;             // <= SB_ROWS rows: reduce+LN / combine folded into the consuming GEMM (VX_SB_FUSE=0: the general chain)
  float *p_qkv = nullptr, *p_o = nullptr, *p_oh = nullptr, *p_logits = nullptr, *part_o = nullptr, *part_ml = nullptr;
  std::map<const unsigned short*, int> w_shift;   // f16x2: power-of-two scale exponent of every weight's planes
  bool balance_rows = true;        // dec_attn launch order pairs long with short contexts per CU (VX_BALANCE_ROWS=0: batch order)
  bool fuse_out = true;            // out_proj folded into dec_attn when nsplit == 1 (VX_FUSE_OUT=0: separate skinny GEMM)
  float *d_logits = nullptr, *d_uniforms = nullptr, *sum_logp = nullptr;
  long uniforms_cap = 0;
  int *cur_tok = nullptr, *cur_pos = nullptr, *ctx_len = nullptr, *n_gen = nullptr, *active = nullptr,
      *text_len = nullptr, *gen = nullptr, *force_tok = nullptr, *n_active = nullptr, *slot_meta = nullptr, *slot_of = nullptr;
  int gen_stride = 0;
  int cur_batch = 0;
  int nsplit = 1;
  std::vector<int> h_L;            // prefill lengths of the current micro-batch

  // graph
  hipGraphExec_t graph_exec = nullptr;
  std::string graph_sig;

  // taps
  std::map<std::string, Tensor> taps;

  // profiling / stats
  int prof_on = 0;                 // 0 off, 1 every class (AR step runs eagerly), 2 full-sequence classes only
  ProfClass prof[5];
  int64_t st_steps = 0, st_frames = 0;
  int st_truncated = 0;            // rows of the last vx_infer cut by the arena (max_new) before the reference's stop rule
  double st_ar_ms = 0, st_nar_ms = 0;

  // EnCodec decoder (optional)
  bool has_encodec = false;
  float *ec_codebook = nullptr, *ec_w0 = nullptr, *ec_lstm_b[2] = {nullptr, nullptr}, *ec_whh_p[2] = {nullptr, nullptr};
  float *ec_wT[4] = {}, *ec_bT[4] = {}, *ec_w1[4] = {}, *ec_w3[4] = {};
  float *ec_e0 = nullptr, *ec_x0 = nullptr, *ec_y1 = nullptr, *ec_y2 = nullptr, *ec_xg = nullptr, *ec_col = nullptr,
        *ec_a = nullptr, *ec_sc = nullptr, *ec_out = nullptr, *ec_h = nullptr, *ec_audio = nullptr, *ec_hp = nullptr,
        *ec_c = nullptr, *ec_pg = nullptr;
  long ec_frames_cap = 0;
  // EnCodec SEANet encoder + RVQ encode (prompt enrolment; shares the decoder's arena)
  bool has_encodec_enc = false;
  float *en_w1[4] = {}, *en_w3[4] = {}, *en_wd[4] = {}, *en_w15 = nullptr, *en_lstm_b[2] = {nullptr, nullptr},
        *en_whh_p[2] = {nullptr, nullptr}, *en_e2 = nullptr, *en_scores = nullptr;
  long long* en_codes = nullptr;

  // vocos arena
  float *vfeat = nullptr, *vcol = nullptr, *vx0 = nullptr, *vx1 = nullptr, *vhid = nullptr, *vo = nullptr,
        *vreim = nullptr, *vframes = nullptr, *vaudio = nullptr;
  long v_rows_cap = 0;
};

namespace {

#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) {                                                                            \
      char _buf[512];                                                                                  \
      snprintf(_buf, sizeof _buf, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      c->err = _buf;                                                                                   \
      return VX_EHIP;                                                                                  \
    }                                                                                                  \
  } while (0)

#define FAIL(code, ...)                         \
  do {                                          \
    char _buf[512];                             \
    snprintf(_buf, sizeof _buf, __VA_ARGS__);   \
    c->err = _buf;                              \
    return (code);                              \
  } while (0)

template <typename T>
int dev_alloc(vx_ctx* c, T** p, size_t count, bool zero = true) {
  void* q = nullptr;
  HIPCHK(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
  c->allocs.push_back(q);
  if (zero) HIPCHK(hipMemsetAsync(q, 0, std::max<size_t>(count, 1) * sizeof(T), c->stream));
  *p = reinterpret_cast<T*>(q);
  return VX_OK;
}

const float* W(vx_ctx* c, const std::string& name) {
  auto it = c->w.find(name);
  return it == c->w.end() ? nullptr : it->second.d;
}

// ---- profiling helpers: an event pair around one launch --------------------------------------------------
struct ProfScope {
  vx_ctx* c;
  int which;
  bool on;
  ProfScope(vx_ctx* c_, int w) : c(c_), which(w), on(c_->prof_on == 1 || (c_->prof_on == 2 && w >= 2)) {
    if (!on) return;
    ProfClass& p = c->prof[which];
    if (p.used + 2 > p.ev.size()) {
      for (int i = 0; i < 2; ++i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { on = false; return; }
        p.ev.push_back(e);
      }
    }
    (void)hipEventRecord(p.ev[p.used], c->stream);
  }
  ~ProfScope() {
    if (!on) return;
    ProfClass& p = c->prof[which];
    (void)hipEventRecord(p.ev[p.used + 1], c->stream);
    p.used += 2;
  }
};

// ---- int metadata upload ---------------------------------------------------------------------------------
int upload_meta(vx_ctx* c) {
  if ((long)c->hmeta.size() > c->imeta_cap) FAIL(VX_EINVAL, "row metadata overflow (%zu > %ld)", c->hmeta.size(), c->imeta_cap);
  HIPCHK(hipMemcpyAsync(c->imeta, c->hmeta.data(), c->hmeta.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  // the host vector is reused by the next call: make the copy complete first
  HIPCHK(hipStreamSynchronize(c->stream));
  return VX_OK;
}

struct MetaBuilder {
  vx_ctx* c;
  explicit MetaBuilder(vx_ctx* c_) : c(c_) { c->hmeta.clear(); }
  // reserve n ints, return offset
  long add(const std::vector<int>& v) {
    long off = (long)c->hmeta.size();
    c->hmeta.insert(c->hmeta.end(), v.begin(), v.end());
    while (c->hmeta.size() % 4) c->hmeta.push_back(0);
    return off;
  }
  const int* dev(long off) const { return c->imeta + off; }
};

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
          const float* colscale, float* C, int ldc, long M, int N, int K, int act, const int* gather = nullptr, int cls = 4) {
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = Wt; g.ldw = ldw; g.bias = bias; g.resid = resid; g.ldr = ldr; g.colscale = colscale;
  g.C = C; g.ldc = ldc; g.M = (int)M; g.N = N; g.K = K; g.act = act; g.row_gather = gather;
  ProfScope ps(c, cls);               // class 2 = transformer projections (proj), 4 = the fp32 GEMMs of Vocos / EnCodec
  if (c->prof_on) c->prof[cls].bytes += 2.0 * (double)M * N * K;      // flops for this class
  launch_gemm_f32(g, c->stream);
}

// transformer projection (F.linear of modules/activation.py:144,166, modules/transformer.py:371-373, models/vallex.py:677)
// f16x2 mode extras: `a_pre` = A operand already in plane form (skip the split pass); `out_pl` = write the result as the next
// GEMM's A planes (K = N) instead of fp32 rows (C may then be null).
void proj(vx_ctx* c, const float* A, int lda, const float* Wf, const unsigned short* W3, const float* bias,
          const float* resid, int ldr, float* C, int ldc, long M, int N, int K, int act, const int* gather = nullptr,
          const unsigned short* a_pre = nullptr, unsigned short* out_pl = nullptr) {
  if (c->gemm_mode == 2 || !W3) {
    gemm(c, A, lda, Wf, K, bias, resid, ldr, nullptr, C, ldc, M, N, K, act, gather, 2);
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
  if (c->gemm_mode == 0) g.descale = ldexpf(1.0f, -(H2_ACT_SHIFT + c->w_shift.at(W3)));
  ProfScope ps(c, 2);
  if (c->prof_on) c->prof[2].bytes += 2.0 * (double)M * N * K;
  if (c->gemm_mode == 0) launch_gemm_f16x2(g, c->stream);
  else if (M >= 1024) launch_gemm_bf16x3_dma(g, c->stream);        // 256-row tiles: not for short row sets
  else launch_gemm_bf16x3(g, c->stream);
}

// one pre-norm block on packed rows (modules/transformer.py:296-302 / :337-347) -- shared by AR prefill and NAR
int full_layer(vx_ctx* c, const LayerW& L, long M, const int* seq_off, const int* seq_len, const int* prefix_len,
               int batch, int max_len, const float* ada1, const float* ada2, float* kcl, float* vcl,
               const int* row_b, const int* row_t, double attn_flops) {
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

struct F32Scope {            // the full-sequence path on the exact-fp32 kernels for the lifetime of the object
  vx_ctx* c;
  int gm;
  bool ax;
  explicit F32Scope(vx_ctx* c_) : c(c_), gm(c_->gemm_mode), ax(c_->attn_x3) { c->gemm_mode = 2; c->attn_x3 = false; }
  ~F32Scope() { c->gemm_mode = gm; c->attn_x3 = ax; }
};

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
  HIPCHK(hipMemcpyAsync(&flag, c->range_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
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
             o_rb = mb.add(row_b), o_rt = mb.add(row_t);
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
  if (c->balance_rows) std::stable_sort(by_len.begin(), by_len.end(), [&](int a, int b2) { return st_ctx[a] > st_ctx[b2]; });
  if (!c->balance_rows) st_ord = by_len;                          // VX_BALANCE_ROWS=0: batch order
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
  const long o_sp = mb.add(st_pos), o_sc = mb.add(st_ctx), o_z = mb.add(st_zero), o_1 = mb.add(st_one), o_sS = mb.add(st_S),
             o_meta = mb.add(st_meta), o_slot = mb.add(st_slot);
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
  HIPCHK(hipMemcpyAsync(c->n_active, &nrows, sizeof(int), hipMemcpyHostToDevice, c->stream));
  c->cur_batch = nrows;
  // enough (row, head, split) 8-wave workgroups to put >= 2 on every CU; one split (no combine launch) from 32 rows up
  c->nsplit = std::max(1, std::min(16, 512 / (nrows * N_HEAD)));
  if (c->sb_fuse && nrows <= SB_ROWS) c->nsplit = nrows <= 2 ? 16 : 8;      // the small-batch out_proj prologue compiles the split count in

  launch_embed_rows(c->fx, mb.dev(o_dt), W(c, "ar_text_embedding.word_embeddings.weight"), mb.dev(o_it),
                    W(c, "ar_language_embedding.word_embeddings.weight"), mb.dev(o_lt), W(c, "ar_text_position.alpha"),
                    c->pe, mb.dev(o_pt), (int)dst_t.size(), c->stream);
  launch_embed_rows(c->fx, mb.dev(o_da), W(c, "ar_audio_embedding.word_embeddings.weight"), mb.dev(o_ia), nullptr,
                    nullptr, W(c, "ar_audio_position.alpha"), c->pe, mb.dev(o_pa), (int)dst_a.size(), c->stream);
  if (int e = tap_store(c, "ar_prefill_in", c->fx, (size_t)M * D_MODEL)) return e;

  double attn_flops = 0;
  for (int i = 0; i < nb; ++i) attn_flops += 4.0 * seq_len[i] * (double)seq_len[i] * D_MODEL;
  const size_t cache_layer = (size_t)c->mbr * N_HEAD * c->Tmax * D_HEAD;
  for (int l = 0; l < NL; ++l) {
    if (int e = full_layer(c, c->ar[l], M, mb.dev(o_off), mb.dev(o_len), mb.dev(o_S), nb, max_len, nullptr, nullptr,
                           c->kc + l * cache_layer, c->vc + l * cache_layer, mb.dev(o_rb), mb.dev(o_rt), attn_flops))
      return e;
    if (c->cfg.debug_taps) {
      char nm[64];
      snprintf(nm, sizeof nm, "ar_layer_out.%d", l);
      if (int e = tap_store(c, nm, c->fx, (size_t)M * D_MODEL)) return e;
    }
  }
  // last row of every sequence -> decode residual stream h[b]
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
  if (c->sb_fuse && nb <= SB_ROWS && c->nsplit > 1) {
    // small batch (BASELINE config 2: one utterance): 5 launches per layer -- QKV | dec_attn partials | out_proj | linear1 | linear2 --
    // every reduce + residual + LayerNorm and the context-split combine run in the prologue of the GEMM that consumes them
    // (decode.hip: skinny_gemm_sb_kernel, skinny16_sb_kernel).  The residual stream alternates between dh and dh2.
    float *hr = c->dh, *hw = c->dh2;            // the sampler / embed kernel left h in dh
    for (int l = 0; l < NL; ++l) {
      const LayerW& L = c->ar[l];
      {
        ProfScope ps(c, 1);
        if (l == 0) launch_skinny_gemm(L.in_wp, c->xp, c->p_qkv, 3 * D_MODEL, D_MODEL, SK_QKV, st);      // xp = norm1(h) from the sampler
        else {
          launch_skinny_gemm_sb_ln(L.in_wp, c->p_qkv, 3 * D_MODEL, SK_QKV, c->p_o, SK_L2, c->ar[l - 1].l2_b, hr, hw, L.n1_w, L.n1_b, nb, st);
          std::swap(hr, hw);
        }
      }
      {
        ProfScope ps(c, 0);
        launch_dec_attn(c->p_qkv, SK_QKV, L.in_b, c->kc + l * cache_layer, c->vc + l * cache_layer, c->Tmax, c->slot_meta,
                        c->xp_att, c->part_o, c->part_ml, c->nsplit, nb, nullptr, c->p_oh, st);
      }
      { ProfScope ps(c, 1); launch_skinny_gemm_sb_combine(L.out_wp, c->p_o, D_MODEL, SK_OUT, c->part_o, c->part_ml, c->nsplit, nb, st); }
      {
        ProfScope ps(c, 1);
        launch_skinny16_sb_ln(L.l1_wp, L.l1_b, c->xp4, D_FF, c->p_o, SK_OUT, L.out_b, hr, hw, L.n2_w, L.n2_b, nb, st);
        std::swap(hr, hw);
      }
      { ProfScope ps(c, 1); launch_skinny_gemm(L.l2_wp, c->xp4, c->p_o, D_MODEL, D_FF, SK_L2, st); }
    }
    {
      ProfScope ps(c, 1);
      launch_skinny_gemm_sb_ln(c->pred_wp, c->p_logits, PRED_NPAD, SK_PRED, c->p_o, SK_L2, c->ar[NL - 1].l2_b, hr, nullptr,
                               W(c, "ar_decoder.norm.weight"), W(c, "ar_decoder.norm.bias"), nb, st);
    }
    if (sa) launch_dec_sample(*sa, st);
    return;
  }
  for (int l = 0; l < NL; ++l) {
    const LayerW& L = c->ar[l];
    { ProfScope ps(c, 1); launch_skinny_gemm(L.in_wp, c->xp, c->p_qkv, 3 * D_MODEL, D_MODEL, SK_QKV, st); }
    const bool fused = c->fuse_out && c->nsplit == 1;
    {
      ProfScope ps(c, 0);
      launch_dec_attn(c->p_qkv, SK_QKV, L.in_b, c->kc + l * cache_layer, c->vc + l * cache_layer, c->Tmax, c->slot_meta,
                      c->xp_att, c->part_o, c->part_ml, c->nsplit, nb, fused ? L.out_wh : nullptr, c->p_oh, st);
    }
    if (fused) {
      launch_dec_reduce_ln_pack(c->p_oh, N_HEAD, D_MODEL, L.out_b, c->dh, c->dh, L.n2_w, L.n2_b, c->xp, nb, st);
    } else {
      if (c->nsplit > 1) launch_dec_attn_combine(c->part_o, c->part_ml, c->nsplit, c->active, c->xp_att, nb, st);
      { ProfScope ps(c, 1); launch_skinny_gemm(L.out_wp, c->xp_att, c->p_o, D_MODEL, D_MODEL, SK_OUT, st); }
      launch_dec_reduce_ln_pack(c->p_o, SK_OUT, D_MODEL, L.out_b, c->dh, c->dh, L.n2_w, L.n2_b, c->xp, nb, st);
    }
    { ProfScope ps(c, 1); launch_skinny16_relu_pack(L.l1_wp, c->xp, L.l1_b, c->xp4, D_FF, D_MODEL, st); }
    { ProfScope ps(c, 1); launch_skinny_gemm(L.l2_wp, c->xp4, c->p_o, D_MODEL, D_FF, SK_L2, st); }
    const float* ng = (l + 1 < NL) ? c->ar[l + 1].n1_w : W(c, "ar_decoder.norm.weight");
    const float* nbp = (l + 1 < NL) ? c->ar[l + 1].n1_b : W(c, "ar_decoder.norm.bias");
    launch_dec_reduce_ln_pack(c->p_o, SK_L2, D_MODEL, L.l2_b, c->dh, c->dh, ng, nbp, c->xp, nb, st);
  }
  { ProfScope ps(c, 1); launch_skinny_gemm(c->pred_wp, c->xp, c->p_logits, PRED_NPAD, D_MODEL, SK_PRED, st); }
  if (sa) launch_dec_sample(*sa, st);
}

int ar_step_run(vx_ctx* c, const SampleArgs* sa, const std::string& sig) {
  if (!c->cfg.use_graph || c->prof_on == 1 || !sa) {
    ar_step_launches(c, sa);
    HIPCHK(hipGetLastError());
    return VX_OK;
  }
  if (!c->graph_exec || c->graph_sig != sig) {
    if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    ar_step_launches(c, sa);
    HIPCHK(hipStreamEndCapture(c->stream, &g));
    HIPCHK(hipGraphInstantiate(&c->graph_exec, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    c->graph_sig = sig;
  }
  HIPCHK(hipGraphLaunch(c->graph_exec, c->stream));
  return VX_OK;
}

// ---- AR generation for one micro-batch -----------------------------------------------------------------------
int ar_generate(vx_ctx* c, const vx_batch* b, const vx_sampling* s, int r0, int nb, std::vector<int>& n_gen,
                std::vector<int>& gen, int beams = 1) {
  const int nb_rows = nb;                            // rows of the caller's batch that this micro-batch prefills
  if (int e = ar_prefill(c, b, r0, nb_rows, beams)) return e;
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
    HIPCHK(hipMemcpyAsync(c->d_uniforms, u.data(), u.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  // the seed of the counter-based sampler lives in a device word: a new seed per call (the reference's contract, every call
  // draws from torch's generator) does not change the captured step graph
  const unsigned long long seed = s->seed;
  HIPCHK(hipMemcpyAsync(c->seed_dev, &seed, sizeof seed, hipMemcpyHostToDevice, c->stream));
  SampleArgs sa = make_sample_args(c, s, 1, nullptr);
  std::vector<int> act(nb);
  bool any = true, raised = false;
  // first token from the prefill logits; the host sync that tells whether anything is still active also brings the range
  // flag of the prefill back (f16x2 guard, see F32Scope)
  auto first_sample = [&]() -> int {
    int flag = 0;
    HIPCHK(hipMemsetAsync(c->sum_logp, 0, MB * sizeof(float), c->stream));
    launch_dec_sample(sa, c->stream);
    HIPCHK(hipMemcpyAsync(act.data(), c->active, nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (range_guarded(c)) HIPCHK(hipMemcpyAsync(&flag, c->range_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    raised = flag != 0;
    any = std::any_of(act.begin(), act.end(), [](int v) { return v != 0; });
    return VX_OK;
  };
  if (int e = first_sample()) return e;
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
  snprintf(sig, sizeof sig, "b%d ns%d k%d t%a u%d f%d l%d", nb, c->nsplit, sa.top_k, sa.temperature,
           sa.uniforms != nullptr, sa.force_eos_at, sa.sum_logp != nullptr);
  const int sync_every = s->sync_every > 0 ? s->sync_every : 8;
  // with a forced EOS every row is inactive after force_eos_at steps: do not run on to the next host poll
  const int hard_cap = s->force_eos_at >= 0 ? std::min(c->gen_stride + 2, s->force_eos_at) : c->gen_stride + 2;
  int steps = 0;
  while (any && steps < hard_cap) {
    if (int e = ar_step_run(c, &sa, sig)) return e;
    ++steps;
    if (steps % sync_every == 0) {
      HIPCHK(hipMemcpyAsync(act.data(), c->active, nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      any = std::any_of(act.begin(), act.end(), [](int v) { return v != 0; });
    }
  }
  n_gen.resize(nb);
  gen.resize((size_t)nb * c->gen_stride);
  HIPCHK(hipMemcpyAsync(n_gen.data(), c->n_gen, nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(gen.data(), c->gen, gen.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
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
  std::vector<int> seq_off(nb), seq_len(nb), dst_t, id_t, lang_t, pos_t, ycodes, ynj, ydst, ypos, gen_rows, gen_y;
  long M = 0, Y = 0, sumT = 0;
  int max_len = 0;
  for (int i = 0; i < nb; ++i) {
    const int r = r0 + i, S = b->text_lens[r], Tp = b->prompt_lens[r];
    seq_off[i] = (int)M; seq_len[i] = S + Tp + T[i];
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
             o_yp = mb.add(ypos), o_gr = mb.add(gen_rows), o_gy = mb.add(gen_y);
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
    for (int l = 0; l < NL; ++l) {
      if (int e = full_layer(c, c->nar[l], M, mb.dev(o_off), mb.dev(o_len), nullptr, nb, max_len,
                             ada + (size_t)(2 * l) * 2 * D_MODEL, ada + (size_t)(2 * l + 1) * 2 * D_MODEL, nullptr, nullptr,
                             nullptr, nullptr, attn_flops))
        return e;
      if (c->cfg.debug_taps && st == 0) {
        char nm[64];
        snprintf(nm, sizeof nm, "nar_layer_out.%d", l);
        if (int e = tap_store(c, nm, c->fx, (size_t)M * D_MODEL)) return e;
      }
    }
    const float* adaf = ada + (size_t)(2 * NL) * 2 * D_MODEL;
    launch_layernorm(c->fx, D_MODEL, c->fxn, D_MODEL, (int)M, D_MODEL, LN_EPS, W(c, "nar_decoder.norm.norm.weight"),
                     W(c, "nar_decoder.norm.norm.bias"), adaf, adaf + D_MODEL, c->stream);
    char nm[64];
    snprintf(nm, sizeof nm, "nar_predict_layers.%d.weight", st);
    proj(c, c->fxn, D_MODEL, W(c, nm), c->pred_w3[st], nullptr, nullptr, 0, c->flogits, AUDIO_VOCAB, sumT, AUDIO_VOCAB, D_MODEL,
         ACT_NONE, mb.dev(o_gr));
    if (c->cfg.debug_taps) {                             // "nar_logits0" .. "nar_logits6": every stage's logits of the generated rows
      snprintf(nm, sizeof nm, "nar_logits%d", st);
      if (int e = tap_store(c, nm, c->flogits, (size_t)sumT * AUDIO_VOCAB)) return e;
    }
    int* samples = c->imeta + o_samples + (long)st * sumT;
    launch_argmax_rows(c->flogits, AUDIO_VOCAB, (int)sumT, AUDIO_VOCAB, samples, c->stream);
    if (st < N_Q - 2) {
      snprintf(nm, sizeof nm, "nar_audio_embeddings.%d.word_embeddings.weight", st + 1);
      launch_embed_accum(c->fyemb, mb.dev(o_gy), W(c, nm), samples, (int)sumT, c->stream);
    }
  }
  HIPCHK(hipMemcpyAsync(out_codes.data(), c->imeta + o_samples, out_codes.size() * sizeof(int), hipMemcpyDeviceToHost,
                        c->stream));
  int flag = 0;                        // the range flag rides on the sync that brings the ids back
  if (range_guarded(c)) HIPCHK(hipMemcpyAsync(&flag, c->range_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (flag) {
    HIPCHK(hipMemsetAsync(c->range_flag, 0, sizeof(int), c->stream));
    return VX_RETRY_F32;
  }
  return VX_OK;
}

int nar_generate(vx_ctx* c, const vx_batch* b, int r0, int nb, const std::vector<int>& T, const int* codes0,
                 long codes0_stride, std::vector<int>& out_codes /* [7][sumT] */, long& sumT_out) {
  int e = nar_generate_once(c, b, r0, nb, T, codes0, codes0_stride, out_codes, sumT_out);
  if (e != VX_RETRY_F32) return e;
  // an operand of one of the 7 stages left the fp16 range: the whole phase again on the exact-fp32 kernels (F32Scope)
  ++c->st_fb_nar; ++c->fb_total;
  if ((e = ensure_f32_buffers(c))) return e;
  F32Scope f32(c);
  return nar_generate_once(c, b, r0, nb, T, codes0, codes0_stride, out_codes, sumT_out);
}

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

// Kernel-development aid (VX_BENCH_CLOCK=1 in vx_bench_gemm): one wave that sits next to the kernel under test for `ref_ticks` of
// the constant 100 MHz counter and reports how many shader-clock ticks (s_memtime) went by -> the clock the chip actually
// holds under that load.  Bounded by the real-time counter, so it always terminates.
__global__ void clock_probe_kernel(unsigned long long* out, unsigned long long ref_ticks) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
  while (__builtin_amdgcn_s_memrealtime() - r0 < ref_ticks) __builtin_amdgcn_s_sleep(16);
  if (threadIdx.x == 0) {
    out[0] = __builtin_readcyclecounter() - c0;
    out[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

}  // namespace

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
  *out = c;
  return VX_OK;
}

void vx_destroy(vx_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->dev);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
  for (auto& p : c->prof) for (auto ev : p.ev) (void)hipEventDestroy(ev);
  for (auto ev : c->ev_t) if (ev) (void)hipEventDestroy(ev);
  for (void* p : c->allocs) (void)hipFree(p);
  for (auto& kv : c->w) (void)hipFree(kv.second.d);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int vx_synchronize(vx_ctx* c) {
  if (!c) return VX_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  HIPCHK(hipStreamSynchronize(c->stream));
  return VX_OK;
}

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
  HIPCHK(hipMemcpy(t.d, data, t.n * sizeof(float), hipMemcpyHostToDevice));
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
  if ((e = dev_alloc(c, &c->fyemb, (size_t)M * d))) return e;
  if ((e = dev_alloc(c, &c->flogits, (size_t)((long)c->mbr * c->cfg.max_new + 128) * AUDIO_VOCAB))) return e;
  c->imeta_cap = std::max(M * 24, (long)c->cfg.max_batch * c->cfg.max_new * 12) + 65536;
  if ((e = dev_alloc(c, &c->imeta, (size_t)c->imeta_cap))) return e;
  const size_t cache = (size_t)NL * c->mbr * N_HEAD * c->Tmax * D_HEAD;
  if ((e = dev_alloc(c, &c->kc, cache, false))) return e;
  if ((e = dev_alloc(c, &c->vc, cache, false))) return e;
  if ((e = dev_alloc(c, &c->dh, (size_t)MB * d))) return e;
  if ((e = dev_alloc(c, &c->dh2, (size_t)MB * d))) return e;
  if (const char* ev = getenv("VX_SB_FUSE")) c->sb_fuse = !(ev[0] == '0');
  if ((e = dev_alloc(c, &c->xp, (size_t)MB * d))) return e;
  if ((e = dev_alloc(c, &c->xp_att, (size_t)MB * d))) return e;
  if ((e = dev_alloc(c, &c->xp4, (size_t)2 * MB * f))) return e;   // linear1's two split-K slabs, packed image
  if ((e = dev_alloc(c, &c->p_qkv, (size_t)SK_QKV * MB * 3 * d))) return e;
  if ((e = dev_alloc(c, &c->p_o, (size_t)std::max(SK_OUT, SK_L2) * MB * d))) return e;
  if ((e = dev_alloc(c, &c->p_oh, (size_t)N_HEAD * MB * d))) return e;
  if (const char* ev = getenv("VX_FUSE_OUT")) c->fuse_out = !(ev[0] == '0');
  if (const char* ev = getenv("VX_BALANCE_ROWS")) c->balance_rows = !(ev[0] == '0');
  if ((e = dev_alloc(c, &c->p_logits, (size_t)SK_PRED * MB * PRED_NPAD))) return e;
  if ((e = dev_alloc(c, &c->part_o, (size_t)MB * N_HEAD * 16 * D_HEAD))) return e;
  if ((e = dev_alloc(c, &c->part_ml, (size_t)MB * N_HEAD * 16 * 2))) return e;
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
    HIPCHK(hipMemcpy(c->pe, pe.data(), pe.size() * sizeof(float), hipMemcpyHostToDevice));
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
        HIPCHK(hipMemcpyAsync(&bits, d_max, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
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
    HIPCHK(hipMemcpy((void*)tmp, tabs.data(), N_Q * sizeof(float*), hipMemcpyHostToDevice));
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
      HIPCHK(hipMemcpy(w.data(), W(c, "vocos.backbone.embed.weight"), w.size() * sizeof(float), hipMemcpyDeviceToHost));
      for (int o = 0; o < C; ++o)
        for (int ch = 0; ch < 128; ++ch)
          for (int tap = 0; tap < 7; ++tap) w2[(size_t)o * 896 + tap * 128 + ch] = w[((size_t)o * 128 + ch) * 7 + tap];
      if ((e = dev_alloc(c, &c->vc_embed_w, w2.size(), false))) return e;
      HIPCHK(hipMemcpy(c->vc_embed_w, w2.data(), w2.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    // head weight/bias padded 1282 -> 1408 rows (GEMM N multiple of 128)
    if ((e = dev_alloc(c, &c->vc_head_w, (size_t)NBP * C))) return e;
    if ((e = dev_alloc(c, &c->vc_head_b, NBP))) return e;
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(c->vc_head_w, W(c, "vocos.head.out.weight"), (size_t)NB * C * sizeof(float), hipMemcpyDeviceToDevice));
    HIPCHK(hipMemcpy(c->vc_head_b, W(c, "vocos.head.out.bias"), (size_t)NB * sizeof(float), hipMemcpyDeviceToDevice));
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
      HIPCHK(hipMemcpy(c->vc_dft, dft.data(), dft.size() * sizeof(float), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(c->vc_win2, win2.data(), win2.size() * sizeof(float), hipMemcpyHostToDevice));
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
      HIPCHK(hipMemcpy(host.data(), t.d, t.n * sizeof(float), hipMemcpyDeviceToHost));
      return VX_OK;
    };
    auto upload = [&](const std::vector<float>& host, float** dev) -> int {
      if (int e2 = dev_alloc(c, dev, host.size(), false)) return e2;
      HIPCHK(hipMemcpy(*dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
      return VX_OK;
    };
    std::vector<float> w, w2, b, b2;
    // RVQ codebooks, concatenated [8*1024][128]
    if ((e = dev_alloc(c, &c->ec_codebook, (size_t)N_Q * 1024 * 128, false))) return e;
    for (int q = 0; q < N_Q; ++q)
      HIPCHK(hipMemcpy(c->ec_codebook + (size_t)q * 1024 * 128, W(c, "encodec.quantizer." + std::to_string(q) + ".embed"),
                       (size_t)1024 * 128 * sizeof(float), hipMemcpyDeviceToDevice));
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
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipGetLastError());
  {
    bool raised = false;          // weights are scaled from their own maximum: only a non-finite weight can raise the flag here
    if ((e = take_range_flag(c, &raised))) return e;
    if (raised) FAIL(VX_EINVAL, "vx_finalize_weights: a projection weight is not finite (NaN / inf in the state-dict)");
  }
  c->finalized = true;
  return VX_OK;
}

int vx_ar_prefill(vx_ctx* c, const vx_batch* b) {
  if (!c) return VX_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  if (int e = check_batch(c, b, c->mbr)) return e;
  c->st_fb_prefill = c->st_fb_nar = 0;
  if (int e = ar_prefill(c, b, 0, b->batch)) return e;
  bool raised = false;
  if (int e = take_range_flag(c, &raised)) return e;       // syncs when the mode is guarded
  if (raised) {
    ++c->st_fb_prefill; ++c->fb_total;
    if (int e = ensure_f32_buffers(c)) return e;
    F32Scope f32(c);
    if (int e = ar_prefill(c, b, 0, b->batch)) return e;
  }
  HIPCHK(hipStreamSynchronize(c->stream));                 // the seam returns with the prefill complete in every mode
  HIPCHK(hipGetLastError());
  return VX_OK;
}

int vx_ar_logits(vx_ctx* c, float* out) {
  if (!c || !out) return VX_EINVAL;
  if (c->cur_batch <= 0) FAIL(VX_ESTATE, "no prefill has run");
  HIPCHK(hipSetDevice(c->dev));
  SampleArgs sa = make_sample_args(c, nullptr, 0, c->d_logits);
  launch_dec_sample(sa, c->stream);
  HIPCHK(hipMemcpyAsync(out, c->d_logits, (size_t)c->cur_batch * AR_LOGITS * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return VX_OK;
}

int vx_ar_step(vx_ctx* c, const int32_t* tokens) {
  if (!c || !tokens) return VX_EINVAL;
  if (c->cur_batch <= 0) FAIL(VX_ESTATE, "no prefill has run");
  HIPCHK(hipSetDevice(c->dev));
  HIPCHK(hipMemcpyAsync(c->force_tok, tokens, c->cur_batch * sizeof(int), hipMemcpyHostToDevice, c->stream));
  launch_dec_force_token(c->force_tok, c->cur_tok, c->cur_pos, c->ctx_len, c->n_gen, c->gen, c->gen_stride, c->active,
                         c->cur_batch, c->slot_meta, c->slot_of, c->stream);
  if (int e = ar_step_run(c, nullptr, "")) return e;
  HIPCHK(hipStreamSynchronize(c->stream));
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
    HIPCHK(hipMemcpy(slp.data(), c->sum_logp, N * sizeof(float), hipMemcpyDeviceToHost));
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
      HIPCHK(hipMemcpyAsync(audio + (long)jb.row * audio_stride + (long)jb.a * 320,
                            c->vaudio + ((long)seq_off[j] + (jb.a - jb.lo)) * 320, (size_t)(jb.b - jb.a) * 320 * sizeof(float),
                            hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
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
      for (int t = 0; t < maxT; ++t) {
        launch_skinny_gemm(c->ec_whh_p[l], c->ec_hp, c->ec_pg, 2048, 512, 2, st);
        launch_lstm_cell(c->ec_pg, 2, c->ec_xg, d_off, d_len, t, c->ec_c, c->ec_hp, yout, l == 1 ? c->ec_x0 : nullptr, nb, st);
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
      HIPCHK(hipMemcpyAsync(audio + (long)(r0 + i) * audio_stride, c->ec_audio + (long)i * astride,
                            (size_t)seq_len[i] * 320 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
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
      HIPCHK(hipMemcpyAsync(c->ec_audio, wav + (long)(r0 + i) * wav_stride, (size_t)Ls[i][0] * sizeof(float),
                            hipMemcpyHostToDevice, st));
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
    HIPCHK(hipMemcpyAsync(hc.data(), c->en_codes, hc.size() * sizeof(long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
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

int64_t vx_read_tap(vx_ctx* c, const char* name, float* dst, int64_t max_floats) {
  if (!c || !name || !dst) return VX_EINVAL;
  auto it = c->taps.find(name);
  if (it == c->taps.end()) FAIL(VX_ENOTFOUND, "no tap '%s'", name);
  const int64_t n = std::min<int64_t>((int64_t)it->second.n, max_floats);
  HIPCHK(hipSetDevice(c->dev));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(dst, it->second.d, n * sizeof(float), hipMemcpyDeviceToHost));
  return n;
}

int vx_prof_enable(vx_ctx* c, int32_t on) {
  if (!c) return VX_EINVAL;
  c->prof_on = on;
  return VX_OK;
}

int vx_prof_reset(vx_ctx* c) {
  if (!c) return VX_EINVAL;
  for (auto& p : c->prof) { p.used = 0; p.bytes = 0; }
  return VX_OK;
}

int vx_prof_get(vx_ctx* c, int32_t which, double* total_ms, int64_t* launches, double* algo_bytes) {
  if (!c || which < 0 || which > 4) return VX_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  HIPCHK(hipStreamSynchronize(c->stream));
  ProfClass& p = c->prof[which];
  double tot = 0;
  for (size_t i = 0; i + 1 < p.used; i += 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]));
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = (int64_t)(p.used / 2);
  if (algo_bytes) *algo_bytes = p.bytes;
  return VX_OK;
}

// Back-to-back replays of ONE decode kernel on the live state of the last AR run, bracketed by a single HIP event
// pair on the engine stream (GPU-bound: no host gaps inside the interval).  which 0: dec_attn of layer 0 with every
// row's context set to prefill_len + gen_offset; which 1: the five weight-streaming GEMMs of a step's layer 0
// (+ predict layer), reported per launch.
int vx_bench_kernel(vx_ctx* c, int32_t which, int32_t reps, int32_t gen_offset, double* avg_us, double* algo_bytes) {
  if (!c || reps <= 0 || !avg_us || !algo_bytes) return VX_EINVAL;
  if (c->cur_batch <= 0) FAIL(VX_ESTATE, "no AR run to replay");
  HIPCHK(hipSetDevice(c->dev));
  const int nb = c->cur_batch;
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  double bytes = 0;
  int launches = 0;
  if (which == 0) {
    std::vector<int> ctx(nb), one(nb, 1);
    for (int i = 0; i < nb; ++i) {
      ctx[i] = std::min(c->h_L[i] + std::max(gen_offset, 1), c->Tmax - 1);
      bytes += (double)ctx[i] * 2.0 * D_MODEL * 4.0;
    }
    if (c->fuse_out && c->nsplit == 1) bytes += (double)D_MODEL * D_MODEL * 4.0;     // + W_o, streamed once (fused out_proj)
    // the replay's contexts go into the per-slot view dec_attn reads (the row order of the last prefill is kept)
    std::vector<int> meta(4 * nb);
    HIPCHK(hipMemcpyAsync(meta.data(), c->slot_meta, meta.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int y = 0; y < nb; ++y) { meta[4 * y + 1] = ctx[meta[4 * y]]; meta[4 * y + 2] = 1; }
    HIPCHK(hipMemcpyAsync(c->slot_meta, meta.data(), meta.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->ctx_len, ctx.data(), nb * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->active, one.data(), nb * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // rotate over the layers' KV arenas like the real step does: the working set (NL x ~178 MB at batch 32) is far beyond
    // the 256 MiB Infinity Cache, so no launch is served from it
    const size_t cache_layer = (size_t)c->mbr * N_HEAD * c->Tmax * D_HEAD;
    auto attn_l = [&](int r) {
      const int l = r % c->NL;
      const bool fused = c->fuse_out && c->nsplit == 1;
      launch_dec_attn(c->p_qkv, SK_QKV, c->ar[l].in_b, c->kc + l * cache_layer, c->vc + l * cache_layer, c->Tmax,
                      c->slot_meta, c->xp_att, c->part_o, c->part_ml, c->nsplit, nb, fused ? c->ar[l].out_wh : nullptr, c->p_oh,
                      c->stream);
    };
    for (int w = 0; w < 3; ++w) attn_l(w);
    HIPCHK(hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps; ++r) attn_l(r);
    HIPCHK(hipEventRecord(e1, c->stream));
    launches = reps;
  } else if (which == 1) {
    const LayerW& L = c->ar[0];
    auto seq = [&]() {
      launch_skinny_gemm(L.in_wp, c->xp, c->p_qkv, 3 * D_MODEL, D_MODEL, SK_QKV, c->stream);
      launch_skinny_gemm(L.out_wp, c->xp_att, c->p_o, D_MODEL, D_MODEL, SK_OUT, c->stream);
      launch_skinny16_relu_pack(L.l1_wp, c->xp, L.l1_b, c->xp4, D_FF, D_MODEL, c->stream);
      launch_skinny_gemm(L.l2_wp, c->xp4, c->p_o, D_MODEL, D_FF, SK_L2, c->stream);
    };
    seq();
    HIPCHK(hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps; ++r) seq();
    HIPCHK(hipEventRecord(e1, c->stream));
    launches = reps * 4;
    bytes = 12.0 * D_MODEL * D_MODEL * 4.0 / 4.0;     // per launch: a layer's 12 d^2 weights over its 4 GEMMs
  } else if (which == 2) {
    // cache-retention probe: the SAME weight-streaming GEMM (layer 0 QKV, 12.6 MB) back to back -- what a launch costs when
    // its weights were read a moment ago (memory-side cache hits) instead of coming cold from HBM (which 1)
    const LayerW& L = c->ar[0];
    auto one = [&]() { launch_skinny_gemm(L.in_wp, c->xp, c->p_qkv, 3 * D_MODEL, D_MODEL, SK_QKV, c->stream); };
    one();
    HIPCHK(hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps; ++r) one();
    HIPCHK(hipEventRecord(e1, c->stream));
    launches = reps;
    bytes = 3.0 * D_MODEL * D_MODEL * 4.0;
#ifdef VX_DEV_PROBES
  } else if (which == 3) {
    // development timeline (tools/step_timeline.py): `reps` graph replays of a ONE-layer decode step (QKV | attention |
    // reduce+LN | linear1 | linear2 | reduce+LN | predict | sampler) on the live state; the kernels stamp the wall clock
    // (decode.hip) and the caller fetches the stamps of the last replay with vx_dev_stamps.
    vx_sampling sp{};
    sp.struct_size = sizeof(vx_sampling); sp.top_k = 10; sp.temperature = 1.0f; sp.seed = 1; sp.force_eos_at = -1; sp.best_of = 1;
    SampleArgs sa = make_sample_args(c, &sp, 1, nullptr);
    const int nl_keep = c->NL;
    c->NL = 1;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    ar_step_launches(c, &sa);
    HIPCHK(hipStreamEndCapture(c->stream, &g));
    c->NL = nl_keep;
    HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    for (int w = 0; w < 3; ++w) HIPCHK(hipGraphLaunch(ge, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    dev_clear_stamps();
    HIPCHK(hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps; ++r) HIPCHK(hipGraphLaunch(ge, c->stream));
    HIPCHK(hipEventRecord(e1, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)hipGraphExecDestroy(ge);
    launches = reps;
    bytes = 0;
#endif
  } else {
    FAIL(VX_EINVAL, "which must be 0, 1 or 2");
  }
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_us = (double)ms * 1e3 / launches;
  *algo_bytes = bytes;
  HIPCHK(hipGetLastError());
  return VX_OK;
}

#ifdef VX_DEV_PROBES
extern "C" int vx_dev_stamps(unsigned long long* out) { dev_read_stamps(out); return VX_OK; }
extern "C" int vx_dev_gemm_stamps(unsigned long long* out) { dev_read_gemm_stamps(out); return VX_OK; }
#endif

// Stand-alone GEMM micro-benchmark on scratch buffers (kernel development aid; never on the product path):
// kernel 0 = gemm_f32, 1 = gemm_bf16x3, 2 = gemm_bf16x3_dma, 6 = gemm_f16x2 (the default of the model path);
// 11-13 / 21-24 = timing probes of the bf16x3 kernels (VX_DEV_PROBES builds only).  Reports the average launch time and the max abs
// difference of the first and last 256 output rows against the fp32-MFMA kernel.
int vx_bench_gemm(vx_ctx* c, int32_t M, int32_t N, int32_t K, int32_t kernel, int32_t reps, double* avg_us,
                  double* max_abs_diff) {
  if (!c || M <= 0 || N <= 0 || K <= 0 || K % 32 || N % 4 || reps <= 0 || !avg_us || !max_abs_diff) return VX_EINVAL;
#ifndef VX_DEV_PROBES
  if (kernel != 0 && kernel != 1 && kernel != 2 && (kernel < 6 || kernel > 10))
    FAIL(VX_EINVAL, "kernel must be 0, 1, 2 or 6 .. 10 (probes need a VX_DEV_PROBES build)");
#endif
  HIPCHK(hipSetDevice(c->dev));
  float *A = nullptr, *Wt = nullptr, *C0 = nullptr, *C1 = nullptr;
  unsigned short *A3 = nullptr, *W3 = nullptr;
  auto cleanup = [&]() { for (void* p : {(void*)A, (void*)Wt, (void*)C0, (void*)C1, (void*)A3, (void*)W3}) if (p) (void)hipFree(p); };
  hipError_t he;
#define TRY(x) if ((he = (x)) != hipSuccess) { cleanup(); c->err = std::string(#x) + ": " + hipGetErrorString(he); return VX_EHIP; }
  TRY(hipMalloc((void**)&A, (size_t)M * K * 4));
  TRY(hipMalloc((void**)&Wt, (size_t)N * K * 4));
  TRY(hipMalloc((void**)&C0, (size_t)M * N * 4));
  TRY(hipMalloc((void**)&C1, (size_t)M * N * 4));
  TRY(hipMalloc((void**)&A3, (size_t)3 * h2_plane(M, K, H2_TILE_A) * 2));
  TRY(hipMalloc((void**)&W3, (size_t)3 * h2_plane(N, K, H2_TILE_W) * 2));
  TRY(hipMemset(A3, 0, (size_t)3 * h2_plane(M, K, H2_TILE_A) * 2));
  TRY(hipMemset(W3, 0, (size_t)3 * h2_plane(N, K, H2_TILE_W) * 2));
  {
    std::vector<float> h((size_t)std::max(M, N) * K);
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    auto fill = [&](size_t n) { for (size_t i = 0; i < n; ++i) { st = st * 6364136223846793005ull + 1442695040888963407ull; h[i] = (float)((st >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; } };
    fill((size_t)M * K);
    TRY(hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice));
    fill((size_t)N * K);
    TRY(hipMemcpy(Wt, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
  }
  GemmArgs g0{};
  g0.A = A; g0.lda = K; g0.W = Wt; g0.ldw = K; g0.C = C0; g0.ldc = N; g0.M = M; g0.N = N; g0.K = K; g0.act = ACT_NONE;
  launch_gemm_f32(g0, c->stream);
  if ((kernel >= 6 && kernel <= 10) || kernel >= 61) {   // fp16 head / tail planes
    launch_split2h(A, K, M, K, nullptr, A3, h2_plane(M, K, H2_TILE_A), H2_TILE_A, nullptr, H2_ACT_SCALE, c->stream);
    launch_split2h(Wt, K, N, K, nullptr, W3, h2_plane(N, K, H2_TILE_W), H2_TILE_W, nullptr, 16384.0f, c->stream);   // |w| < 1
  } else {
    launch_split3(A, K, M, K, nullptr, A3, (long)M * K, c->stream);
    launch_split3(Wt, K, N, K, nullptr, W3, (long)N * K, c->stream);
  }
  GemmX3Args gx{};
  const bool h2 = (kernel >= 6 && kernel <= 10) || kernel >= 61;
  gx.A = A3; gx.a_plane = h2 ? h2_plane(M, K, H2_TILE_A) : (long)M * K; gx.W = W3; gx.w_plane = h2 ? h2_plane(N, K, H2_TILE_W) : (long)N * K; gx.C = C1; gx.ldc = N; gx.M = M; gx.N = N; gx.K = K;
  gx.act = ACT_NONE;
  gx.descale = ldexpf(1.0f, -(H2_ACT_SHIFT + 14));
  GemmArgs g1 = g0;
  g1.C = C1;
  auto run = [&]() {
    if (kernel == 0) launch_gemm_f32(g1, c->stream);
    else if (kernel == 1) launch_gemm_bf16x3(gx, c->stream);
    else if (kernel == 2) launch_gemm_bf16x3_dma(gx, c->stream);
    else if (kernel == 6) launch_gemm_f16x2(gx, c->stream);              // the product's choice of tile
    else if (kernel == 7) launch_gemm_f16x2(gx, c->stream, 128);
    else if (kernel == 8) launch_gemm_f16x2(gx, c->stream, 256);
    else if (kernel == 9) launch_gemm_f16x2(gx, c->stream, -128);         // 128 x 128 tiles (the short-row-set kernel) forced
    else if (kernel == 10) launch_gemm_f16x2(gx, c->stream, -129);        // ... with two LDS stages forced (A/B of the four-stage ring)
#ifdef VX_DEV_PROBES
    else if (kernel >= 61) launch_gemm_f16x2_probe(gx, kernel - 60, c->stream);         // 61-64: probes of the f16x2 kernel
    else if (kernel >= 21) launch_gemm_bf16x3_dma_probe(gx, kernel - 20, c->stream);   // 21-24: probes of the DMA kernel
    else launch_gemm_bf16x3_probe(gx, kernel - 10, c->stream);      // 11 / 12 / 13: timing probes
#endif
  };
  run();
  // VX_BENCH_CLOCK=1: sample the shader clock on a second stream while the timed launches run (power / clock ceiling check)
  const char* want_clock = getenv("VX_BENCH_CLOCK");
  hipStream_t s2 = nullptr;
  unsigned long long* d_clk = nullptr;
  if (want_clock && want_clock[0] == '1') {
    TRY(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    TRY(hipMalloc((void**)&d_clk, 16));
    TRY(hipStreamSynchronize(c->stream));
  }
  hipEvent_t e0, e1;
  TRY(hipEventCreate(&e0));
  TRY(hipEventCreate(&e1));
  TRY(hipEventRecord(e0, c->stream));
  run();                                                           // the probe starts once the device is busy
  if (s2) hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, s2, d_clk, 100ull * 2000ull);   // 2 ms at 100 MHz
  for (int r = 1; r < reps; ++r) run();
  TRY(hipEventRecord(e1, c->stream));
  TRY(hipEventSynchronize(e1));
  float ms = 0;
  TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_us = (double)ms * 1e3 / reps;
  if (s2) {
    unsigned long long hclk[2] = {0, 0};
    TRY(hipStreamSynchronize(s2));
    TRY(hipMemcpy(hclk, d_clk, 16, hipMemcpyDeviceToHost));
    fprintf(stderr, "[vx_bench_gemm] kernel %d M=%d N=%d K=%d: shader clock while running = %.0f MHz (%llu ticks in %.3f ms)\n", kernel,
            M, N, K, hclk[1] ? (double)hclk[0] / ((double)hclk[1] / 100.0) : 0.0, hclk[0], (double)hclk[1] / 1e5);
    (void)hipFree(d_clk);
    (void)hipStreamDestroy(s2);
  }
  const int rows = std::min(M, 256);
  std::vector<float> h0((size_t)rows * N), h1((size_t)rows * N);
  TRY(hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost));
  TRY(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
  // also the LAST rows (tile tails)
  double md = 0;
  for (size_t i = 0; i < h0.size(); ++i) md = std::max(md, (double)fabsf(h0[i] - h1[i]));
  TRY(hipMemcpy(h0.data(), C0 + (size_t)(M - rows) * N, h0.size() * 4, hipMemcpyDeviceToHost));
  TRY(hipMemcpy(h1.data(), C1 + (size_t)(M - rows) * N, h1.size() * 4, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < h0.size(); ++i) md = std::max(md, (double)fabsf(h0[i] - h1[i]));
  *max_abs_diff = md;
#undef TRY
  cleanup();
  HIPCHK(hipGetLastError());
  return VX_OK;
}

// kernel-development aid: time attn_full (variant 0) or one of its probes (1 no staging, 2 no MFMA, 3 no softmax) on
// random q|k|v for `batch` sequences of length `len`, unmasked (NAR) or prefix-LM with prefix = len/3 (causal != 0).
int vx_bench_attn(vx_ctx* c, int32_t batch, int32_t len, int32_t causal, int32_t variant, int32_t reps, double* avg_us,
                  double* max_diff) {
  // variant: 0 fp32 kernel, 1-3 its probes; 10 bf16x3 kernel, 11-13 its probes.  max_diff (optional) = max |out - out of
  // the fp32 kernel| for the product variants (0 / 10), -1 for probes.
  if (!c || batch <= 0 || len <= 0 || reps <= 0 || !avg_us) return VX_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  const long M = (long)batch * len;
  float *qkv = nullptr, *out = nullptr, *ref = nullptr;
  int* meta = nullptr;
  auto cleanup = [&]() { for (void* p : {(void*)qkv, (void*)out, (void*)ref, (void*)meta}) if (p) (void)hipFree(p); };
  hipError_t he;
#define TRY(x) if ((he = (x)) != hipSuccess) { cleanup(); c->err = std::string(#x) + ": " + hipGetErrorString(he); return VX_EHIP; }
  TRY(hipMalloc((void**)&qkv, (size_t)M * 3 * D_MODEL * 4));
  TRY(hipMalloc((void**)&out, (size_t)M * D_MODEL * 4));
  TRY(hipMalloc((void**)&ref, (size_t)M * D_MODEL * 4));
  TRY(hipMalloc((void**)&meta, (size_t)3 * batch * 4));
  {
    std::vector<float> h((size_t)M * 3 * D_MODEL);
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    for (auto& v : h) { st = st * 6364136223846793005ull + 1442695040888963407ull; v = (float)((st >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; }
    // Q columns x4: scores of a few units instead of ~0.3, so the softmax is not nearly uniform
    for (long r = 0; r < M; ++r) for (int k = 0; k < D_MODEL; ++k) h[(size_t)r * 3 * D_MODEL + k] *= 4.0f;
    TRY(hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<int> m(3 * batch);
    for (int i = 0; i < batch; ++i) { m[i] = i * len; m[batch + i] = len; m[2 * batch + i] = len / 3; }
    TRY(hipMemcpy(meta, m.data(), m.size() * 4, hipMemcpyHostToDevice));
  }
  const int* pre = causal ? meta + 2 * batch : nullptr;
  auto run = [&]() {
    if (variant == 0) launch_attn_full(qkv, out, meta, meta + batch, pre, batch, len, c->stream);
#ifdef VX_DEV_PROBES
    else if (variant < 10) launch_attn_full_probe(qkv, out, meta, meta + batch, pre, batch, len, variant, c->stream);
#else
    else if (variant < 10) return;
#endif
    else if (variant == 20) launch_attn_full_h2(qkv, out, meta, meta + batch, pre, batch, len, c->stream, nullptr, 0, nullptr);
    else launch_attn_full_x3(qkv, out, meta, meta + batch, pre, batch, len, variant - 10, c->stream);
  };
  run();
  hipEvent_t e0, e1;
  TRY(hipEventCreate(&e0));
  TRY(hipEventCreate(&e1));
  TRY(hipEventRecord(e0, c->stream));
  for (int r = 0; r < reps; ++r) run();
  TRY(hipEventRecord(e1, c->stream));
  TRY(hipEventSynchronize(e1));
  float ms = 0;
  TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_us = (double)ms * 1e3 / reps;
  if (max_diff) {
    *max_diff = -1.0;
    if (variant == 0 || variant == 10 || variant == 20) {
      launch_attn_full(qkv, ref, meta, meta + batch, pre, batch, len, c->stream);
      TRY(hipStreamSynchronize(c->stream));
      std::vector<float> ho((size_t)M * D_MODEL), hr((size_t)M * D_MODEL);
      TRY(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
      TRY(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
      double md = 0;
      for (size_t i = 0; i < ho.size(); ++i) {
        const double d = std::fabs((double)ho[i] - (double)hr[i]);
        md = (d > md || d != d) ? (d != d ? 1e30 : d) : md;
      }
      *max_diff = md;
    }
  }
#undef TRY
  cleanup();
  HIPCHK(hipGetLastError());
  return VX_OK;
}

int vx_last_fallbacks(vx_ctx* c, int32_t* prefill_phases, int32_t* nar_phases, int64_t* lifetime) {
  if (!c) return VX_EINVAL;
  if (prefill_phases) *prefill_phases = c->st_fb_prefill;
  if (nar_phases) *nar_phases = c->st_fb_nar;
  if (lifetime) *lifetime = c->fb_total;
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
