// Internal declarations shared by the engine's translation units (engine.hip: context, AR / NAR drivers and the hot-path ABI;
// weights.hip: ingest of the reference state-dict; vocoders.hip: Vocos head, EnCodec decoder / encoder drivers;
// bench_harness.hip: the measurement entries of include/vallex_hip_dev.h).  Not part of the public C ABI.
#pragma once

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vallex_hip.h"
#include "vx_common.h"

using namespace vx;

namespace vxe {

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

// Host <-> device transfers.  Every byte that crosses the boundary goes through a context-owned PINNED ring (hipHostMalloc, one per
// context = per GPU = per rank): a host source is copied into the ring and DMA'd from there, a host destination is filled from the
// ring behind the next stream sync.  No call of the library hands a caller's (or its own) pageable pages to the runtime, so nothing
// depends on the driver pinning user memory on the fly (the userptr path -- the one a box-level fault of round 5 died in, inside the
// first weight upload, replacing load_state_dict + .to(device) of utils/generation.py:79-83).  Transfers larger than a chunk are cut
// into chunks, so the CPU copy of chunk i + 1 overlaps the DMA of chunk i; the ring wraps by synchronising the stream.
struct PinRing {
  char* base = nullptr;
  size_t cap = 0, head = 0;
  struct Pend { void* dst; const char* src; size_t n; };
  std::vector<Pend> pend;          // device -> host copies in flight: ring slot -> caller memory at the next xfer_sync
};
constexpr size_t XFER_CHUNK = 8u << 20;

struct ProfClass {
  std::vector<hipEvent_t> ev;   // pairs
  size_t used = 0;
  double bytes = 0;
};

constexpr int SK_QKV = 4, SK_OUT = 4, SK_L2 = 8, SK_PRED = 4;
constexpr int PRED_NPAD = 1056;
constexpr int FB_STICKY_AFTER = 2;
constexpr int FB_STICKY_PROBE_EVERY = 32;

}  // namespace vxe
using namespace vxe;

struct vx_ctx {
  vx_config cfg{};
  int dev = 0;
  hipStream_t stream = nullptr;
  PinRing ring;                    // pinned staging of every host transfer (xfer_h2d / xfer_d2h / xfer_sync)
  hipEvent_t ev_t[3] = {nullptr, nullptr, nullptr};   // AR / NAR phase timing of vx_infer (created once, vx_create)
  std::string err;
  const char* launch_fail = nullptr;   // a launcher refused a configuration that is not compiled in (set by LAUNCH, read by the ABI call)
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
  // sticky fallback: a checkpoint whose operands leave the fp16 range on (nearly) every call would pay an f16x2 pass AND an fp32
  // pass each time.  After FB_STICKY_AFTER CONSECUTIVE raises of a phase kind (a clean f16x2 pass of that kind resets the count: two
  // outlier inputs days apart in a long-running server never add up) that kind runs on the exact-fp32 kernels directly (still
  // counted by vx_last_fallbacks); while the count is non-zero the NAR phase polls the flag right behind stage 0 instead of behind
  // stage 6.  Sticky mode is not for ever: every FB_STICKY_PROBE_EVERY-th phase of the kind is tried on f16x2 again and a clean pass
  // leaves it (a burst of outlier inputs costs at most that many fp32 phases).  vx_fallback_state reports it, vx_fallback_reset
  // clears it.
  int fb_prefill_raises = 0, fb_nar_raises = 0;
  bool sticky_prefill_f32 = false, sticky_nar_f32 = false;
  int sticky_prefill_age = 0, sticky_nar_age = 0;     // fp32-direct phases since sticky mode engaged / since the last probe
  long sticky_engaged = 0;                            // how many times either kind ENTERED sticky mode since vx_create
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
  bool sb_qkv = false;             // ... on the small-batch chain with norm1 + QKV folded into the attention launch
  int sb_qkv_rows = 4, sb_qkv_nsplit = 0;   // sb_qkv up to this many rows (VX_SB_QKV=n, 0 = off), forced split count (VX_SB_QKV_NSPLIT)
  bool sb_chain = false;           // the current micro-batch decodes on the small-batch chain (set by ar_prefill)
  bool sb_fuse = true;             // <= SB_ROWS rows: reduce+LN / combine folded into the consuming GEMM (VX_SB_FUSE=0: the general chain)
  bool qkv_bal = true;             // the decode in_proj GEMM on 512 workgroups (8 K slices of q, 4 of k / v; VX_QKV_BALANCED=0: 384 x 4 slices)
  float* qk_new = nullptr;         // [MB][16][2][64]: q / 8 and k_new of the step's new token (dec_attn_qkv_kernel -> out_proj prologue)
  float *p_qkv = nullptr, *p_o = nullptr, *p_oh = nullptr, *p_logits = nullptr, *part_o = nullptr, *part_ml = nullptr;
  std::map<const unsigned short*, int> w_shift;   // f16x2: power-of-two scale exponent of every weight's planes
  bool nar_trim = true;            // last NAR layer computes only the generated rows (VX_NAR_TRIM=1; engine.hip struct Trim)
  bool balance_rows = true;        // dec_attn launch order pairs long with short contexts per CU (VX_BALANCE_ROWS=0: batch order)
  bool fuse_out = true;            // out_proj folded into dec_attn when nsplit == 1 (VX_FUSE_OUT=0: separate skinny GEMM)
  int fuse_split = 1;              // ... and, 8 .. 16 rows, with 2 .. 4 context splits: slabs per (head, split) weighed by the consumer (VX_FUSE_SPLIT=0: dec_attn | combine | out_proj; 2: whenever VX_ATT_NSPLIT forces 2 .. 4 splits)
  bool split_fused = false;        // decided per micro-batch by the prefill
  float *d_logits = nullptr, *d_uniforms = nullptr, *sum_logp = nullptr;
  long uniforms_cap = 0;
  int *cur_tok = nullptr, *cur_pos = nullptr, *ctx_len = nullptr, *n_gen = nullptr, *active = nullptr,
      *text_len = nullptr, *gen = nullptr, *force_tok = nullptr, *n_active = nullptr, *slot_meta = nullptr, *slot_of = nullptr;
  int gen_stride = 0;
  int cur_batch = 0;
  int nsplit = 1;
  int att_nsplit_force = 0;        // VX_ATT_NSPLIT=n: context splits of dec_attn on the general chain (0 = 512 / (rows x 16) workgroups rule)
  std::vector<int> h_L;            // prefill lengths of the current micro-batch

  // graph
  hipGraphExec_t graph_exec = nullptr, graph_exec_n = nullptr;   // one decode step / GRAPH_STEPS steps per launch
  bool graph_multi = true;         // several steps per graph launch (VX_GRAPH_MULTI=0: one)
  std::string graph_sig;

  // taps
  std::map<std::string, Tensor> taps;

  // profiling / stats
  int prof_on = 0;                 // 0 off, 1 every class (AR step runs eagerly), 2 full-sequence classes only
  ProfClass prof[6];             // 0 dec_attn, 1 skinny GEMMs, 2 projections, 3 full-seq attention, 4 vocoder GEMMs, 5 LSTM recurrences
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

// launchers that compile a split count in return false instead of launching an uninstantiated configuration
#define LAUNCH(call)                                      \
  do {                                                    \
    if (!(call) && !c->launch_fail) c->launch_fail = #call; \
  } while (0)

namespace vxe {

int xfer_h2d(vx_ctx* c, void* dst_dev, const void* src_host, size_t bytes);    // asynchronous on c->stream; src is free on return
int xfer_d2h(vx_ctx* c, void* dst_host, const void* src_dev, size_t bytes);    // dst is valid after the next xfer_sync
int xfer_sync(vx_ctx* c);                                                      // stream sync + delivery of pending d2h + ring reset
#define H2D(dst, src, bytes) do { if (int _e = xfer_h2d(c, (dst), (src), (bytes))) return _e; } while (0)
#define D2H(dst, src, bytes) do { if (int _e = xfer_d2h(c, (dst), (src), (bytes))) return _e; } while (0)
#define SYNC() do { if (int _e = xfer_sync(c)) return _e; } while (0)

template <typename T>
int dev_alloc(vx_ctx* c, T** p, size_t count, bool zero = true) {
  void* q = nullptr;
  HIPCHK(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
  c->allocs.push_back(q);
  if (zero) HIPCHK(hipMemsetAsync(q, 0, std::max<size_t>(count, 1) * sizeof(T), c->stream));
  *p = reinterpret_cast<T*>(q);
  return VX_OK;
}

const float* W(vx_ctx* c, const std::string& name);

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

int upload_meta(vx_ctx* c);

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

int tap_store(vx_ctx* c, const std::string& name, const float* src, size_t n);
// C = resid + colscale * act(A W^T + bias) on the fp32 MFMA (cls: profiling class, 2 = transformer projections, 4 = vocoders)
void gemm(vx_ctx* c, const float* A, int lda, const float* Wt, int ldw, const float* bias, const float* resid, int ldr,
          const float* colscale, float* C, int ldc, long M, int N, int K, int act, const int* gather = nullptr, int cls = 4,
          const int* resid_rows = nullptr);
bool range_guarded(const vx_ctx* c);
int ensure_f32_buffers(vx_ctx* c);
int take_range_flag(vx_ctx* c, bool* raised);
int check_batch(vx_ctx* c, const vx_batch* b, int max_rows);
SampleArgs make_sample_args(vx_ctx* c, const vx_sampling* s, int commit, float* logits_out);
void ar_step_launches(vx_ctx* c, const SampleArgs* sa);
int launch_status(vx_ctx* c);      // VX_EINVAL (+ message) if a launcher refused since the last check

struct F32Scope {            // the full-sequence path on the exact-fp32 kernels for the lifetime of the object
  vx_ctx* c;
  int gm;
  bool ax;
  explicit F32Scope(vx_ctx* c_) : c(c_), gm(c_->gemm_mode), ax(c_->attn_x3) { c->gemm_mode = 2; c->attn_x3 = false; }
  ~F32Scope() { c->gemm_mode = gm; c->attn_x3 = ax; }
};

}  // namespace vxe
