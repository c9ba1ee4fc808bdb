// Shared declarations for the gfx950 VALL-E X engine (internal; the public C-ABI is include/vallex_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vx {

constexpr int D_MODEL = 1024;
constexpr int N_HEAD = 16;
constexpr int D_HEAD = 64;
constexpr int D_FF = 4096;
constexpr int N_Q = 8;                 // codebooks
constexpr int AUDIO_VOCAB = 1024;
constexpr int EOS_ID = 1024;
constexpr int BOS_ID = 1025;
constexpr int AR_LOGITS = 1025;
constexpr int MB = 32;                 // AR micro-batch rows held by one MFMA column block
constexpr float LN_EPS = 1e-5f;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_ELU = 3 };

// ---- big fp32 MFMA GEMM (gemm_f32.hip) -------------------------------------------------------
// C[m][n] = resid[m][n] + colscale[n] * act(sum_k A[m][k] * W[n][k] + bias[n]),  m<M, n<N
// N % 4 == 0, K % 32 == 0, all leading dims % 4 == 0; A must be readable for rows < M.
struct GemmArgs {
  const float* A; int lda;
  const float* W; int ldw;
  const float* bias;        // [N] or null
  const float* resid; int ldr;   // [M][ldr] or null
  const float* colscale;    // [N] or null
  float* C; int ldc;
  int M, N, K;
  int act;
  const int* row_gather;    // optional: A row index per output row (null = identity)
  int walk;                 // tile order override (tile_walk below): 0 = the kernel's own; else group depth | column-fastest << 8
  const int* resid_rows;    // optional: row of `resid` read for output row m (null = m): compacted row sets (trimmed last NAR layer)
};
void launch_gemm_f32(const GemmArgs& g, hipStream_t s, int variant = 0);   // variant: gemm_f32.hip (0 = product choice)

// ---- fp32-class GEMMs on the 16-bit matrix cores ----------------------------------------------------------------
// Operands are pre-split planes, each K-tile-major [K/32][rows][32] 16-bit; same epilogue contract as GemmArgs.  K % 32 == 0.
//   f16x2  (gemm_f16x2.hip, DEFAULT): 2 planes  x = h + t/2048 (fp16 head + fp16 tail), 3 f16 MFMAs per product block;
//          its planes are TILE-major (see launch_split2h), not K-tile-major
//   bf16x3 (gemm_bf16x3*.hip, VX_GEMM_X3=1): 3 bf16 planes x = x1 + x2 + x3, 6 bf16 MFMAs per product block
struct GemmX3Args {
  const unsigned short* A; long a_plane;   // [P][M][K], plane stride in elements (P = 2 or 3)
  const unsigned short* W; long w_plane;   // [P][N][K]
  const float* bias; const float* resid; int ldr; const float* colscale;
  float* C; int ldc;
  int M, N, K;
  int act;
  // f16x2 only, optional: write the result as the f16x2 A planes of the NEXT GEMM (tile-major, K = this N) instead of fp32 rows
  unsigned short* out_planes; long out_plane;   // plane stride in elements: h2_plane(M, N, H2_TILE_A)
  int* range_flag;
  float descale;                                // f16x2: 2^-(shift of A + shift of W), applied to the accumulator (exact)
  const int* resid_rows;                        // f16x2 only, optional: row of `resid` for output row m (null = m): compacted row sets
  int dev_variant;                              // tools builds only (-DVX_DEV_PROBES): > 0 selects a wave-priority variant of the 256 x 256 kernel
  int walk;                                     // tile order override (tile_walk below): 0 = the kernel's own; else group depth | column-fastest << 8
};
void launch_gemm_f16x2(const GemmX3Args& g, hipStream_t s, int tn = 0);     // 256 x (256 | 128) x 32 tiles, async LDS fill, any M
// f16x2 planes are TILE-major: [rows / tile_rows][K/32][tile_rows][32], tile_rows = 256 (both operands);
// plane_stride >= roundup(rows, tile_rows) * K.  *range_flag = 1 if some |x| does not fit fp16.
constexpr int H2_TILE_A = 256, H2_TILE_W = 256;
// f16x2 operand format: X = x * 2^shift,  head = fp16(X),  tail = fp16(X - head)  (both planes at the SAME scale, so all three
// products of a block go into ONE fp32 accumulator).  Activations use one fixed shift (|x| < 65504 / 32 = 2047: LayerNorm outputs,
// attention outputs, ReLU'd FFN activations); every weight tensor gets its own from max |w| at load.  A tail below 2^-14 is an
// fp16 subnormal (v_mfma honours them): absolute error <= 2^-25 * 2^-shift, far below fp32 resolution of the sums.
constexpr int H2_ACT_SHIFT = 5;
constexpr float H2_ACT_SCALE = 32.0f;
// VX_GEMM_WALK="<group depth>[,c]" (read once per process by the GEMM launchers; kernel development: tools/gemm_walk_sweep.py)
int gemm_walk_env();
#if defined(__HIPCC__)
// Tile order of the dense GEMM kernels.  (1) XCD-aware: consecutive workgroup ids land on different XCDs (id % 8), each with a
// private 4 MiB L2, so ids are remapped to give every XCD one contiguous run of the order.  (2) Inside that order tiles come in
// groups of `gm` row tiles x all column tiles; inside a group the row tile runs fastest (walk >> 8 == 0: the ~32 workgroups an XCD
// keeps resident cover gm row tiles x 32 / gm column tiles) or the column tile does (walk >> 8 == 1: 32 / tiles_n row tiles x all
// column tiles).  walk == 0: the measured default, see below.
__device__ __forceinline__ void tile_walk(int wg, int tiles_m, int tiles_n, int gm_default, int walk, int& tm, int& tn) {
  const int nwg = tiles_m * tiles_n;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // walk == 0: N of at most four column tiles (out_proj, linear2: N = 1024) walks column-fastest -- a row tile's column tiles start
  // back to back and share its A panel while it streams (round 6 sweep, profiles/r06_gemm_walk_*: FETCH_SIZE 691 -> 491 MB per launch on
  // those shapes, -3..5 % time on the f16x2 kernel); wider N keeps the row-fastest order, which measured best there
  if (!walk) walk = gm_default | (tiles_n <= 4 ? 256 : 0);
  const int GM = walk & 255;
  const int per_group = GM * tiles_n;
  const int grp = wg / per_group, in_grp = wg - grp * per_group;
  const int gm0 = grp * GM;
  const int gm_rows = (tiles_m - gm0 < GM) ? tiles_m - gm0 : GM;   // last group may be short
  if (walk >> 8) { tm = gm0 + in_grp / tiles_n; tn = in_grp - (in_grp / tiles_n) * tiles_n; }
  else { tm = gm0 + in_grp % gm_rows; tn = in_grp / gm_rows; }
}
__device__ __forceinline__ void h2_split(float x, float scale, _Float16& h, _Float16& t, bool& bad) {
  const float X = x * scale;                    // power of two: exact
  bad |= !(fabsf(X) < 65504.0f);
  h = (_Float16)X;                              // RNE
  t = (_Float16)(X - (float)h);                 // exact difference, rounded once
}
#endif
inline long h2_plane(long rows, int K, int tile_rows) { return (rows + tile_rows - 1) / tile_rows * tile_rows * (long)K; }
void launch_split2h(const float* x, int ldx, long rows, int K, const int* gather, unsigned short* planes, long plane_stride,
                    int tile_rows, int* range_flag, float scale, hipStream_t s);
// max |x| over a tensor, as the bit pattern of the float (device word, atomicMax; zero it first)
void launch_absmax(const float* x, long n, unsigned* out_bits, hipStream_t s);
void launch_gemm_bf16x3(const GemmX3Args& g, hipStream_t s);                // register-staged 128 x 128 tile (short row sets)
void launch_gemm_bf16x3_dma(const GemmX3Args& g, hipStream_t s);            // 256 x 128 tile, async LDS fill (M >= 1024)
void launch_split3(const float* x, int ldx, long rows, int K, const int* gather, unsigned short* planes,
                   long plane_stride, hipStream_t s);
#ifdef VX_DEV_PROBES
void launch_gemm_bf16x3_probe(const GemmX3Args& g, int variant, hipStream_t s);
void launch_gemm_bf16x3_dma_probe(const GemmX3Args& g, int variant, hipStream_t s);
void launch_gemm_f16x2_probe(const GemmX3Args& g, int variant, hipStream_t s);
#endif

// ---- row-wise ops (rows.hip) --------------------------------------------------------------------
// y = (LN(x) * g + b) [* ada_w + ada_b]; any of g/b/ada_* may be null.  C in {1024, 384}.
// planes (C = 1024 only, optional; y may then be null): also write the f16x2 A planes of the consuming GEMM (tile-major, K = C)
void launch_layernorm(const float* x, int ldx, float* y, int ldy, int rows, int C, float eps, const float* g,
                      const float* b, const float* ada_w, const float* ada_b, hipStream_t s,
                      unsigned short* planes = nullptr, long plane_stride = 0, int* range_flag = nullptr);
// out[dst? dst[r] : r] = tabA[idA[r]] (+ tabB[idB[r]] if idB && idB[r] >= 0) + alpha * pe[pos[r]]   (rows of 1024)
void launch_embed_rows(float* out, const int* dst, const float* tabA, const int* idA, const float* tabB,
                       const int* idB, const float* alpha, const float* pe, const int* pos, int rows, hipStream_t s);
// y_emb[r] = sum_{j < nj[r]} tabs[j][codes[r*8+j]]   (tabs: 8 table pointers in a device array)
void launch_nar_yemb_init(float* yemb, const float* const* tabs, const int* codes, const int* nj, int rows,
                          hipStream_t s);
// out[dst[r]] = yemb[r] + alpha * pe[pos[r]]
void launch_add_pe_scatter(float* out, const int* dst, const float* yemb, const float* alpha, const float* pe,
                           const int* pos, int rows, hipStream_t s);
// yemb[rowidx[i]] += tab[tok[i]]
void launch_embed_accum(float* yemb, const int* rowidx, const float* tab, const int* tok, int n, hipStream_t s);
// idx[r] = argmax_n x[r][n] (first max), n < N
void launch_argmax_rows(const float* x, int ldx, int rows, int N, int* idx, hipStream_t s);
// copy K,V of packed prefill rows into the cache: qkv [M][3072] -> cache[(b*H+h)*Tmax + t][64]
void launch_kv_scatter(const float* qkv, const int* row_b, const int* row_t, int M, float* kc, float* vc, int Tmax,
                       hipStream_t s);
// out[r] = W[r] . e + b[r]  (tiny GEMV used once at load for the AdaLN stage projections)
void launch_gemv(const float* W, const float* e, const float* b, float* out, int N, int K, hipStream_t s);

// ---- full-sequence attention (attn_full.hip) ---------------------------------------------------
// qkv packed rows [M][3072] (q|k|v, heads = contiguous 64-wide slices, modules/activation.py:144-147).
// Sequence b owns rows [seq_off[b], seq_off[b]+seq_len[b]).  prefix_len[b] = S_b for the AR prefix-LM mask
// (text rows see text only; audio rows see all text + causal audio, models/vallex.py:535-549), or null for
// the unmasked NAR attention.
void launch_attn_full(const float* qkv, float* out, const int* seq_off, const int* seq_len, const int* prefix_len,
                      int batch, int max_len, hipStream_t s, const int* q_first = nullptr, const int* c_off = nullptr, int nbuf = 0);

// bf16x3 version (attn_full_x3.hip), the product path; variant 0 = product (1-5 = timing probes, VX_DEV_PROBES builds only)
// planes (optional; out may then be null): write the result as the f16x2 A planes of out_proj (tile-major, K = 1024)
void launch_attn_full_x3(const float* qkv, float* out, const int* seq_off, const int* seq_len, const int* prefix_len,
                         int batch, int max_len, int variant, hipStream_t s, unsigned short* planes = nullptr,
                         long plane_stride = 0);
// f16x2 version (attn_full_h2.hip): three f16 MFMAs per block, operands scaled by powers of two (see the file); *range_flag = 1
// if an operand head did not fit fp16 (the output is then non-finite)
void launch_attn_full_h2(const float* qkv, float* out, const int* seq_off, const int* seq_len, const int* prefix_len,
                         int batch, int max_len, hipStream_t s, unsigned short* planes, long plane_stride, int* range_flag,
                         int prio = -1,       // prio >= 0: wave-priority variant (tools builds only; see the kernel)
                         const int* q_first = nullptr, const int* c_off = nullptr);   // row trimming (planes mode; see the kernel)
#ifdef VX_DEV_PROBES
void launch_attn_full_probe(const float* qkv, float* out, const int* seq_off, const int* seq_len, const int* prefix_len,
                            int batch, int max_len, int variant, hipStream_t s);   // timing probes (tools/attn_bench.py)
#endif

// ---- AR decode step (decode.hip) ---------------------------------------------------------------
// packed skinny-GEMM weight image: tiles of 32 n-rows x 8 k, lane-linear (see decode.hip)
void launch_pack_weight(const float* W, int N, int K, float* Wp, int Npad, hipStream_t s);
// partial[ks][b][n] = sum_{k in slice ks} x[b][k] * W[n][k];  xp is the packed activation image.
void launch_skinny_gemm(const float* Wp, const float* xp, float* partial, int Npad, int K, int splitk, hipStream_t s, bool wt = false);
// slab counts of the balanced in_proj (skinny_qkv_bal_kernel): q in 8 K slices, k / v in 4.  dec_attn_kernel's SK template parameter
// carries both as SKQ * 10 + SKV (a plain count <= 10 means the same count for q, k and v: skinny_gemm_kernel with splitk = 4).
constexpr int SK_QKV_BAL_Q = 8, SK_QKV_BAL_KV = 4;
constexpr int SK_QKV_BALANCED = SK_QKV_BAL_Q * 10 + SK_QKV_BAL_KV;
void launch_skinny_qkv_balanced(const float* Wp, const float* xp, float* partial, hipStream_t s);   // 512 workgroups: q in 8 K slices, k / v in 4 (default since round 5)
// h[b] = (resid? resid[b] : 0) + sum_ks partial[ks][b] + bias ; xp = pack(LN(h)*g+b)   (N = 1024)
bool launch_dec_reduce_ln_split(const float* slabs, const float* part_ml, int nsplit, const float* bias, const float* resid, float* h,
                                const float* g, const float* b, float* xp, int batch, hipStream_t s);      // consumer of the fused out_proj with context splits (2 .. 4)
void launch_dec_reduce_ln_pack(const float* partial, int splitk, int npad, const float* bias, const float* resid,
                               float* h, const float* g, const float* b, float* xp, int batch, hipStream_t s);
// linear1 with fused bias+ReLU+pack on 16-row tiles (v_mfma_f32_16x16x4_f32), 256 workgroups, no split-K
void launch_pack_weight16(const float* W, int N, int K, float* Wp, hipStream_t s);
void launch_skinny16_relu_pack(const float* W16, const float* xp, const float* bias, float* xp_out, int N, int K,
                               hipStream_t s);
// small batches (<= SB_ROWS rows): the consumer GEMM computes its own input rows in its prologue (decode.hip)
constexpr int SB_ROWS = 4;
// The launchers below that return bool compile a producer's split count into the kernel: false = that configuration is not
// instantiated and NOTHING was launched (the engine turns it into VX_EINVAL; a library must never abort() its host process).
// sb_chain_supported: may the small-batch chain run with these split counts?  (otherwise the engine takes the general chain)
bool sb_chain_supported(int sk_l2, int sk_out, int nsplit, int batch);
// small batches with norm1 + the QKV projection folded into the split attention launch (decode.hip: dec_attn_qkv_kernel): grid
// (head, slot, nsplit), nsplit in {4, 8, 16}; part_o / part_ml hold nsplit + 1 partials per (row, head) -- the last one is the new
// token's: its output (v_new) is written here, its score q . k_new is formed by the consumer from qk_new [MB][16][2][64] = (q / 8, k_new)
// (launch_skinny_gemm_sb_combine with nsplit + 1 and qk_new).  skp = slabs of the previous layer's linear2 (partial_in, + pbias +
// resid -> LayerNorm with g, b; the workgroup of head 0, split 0 writes h_out), or 0: x is the packed image xp.
bool sb_qkv_chain_supported(int sk_l2, int sk_out, int nsplit, int batch);
bool launch_dec_attn_qkv(const float* in_w, const float* in_b, float* kc, float* vc, int Tmax, const int* slot_meta, float* part_o,
                         float* part_ml, float* qk_new, int nsplit, int batch, const float* partial_in, int skp, const float* pbias,
                         const float* resid, float* h_out, const float* g, const float* b, const float* xp, hipStream_t s);
// partial_out[ks][b][n] = sum_k LN(resid[b] + sum_ks' partial_in[ks'][b] + bias)[k] W[n][k]   (K = 1024); workgroup 0 writes h_out
bool launch_skinny_gemm_sb_ln(const float* Wp, float* partial_out, int Npad, int splitk, const float* partial_in, int sk_in,
                              const float* bias, const float* resid, float* h_out, const float* g, const float* b, int batch,
                              hipStream_t s);
// the same GEMM on the combine of dec_attn's context-split partials (out_proj, K = 1024)
bool launch_skinny_gemm_sb_combine(const float* Wp, float* partial_out, int Npad, int splitk, const float* part_o, const float* part_ml,
                                   int nsplit, int batch, hipStream_t s, const float* qk_new = nullptr);
// linear1 (bias + ReLU + pack fused, 16-row tiles) on LN(resid + sum of the out_proj slabs + pbias)
bool launch_skinny16_sb_ln(const float* W16, const float* bias, float* xp_out, int N, const float* partial_in, int sk_in,
                           const float* pbias, const float* resid, float* h_out, const float* g, const float* b, int batch,
                           hipStream_t s);
// h[b] = tab[tok[b]] + alpha*pe[pos[b]] ; xp = pack(LN(h))  -- start of a decode step
void launch_dec_embed_ln_pack(const int* tok, const int* pos, const float* tab, const float* alpha, const float* pe,
                              float* h, const float* g, const float* b, float* xp, int batch, hipStream_t s);
// one-token attention over the cache, per (b, head, split); appends the new k/v (from the QKV partials).
// wo_heads != null and nsplit == 1: out_proj fused into the epilogue, per-head partial slabs out_heads[h][MB][1024]
constexpr int DEC_ATTN_TILE = 128;   // rows of one dec_attn tile (8 waves x 4 lane groups x 4 rows); the fused variant needs Tmax >= this
// slot_meta [batch][4] = {row, cached rows incl. the new token, active, -} per launch slot (kept current by dec_sample /
// dec_force_token through slot_of[row])
bool launch_dec_attn(const float* qkv_partial, int splitk, const float* qkv_bias, float* kc, float* vc, int Tmax,
                     const int* slot_meta, float* xp_out, float* part_o, float* part_ml, int nsplit, int batch,
                     const float* wo_heads, float* out_heads, hipStream_t s);
void launch_pack_wo_heads(const float* W, float* out, hipStream_t s);
void launch_dec_attn_combine(const float* part_o, const float* part_ml, int nsplit, const int* active, float* xp_out,
                             int batch, hipStream_t s);
struct SampleArgs {
  const float* partial; int splitk; int npad;     // logits partials [splitk][MB][npad]
  int top_k; float temperature;
  const float* uniforms; int uniforms_stride;     // [steps][batch] or null
  const unsigned long long* seed_dev;             // device word (not a launch constant: a new seed must not re-capture the graph)
  int force_eos_at;
  int commit;                                      // 0: only reduce (and export) the logits
  int* cur_tok; int* cur_pos; int* ctx_len; int* n_gen; int* active; int* n_active; const int* text_len;
  int* slot_meta; const int* slot_of;              // dec_attn's per-slot view of (row, ctx_len, active); slot of each row
  int* gen; int gen_stride;                        // generated ids [b][gen_stride]
  float* logits_out;                               // optional [MB][1025] copy of the reduced logits
  float* sum_logp;                                 // optional [MB] running sum of log p(pick) per row (beam selection)
  int batch;
  // optional fused start of the next step (dec_embed_ln_pack for the committed token); emb_tab == null: off
  const float* emb_tab; const float* emb_alpha; const float* pe; const float* ln_g; const float* ln_b;
  float* emb_h; float* emb_xp;
  int wt;                                          // write-through stores of emb_h / emb_xp (decode.hip store_result: the 5 .. 32-row chain)
};
bool launch_dec_sample(const SampleArgs& a, hipStream_t s);
#ifdef VX_DEV_PROBES
void dev_read_gemm_stamps(unsigned long long* out);     // gemm_f16x2.hip: shader-clock stamps around two k-steps, [256][16]
// development timeline of the decode kernels (decode.hip: vx_stamps[8][512][8], 100 MHz wall-clock ticks)
void dev_read_stamps(unsigned long long* out);
void dev_clear_stamps();
#endif
// best_of: copy row 0's prefilled K/V (L rows per head, every layer) to rows 1 .. beams-1
void launch_beam_kv_broadcast(float* kc, float* vc, long cache_layer, int layers, int Tmax, int L, int beams, hipStream_t s);
void launch_dec_force_token(const int* tok, int* cur_tok, int* cur_pos, int* ctx_len, int* n_gen, int* gen,
                            int gen_stride, const int* active, int batch, int* slot_meta, const int* slot_of,
                            hipStream_t s);

// ---- EnCodec SEANet decoder glue (encodec.hip) ----------------------------------------------------------------
void launch_im2col_seq(const float* x, int C, int k, int mode, int elu, const int* seq_off, const int* seq_len, int R,
                       float* out, int ldo, int batch, long max_rows, hipStream_t s);
void launch_lstm_cell(const float* part, int splitk, const float* xg, const int* seq_off, const int* seq_len, int t,
                      float* cstate, float* hp, float* y, const float* skip, int batch, hipStream_t s);
void launch_final_conv(const float* x, const float* w, const float* bias, const int* seq_off, const int* seq_len, int R,
                       float* audio, long audio_stride, int batch, long max_rows, hipStream_t s);

// encoder side (encodec.hip): first Conv1d(1,32,k7); ELU + causal reflect padding of a strided conv; one residual-VQ step
void launch_enc_first_conv(const float* wav, long L, const float* w, const float* bias, float* out, hipStream_t s);
void launch_enc_pad_elu(const float* x, long L, long Le, int C, int left, long rows, float* out, hipStream_t s);
void launch_rvq_select(float* resid, const float* scores, const float* e2, const float* codebook, long long* codes, int q,
                       long rows, hipStream_t s);
// ---- Vocos head (vocos.hip) --------------------------------------------------------------------
void launch_codebook_sum(const int* codes, const float* codebook, float* feat, int rows, hipStream_t s);
void launch_im2col7(const float* x, int C, const int* row_t, const int* row_len, float* out, int rows, hipStream_t s);
void launch_dwconv7(const float* x, const float* w, const float* bias, const int* row_t, const int* row_len,
                    float* out, int rows, int C, hipStream_t s);
void launch_istft_prep(const float* o, int ldo, float* reim, int ldr, int rows, hipStream_t s);
void launch_overlap_add(const float* frames, int ldf, const int* seq_off, const int* seq_len, const float* win2,
                        float* audio, long audio_stride, int batch, int max_T, hipStream_t s);

}  // namespace vx
