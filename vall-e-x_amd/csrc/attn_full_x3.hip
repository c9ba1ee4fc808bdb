// Full-sequence self-attention with fp32 accuracy on the bf16 matrix cores (bf16x3, see gemm_bf16x3.hip): the same
// flash-style transposed kernel as attn_full.hip -- same work decomposition, mask rule, online softmax and output
// layout -- with both contractions moved from v_mfma_f32_32x32x2_f32 (64 x 64 cycles per key tile) to six
// v_mfma_f32_32x32x16_bf16 per 32x32x16 block (48 x 32 cycles per key tile, 2.67x less matrix time).  The probes of
// tools/attn_bench.py showed the fp32 kernel bound by exactly that matrix time (staging 5 %, softmax 14 %).
//
// Replaces the same reference code as attn_full.hip: multi_head_attention_forward with the prefix-LM mask for the AR
// prefill (modules/activation.py:142-167, mask models/vallex.py:535-549) and without mask for the 7 NAR stages
// (modules/activation.py:566-585).
//
// Every fp32 operand x is split x = x1 + x2 + x3 (bf16 each, exact to 2^-27 |x|) and the six leading products are
// accumulated in the MFMA's fp32 accumulator, smallest terms first:
//   S^T = K . Q^T    A = K planes from LDS ([key][64 d] bf16, ds_read_b128), B = Q planes held in 48 VGPRs
//   O^T += V^T . P^T A = V^T planes from LDS ([d][32 keys] bf16, ds_read_b128), B = P split in registers
// K/V tiles are split by the staging threads on their way global -> LDS (each value once per workgroup).  V is written
// TRANSPOSED, two keys per ds_write_b32, with the key order inside a row permuted so that the 8 keys one lane contracts
// in k-step s (the C-layout rows 16 s + 4 hi + {0..3, 8..11} of S^T, which are the P registers 8 s .. 8 s + 7) are one
// 16-byte run: P never moves between lanes, exactly as in the fp32 kernel.
//
// Structure of the tile loop (details at the loop): three LDS buffers (tile t: V, tile t+1: K, tile t+2: being written)
// and ONE barrier per tile; phase 1 = 24 slots of [1 QK^T MFMA of tile t+1 | a slice of tile t's softmax], phase 2 =
// 24 slots of [1 PV MFMA of tile t | a slice of the three-way splits of P, of the staged K and of the staged V + their
// LDS stores]; the slots are instantiated with static_for and every slice is pinned in its slot (pin()).  The kernel is
// VALU-issue-bound (about 8 VALU instructions per MFMA), see DESIGN.md section 6.
#include <type_traits>

#include "vx_common.h"

namespace vx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int QB = 128, KT = 32;
constexpr int KP_LD = 144;                     // bytes per key row of a K plane: 128 data + 16 pad -> b128 reads conflict-free
constexpr int KP_SZ = KT * KP_LD;              // 4608
constexpr int VT_SZ = 64 * 64 + 16 * 16;       // V^T plane: row d at d*64 + (d/4)*16 bytes (reads conflict-free, writes 2-way)
constexpr float MASKED = -1e30f;

__device__ __forceinline__ int vt_row(int d) { return d * 64 + (d >> 2) * 16; }

__device__ __forceinline__ float exp_bf(float x) {           // see attn_full.hip
  const float L2E = 1.44269504088896341f, L2E_LO = 1.925963033500649e-08f, LN2 = 0.6931471805599453f;
  const float ph = x * L2E;
  const float pl = fmaf(x, L2E, -ph) + x * L2E_LO;
  const float e = __builtin_amdgcn_exp2f(ph);
  return fmaf(e, pl * LN2, e);
}

// compile-time loop: the slot bodies are instantiated per index (a 24-trip `#pragma unroll` over a body holding every
// slot's code exceeds LLVM's full-unroll budget and falls back to a runtime loop with indexed registers)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Keep a value where the source computes it: an empty volatile asm that "modifies" it is ordered against the other
// pins and the sched_barriers, so neither instruction selection nor the IR sink pass can move the producing arithmetic.
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(unsigned& x) { asm volatile("" : "+v"(x)); }

// (x, y) -> three packed bf16 pairs (x in the low half): w1 + w2 + w3 == (x, y) to 2^-27 relative; pinned.
// v_cvt_pk_bf16_f32 (RNE) once per term, the bf16 -> f32 widenings are a shift and a mask of the packed word.
__device__ __forceinline__ unsigned cvt_pk_bf16(f32x2 v) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ f32x2 widen_pk_bf16(unsigned w) {
  return f32x2{__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
}
__device__ __forceinline__ void pin2(f32x2& x) { asm volatile("" : "+v"(x)); }
// the same in two halves (terms 1-2 and the second residual | term 3), so a split can straddle two MFMA slots
__device__ __forceinline__ void split3_pair_a(float x, float y, unsigned& w1, unsigned& w2, f32x2& res) {
  const f32x2 v = {x, y};
  w1 = cvt_pk_bf16(v);
  const f32x2 r1 = v - widen_pk_bf16(w1);
  w2 = cvt_pk_bf16(r1);
  res = r1 - widen_pk_bf16(w2);
  pin(w1); pin(w2); pin2(res);
}
__device__ __forceinline__ void split3_pair_b(f32x2 res, unsigned& w3) {
  w3 = cvt_pk_bf16(res);
  pin(w3);
}
__device__ __forceinline__ void split3_pair(float x, float y, unsigned& w1, unsigned& w2, unsigned& w3) {
  const f32x2 v = {x, y};
  w1 = cvt_pk_bf16(v);
  const f32x2 r1 = v - widen_pk_bf16(w1);                      // exact
  w2 = cvt_pk_bf16(r1);
  w3 = cvt_pk_bf16(r1 - widen_pk_bf16(w2));                    // exact difference, rounded once
  pin(w1); pin(w2); pin(w3);
}

// exp_bf on a pair (packed f32 arithmetic around the two v_exp_f32)
__device__ __forceinline__ f32x2 exp_bf2(f32x2 x) {
  const float L2E = 1.44269504088896341f, L2E_LO = 1.925963033500649e-08f, LN2 = 0.6931471805599453f;
  const f32x2 ph = x * L2E;
  f32x2 pl = {fmaf(x[0], L2E, -ph[0]), fmaf(x[1], L2E, -ph[1])};
  pl = pl + x * L2E_LO;
  const f32x2 e = {__builtin_amdgcn_exp2f(ph[0]), __builtin_amdgcn_exp2f(ph[1])};
  const f32x2 q = pl * LN2;
  return f32x2{fmaf(e[0], q[0], e[0]), fmaf(e[1], q[1], e[1])};
}

__device__ __forceinline__ void split3(float v, __bf16& a1, __bf16& a2, __bf16& a3) {
  a1 = (__bf16)v;                                              // RNE
  const float r1 = v - (float)a1;                              // exact
  a2 = (__bf16)r1;
  a3 = (__bf16)(r1 - (float)a2);                               // exact difference, rounded once
}

// the six products, smallest first; first index = plane of the LDS (A) operand, second = plane of the register (B) one
#define VX_X3_MFMA6(acc, A, B)                                                     \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], B[0], acc, 0, 0, 0);         \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[2], acc, 0, 0, 0);         \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[1], acc, 0, 0, 0);         \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[0], acc, 0, 0, 0);         \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[1], acc, 0, 0, 0);         \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[0], acc, 0, 0, 0);

}  // namespace

// V = 0 product kernel.  Timing probes (tools/attn_bench.py, results meaningless): V = 1 no K/V staging after the first
// two tiles (V = 4: only the LDS stores dropped, V = 5: only the global loads); V = 2 no MFMAs; V = 3 no softmax arithmetic.
template <int V>
__global__ __launch_bounds__(256, 2) void attn_full_x3_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                             const int* __restrict__ seq_off,
                                                             const int* __restrict__ seq_len,
                                                             const int* __restrict__ prefix_len, int nqb,
                                                             unsigned short* __restrict__ planes, long plane_stride) {
  __shared__ __attribute__((aligned(16))) unsigned char Kp[3][3][KP_SZ];   // [buffer][plane]
  __shared__ __attribute__((aligned(16))) unsigned char Vt[3][3][VT_SZ];

  // XCD-aware work order: a unit's query blocks share one XCD's L2 (see attn_full.hip)
  const int id = blockIdx.x, per8 = 8 * nqb;
  const int grp = id / per8, rem = id - grp * per8;
  const int u = grp * 8 + (rem & 7);
  const int b = u / N_HEAD, h = u - b * N_HEAD, q0 = (rem >> 3) * QB;
  const int len = seq_len[b];
  if (q0 >= len) return;
  const long row0 = seq_off[b];
  const int S = prefix_len ? prefix_len[b] : 0x7fffffff;       // keys < S are visible to everyone
  const bool causal = prefix_len != nullptr;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int qi = q0 + wid * 32 + l31;                          // this lane's query (sequence-local index)
  const int qc = qi < len ? qi : len - 1;

  // Q planes: k-step s covers d = 16 s + 8 hi + 0..7; scaled by 1/sqrt(64) before the split (power of two: exact)
  bf16x8 qp[4][3];
  {
    const float* qptr = qkv + (row0 + qc) * (long)(3 * D_MODEL) + h * D_HEAD + hi * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(qptr + 16 * s);
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(qptr + 16 * s + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        __bf16 a1, a2, a3;
        split3((e < 4 ? t0[e & 3] : t1[e & 3]) * 0.125f, a1, a2, a3);
        qp[s][0][e] = a1; qp[s][1][e] = a2; qp[s][2][e] = a3;
      }
    }
  }

  const int q_last = (q0 + QB - 1 < len ? q0 + QB - 1 : len - 1);
  int kv_end = len;
  if (causal) kv_end = (q_last < S) ? S : (q_last + 1 < len ? q_last + 1 : len);

  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  float m_run = MASKED, l_run = 0.f;

  // staging maps.  K: float4 index f = tid + 256 i -> key f/16, dims 4 (f%16)..  V: thread -> key pair kp = tid/16
  // (keys 2 kp, 2 kp + 1), dims 4 (tid%16)..: the pair lands in one 32-bit word of each V^T row.
  const float* kbase = qkv + row0 * (long)(3 * D_MODEL) + D_MODEL + h * D_HEAD;
  const float* vbase = kbase + D_MODEL;
  const int c4 = (tid & 15) * 4, kp2 = (tid >> 4) * 2;
  // position of key 2 kp inside a V^T row: keys of (s, hi) = 16 s + 4 hi + {0,1,2,3,8,9,10,11} are positions 8 (2 s + hi) + j
  const int vpos = ((kp2 >> 4) * 2 + ((kp2 >> 2) & 1)) * 8 + (kp2 & 3) + 4 * ((kp2 >> 3) & 1);
  f32x4 rk[2], rv[2];
  unsigned kw[2][3][2];                                        // split K of the staged tile: [float4 i][plane][pair]
  unsigned vw[4][3];                                           // split V: [dim e][plane] = (key 2 kp, key 2 kp + 1)
  auto issue = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int kk = k0 + ((tid + 256 * i) >> 4);
      kk = kk < len ? kk : len - 1;
      rk[i] = *reinterpret_cast<const f32x4*>(kbase + kk * (long)(3 * D_MODEL) + c4);
      int kv = k0 + kp2 + i;
      kv = kv < len ? kv : len - 1;
      rv[i] = *reinterpret_cast<const f32x4*>(vbase + kv * (long)(3 * D_MODEL) + c4);
    }
  };
  // split slices, each pinned where it is written (see pin3)
  auto split_k = [&](int i, int pr) { split3_pair(rk[i][2 * pr], rk[i][2 * pr + 1], kw[i][0][pr], kw[i][1][pr], kw[i][2][pr]); };
  auto split_v = [&](int e) { split3_pair(rv[0][e], rv[1][e], vw[e][0], vw[e][1], vw[e][2]); };
  auto split_all = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) { split_k(i, 0); split_k(i, 1); }
#pragma unroll
    for (int e = 0; e < 4; ++e) split_v(e);
  };
  // LDS stores of the split tile (the split itself happens under the MFMAs)
  auto write_k = [&](int buf, int i) {
    const int off = ((tid + 256 * i) >> 4) * KP_LD + c4 * 2;
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(&Kp[buf][p][off]) = u32x2{kw[i][p][0], kw[i][p][1]};
  };
  auto write_v = [&](int buf, int e) {
    const int off = vt_row(c4 + e) + vpos * 2;
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<unsigned*>(&Vt[buf][p][off]) = vw[e][p];
  };
  auto stage_write = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) write_k(buf, i);
#pragma unroll
    for (int e = 0; e < 4; ++e) write_v(buf, e);
  };
  auto kfrag = [&](int buf, int s, bf16x8 (&kf)[3]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
      kf[p] = *reinterpret_cast<const bf16x8*>(&Kp[buf][p][l31 * KP_LD + (2 * s + hi) * 16]);
  };
  auto vfrag = [&](int buf, int s, bf16x8 (&v0)[3], bf16x8 (&v1)[3]) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      v0[p] = *reinterpret_cast<const bf16x8*>(&Vt[buf][p][vt_row(l31) + (2 * s + hi) * 16]);
      v1[p] = *reinterpret_cast<const bf16x8*>(&Vt[buf][p][vt_row(l31 + 32) + (2 * s + hi) * 16]);
    }
  };
  auto qk = [&](int buf) {
    f32x16 s16;
#pragma unroll
    for (int r = 0; r < 16; ++r) s16[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 kf[3];
      kfrag(buf, s, kf);
      VX_X3_MFMA6(s16, kf, qp[s]);
    }
    return s16;
  };

  // Visibility as ONE per-lane key limit: key kj is visible to query qi iff kj < lim, lim = min(len, qi >= S ? qi + 1 : S)
  // (text queries see the text, audio queries everything up to themselves; S = INT_MAX without a mask).
  const int lim = min(len, qi >= S ? qi + 1 : S);

  // Software pipeline as in attn_full.hip (QK^T of tile t+1 is issued before the softmax of tile t), plus: tile t+2
  // travels global -> registers during iteration t, is SPLIT under the PV MFMAs of iteration t and stored to LDS at its
  // end; with three LDS buffers the stores go out under those MFMAs too and one barrier per tile is enough.
  //
  // Issue order inside a wave is what makes the matrix pipe and the VALU overlap: the wave issues in order, an MFMA
  // occupies the pipe for 32 cycles, and an MFMA waiting for the pipe blocks everything behind it.  Each phase is
  // therefore 24 slots of [1 MFMA | <= ~10 VALU/LDS instructions]; LDS fragments are requested one k-step (6 slots)
  // before their MFMAs.  The slices are pure arithmetic, which LLVM places wherever it likes (it sank them below the
  // MFMAs, or into a later conditional block): every slice ends in pin()s -- empty volatile asm that reads and writes
  // the values just produced -- and every slot in sched_barrier(0).
  const int ntiles = (kv_end + KT - 1) / KT;
  issue(0);
  split_all();
  stage_write(0);
  if (1 < ntiles) { issue(KT); split_all(); stage_write(1); }
  __syncthreads();
  if (2 < ntiles) issue(2 * KT);
  f32x16 s_cur = qk(0);
  bf16x8 kfa[3], kfb[3];                                       // K fragments of even / odd k-steps
  kfrag(1, 0, kfa);                                            // (tile 1; a stale buffer if there is none: discarded)
  // wave-uniform "every key of tile t is visible to every query of this wave" test: (t + 1) * KT <= min over lanes of lim
  int lim_min = lim;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) lim_min = min(lim_min, __shfl_xor(lim_min, m, 64));
  lim_min = __builtin_amdgcn_readfirstlane(lim_min);

  // One tile iteration.  The loop is unrolled by two so that the score registers alternate roles (s_cur: scores of tile
  // t, consumed; sA: scores of tile t+1, produced) without a 16-register copy, and the LDS buffer index is a constant.
  auto tile = [&](int t, int cur, int nxt, int wr, f32x16& s_cur, f32x16& sA) {
    const int lim_t = lim - t * KT - 4 * hi;                   // register r holds key offset (r&3) + 8 (r>>2) of this lane
    const bool need_mask = (t + 1) * KT > lim_min;             // scalar
    bf16x8 v0[3], v1[3], w0[3], w1[3];                         // V^T fragments of k-step 0 / 1, both halves of d
    unsigned pw[2][3][4];                                      // P planes of k-step s as packed pairs: [s][plane][pair]
    f32x2 res;                                                 // second residual of the split in flight
    float m_new = m_run, alpha = 1.f, psum = 0.f, m_tile = MASKED;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // a split is cut in two halves (terms 1-2 | term 3) that go behind consecutive MFMAs
    auto split_p_a = [&](int s, int pr) { split3_pair_a(s_cur[8 * s + 2 * pr], s_cur[8 * s + 2 * pr + 1], pw[s][0][pr], pw[s][1][pr], res); };
    auto split_p_b = [&](int s, int pr) { split3_pair_b(res, pw[s][2][pr]); };
    // ---- phase 1: S^T of tile t+1 (24 MFMAs, one chain) with the softmax of tile t threaded through it
    //   slots 0-3 mask | 4-5 running max, alpha | 6,8,..,20 exp pairs | 7,9,11,13 rescale O | 15-22 split P (k-step 0)
    static_for<0, 24>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int s = i / 6, j = i - 6 * s;
      constexpr int pa = j == 0 ? 2 : (j == 2 || j == 3) ? 1 : 0;     // plane of K
      constexpr int pb = j == 1 ? 2 : (j == 2 || j == 4) ? 1 : 0;     // plane of Q
      if constexpr (j == 0 && s < 3) kfrag(nxt, s + 1, (s & 1) ? kfa : kfb);   // next k-step's fragments
      if constexpr (i == 18) vfrag(cur, 0, v0, v1);
      if constexpr (V == 2) {
        sA[0] = (i ? sA[0] : 0.f) + (float)((s & 1) ? kfb : kfa)[pa][0] * (float)qp[s][pb][0];
      } else {
        sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16((s & 1) ? kfb[pa] : kfa[pa], qp[s][pb], i ? sA : zero, 0, 0, 0);
      }
      if constexpr (V == 3) {
        if constexpr (i == 21) {
#pragma unroll
          for (int pr = 0; pr < 4; ++pr) { split_p_a(0, pr); split_p_b(0, pr); }
        }
      } else if constexpr (i < 4) {                            // visibility, 4 keys per slot (interior tiles skip it)
        if (need_mask) {
          float mv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) mv[r] = ((r + 8 * i) < lim_t) ? s_cur[4 * i + r] : MASKED;   // key offset of reg 4i+r
#pragma unroll
          for (int r = 0; r < 4; ++r) { pin(mv[r]); s_cur[4 * i + r] = mv[r]; }
        }
      } else if constexpr (i == 4) {                           // tile max of this lane's 16 keys
#pragma unroll
        for (int r = 0; r < 16; ++r) m_tile = fmaxf(m_tile, s_cur[r]);
        pin(m_tile);
      } else if constexpr (i == 5) {                           // the other 16 keys live in lane ^ 32
        // the other 16 keys of the query live in lane ^ 32.  Inline asm on two distinct registers: the compiler folds the two
        // results of __builtin_amdgcn_permlane32_swap into one (ROCm 7.2), which silently left the partner's half out of the
        // maximum -- still a common, valid stabiliser for the pair, but p could exceed 1 (and overflow an fp16 head)
        unsigned ua = __builtin_bit_cast(unsigned, m_tile), ub = ua;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ua), "+v"(ub));
        m_tile = fmaxf(__builtin_bit_cast(float, ua), __builtin_bit_cast(float, ub));
        m_new = fmaxf(m_run, m_tile);                          // finite: key 0 is visible to every query
        alpha = exp_bf(m_run - m_new);
        pin(m_new);
        pin(alpha);
      } else if constexpr (i < 22 && !(i & 1)) {               // slots 6, 8, .., 20: one pair of exps
        constexpr int r = i - 6;
        f32x2 ev = exp_bf2(f32x2{s_cur[r] - m_new, s_cur[r + 1] - m_new});
        pin2(ev);
        s_cur[r] = ev[0];
        s_cur[r + 1] = ev[1];
        psum += ev[0] + ev[1];
        pin(psum);
        if constexpr (i >= 16) split_p_b(0, (i - 16) >> 1);    // second half of the split begun in slot i - 1
      } else if constexpr (i < 14) {                           // slots 7, 9, 11, 13: rescale a quarter of O
        constexpr int q = (i - 7) >> 1;
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
          f32x2 v = f32x2{o[q >> 1][8 * (q & 1) + r], o[q >> 1][8 * (q & 1) + r + 1]} * alpha;
          pin2(v);
          o[q >> 1][8 * (q & 1) + r] = v[0];
          o[q >> 1][8 * (q & 1) + r + 1] = v[1];
        }
      } else if constexpr (i < 22) {                           // slots 15, 17, 19, 21: first half of a P split (k-step 0)
        split_p_a(0, (i - 15) >> 1);
      } else if constexpr (i == 22) {
        split_p_b(0, 3);
        l_run = l_run * alpha + psum;
        m_run = m_new;
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // ---- phase 2: O^T += V^T . P^T, 24 MFMAs in two chains (o[0], o[1]); a split pair per two slots:
    //   slots 0-7 P (k-step 1) | 8-15 the staged K tile | 16-23 the staged V tile (registers hold tile t+2; past the
    //   last tile they are stale and the result is not stored)
    static_for<0, 24>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int s = i / 12, j = (i - 12 * s) >> 1, half = i & 1;
      constexpr int pa = j == 0 ? 2 : (j == 2 || j == 3) ? 1 : 0;     // plane of V^T
      constexpr int pb = j == 1 ? 2 : (j == 2 || j == 4) ? 1 : 0;     // plane of P
      if constexpr (i == 1) vfrag(cur, 1, w0, w1);
      const bf16x8 pf = __builtin_bit_cast(bf16x8, u32x4{pw[s][pb][0], pw[s][pb][1], pw[s][pb][2], pw[s][pb][3]});
      const bf16x8 vf = s ? (half ? w1[pa] : w0[pa]) : (half ? v1[pa] : v0[pa]);
      if constexpr (V == 2) o[half][0] += (float)vf[0] * (float)pf[0];
      else o[half] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[half], 0, 0, 0);
      constexpr int q = i >> 1;
      if constexpr (q < 4) {
        if constexpr (!half) split_p_a(1, q); else split_p_b(1, q);
      } else if constexpr (q < 8) {
        constexpr int ki = (q - 4) >> 1, pr = (q - 4) & 1;
        if constexpr (!half) split3_pair_a(rk[ki][2 * pr], rk[ki][2 * pr + 1], kw[ki][0][pr], kw[ki][1][pr], res);
        else split3_pair_b(res, kw[ki][2][pr]);
        if constexpr (V != 1 && V != 4 && half && pr == 1) write_k(wr, ki);
      } else {
        constexpr int e = q - 8;
        if constexpr (!half) split3_pair_a(rv[0][e], rv[1][e], vw[e][0], vw[e][1], res);
        else split3_pair_b(res, vw[e][2]);
        if constexpr (V != 1 && V != 4 && half) write_v(wr, e);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // One barrier per tile: buffer `wr` (tile t+2; past the last tile a stale copy nobody reads) was last read in
    // iteration t-1, is written above, and is first read below / in iteration t+1.
    __syncthreads();
    kfrag(wr, 0, kfa);                                         // first K fragments of the next iteration's tile (t+2)
    if (V != 1 && V != 5 && t + 3 < ntiles) issue((t + 3) * KT);
  };

  f32x16 s_odd;
  int b0 = 0, b1 = 1, b2 = 2;                                  // buffers of tiles t, t+1, t+2
  for (int t = 0; t < ntiles; t += 2) {
    tile(t, b0, b1, b2, s_cur, s_odd);
    if (t + 1 < ntiles) tile(t + 1, b1, b2, b0, s_odd, s_cur);
    const int r0 = b0;                                         // advance by two tiles
    b0 = b2; b2 = b1; b1 = r0;
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (qi < len) {
    const float inv = 1.0f / l_tot;
    if (!planes) {
      float* op = out + (row0 + qi) * (long)D_MODEL + h * D_HEAD + 4 * hi;
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f32x4 t;
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = o[half][g4 * 4 + e] * inv;
          *reinterpret_cast<f32x4*>(op + half * 32 + g4 * 8) = t;   // d = 32*half + 8*g4 + 4*hi + e
        }
    } else {
      // the attention output only feeds out_proj: write it as that GEMM's f16x2 A planes (tile-major, K = 1024; the 32 dims of
      // `half` are one K tile, index 2 h + half).  Lanes l and l ^ 32 hold complementary 4-dim halves of every 8-dim group and
      // trade them, so each lane stores 16 contiguous bytes per plane (same scheme as the GEMM's plane epilogue).
      typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
      const long row = row0 + qi;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        unsigned hw[4][2], tw[4][2];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            h2_t h2, t2;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              _Float16 hk, tk;
              bool bad = false;                     // |attention output| <= max |v|: the range is checked where V was produced
              h2_split(o[half][g4 * 4 + 2 * pr + k] * inv, H2_ACT_SCALE, hk, tk, bad);
              h2[k] = hk;
              t2[k] = tk;
            }
            hw[g4][pr] = __builtin_bit_cast(unsigned, h2);
            tw[g4][pr] = __builtin_bit_cast(unsigned, t2);
          }
        unsigned short* blk = planes + (((row >> 8) * (D_MODEL / 32) + (2 * h + half)) * 256 + (row & 255)) * 32;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int keep = 2 * j + hi, give = 2 * j + 1 - hi;
          unsigned rh[2], rt[2];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            rh[k] = (unsigned)__shfl_xor((int)hw[give][k], 32, 64);
            rt[k] = (unsigned)__shfl_xor((int)tw[give][k], 32, 64);
          }
          const u32x4 oh = hi ? u32x4{rh[0], rh[1], hw[keep][0], hw[keep][1]} : u32x4{hw[keep][0], hw[keep][1], rh[0], rh[1]};
          const u32x4 ot = hi ? u32x4{rt[0], rt[1], tw[keep][0], tw[keep][1]} : u32x4{tw[keep][0], tw[keep][1], rt[0], rt[1]};
          *reinterpret_cast<u32x4*>(blk + 16 * j + 8 * hi) = oh;
          *reinterpret_cast<u32x4*>(blk + 16 * j + 8 * hi + plane_stride) = ot;
        }
      }
    }
  }
}

void launch_attn_full_x3(const float* qkv, float* out, const int* seq_off, const int* seq_len, const int* prefix_len,
                         int batch, int max_len, int variant, hipStream_t s, unsigned short* planes, long plane_stride) {
  if (batch <= 0 || max_len <= 0) return;
  const int nqb = (max_len + QB - 1) / QB;                   // batch * N_HEAD is a multiple of 8 (16 heads)
  const dim3 grid(nqb * N_HEAD * batch), block(256);
#ifdef VX_DEV_PROBES
  if (variant == 1) hipLaunchKernelGGL(attn_full_x3_kernel<1>, grid, block, 0, s, qkv, out, seq_off, seq_len, prefix_len, nqb, planes, plane_stride);
  else if (variant == 2) hipLaunchKernelGGL(attn_full_x3_kernel<2>, grid, block, 0, s, qkv, out, seq_off, seq_len, prefix_len, nqb, planes, plane_stride);
  else if (variant == 3) hipLaunchKernelGGL(attn_full_x3_kernel<3>, grid, block, 0, s, qkv, out, seq_off, seq_len, prefix_len, nqb, planes, plane_stride);
  else if (variant == 4) hipLaunchKernelGGL(attn_full_x3_kernel<4>, grid, block, 0, s, qkv, out, seq_off, seq_len, prefix_len, nqb, planes, plane_stride);
  else if (variant == 5) hipLaunchKernelGGL(attn_full_x3_kernel<5>, grid, block, 0, s, qkv, out, seq_off, seq_len, prefix_len, nqb, planes, plane_stride);
  else
#endif
  hipLaunchKernelGGL(attn_full_x3_kernel<0>, grid, block, 0, s, qkv, out, seq_off, seq_len, prefix_len, nqb, planes, plane_stride);
}

}  // namespace vx
