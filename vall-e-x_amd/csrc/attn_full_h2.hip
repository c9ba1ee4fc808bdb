// Full-sequence self-attention with fp32-class accuracy on the f16 matrix cores ("f16x2", the arithmetic of gemm_f16x2.hip):
// the same flash-style transposed kernel as attn_full_x3.hip -- same work decomposition, mask rule, online softmax, LDS image and
// output layout -- with THREE v_mfma_f32_32x32x16_f16 per 32x32x16 block instead of six bf16 ones (24 instead of 48 per key
// tile) and two-way instead of three-way operand splits (the kernel is VALU-bound, so both halves of that matter).
//
// Replaces the same reference code: multi_head_attention_forward with the prefix-LM mask for the AR prefill
// (modules/activation.py:142-167, mask models/vallex.py:535-549) and without mask for the 7 NAR stages
// (modules/activation.py:566-585).
//
// Every fp32 operand x is scaled by a power of two and split X = x * 2^s = h + t (fp16 head, fp16 tail at the same scale, 22
// significant bits) and the three leading products go into ONE fp32 accumulator, small terms first (vx_common.h):
//   S'^T = K . Q^T    K * 2^5 from LDS ([key][64 d] fp16, ds_read_b128), Q * 0.125 * 2^5 held in 32 VGPRs      => S' = 2^10 S
//   O'^T += V^T . P^T V * 2^5 from LDS ([d][32 keys] fp16), P * 2^14 split in registers                      => O' = 2^19 sum p v
// The softmax runs on the scaled scores: exp((S' - M') 2^-10) with the 2^-10 folded into the exp constants (exact), and the
// 2^14 of P comes from lowering the subtracted maximum by 14 ln2 (a common factor of a tile's p and of the running sum: it
// cancels in O / l).  P <= 2^14 and |q|/8, |k|, |v| < 2047 keep every head inside fp16; a head that does not fit (inf) turns the
// output into NaN, which the epilogue reports through the range flag -- the engine then re-runs the phase on the exact-fp32 kernels.
// K/V tiles are split by the staging threads on their way global -> LDS (each value once per workgroup).  V is written
// TRANSPOSED, two keys per ds_write_b32, with the key order inside a row permuted so that the 8 keys one lane contracts
// in k-step s (the C-layout rows 16 s + 4 hi + {0..3, 8..11} of S^T, which are the P registers 8 s .. 8 s + 7) are one
// 16-byte run: P never moves between lanes.
//
// Structure of the tile loop: three LDS buffers (tile t: V, tile t+1: K, tile t+2: being written) and ONE barrier per tile;
// phase 1 = 12 slots of [1 QK^T MFMA of tile t+1 | a slice of tile t's softmax], phase 2 = 12 slots of [1 PV MFMA of tile t | a
// slice of the splits of P, of the staged K and of the staged V + their LDS stores]; every slice is pinned in its slot.
#include <stdlib.h>

#include <type_traits>

#include "vx_common.h"

namespace vx {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int QB = 128, KT = 32;
constexpr int KP_LD = 144;                     // bytes per key row of a K plane: 128 data + 16 pad -> b128 reads conflict-free
constexpr int KP_SZ = KT * KP_LD;              // 4608
constexpr int VT_SZ = 64 * 64 + 16 * 16;       // V^T plane: row d at d*64 + (d/4)*16 bytes (reads conflict-free, writes 2-way)
constexpr float MASKED = -1e30f;
constexpr float QK_SCALE = 32.0f;              // K and Q/8 are split at 2^5: scores come out of the MFMAs as S' = 2^10 S
constexpr float S_INV = 1.0f / 1024.0f;
constexpr float V_SCALE = 32.0f, V_INV = 1.0f / 32.0f;
constexpr float P_SHIFT = 14.0f * 0.6931471805599453f * 1024.0f;    // 14 ln2 in units of S': p comes out as p * 2^14

__device__ __forceinline__ int vt_row(int d) { return d * 64 + (d >> 2) * 16; }

// exp(x * 2^-10) for scaled score differences x: the compensated exp2 of attn_full.hip with the 2^-10 folded into its constants
__device__ __forceinline__ float exp_s(float x) {
  const float L2E = 1.44269504088896341f * S_INV, L2E_LO = 1.925963033500649e-08f * S_INV, LN2 = 0.6931471805599453f;
  const float ph = x * L2E;
  const float pl = fmaf(x, L2E, -ph) + x * L2E_LO;
  const float e = __builtin_amdgcn_exp2f(ph);
  return fmaf(e, pl * LN2, e);
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Keep a value where the source computes it (see attn_full_x3.hip)
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(unsigned& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin2(f32x2& x) { asm volatile("" : "+v"(x)); }

// (x, y) (already scaled) -> packed fp16 heads and tails (x in the low half): h + t == (x, y) to 2^-22 relative; pinned
__device__ __forceinline__ unsigned cvt_pk_f16(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2)); }
__device__ __forceinline__ f32x2 widen_pk_f16(unsigned w) { return __builtin_convertvector(__builtin_bit_cast(f16x2, w), f32x2); }
__device__ __forceinline__ void split2_pair(f32x2 v, unsigned& wh, unsigned& wt) {
  wh = cvt_pk_f16(v);                                          // RNE
  wt = cvt_pk_f16(v - widen_pk_f16(wh));                       // exact difference, rounded once
  pin(wh); pin(wt);
}

// exp_s on a pair (packed f32 arithmetic around the two v_exp_f32)
__device__ __forceinline__ f32x2 exp_s2(f32x2 x) {
  const float L2E = 1.44269504088896341f * S_INV, L2E_LO = 1.925963033500649e-08f * S_INV, LN2 = 0.6931471805599453f;
  const f32x2 ph = x * L2E;
  f32x2 pl = {fmaf(x[0], L2E, -ph[0]), fmaf(x[1], L2E, -ph[1])};
  pl = pl + x * L2E_LO;
  const f32x2 e = {__builtin_amdgcn_exp2f(ph[0]), __builtin_amdgcn_exp2f(ph[1])};
  const f32x2 q = pl * LN2;
  return f32x2{fmaf(e[0], q[0], e[0]), fmaf(e[1], q[1], e[1])};
}

}  // namespace

// PRIO: wave-priority policy (guide T5; two 4-wave workgroups share a CU = two independent waves per SIMD whose MFMAs compete with
// the partner's softmax / split VALU):  0 none;  1 s_setprio(1) around every MFMA of the tile loop;  2 static: the workgroups
// with an odd block id run at priority 1 throughout;  3 priority 1 through phase 2 (the PV products + operand splits) only.
// The product instantiates ONE value (VX_ATTN_PRIO below); tools builds (-DVX_DEV_PROBES) instantiate all four for A/B
// (vx_bench_attn variants 20..23).  Results do not depend on it.
#ifndef VX_ATTN_PRIO
#define VX_ATTN_PRIO 0
#endif
template <int PRIO>
__global__ __launch_bounds__(256, 2) void attn_full_h2_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                             const int* __restrict__ seq_off,
                                                             const int* __restrict__ seq_len,
                                                             const int* __restrict__ prefix_len, int nqb,
                                                             unsigned short* __restrict__ planes, long plane_stride,
                                                             int* __restrict__ range_flag,
                                                             const int* __restrict__ q_first,
                                                             const int* __restrict__ c_off) {
  // q_first / c_off (planes mode only, optional): ROW TRIMMING for a layer whose output is only needed for the rows
  // [q_first[b], len) of every sequence (the last decoder layer of a NAR stage: only generated frames reach a predict layer,
  // models/vallex.py:672-679).  Query blocks that lie entirely before q_first[b] are skipped, and the output planes are written
  // COMPACTED: sequence-local query qi >= q_first[b] lands in plane row c_off[b] + qi - q_first[b].  Keys / values are still
  // all of the sequence.  Every stored value is computed exactly as without trimming.
  __shared__ __attribute__((aligned(16))) unsigned char Kp[3][2][KP_SZ];   // [buffer][plane: head, tail]
  __shared__ __attribute__((aligned(16))) unsigned char Vt[3][2][VT_SZ];

  // XCD-aware work order: a unit's query blocks share one XCD's L2 (see attn_full.hip)
  const int id = blockIdx.x, per8 = 8 * nqb;
  const int grp = id / per8, rem = id - grp * per8;
  const int u = grp * 8 + (rem & 7);
  const int b = u / N_HEAD, h = u - b * N_HEAD, q0 = (rem >> 3) * QB;
  const int len = seq_len[b];
  if (q0 >= len) return;
  const int qf = q_first ? q_first[b] : 0;
  if (q0 + QB <= qf) return;                                    // no query of this block is needed
  if constexpr (PRIO == 2) {                                    // s_setprio ignores EXEC: the guard must be scalar (guide T5)
    if (__builtin_amdgcn_readfirstlane((int)blockIdx.x) & 1) __builtin_amdgcn_s_setprio(1);
  }
  const long row0 = seq_off[b];
  const int S = prefix_len ? prefix_len[b] : 0x7fffffff;       // keys < S are visible to everyone
  const bool causal = prefix_len != nullptr;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int qi = q0 + wid * 32 + l31;                          // this lane's query (sequence-local index)
  const int qc = qi < len ? qi : len - 1;

  // Q planes: k-step s covers d = 16 s + 8 hi + 0..7; scaled by 1/sqrt(64) * 2^5 = 4 before the split (power of two: exact)
  f16x8 qp[4][2];
  {
    const float* qptr = qkv + (row0 + qc) * (long)(3 * D_MODEL) + h * D_HEAD + hi * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(qptr + 16 * s);
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(qptr + 16 * s + 4);
      unsigned wh[4], wt[4];
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const f32x2 v = f32x2{pr < 2 ? t0[2 * pr] : t1[2 * pr - 4], pr < 2 ? t0[2 * pr + 1] : t1[2 * pr - 3]} * (0.125f * QK_SCALE);
        split2_pair(v, wh[pr], wt[pr]);
      }
      qp[s][0] = __builtin_bit_cast(f16x8, u32x4{wh[0], wh[1], wh[2], wh[3]});
      qp[s][1] = __builtin_bit_cast(f16x8, u32x4{wt[0], wt[1], wt[2], wt[3]});
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
  unsigned kw[2][2][2];                                        // split K of the staged tile: [float4 i][plane][pair]
  unsigned vw[4][2];                                           // split V: [dim e][plane] = (key 2 kp, key 2 kp + 1)
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
  auto split_k = [&](int i, int pr) { split2_pair(f32x2{rk[i][2 * pr], rk[i][2 * pr + 1]} * QK_SCALE, kw[i][0][pr], kw[i][1][pr]); };
  auto split_v = [&](int e) { split2_pair(f32x2{rv[0][e], rv[1][e]} * V_SCALE, vw[e][0], vw[e][1]); };
  auto write_k = [&](int buf, int i) {
    const int off = ((tid + 256 * i) >> 4) * KP_LD + c4 * 2;
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x2*>(&Kp[buf][p][off]) = u32x2{kw[i][p][0], kw[i][p][1]};
  };
  auto write_v = [&](int buf, int e) {
    const int off = vt_row(c4 + e) + vpos * 2;
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<unsigned*>(&Vt[buf][p][off]) = vw[e][p];
  };
  auto stage_all = [&](int buf) {                              // prologue: split + store a whole tile
#pragma unroll
    for (int i = 0; i < 2; ++i) { split_k(i, 0); split_k(i, 1); write_k(buf, i); }
#pragma unroll
    for (int e = 0; e < 4; ++e) { split_v(e); write_v(buf, e); }
  };
  auto kfrag = [&](int buf, int s, f16x8 (&kf)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) kf[p] = *reinterpret_cast<const f16x8*>(&Kp[buf][p][l31 * KP_LD + (2 * s + hi) * 16]);
  };
  auto vfrag = [&](int buf, int s, f16x8 (&v0)[2], f16x8 (&v1)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      v0[p] = *reinterpret_cast<const f16x8*>(&Vt[buf][p][vt_row(l31) + (2 * s + hi) * 16]);
      v1[p] = *reinterpret_cast<const f16x8*>(&Vt[buf][p][vt_row(l31 + 32) + (2 * s + hi) * 16]);
    }
  };
  // the three products of a block, small terms first: (tail, head), (head, tail), (head, head); first index = LDS operand
  auto qk = [&](int buf) {
    f32x16 s16;
#pragma unroll
    for (int r = 0; r < 16; ++r) s16[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f16x8 kf[2];
      kfrag(buf, s, kf);
      s16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], qp[s][0], s16, 0, 0, 0);
      s16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qp[s][1], s16, 0, 0, 0);
      s16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qp[s][0], s16, 0, 0, 0);
    }
    return s16;
  };

  // Visibility as ONE per-lane key limit: key kj is visible to query qi iff kj < lim, lim = min(len, qi >= S ? qi + 1 : S)
  const int lim = min(len, qi >= S ? qi + 1 : S);

  const int ntiles = (kv_end + KT - 1) / KT;
  issue(0);
  stage_all(0);
  if (1 < ntiles) { issue(KT); stage_all(1); }
  __syncthreads();
  if (2 < ntiles) issue(2 * KT);
  f32x16 s_cur = qk(0);
  f16x8 kfa[2], kfb[2];                                        // K fragments of even / odd k-steps
  kfrag(1, 0, kfa);                                            // (tile 1; a stale buffer if there is none: discarded)
  int lim_min = lim;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) lim_min = min(lim_min, __shfl_xor(lim_min, m, 64));
  lim_min = __builtin_amdgcn_readfirstlane(lim_min);

  // One tile iteration (unrolled by two by the caller so that the score registers alternate roles and the LDS buffer
  // indices are constants): s_cur = scaled scores of tile t (consumed), sA = scores of tile t+1 (produced).
  auto tile = [&](int t, int cur, int nxt, int wr, f32x16& s_cur, f32x16& sA) {
    const int lim_t = lim - t * KT - 4 * hi;                   // register r holds key offset (r&3) + 8 (r>>2) of this lane
    const bool need_mask = (t + 1) * KT > lim_min;             // scalar
    f16x8 v0[2], v1[2], w0[2], w1[2];                          // V^T fragments of k-step 0 / 1, both halves of d
    unsigned pw[2][2][4];                                      // P planes of k-step s as packed pairs: [s][plane][pair]
    float m_new = m_run, m_sub = m_run, alpha = 1.f, psum = 0.f, m_tile = MASKED;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto split_p = [&](int s, int pr) { split2_pair(f32x2{s_cur[8 * s + 2 * pr], s_cur[8 * s + 2 * pr + 1]}, pw[s][0][pr], pw[s][1][pr]); };
    // ---- phase 1: S'^T of tile t+1 (12 MFMAs, one chain) with the softmax of tile t threaded through it
    //   slots 0-1 mask | 2-3 running max, alpha | 4-11 one exp pair each; 4-7 also rescale a quarter of O; 8-11 split P (k-step 0)
    static_for<0, 12>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int s = i / 3, j = i - 3 * s;
      constexpr int pa = j == 0 ? 1 : 0;                       // plane of K: tail first
      constexpr int pb = j == 1 ? 1 : 0;                       // plane of Q
      if constexpr (j == 0 && s < 3) kfrag(nxt, s + 1, (s & 1) ? kfa : kfb);   // next k-step's fragments
      if constexpr (i == 9) vfrag(cur, 0, v0, v1);
      if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);
      sA = __builtin_amdgcn_mfma_f32_32x32x16_f16((s & 1) ? kfb[pa] : kfa[pa], qp[s][pb], i ? sA : zero, 0, 0, 0);
      if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
      if constexpr (i < 2) {                                   // visibility, 8 keys per slot (interior tiles skip it)
        if (need_mask) {
          float mv[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) mv[r] = ((r & 3) + 8 * ((8 * i + r) >> 2) < lim_t) ? s_cur[8 * i + r] : MASKED;
#pragma unroll
          for (int r = 0; r < 8; ++r) { pin(mv[r]); s_cur[8 * i + r] = mv[r]; }
        }
      } else if constexpr (i == 2) {                           // tile max of this lane's 16 keys
#pragma unroll
        for (int r = 0; r < 16; ++r) m_tile = fmaxf(m_tile, s_cur[r]);
        pin(m_tile);
      } else if constexpr (i == 3) {                           // the other 16 keys live in lane ^ 32
        // the other 16 keys of the query live in lane ^ 32.  Inline asm on two distinct registers: the compiler folds the two
        // results of __builtin_amdgcn_permlane32_swap into one (ROCm 7.2), which silently left the partner's half out of the
        // maximum -- still a common, valid stabiliser for the pair, but p could exceed 1 (and overflow an fp16 head)
        unsigned ua = __builtin_bit_cast(unsigned, m_tile), ub = ua;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ua), "+v"(ub));
        m_tile = fmaxf(__builtin_bit_cast(float, ua), __builtin_bit_cast(float, ub));
        m_new = fmaxf(m_run, m_tile);                          // finite: key 0 is visible to every query
        alpha = exp_s(m_run - m_new);
        m_sub = m_new - P_SHIFT;                               // p comes out scaled by 2^14
        pin(m_new);
        pin(alpha);
        pin(m_sub);
      } else {                                                 // slots 4 .. 11: one pair of exps each
        constexpr int r = 2 * (i - 4);
        f32x2 ev = exp_s2(f32x2{s_cur[r] - m_sub, s_cur[r + 1] - m_sub});
        pin2(ev);
        s_cur[r] = ev[0];
        s_cur[r + 1] = ev[1];
        psum += ev[0] + ev[1];
        pin(psum);
        if constexpr (i < 8) {                                 // rescale a quarter of O
          constexpr int q = i - 4;
#pragma unroll
          for (int rr = 0; rr < 8; rr += 2) {
            f32x2 v = f32x2{o[q >> 1][8 * (q & 1) + rr], o[q >> 1][8 * (q & 1) + rr + 1]} * alpha;
            pin2(v);
            o[q >> 1][8 * (q & 1) + rr] = v[0];
            o[q >> 1][8 * (q & 1) + rr + 1] = v[1];
          }
        } else {
          split_p(0, i - 8);                                   // P of k-step 0: registers 0..7, exps of slots 4..7
        }
        if constexpr (i == 11) {
          l_run = l_run * alpha + psum;
          m_run = m_new;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // ---- phase 2: O'^T += V^T . P^T, 12 MFMAs in two chains (o[0], o[1]);
    //   slots 0-3 split P (k-step 1) | 4-7 the staged K tile | 8-11 the staged V tile (registers hold tile t+2; past the
    //   last tile they are stale and the result is not read)
    static_for<0, 12>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int s = i / 6, j = (i - 6 * s) >> 1, half = i & 1;
      constexpr int pa = j == 0 ? 1 : 0;                       // plane of V^T: tail first
      constexpr int pb = j == 1 ? 1 : 0;                       // plane of P
      if constexpr (i == 1) vfrag(cur, 1, w0, w1);
      const f16x8 pf = __builtin_bit_cast(f16x8, u32x4{pw[s][pb][0], pw[s][pb][1], pw[s][pb][2], pw[s][pb][3]});
      const f16x8 vf = s ? (half ? w1[pa] : w0[pa]) : (half ? v1[pa] : v0[pa]);
      if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);
      if constexpr (PRIO == 3 && i == 0) __builtin_amdgcn_s_setprio(1);
      o[half] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[half], 0, 0, 0);
      if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
      if constexpr (i < 4) {
        split_p(1, i);
      } else if constexpr (i < 8) {
        constexpr int ki = (i - 4) >> 1, pr = (i - 4) & 1;
        split_k(ki, pr);
        if constexpr (pr == 1) write_k(wr, ki);
      } else {
        constexpr int e = i - 8;
        split_v(e);
        write_v(wr, e);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (PRIO == 3) __builtin_amdgcn_s_setprio(0);
    // One barrier per tile: buffer `wr` (tile t+2; past the last tile a stale copy nobody reads) was last read in
    // iteration t-1, is written above, and is first read below / in iteration t+1.
    __syncthreads();
    kfrag(wr, 0, kfa);                                         // first K fragments of the next iteration's tile (t+2)
    if (t + 3 < ntiles) issue((t + 3) * KT);
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
  if (qi < len && qi >= qf) {
    const float inv = V_INV / l_tot;                           // O' / l' = 2^5 O / l
    bool nonfinite = false;                                    // an operand head that did not fit fp16 (inf) ends up here as NaN
    if (!planes) {
      float* op = out + (row0 + qi) * (long)D_MODEL + h * D_HEAD + 4 * hi;
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f32x4 t;
#pragma unroll
          for (int e = 0; e < 4; ++e) { t[e] = o[half][g4 * 4 + e] * inv; nonfinite |= !(fabsf(t[e]) < 3.0e38f); }
          *reinterpret_cast<f32x4*>(op + half * 32 + g4 * 8) = t;   // d = 32*half + 8*g4 + 4*hi + e
        }
    } else {
      // the attention output only feeds out_proj: write it as that GEMM's f16x2 A planes (tile-major, K = 1024; the 32 dims of
      // `half` are one K tile, index 2 h + half).  Lanes l and l ^ 32 hold complementary 4-dim halves of every 8-dim group and
      // trade them, so each lane stores 16 contiguous bytes per plane (same scheme as the GEMM's plane epilogue).
      const long row = c_off ? (long)c_off[b] + (qi - qf) : row0 + qi;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        unsigned hw[4][2], tw[4][2];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            f16x2 h2, t2;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              _Float16 hk, tk;
              h2_split(o[half][g4 * 4 + 2 * pr + k] * inv, H2_ACT_SCALE, hk, tk, nonfinite);
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
    if (nonfinite && range_flag) *range_flag = 1;
  }
}

void launch_attn_full_h2(const float* qkv, float* out, const int* seq_off, const int* seq_len, const int* prefix_len,
                         int batch, int max_len, hipStream_t s, unsigned short* planes, long plane_stride, int* range_flag,
                         int prio, const int* q_first, const int* c_off) {
  if (batch <= 0 || max_len <= 0) return;
  const int nqb = (max_len + QB - 1) / QB;                   // batch * N_HEAD is a multiple of 8 (16 heads)
  const dim3 grid(nqb * N_HEAD * batch), block(256);
#define VX_ATTN_H2_GO(P) hipLaunchKernelGGL(attn_full_h2_kernel<P>, grid, block, 0, s, qkv, out, seq_off, seq_len, prefix_len, nqb, planes, plane_stride, range_flag, planes ? q_first : nullptr, planes ? c_off : nullptr)
#ifdef VX_DEV_PROBES
  static const int env_prio = [] { const char* e = getenv("VX_ATTN_PRIO_RT"); return e ? atoi(e) : -1; }();
  const int p = prio >= 0 ? prio : (env_prio >= 0 ? env_prio : VX_ATTN_PRIO);
  if (p == 1) VX_ATTN_H2_GO(1);
  else if (p == 2) VX_ATTN_H2_GO(2);
  else if (p == 3) VX_ATTN_H2_GO(3);
  else VX_ATTN_H2_GO(0);
#else
  (void)prio;
  VX_ATTN_H2_GO(VX_ATTN_PRIO);
#endif
#undef VX_ATTN_H2_GO
}

}  // namespace vx
