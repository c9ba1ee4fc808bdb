// Full-sequence self-attention in exact fp32 on the f32 MFMA (v_mfma_f32_32x32x2_f32), flash-style.
//
// Replaces, on packed ragged rows:
//   * AR prefill: multi_head_attention_forward with the prefix-LM mask (modules/activation.py:142-167; mask built
//     at models/vallex.py:535-549: text rows see text only, audio rows see all text + causal audio), and
//   * NAR stages: MultiheadAttention.forward -> F.multi_head_attention_forward, no mask
//     (modules/activation.py:566-585), 12 layers x 7 stages.
// The mask is never materialised: visibility is computed from (query index, key index, S_b).
//
// Work decomposition: one 256-thread workgroup = 128 query rows of one (sequence, head); each of the 4 waves owns
// 32 query rows.  K/V tiles of 32 keys are staged through LDS (shared by the 4 waves).  Per wave and tile:
//   S^T = K . Q^T      32 MFMAs  (A = K tile rows from LDS via ds_read_b128, B = Q held in 32 VGPRs)
//   online softmax     the transposed product leaves ONE query per lane (col = lane&31): the row max / sum are
//                      15 in-register ops + one xor-32 shuffle, and the rescale of O is lane-local
//   O^T += V^T . P^T   32 MFMAs  (A = V columns from LDS via ds_read_b32, B = the P registers as they are)
// The C-layout row map (r&3)+8*(r>>2)+4*(lane>>5) of S^T is used directly as the key order of the second
// contraction, so P never moves between lanes.
#include <stdlib.h>

#include "vx_common.h"

namespace vx {

constexpr int QB = 128, KT = 32, K_LD = 68, V_LD = 64;
constexpr float MASKED = -1e30f;     // finite stand-in for -inf: exp_bf() maps it to exactly 0 without a NaN path

// Branch-free exp for softmax arguments (x <= 0): exp(x) = 2^(x*log2e) with the product carried in two floats
// (hi + lo) so the error stays at the ulp level of v_exp_f32 for |x| up to ~100 instead of growing with |x|.
// libm's expf compiles to a conditional block per call, which splits the tile loop into 19 basic blocks and stops the
// scheduler from overlapping the softmax with the MFMAs.
__device__ __forceinline__ float exp_bf(float x) {
  const float L2E = 1.44269504088896341f, L2E_LO = 1.925963033500649e-08f, LN2 = 0.6931471805599453f;
  const float ph = x * L2E;
  const float pl = fmaf(x, L2E, -ph) + x * L2E_LO;
  const float e = __builtin_amdgcn_exp2f(ph);
  return fmaf(e, pl * LN2, e);
}

// V = 0 product kernel.  Timing probes (tools/attn_bench.py, results meaningless): V = 1 no K/V staging after the first two
// tiles; V = 2 no MFMAs; V = 3 no softmax arithmetic.
// NB: LDS buffers of the K / V tile ring.  2 = the product (two barriers per tile: everyone done reading `cur` | tile
// t + 2 written into it).  3 (round-6 experiment, as attn_full_h2.hip always had it; measured null, see launch_attn_full): tile t + 2 goes into the THIRD buffer -- the one tile t - 1
// occupied, which every wave left before the barrier that ended iteration t - 1 -- so ONE barrier per tile is enough.  Same
// operations on the same values in the same order: bit-identical output.
template <int V, int NB>
__global__ __launch_bounds__(256, 2) void attn_full_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                          const int* __restrict__ seq_off,
                                                          const int* __restrict__ seq_len,
                                                          const int* __restrict__ prefix_len, int nqb,
                                                          const int* __restrict__ q_first, const int* __restrict__ c_off,
                                                          int mask_all) {
  // q_first / c_off (optional, round 6): ROW TRIMMING for the last decoder layer of a NAR stage in the reference-arithmetic mode -- only
  // the generated frames reach a predict layer (models/vallex.py:672-679), so only the queries [q_first[b], len) of every sequence
  // are needed.  Query blocks entirely before q_first[b] are skipped and the output is written COMPACTED: sequence-local query qi
  // lands in row c_off[b] + qi - q_first[b] of `out`.  Keys / values are all of the sequence and a kept query sees the same tiles
  // in the same order as without trimming: the same bits (the scheme of attn_full_h2.hip).
  __shared__ __attribute__((aligned(16))) float Ks[NB][KT * K_LD];
  __shared__ __attribute__((aligned(16))) float Vs[NB][KT * V_LD];

  // XCD-aware work order.  Workgroup ids are dealt round-robin to the 8 XCDs (id % 8), each with a private L2, and the
  // K/V of one (sequence, head) are re-read by every query block of that unit.  Units are therefore dealt 8 at a time
  // (one per XCD) and a unit's query blocks take the ids base + 8*q + (u % 8): same XCD, back to back, so K/V is fetched
  // into that L2 once instead of once per query block (measured L2 hit rate of the naive order: 23 %).
  const int id = blockIdx.x, per8 = 8 * nqb;
  const int grp = id / per8, rem = id - grp * per8;
  const int u = grp * 8 + (rem & 7);
  const int b = u / N_HEAD, h = u - b * N_HEAD, q0 = (rem >> 3) * QB;
  const int len = seq_len[b];
  if (q0 >= len) return;
  const int qf = q_first ? q_first[b] : 0;
  if (q0 + QB <= qf) return;                                    // no query of this block is needed
  const long row0 = seq_off[b];
  const int S = prefix_len ? prefix_len[b] : 0x7fffffff;       // keys < S are visible to everyone
  const bool causal = prefix_len != nullptr;   // only narrows the block's key range below

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int qi = q0 + wid * 32 + l31;                          // this lane's query (sequence-local index)
  const int qc = qi < len ? qi : len - 1;

  // Q fragment: Q[q][8c + 4hi + j] * 1/sqrt(64)  (power of two: exact, same as scaling the scores)
  float qreg[32];
  {
    const float* qp = qkv + (row0 + qc) * (long)(3 * D_MODEL) + h * D_HEAD + hi * 4;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(qp + c * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) qreg[c * 4 + j] = t[j] * 0.125f;
    }
  }

  // block-uniform key range
  const int q_last = (q0 + QB - 1 < len ? q0 + QB - 1 : len - 1);
  int kv_end = len;
  if (causal) kv_end = (q_last < S) ? S : (q_last + 1 < len ? q_last + 1 : len);

  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  float m_run = MASKED, l_run = 0.f;

  // staging map: float4 index f = tid + 256*i over 32 keys x 16 float4
  const float* kbase = qkv + row0 * (long)(3 * D_MODEL) + D_MODEL + h * D_HEAD;
  const float* vbase = kbase + D_MODEL;
  f32x4 rk[2], rv[2];
  auto issue = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + 256 * i, key = f >> 4, c4 = (f & 15) * 4;
      int kk = k0 + key;
      kk = kk < len ? kk : len - 1;
      rk[i] = *reinterpret_cast<const f32x4*>(kbase + kk * (long)(3 * D_MODEL) + c4);
      rv[i] = *reinterpret_cast<const f32x4*>(vbase + kk * (long)(3 * D_MODEL) + c4);
    }
  };
  auto stage_write = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + 256 * i, key = f >> 4, c4 = (f & 15) * 4;
      *reinterpret_cast<f32x4*>(&Ks[buf][key * K_LD + c4]) = rk[i];
      *reinterpret_cast<f32x4*>(&Vs[buf][key * V_LD + c4]) = rv[i];
    }
  };
  // S^T tile = K_tile . Q^T: 32 MFMAs, A fragment from LDS, B = Q registers
  auto qk = [&](int buf) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[buf][l31 * K_LD + c * 8 + hi * 4]);
#pragma unroll
      for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qreg[c * 4 + j], s, 0, 0, 0);
    }
    return s;
  };

  // Software pipeline (two LDS buffers): iteration t issues the QK^T MFMAs of tile t+1 BEFORE the softmax of tile t,
  // so the ~500 VALU instructions of the softmax sit in the shadow of 32 independent MFMAs of the same wave instead
  // of idling the matrix pipe (all waves of a CU run in lock-step through identical tiles, so other waves do not fill
  // that gap by themselves).  Tile t+2 travels global -> registers during iteration t and is written to LDS at its end.
  const int ntiles = (kv_end + KT - 1) / KT;
  issue(0);
  stage_write(0);
  if (NB == 2) __syncthreads();
  if (1 < ntiles) issue(KT);
  f32x16 s_cur;
  if (NB == 2) s_cur = qk(0);
  if (1 < ntiles) stage_write(1);
  __syncthreads();
  if (2 < ntiles) issue(2 * KT);
  if (NB == 3) s_cur = qk(0);

  // Round 6: the visibility mask costs ~100 of the ~500 VALU instructions of a tile's softmax and is the identity on every tile that
  // lies entirely inside what ALL 32 queries of the wave may see: keys below `len`, and -- prefix-LM mask of the prefill -- below S when
  // the wave holds text queries, below its first query + 1 when it holds audio queries only.  The NAR stages (no mask) need it on
  // the last, ragged tile of a sequence only.  A wave-uniform test per tile skips it (the h2 kernel applies its mask the same way).
  const int q_w0 = q0 + wid * 32;
  const int lim_min = causal ? (q_w0 < S ? S : q_w0 + 1) : 0x7fffffff;
  // (mask_all: VX_ATTN_F32_MASKALL=1, the kernel of rounds 1-5 for A/B -- the mask on every tile)
  const int full_keys = mask_all ? 0 : __builtin_amdgcn_readfirstlane(len < lim_min ? len : lim_min);   // tiles ending at or below: no mask

  for (int t = 0; t < ntiles; ++t) {
    // buffers of tile t (its V is read in phase 2), of tile t + 1 (its K in phase 1) and the one tile t + 2 is written into
    const int k0 = t * KT, cur = NB == 3 ? t % 3 : (t & 1), nxt = NB == 3 ? (t + 1) % 3 : (cur ^ 1), wr = NB == 3 ? (t + 2) % 3 : cur;
    // ---- phase 1: QK^T of tile t+1 (32 MFMAs) with the softmax of tile t threaded through it.
    // In-order issue + dependent MFMA chains mean VALU work only overlaps the matrix pipe if it physically sits between
    // the MFMAs, so the tile is cut into 16 steps of [2 MFMAs | one slice of the softmax], fenced by sched_barrier(0).
    // (When there is no tile t+1 the product reads a stale buffer and is discarded; fully hidden tiles contribute
    // exp(-1e30) = 0.)   s_cur[r] = score(q = qi, key = k0 + (r&3) + 8*(r>>2) + 4*hi)
    f32x16 sA;                                                 // one chain: dependent latency == issue interval (64 cyc)
#pragma unroll
    for (int r = 0; r < 16; ++r) sA[r] = 0.f;
    f32x4 kfA, kfB;
    float m_new = m_run, alpha = 1.f, psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int cp = i >> 2, j = i & 3;
      if (j == 0) {
        kfA = *reinterpret_cast<const f32x4*>(&Ks[nxt][l31 * K_LD + (2 * cp) * 8 + hi * 4]);
        kfB = *reinterpret_cast<const f32x4*>(&Ks[nxt][l31 * K_LD + (2 * cp + 1) * 8 + hi * 4]);
      }
      if (V == 2) {
        sA[0] += kfA[j] * qreg[(2 * cp) * 4 + j] + kfB[j] * qreg[(2 * cp + 1) * 4 + j];
      } else {
        sA = __builtin_amdgcn_mfma_f32_32x32x2f32(kfA[j], qreg[(2 * cp) * 4 + j], sA, 0, 0, 0);
        sA = __builtin_amdgcn_mfma_f32_32x32x2f32(kfB[j], qreg[(2 * cp + 1) * 4 + j], sA, 0, 0, 0);
      }
      if (V == 3) {
        // probe: no softmax work at all
      } else if (i < 4) {                                      // steps 0-3: visibility mask, 4 keys per step (uniform skip: see above)
        if (k0 + KT > full_keys)
#pragma unroll
        for (int r = 4 * i; r < 4 * i + 4; ++r) {
          const int kj = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          // branch-free (S = INT_MAX when unmasked): in range, and (text key, or causal audio key)
          const bool vis = (kj < len) & ((kj < S) | ((qi >= S) & (kj <= qi)));
          s_cur[r] = vis ? s_cur[r] : MASKED;
        }
      } else if (i == 4) {                                     // step 4: running max (other 16 keys live in lane ^ 32)
        float m_tile = MASKED;
#pragma unroll
        for (int r = 0; r < 16; ++r) m_tile = fmaxf(m_tile, s_cur[r]);
        // inline asm on two distinct registers: the compiler folds the two results of __builtin_amdgcn_permlane32_swap into one
        // (ROCm 7.2), which left the partner's keys out of the maximum (see attn_full_h2.hip)
        unsigned ua = __builtin_bit_cast(unsigned, m_tile), ub = ua;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ua), "+v"(ub));   // VALU op, no LDS round trip
        m_tile = fmaxf(__builtin_bit_cast(float, ua), __builtin_bit_cast(float, ub));
        m_new = fmaxf(m_run, m_tile);                          // finite: key 0 is visible to every query
        alpha = exp_bf(m_run - m_new);
      } else if (i < 13) {                                     // steps 5-12: two exps per step
#pragma unroll
        for (int r = 2 * (i - 5); r < 2 * (i - 5) + 2; ++r) { s_cur[r] = exp_bf(s_cur[r] - m_new); psum += s_cur[r]; }
      } else if (i == 13) {
        l_run = l_run * alpha + psum;
        m_run = m_new;
      } else {                                                 // steps 14-15: rescale the two halves of O
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i - 14][r] *= alpha;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- phase 2: O^T += V^T . P^T, 32 MFMAs fed by ds_read of V
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float v0 = Vs[cur][key * V_LD + l31];
      const float v1 = Vs[cur][key * V_LD + 32 + l31];
      if (V == 2) {
        o[0][r] += v0 * s_cur[r];
        o[1][r] += v1 * s_cur[r];
      } else {
        o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s_cur[r], o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s_cur[r], o[1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s_cur[r] = sA[r];
    if (NB == 2) __syncthreads();                              // two buffers: everyone is done reading buffer `cur` (= wr)
    if (V != 1 && t + 2 < ntiles) stage_write(wr);             // registers hold tile t+2
    __syncthreads();
    if (V != 1 && t + 3 < ntiles) issue((t + 3) * KT);
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (qi < len && qi >= qf) {
    const float inv = 1.0f / l_tot;
    const long orow = c_off ? (long)c_off[b] + (qi - qf) : row0 + qi;
    float* op = out + orow * (long)D_MODEL + h * D_HEAD + 4 * hi;
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = o[half][g4 * 4 + e] * inv;
        *reinterpret_cast<f32x4*>(op + half * 32 + g4 * 8) = t;   // d = 32*half + 8*g4 + 4*hi + e
      }
  }
}

void launch_attn_full(const float* qkv, float* out, const int* seq_off, const int* seq_len, const int* prefix_len,
                      int batch, int max_len, hipStream_t s, const int* q_first, const int* c_off, int nbuf) {
  if (batch <= 0 || max_len <= 0) return;
  const int nqb = (max_len + QB - 1) / QB;                   // batch * N_HEAD is a multiple of 8 (16 heads)
  // nbuf 0: the product's choice = two LDS buffers.  The three-buffer / one-barrier variant measured -0.7 % at the NAR shape, +0.4 % at the
  // prefill shape and nothing end to end (profiles/r06_attn_f32_nbuf_ab.log: NAR 614.7 / 615.2 vs 615.5 / 616.7 ms): below the 3 % bar, so
  // the kernel of rounds 1-5 stays; VX_ATTN_F32_NBUF=3 selects the variant (bit-identical) for A/B
  static const int env_nb = [] { const char* e = getenv("VX_ATTN_F32_NBUF"); return (e && e[0] == '3') ? 3 : 2; }();
  if (nbuf == 0) nbuf = env_nb;
  static const int mask_all = [] { const char* e = getenv("VX_ATTN_F32_MASKALL"); return (e && e[0] == '1') ? 1 : 0; }();
  if (nbuf == 2)
    hipLaunchKernelGGL((attn_full_kernel<0, 2>), dim3(nqb * N_HEAD * batch), dim3(256), 0, s, qkv, out, seq_off, seq_len, prefix_len, nqb,
                       q_first, c_off, mask_all);
  else
    hipLaunchKernelGGL((attn_full_kernel<0, 3>), dim3(nqb * N_HEAD * batch), dim3(256), 0, s, qkv, out, seq_off, seq_len, prefix_len, nqb,
                       q_first, c_off, mask_all);
}

#ifdef VX_DEV_PROBES   // timing probes: tools-only build (vall-e-x_amd/_build.py --dev), never in the product library
void launch_attn_full_probe(const float* qkv, float* out, const int* seq_off, const int* seq_len, const int* prefix_len,
                            int batch, int max_len, int variant, hipStream_t s) {
  const int nqb = (max_len + QB - 1) / QB;
  const dim3 grid(nqb * N_HEAD * batch), block(256);
  if (variant == 1) hipLaunchKernelGGL((attn_full_kernel<1, 2>), grid, block, 0, s, qkv, out, seq_off, seq_len, prefix_len, nqb, (const int*)nullptr, (const int*)nullptr, 1);
  else if (variant == 2) hipLaunchKernelGGL((attn_full_kernel<2, 2>), grid, block, 0, s, qkv, out, seq_off, seq_len, prefix_len, nqb, (const int*)nullptr, (const int*)nullptr, 1);
  else hipLaunchKernelGGL((attn_full_kernel<3, 2>), grid, block, 0, s, qkv, out, seq_off, seq_len, prefix_len, nqb, (const int*)nullptr, (const int*)nullptr, 1);
}

#endif

}  // namespace vx
