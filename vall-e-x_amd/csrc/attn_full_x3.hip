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
#include "vx_common.h"

namespace vx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

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
// two tiles; V = 2 no MFMAs; V = 3 no softmax arithmetic.
template <int V>
__global__ __launch_bounds__(256, 2) void attn_full_x3_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                             const int* __restrict__ seq_off,
                                                             const int* __restrict__ seq_len,
                                                             const int* __restrict__ prefix_len, int nqb) {
  __shared__ __attribute__((aligned(16))) unsigned char Kp[2][3][KP_SZ];
  __shared__ __attribute__((aligned(16))) unsigned char Vt[2][3][VT_SZ];

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
  auto stage_write = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = (tid + 256 * i) >> 4;
      bf16x4 p1, p2, p3;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        __bf16 a1, a2, a3;
        split3(rk[i][e], a1, a2, a3);
        p1[e] = a1; p2[e] = a2; p3[e] = a3;
      }
      const int off = key * KP_LD + c4 * 2;
      *reinterpret_cast<bf16x4*>(&Kp[buf][0][off]) = p1;
      *reinterpret_cast<bf16x4*>(&Kp[buf][1][off]) = p2;
      *reinterpret_cast<bf16x4*>(&Kp[buf][2][off]) = p3;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bf16x2 p1, p2, p3;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        __bf16 a1, a2, a3;
        split3(rv[i][e], a1, a2, a3);
        p1[i] = a1; p2[i] = a2; p3[i] = a3;
      }
      const int off = vt_row(c4 + e) + vpos * 2;
      *reinterpret_cast<bf16x2*>(&Vt[buf][0][off]) = p1;
      *reinterpret_cast<bf16x2*>(&Vt[buf][1][off]) = p2;
      *reinterpret_cast<bf16x2*>(&Vt[buf][2][off]) = p3;
    }
  };
  auto kfrag = [&](int buf, int s, bf16x8 (&kf)[3]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
      kf[p] = *reinterpret_cast<const bf16x8*>(&Kp[buf][p][l31 * KP_LD + (2 * s + hi) * 16]);
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

  // same software pipeline as attn_full.hip: QK^T of tile t+1 is issued before the softmax of tile t
  const int ntiles = (kv_end + KT - 1) / KT;
  issue(0);
  stage_write(0);
  __syncthreads();
  if (1 < ntiles) issue(KT);
  f32x16 s_cur = qk(0);
  if (1 < ntiles) stage_write(1);
  __syncthreads();
  if (2 < ntiles) issue(2 * KT);

  for (int t = 0; t < ntiles; ++t) {
    const int k0 = t * KT, cur = t & 1, nxt = cur ^ 1;
    // ---- phase 1: S^T of tile t+1 (24 MFMAs) with the softmax of tile t threaded through it: 12 steps of
    // [2 MFMAs | one slice of the softmax], fenced by sched_barrier(0).
    f32x16 sA;
#pragma unroll
    for (int r = 0; r < 16; ++r) sA[r] = 0.f;
    bf16x8 kf[3];
    float m_new = m_run, alpha = 1.f, psum = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int s = i / 3, j = i - 3 * s;
      if (j == 0) kfrag(nxt, s, kf);
      if (V == 2) {
        sA[0] += (float)kf[j][0] * (float)qp[s][j][0];
      } else if (j == 0) {
        sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2], qp[s][0], sA, 0, 0, 0);
        sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qp[s][2], sA, 0, 0, 0);
      } else if (j == 1) {
        sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qp[s][1], sA, 0, 0, 0);
        sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qp[s][0], sA, 0, 0, 0);
      } else {
        sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qp[s][1], sA, 0, 0, 0);
        sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qp[s][0], sA, 0, 0, 0);
      }
      if (V == 3) {
        // probe: no softmax work at all
      } else if (i < 4) {                                      // steps 0-3: visibility mask, 4 keys per step
#pragma unroll
        for (int r = 4 * i; r < 4 * i + 4; ++r) {
          const int kj = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const bool vis = (kj < len) & ((kj < S) | ((qi >= S) & (kj <= qi)));
          s_cur[r] = vis ? s_cur[r] : MASKED;
        }
      } else if (i == 4) {                                     // step 4: running max (other 16 keys live in lane ^ 32)
        float m_tile = MASKED;
#pragma unroll
        for (int r = 0; r < 16; ++r) m_tile = fmaxf(m_tile, s_cur[r]);
        const unsigned um = __builtin_bit_cast(unsigned, m_tile);
        const auto sw = __builtin_amdgcn_permlane32_swap(um, um, false, false);
        m_tile = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1]));
        m_new = fmaxf(m_run, m_tile);                          // finite: key 0 is visible to every query
        alpha = exp_bf(m_run - m_new);
      } else if (i < 11) {                                     // steps 5-10: 3,3,3,3,2,2 exps
        const int r0 = i < 9 ? 3 * (i - 5) : 12 + 2 * (i - 9), r1 = i < 9 ? r0 + 3 : r0 + 2;
#pragma unroll
        for (int r = r0; r < r1; ++r) { s_cur[r] = exp_bf(s_cur[r] - m_new); psum += s_cur[r]; }
      } else {                                                 // step 11: running sum, rescale O
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- phase 2: O^T += V^T . P^T: P split in registers (B operand), V^T planes from LDS, 24 MFMAs in two chains
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 pp[3];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        __bf16 a1, a2, a3;
        split3(s_cur[8 * s + e], a1, a2, a3);
        pp[0][e] = a1; pp[1][e] = a2; pp[2][e] = a3;
      }
      bf16x8 v0[3], v1[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        v0[p] = *reinterpret_cast<const bf16x8*>(&Vt[cur][p][vt_row(l31) + (2 * s + hi) * 16]);
        v1[p] = *reinterpret_cast<const bf16x8*>(&Vt[cur][p][vt_row(l31 + 32) + (2 * s + hi) * 16]);
      }
      if (V == 2) {
        o[0][0] += (float)v0[0][0] * (float)pp[0][0] + (float)v0[1][1] * (float)pp[1][1] + (float)v0[2][2] * (float)pp[2][2];
        o[1][0] += (float)v1[0][0] * (float)pp[0][0] + (float)v1[1][1] * (float)pp[1][1] + (float)v1[2][2] * (float)pp[2][2];
      } else {
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0[2], pp[0], o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1[2], pp[0], o[1], 0, 0, 0);
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0[0], pp[2], o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1[0], pp[2], o[1], 0, 0, 0);
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0[1], pp[1], o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1[1], pp[1], o[1], 0, 0, 0);
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0[1], pp[0], o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1[1], pp[0], o[1], 0, 0, 0);
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0[0], pp[1], o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1[0], pp[1], o[1], 0, 0, 0);
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0[0], pp[0], o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1[0], pp[0], o[1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s_cur[r] = sA[r];
    __syncthreads();                                           // everyone is done reading buffer `cur`
    if (V != 1 && t + 2 < ntiles) stage_write(cur);            // registers hold tile t+2
    __syncthreads();
    if (V != 1 && t + 3 < ntiles) issue((t + 3) * KT);
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (qi < len) {
    const float inv = 1.0f / l_tot;
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
  }
}

void launch_attn_full_x3(const float* qkv, float* out, const int* seq_off, const int* seq_len, const int* prefix_len,
                         int batch, int max_len, int variant, hipStream_t s) {
  if (batch <= 0 || max_len <= 0) return;
  const int nqb = (max_len + QB - 1) / QB;                   // batch * N_HEAD is a multiple of 8 (16 heads)
  const dim3 grid(nqb * N_HEAD * batch), block(256);
  if (variant == 1) hipLaunchKernelGGL(attn_full_x3_kernel<1>, grid, block, 0, s, qkv, out, seq_off, seq_len, prefix_len, nqb);
  else if (variant == 2) hipLaunchKernelGGL(attn_full_x3_kernel<2>, grid, block, 0, s, qkv, out, seq_off, seq_len, prefix_len, nqb);
  else if (variant == 3) hipLaunchKernelGGL(attn_full_x3_kernel<3>, grid, block, 0, s, qkv, out, seq_off, seq_len, prefix_len, nqb);
  else hipLaunchKernelGGL(attn_full_x3_kernel<0>, grid, block, 0, s, qkv, out, seq_off, seq_len, prefix_len, nqb);
}

}  // namespace vx
