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
#include "vx_common.h"

namespace vx {

constexpr int QB = 128, KT = 32, K_LD = 68, V_LD = 64;

__global__ __launch_bounds__(256, 2) void attn_full_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                          const int* __restrict__ seq_off,
                                                          const int* __restrict__ seq_len,
                                                          const int* __restrict__ prefix_len) {
  __shared__ __attribute__((aligned(16))) float Ks[KT * K_LD];
  __shared__ __attribute__((aligned(16))) float Vs[KT * V_LD];

  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB;
  const int len = seq_len[b];
  if (q0 >= len) return;
  const long row0 = seq_off[b];
  const int S = prefix_len ? prefix_len[b] : 0x7fffffff;       // keys < S are visible to everyone
  const bool causal = prefix_len != nullptr;

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

  // block-uniform key range; wave-uniform early-out bound
  const int q_last = (q0 + QB - 1 < len ? q0 + QB - 1 : len - 1);
  int kv_end = len;
  if (causal) kv_end = (q_last < S) ? S : (q_last + 1 < len ? q_last + 1 : len);
  int wq_last = q0 + wid * 32 + 31;
  if (wq_last > len - 1) wq_last = len - 1;
  int wave_kv_end = len;
  if (causal) wave_kv_end = (wq_last < S) ? S : wq_last + 1;   // keys >= this are hidden from the whole wave

  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

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
  issue(0);

  for (int k0 = 0; k0 < kv_end; k0 += KT) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + 256 * i, key = f >> 4, c4 = (f & 15) * 4;
      *reinterpret_cast<f32x4*>(&Ks[key * K_LD + c4]) = rk[i];
      *reinterpret_cast<f32x4*>(&Vs[key * V_LD + c4]) = rv[i];
    }
    __syncthreads();
    if (k0 + KT < kv_end) issue(k0 + KT);
    if (k0 >= wave_kv_end) continue;                           // wave-uniform: nothing visible in this tile

    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[l31 * K_LD + c * 8 + hi * 4]);
#pragma unroll
      for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qreg[c * 4 + j], s, 0, 0, 0);
    }
    // s[r] = score(q = qi, key = k0 + (r&3) + 8*(r>>2) + 4*hi)
    float m_tile = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kj = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      bool vis = kj < len;
      if (causal) vis = vis && (kj < S || (qi >= S && kj <= qi));
      s[r] = vis ? s[r] : -INFINITY;
      m_tile = fmaxf(m_tile, s[r]);
    }
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
    const float m_new = fmaxf(m_run, m_tile);                  // finite: key 0 is visible to every query
    const float alpha = expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = expf(s[r] - m_new); psum += s[r]; }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float v0 = Vs[key * V_LD + l31];
      const float v1 = Vs[key * V_LD + 32 + l31];
      o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[r], o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[r], o[1], 0, 0, 0);
    }
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

void launch_attn_full(const float* qkv, float* out, const int* seq_off, const int* seq_len, const int* prefix_len,
                      int batch, int max_len, hipStream_t s) {
  if (batch <= 0 || max_len <= 0) return;
  dim3 grid((max_len + QB - 1) / QB, N_HEAD, batch);
  hipLaunchKernelGGL(attn_full_kernel, grid, dim3(256), 0, s, qkv, out, seq_off, seq_len, prefix_len);
}

}  // namespace vx
