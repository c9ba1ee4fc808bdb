// AR decode step for gfx950: the HBM-bound core of VALLE.inference (models/vallex.py:528-598).
//
// One step = one new codec frame for every active row of the micro-batch (<= 32 rows).  Per layer the reference
// runs norm1 -> QKV linear -> cat(past_kv) -> q.k^T -> softmax -> .v -> out_proj -> +res -> norm2 -> linear1 -> ReLU
// -> linear2 -> +res on ONE row per sequence (modules/transformer.py:337-347, modules/activation.py:142-167), copying
// the whole KV cache with torch.cat every layer.  Here:
//
//   * the KV cache is an in-place arena  [(b*16 + head)][t][64] fp32 per layer: a head's keys/values are ONE
//     contiguous stream; a step appends one 256-B row per head and reads 512*ctx bytes per (row, head);
//   * dec_attn streams that cache with 16 lanes per 256-B row (float4 per lane, 4 rows = 1 KiB per wave load,
//     16 KiB in flight per wave), DPP row reductions for q.k, online softmax per 16-lane group, and combines
//     groups -> waves -> (optionally) ctx-splits;
//   * the projections are weight-streaming skinny GEMMs on the f32 MFMA: the batch (padded to 32 rows) is the
//     32-wide MFMA column block, weights are pre-packed at load into the lane-linear image the MFMA A operand
//     wants, so every wave load is 1 KiB contiguous and each weight byte is read exactly once per step
//     (non-temporal).  Split-K partial slabs are summed in the consumer's prologue -- no atomics, bit-stable.
//   * per-row lengths / positions / EOS flags live on the device, so the whole step is a fixed launch sequence
//     (hipGraph-capturable) and the host only polls `active` every few steps.
#include "vx_common.h"

namespace vx {

// ------------------------------------------------------------------------------------------------------------
// packed images
//   weight : Wp[((nt*KB + kb)*64 + lane)*4 + j] = W[nt*32 + (lane&31)][kb*8 + 4*(lane>>5) + j]   (KB = K/8)
//   x      : xp[(kb*64 + b + 32*hi)*4 + j]      = x[b][kb*8 + 4*hi + j]                            (b < 32)
// MFMA step j of k-block kb contracts k = kb*8 + 4*(lane>>5) + j on both operands (a bijection of the 8 k's).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ W, int N, int K,
                                                          float* __restrict__ Wp, int Npad) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;         // float4 index
  const int KB = K / 8;
  const long total = (long)(Npad / 32) * KB * 64;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  const long t = i >> 6;
  const int kb = (int)(t % KB), nt = (int)(t / KB);
  const int n = nt * 32 + (lane & 31), k = kb * 8 + 4 * (lane >> 5);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (n < N) v = *reinterpret_cast<const f32x4*>(W + (long)n * K + k);
  *reinterpret_cast<f32x4*>(Wp + i * 4) = v;
}

void launch_pack_weight(const float* W, int N, int K, float* Wp, int Npad, hipStream_t s) {
  const long total = (long)(Npad / 32) * (K / 8) * 64;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, N, K, Wp, Npad);
}

// ------------------------------------------------------------------------------------------------------------
// skinny GEMM: partial[ks][b][n] = sum_{k in slice ks} x[b][k] * W[n][k]
// grid = (Npad/32, splitk); 4 waves split the block's K slice and reduce through LDS.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const float* __restrict__ Wp, const float* __restrict__ xp,
                                                          float* __restrict__ partial, int Npad, int K, int splitk) {
  __shared__ __attribute__((aligned(16))) float red[3 * 16 * 64];
  const int nt = blockIdx.x, ks = blockIdx.y;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int KB = K / 8;
  const int kb_per_wave = KB / (splitk * 4);
  const int kb0 = (ks * 4 + wid) * kb_per_wave;
  const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + ((long)nt * KB + kb0) * 64 + lane;
  const f32x4* xq = reinterpret_cast<const f32x4*>(xp) + (long)kb0 * 64 + lane;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int i = 0; i < kb_per_wave; i += 8) {
    f32x4 w[8], x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = __builtin_nontemporal_load(wp + (long)(i + u) * 64);
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = xq[(long)(i + u) * 64];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[u][j], x[u][j], acc, 0, 0, 0);
  }

  // acc[r] = out[b = lane&31][n = nt*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)]
  if (wid > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wid - 1) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (wid == 0) {
#pragma unroll
    for (int w2 = 0; w2 < 3; ++w2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += red[(w2 * 16 + r) * 64 + lane];
    float* dst = partial + ((long)ks * MB + (lane & 31)) * Npad + nt * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 t = {acc[g4 * 4], acc[g4 * 4 + 1], acc[g4 * 4 + 2], acc[g4 * 4 + 3]};
      *reinterpret_cast<f32x4*>(dst + g4 * 8) = t;
    }
  }
}

void launch_skinny_gemm(const float* Wp, const float* xp, float* partial, int Npad, int K, int splitk,
                        hipStream_t s) {
  hipLaunchKernelGGL(skinny_gemm_kernel, dim3(Npad / 32, splitk), dim3(256), 0, s, Wp, xp, partial, Npad, K, splitk);
}

// ------------------------------------------------------------------------------------------------------------
// row kernels of the step: one 256-thread block per batch row, thread t owns columns 4t..4t+3 of the 1024.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wid = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[wid] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// LayerNorm of the row held as one float4 per thread, written in the packed-x image (K = 1024).
__device__ __forceinline__ void ln_pack_row(f32x4 v, int b, const float* __restrict__ g,
                                            const float* __restrict__ bb, float* __restrict__ xp, float* sh) {
  const int t = threadIdx.x;
  const float mean = block_sum_256(v[0] + v[1] + v[2] + v[3], sh) * (1.0f / D_MODEL);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(block_sum_256(q, sh) * (1.0f / D_MODEL) + LN_EPS);
  const f32x4 gg = *reinterpret_cast<const f32x4*>(g + t * 4);
  const f32x4 be = *reinterpret_cast<const f32x4*>(bb + t * 4);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * gg[e] + be[e];
  // columns 4t..4t+3: kb = t>>1, hi = t&1
  *reinterpret_cast<f32x4*>(xp + (((long)(t >> 1) * 64) + b + 32 * (t & 1)) * 4) = o;
}

// h[b] = resid[b] + (sum_ks partial[ks][b] + bias);  xp = pack(LN(h)).  partial may be null (splitk = 0).
// modules/transformer.py:345-346 (x = x + attn_out ; x = x + ff(norm2(x))) fused with the next norm.
__global__ __launch_bounds__(256) void dec_reduce_ln_pack_kernel(const float* __restrict__ partial, int splitk,
                                                                 int npad, const float* __restrict__ bias,
                                                                 const float* __restrict__ resid,
                                                                 float* __restrict__ h, const float* __restrict__ g,
                                                                 const float* __restrict__ bb,
                                                                 float* __restrict__ xp) {
  __shared__ float sh[4];
  const int b = blockIdx.x, t = threadIdx.x;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (splitk > 0) {
    v = *reinterpret_cast<const f32x4*>(partial + (long)b * npad + t * 4);
    for (int ks = 1; ks < splitk; ++ks) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(partial + ((long)ks * MB + b) * npad + t * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += p[e];
    }
    if (bias) {
      const f32x4 bi = *reinterpret_cast<const f32x4*>(bias + t * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += bi[e];
    }
  }
  if (resid) {
    const f32x4 r = *reinterpret_cast<const f32x4*>(resid + (long)b * D_MODEL + t * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = r[e] + v[e];
  }
  if (h) *reinterpret_cast<f32x4*>(h + (long)b * D_MODEL + t * 4) = v;
  ln_pack_row(v, b, g, bb, xp, sh);
}

void launch_dec_reduce_ln_pack(const float* partial, int splitk, int npad, const float* bias, const float* resid,
                               float* h, const float* g, const float* b, float* xp, int batch, hipStream_t s) {
  hipLaunchKernelGGL(dec_reduce_ln_pack_kernel, dim3(batch), dim3(256), 0, s, partial, splitk, npad, bias, resid, h, g,
                     b, xp);
}

// xp(K=4096) = pack(relu(sum_ks partial + bias))   -- linear1 epilogue (modules/transformer.py:371-373)
__global__ __launch_bounds__(256) void dec_reduce_relu_pack_kernel(const float* __restrict__ partial, int splitk,
                                                                   const float* __restrict__ bias,
                                                                   float* __restrict__ xp) {
  const int b = blockIdx.x, c4 = blockIdx.y * 256 + threadIdx.x;       // float4 column index, 0..1023
  f32x4 v = *reinterpret_cast<const f32x4*>(partial + (long)b * D_FF + c4 * 4);
  for (int ks = 1; ks < splitk; ++ks) {
    const f32x4 p = *reinterpret_cast<const f32x4*>(partial + ((long)ks * MB + b) * D_FF + c4 * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += p[e];
  }
  const f32x4 bi = *reinterpret_cast<const f32x4*>(bias + c4 * 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + bi[e], 0.f);
  *reinterpret_cast<f32x4*>(xp + (((long)(c4 >> 1) * 64) + b + 32 * (c4 & 1)) * 4) = v;
}

void launch_dec_reduce_relu_pack(const float* partial, int splitk, const float* bias, float* xp, int batch,
                                 hipStream_t s) {
  hipLaunchKernelGGL(dec_reduce_relu_pack_kernel, dim3(batch, 4), dim3(256), 0, s, partial, splitk, bias, xp);
}

// Start of a step: embed the newest token of each row at its audio position (the reference re-embeds all of y and
// keeps the last row, models/vallex.py:529-531,552-553), then norm1 of layer 0.
__global__ __launch_bounds__(256) void dec_embed_ln_pack_kernel(const int* __restrict__ tok,
                                                                const int* __restrict__ pos,
                                                                const float* __restrict__ tab,
                                                                const float* __restrict__ alpha,
                                                                const float* __restrict__ pe, float* __restrict__ h,
                                                                const float* __restrict__ g,
                                                                const float* __restrict__ bb,
                                                                float* __restrict__ xp) {
  __shared__ float sh[4];
  const int b = blockIdx.x, t = threadIdx.x;
  f32x4 v = *reinterpret_cast<const f32x4*>(tab + (long)tok[b] * D_MODEL + t * 4);
  const f32x4 p = *reinterpret_cast<const f32x4*>(pe + (long)pos[b] * D_MODEL + t * 4);
  const float a = alpha[0];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(v[e], __fmul_rn(a, p[e]));
  *reinterpret_cast<f32x4*>(h + (long)b * D_MODEL + t * 4) = v;
  ln_pack_row(v, b, g, bb, xp, sh);
}

void launch_dec_embed_ln_pack(const int* tok, const int* pos, const float* tab, const float* alpha, const float* pe,
                              float* h, const float* g, const float* b, float* xp, int batch, hipStream_t s) {
  hipLaunchKernelGGL(dec_embed_ln_pack_kernel, dim3(batch), dim3(256), 0, s, tok, pos, tab, alpha, pe, h, g, b, xp);
}

// ------------------------------------------------------------------------------------------------------------
// dec_attn: softmax(q.K^T/8).V over the cache for one new token per row  (modules/activation.py:148-165 with T=1).
// grid = (head, row, split).  Lane = (g = lane>>4 : row slot, c = lane&15 : float4 chunk of the 64-float row).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_sum16(float x) {
  int v = __builtin_bit_cast(int, x);
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v = __builtin_bit_cast(int, x);
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v = __builtin_bit_cast(int, x);
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));   // row_half_mirror
  v = __builtin_bit_cast(int, x);
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));   // row_mirror
  return x;
}

constexpr float NEG_BIG = -1e30f;
constexpr int ATT_U = 8;              // rows per lane-group in flight (x 4 groups x 4 waves = 128 rows / block iter)

__global__ __launch_bounds__(256) void dec_attn_kernel(const float* __restrict__ qkv_partial, int splitk,
                                                       const float* __restrict__ qkv_bias, float* __restrict__ kc,
                                                       float* __restrict__ vc, int Tmax,
                                                       const int* __restrict__ ctx_len,
                                                       const int* __restrict__ active, float* __restrict__ xp_out,
                                                       float* __restrict__ part_o, float* __restrict__ part_ml,
                                                       int nsplit) {
  __shared__ __attribute__((aligned(16))) float sh_o[4][64];
  __shared__ float sh_m[4], sh_l[4];
  const int h = blockIdx.x, b = blockIdx.y, sp = blockIdx.z;
  if (!active[b]) return;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const int ctx = ctx_len[b];                 // cached rows INCLUDING the new token (at ctx-1)
  const int npast = ctx - 1;

  // q / k_new / v_new of this head: reduce the QKV split-K partials + bias (in_proj, modules/activation.py:144)
  f32x4 q4, k4, v4;
  {
    const int NP = 3 * D_MODEL;
    const float* p = qkv_partial + (long)b * NP + h * D_HEAD + c * 4;
    q4 = *reinterpret_cast<const f32x4*>(p);
    k4 = *reinterpret_cast<const f32x4*>(p + D_MODEL);
    v4 = *reinterpret_cast<const f32x4*>(p + 2 * D_MODEL);
    for (int ks = 1; ks < splitk; ++ks) {
      const float* pk = p + (long)ks * MB * NP;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(pk);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(pk + D_MODEL);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(pk + 2 * D_MODEL);
#pragma unroll
      for (int e = 0; e < 4; ++e) { q4[e] += a0[e]; k4[e] += a1[e]; v4[e] += a2[e]; }
    }
    const float* bp = qkv_bias + h * D_HEAD + c * 4;
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(bp + D_MODEL);
    const f32x4 b2 = *reinterpret_cast<const f32x4*>(bp + 2 * D_MODEL);
#pragma unroll
    for (int e = 0; e < 4; ++e) { q4[e] = (q4[e] + b0[e]) * 0.125f; k4[e] += b1[e]; v4[e] += b2[e]; }
  }
  const long head_base = ((long)(b * N_HEAD + h) * Tmax) * D_HEAD;
  if (sp == 0 && wid == 0 && g == 0) {        // in-place append: present = (k, v) (modules/activation.py:151-157)
    *reinterpret_cast<f32x4*>(kc + head_base + (long)npast * D_HEAD + c * 4) = k4;
    *reinterpret_cast<f32x4*>(vc + head_base + (long)npast * D_HEAD + c * 4) = v4;
  }

  // this block's slice of the past rows
  const int chunk = ((npast + nsplit - 1) / nsplit + 15) & ~15;
  const int t0 = sp * chunk;
  const int t1 = (t0 + chunk < npast) ? t0 + chunk : npast;

  float m = NEG_BIG, l = 0.f;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  const f32x4* kp = reinterpret_cast<const f32x4*>(kc + head_base) + c;
  const f32x4* vp = reinterpret_cast<const f32x4*>(vc + head_base) + c;

  for (int base = t0 + wid * 4 + g; base < t1; base += 16 * ATT_U) {
    f32x4 kk[ATT_U], vv[ATT_U];
#pragma unroll
    for (int u = 0; u < ATT_U; ++u) {
      int t = base + 16 * u;
      t = t < t1 ? t : t1 - 1;
      kk[u] = __builtin_nontemporal_load(kp + (long)t * 16);
    }
#pragma unroll
    for (int u = 0; u < ATT_U; ++u) {
      int t = base + 16 * u;
      t = t < t1 ? t : t1 - 1;
      vv[u] = __builtin_nontemporal_load(vp + (long)t * 16);
    }
    float sc[ATT_U];
    float m_new = m;
#pragma unroll
    for (int u = 0; u < ATT_U; ++u) {
      float d = q4[0] * kk[u][0] + q4[1] * kk[u][1] + q4[2] * kk[u][2] + q4[3] * kk[u][3];
      d = dpp_sum16(d);
      sc[u] = (base + 16 * u < t1) ? d : NEG_BIG;
      m_new = fmaxf(m_new, sc[u]);
    }
    const float alpha = expf(m - m_new);
    l *= alpha;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] *= alpha;
#pragma unroll
    for (int u = 0; u < ATT_U; ++u) {
      const float p = (base + 16 * u < t1) ? expf(sc[u] - m_new) : 0.f;
      l += p;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += p * vv[u][e];
    }
    m = m_new;
  }
  // the new token itself (always visible: last mask row is all False, models/vallex.py:535-549)
  if (sp == nsplit - 1 && wid == 0 && g == 0) {
    float d = q4[0] * k4[0] + q4[1] * k4[1] + q4[2] * k4[2] + q4[3] * k4[3];
    d = dpp_sum16(d);
    const float m_new = fmaxf(m, d);
    const float alpha = expf(m - m_new), p = expf(d - m_new);
    l = l * alpha + p;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = o[e] * alpha + p * v4[e];
    m = m_new;
  }

  // combine the 4 lane-groups of the wave (xor 16, 32), then the 4 waves through LDS
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64);
    f32x4 o2;
#pragma unroll
    for (int e = 0; e < 4; ++e) o2[e] = __shfl_xor(o[e], off, 64);
    const float mn = fmaxf(m, m2), a1 = expf(m - mn), a2 = expf(m2 - mn);
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = o[e] * a1 + o2[e] * a2;
    m = mn;
  }
  if (g == 0) {
    *reinterpret_cast<f32x4*>(&sh_o[wid][c * 4]) = o;
    if (c == 0) { sh_m[wid] = m; sh_l[wid] = l; }
  }
  __syncthreads();
  if (wid == 0 && g == 0) {
    float mt = fmaxf(fmaxf(sh_m[0], sh_m[1]), fmaxf(sh_m[2], sh_m[3]));
    float lt = 0.f;
    f32x4 ot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float a = expf(sh_m[w] - mt);
      lt += sh_l[w] * a;
      const f32x4 ow = *reinterpret_cast<const f32x4*>(&sh_o[w][c * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) ot[e] += ow[e] * a;
    }
    if (nsplit == 1) {
      const float inv = 1.0f / lt;
#pragma unroll
      for (int e = 0; e < 4; ++e) ot[e] *= inv;
      // column k = h*64 + 4c: kb = h*8 + (c>>1), hi = c&1
      *reinterpret_cast<f32x4*>(xp_out + (((long)(h * 8 + (c >> 1)) * 64) + b + 32 * (c & 1)) * 4) = ot;
    } else {
      const long pi = ((long)(b * N_HEAD + h) * nsplit + sp);
      *reinterpret_cast<f32x4*>(part_o + pi * D_HEAD + c * 4) = ot;
      if (c == 0) { part_ml[pi * 2] = mt; part_ml[pi * 2 + 1] = lt; }
    }
  }
}

void launch_dec_attn(const float* qkv_partial, int splitk, const float* qkv_bias, float* kc, float* vc, int Tmax,
                     const int* ctx_len, const int* active, float* xp_out, float* part_o, float* part_ml, int nsplit,
                     int batch, hipStream_t s) {
  hipLaunchKernelGGL(dec_attn_kernel, dim3(N_HEAD, batch, nsplit), dim3(256), 0, s, qkv_partial, splitk, qkv_bias, kc,
                     vc, Tmax, ctx_len, active, xp_out, part_o, part_ml, nsplit);
}

__global__ __launch_bounds__(64) void dec_attn_combine_kernel(const float* __restrict__ part_o,
                                                              const float* __restrict__ part_ml, int nsplit,
                                                              const int* __restrict__ active,
                                                              float* __restrict__ xp_out) {
  const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  if (!active[b]) return;
  const long pi = (long)(b * N_HEAD + h) * nsplit;
  float mt = NEG_BIG;
  for (int s = 0; s < nsplit; ++s) mt = fmaxf(mt, part_ml[(pi + s) * 2]);
  float lt = 0.f, ot = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float a = expf(part_ml[(pi + s) * 2] - mt);
    lt += part_ml[(pi + s) * 2 + 1] * a;
    ot += part_o[(pi + s) * D_HEAD + d] * a;
  }
  const int k = h * D_HEAD + d;
  xp_out[(((long)(k >> 3) * 64) + b + 32 * ((k >> 2) & 1)) * 4 + (k & 3)] = ot / lt;
}

void launch_dec_attn_combine(const float* part_o, const float* part_ml, int nsplit, const int* active, float* xp_out,
                             int batch, hipStream_t s) {
  hipLaunchKernelGGL(dec_attn_combine_kernel, dim3(N_HEAD, batch), dim3(64), 0, s, part_o, part_ml, nsplit, active,
                     xp_out);
}

// ------------------------------------------------------------------------------------------------------------
// dec_sample: ar_predict_layer logits (split-K partials) -> topk_sampling (models/vallex.py:791-853) -> EOS / cap
// bookkeeping (models/vallex.py:572-598), one block per row, everything stays on the device.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_max_256(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void dec_sample_kernel(SampleArgs a) {
  __shared__ float lg[AR_LOGITS + 7];
  __shared__ float sh[4];
  __shared__ int sh_i[4];
  __shared__ int s_tok;
  const int b = blockIdx.x, t = threadIdx.x;
  const bool act = a.active[b] != 0;
  if (!act && !a.logits_out) return;

  for (int n = t; n < AR_LOGITS; n += 256) {
    float v = a.partial[(long)b * a.npad + n];
    for (int ks = 1; ks < a.splitk; ++ks) v += a.partial[((long)ks * MB + b) * a.npad + n];
    lg[n] = v;
    if (a.logits_out) a.logits_out[(long)b * AR_LOGITS + n] = v;
  }
  __syncthreads();
  if (!act || !a.commit) return;

  if (a.temperature != 1.0f) {                                   // :845-846
    for (int n = t; n < AR_LOGITS; n += 256) lg[n] = lg[n] / a.temperature;
    __syncthreads();
  }
  float mx = -INFINITY;
  for (int n = t; n < AR_LOGITS; n += 256) mx = fmaxf(mx, lg[n]);
  mx = block_max_256(mx, sh);

  if (a.top_k > 0) {                                              // :803-809, ties with the k-th value are kept
    const int k = a.top_k < AR_LOGITS ? a.top_k : AR_LOGITS;
    float thr = mx, prev = INFINITY;
    int count = 0;
    while (true) {
      float cur = -INFINITY;
      int cnt = 0;
      for (int n = t; n < AR_LOGITS; n += 256) { const float v = lg[n]; if (v < prev) cur = fmaxf(cur, v); }
      cur = block_max_256(cur, sh);
      for (int n = t; n < AR_LOGITS; n += 256) cnt += (lg[n] == cur);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
      __syncthreads();
      if ((t & 63) == 0) sh_i[t >> 6] = cnt;
      __syncthreads();
      count += sh_i[0] + sh_i[1] + sh_i[2] + sh_i[3];
      thr = cur;
      if (count >= k || cur == -INFINITY) break;
      prev = cur;
    }
    for (int n = t; n < AR_LOGITS; n += 256) if (lg[n] < thr) lg[n] = -INFINITY;
    __syncthreads();
  }
  // softmax numerators (F.softmax: exp(x - max) / sum); probabilities are only needed up to the common 1/sum
  for (int n = t; n < AR_LOGITS; n += 256) lg[n] = expf(lg[n] - mx);
  __syncthreads();

  if (t == 0) {
    const int ngen = a.n_gen[b];
    float u;
    if (a.uniforms) u = a.uniforms[(long)ngen * a.uniforms_stride + b];
    else u = (float)(splitmix64(a.seed ^ ((unsigned long long)ngen << 24) ^ (unsigned long long)b) >> 40) *
             (1.0f / 16777216.0f);
    // inverse CDF over the fp32 running sum in index order (torch.cumsum order): first i with c[i] > u * total
    float total = 0.f;
    int first = -1, last = 0;
    for (int n = 0; n < AR_LOGITS; ++n) {
      const float p = lg[n];
      if (p > 0.f) { total += p; if (first < 0) first = n; last = n; }
    }
    const float thresh = u * total;
    float c = 0.f;
    int tok = last;
    for (int n = first; n <= last; ++n) {
      const float p = lg[n];
      if (p > 0.f) { c += p; if (c > thresh) { tok = n; break; } }
    }
    if (a.force_eos_at >= 0 && ngen >= a.force_eos_at) tok = EOS_ID;
    // stop test: EOS, or (y_len - prompt_len) > 16 * text_len  (models/vallex.py:575-578; y has BOS: 1 + ngen)
    if (tok == EOS_ID || (1 + ngen) > 16 * a.text_len[b] || ngen >= a.gen_stride) {
      a.active[b] = 0;
    } else {
      a.gen[(long)b * a.gen_stride + ngen] = tok;
      a.n_gen[b] = ngen + 1;
      a.cur_tok[b] = tok;
      a.cur_pos[b] += 1;
      a.ctx_len[b] += 1;
    }
  }
}

void launch_dec_sample(const SampleArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(dec_sample_kernel, dim3(a.batch), dim3(256), 0, s, a);
}

// teacher forcing (tests): commit a caller-chosen token exactly like dec_sample would
__global__ void dec_force_token_kernel(const int* __restrict__ tok, int* cur_tok, int* cur_pos, int* ctx_len,
                                       int* n_gen, int* gen, int gen_stride, const int* active, int batch) {
  const int b = threadIdx.x;
  if (b >= batch || !active[b]) return;
  const int ngen = n_gen[b];
  if (ngen >= gen_stride) return;
  gen[(long)b * gen_stride + ngen] = tok[b];
  n_gen[b] = ngen + 1;
  cur_tok[b] = tok[b];
  cur_pos[b] += 1;
  ctx_len[b] += 1;
}

void launch_dec_force_token(const int* tok, int* cur_tok, int* cur_pos, int* ctx_len, int* n_gen, int* gen,
                            int gen_stride, const int* active, int batch, hipStream_t s) {
  hipLaunchKernelGGL(dec_force_token_kernel, dim3(1), dim3(64), 0, s, tok, cur_tok, cur_pos, ctx_len, n_gen, gen,
                     gen_stride, active, batch);
}

}  // namespace vx
