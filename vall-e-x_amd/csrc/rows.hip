// Row-wise HBM-bound kernels: LayerNorm / AdaLN, embedding gathers, argmax, KV scatter, tiny GEMV.
// One 64-lane wave per row, float4 (or float2) accesses, wave-shuffle reductions -- no LDS, no atomics.
#include "vx_common.h"

namespace vx {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm (modules/transformer.py:57-74 -> F.layer_norm) with optional adaptive scale/shift
// (AdaptiveLayerNorm, modules/transformer.py:93-108: weight * LN_affine(x) + bias, the projection of the
// stage embedding is input independent and precomputed at load).  Vocos AdaLayerNorm is the g=b=null case.
// Two-pass statistics from registers (mean, then centred sum of squares) in fp32.
// ---------------------------------------------------------------------------------------------
// planes != null (C = 1024 only): the result is ALSO / INSTEAD (y may be null) written as the f16x2 operand planes of the GEMM that
// consumes it (tile-major [rows/256][C/32][256][32], head and scaled tail; same conversions as split2h_kernel => bit-identical
// planes): the LayerNorm -> QKV / linear1 edges need no fp32 round trip and no split pass.  Adjacent lanes hold adjacent 4-column
// groups; they trade words through one DPP swap so the even lane stores 16 B of the head plane and the odd lane 16 B of the
// tail plane.
template <int C, int VEC>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, int ldx, float* y,
                                                        int ldy, int rows, float eps, const float* __restrict__ g,
                                                        const float* __restrict__ b, const float* __restrict__ aw,
                                                        const float* __restrict__ ab, unsigned short* __restrict__ planes,
                                                        long plane_stride, int* __restrict__ range_flag) {
  constexpr int PER = C / (64 * VEC);
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  float v[PER * VEC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = (i * 64 + lane) * VEC;
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const vec_t t = *reinterpret_cast<const vec_t*>(xr + c);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { v[i * VEC + e] = t[e]; s += t[e]; }
  }
  const float mean = wave_sum(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PER * VEC; ++i) { const float d = v[i] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + eps);
  float* yr = y ? y + (long)row * ldy : nullptr;
  bool bad = false;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = (i * 64 + lane) * VEC;
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    vec_t o, t0, t1;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = (v[i * VEC + e] - mean) * rstd;
    if (g) {
      t0 = *reinterpret_cast<const vec_t*>(g + c);
      t1 = *reinterpret_cast<const vec_t*>(b + c);
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = o[e] * t0[e] + t1[e];
    }
    if (aw) {
      t0 = *reinterpret_cast<const vec_t*>(aw + c);
      t1 = *reinterpret_cast<const vec_t*>(ab + c);
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = t0[e] * o[e] + t1[e];
    }
    if (yr) *reinterpret_cast<vec_t*>(yr + c) = o;
    if constexpr (VEC == 4 && C == 1024) {
      if (planes) {
        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
        typedef unsigned u4_t __attribute__((ext_vector_type(4)));
        unsigned hw[2], tw[2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          h2_t h2, t2;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            _Float16 hk, tk;
            h2_split(o[2 * pr + k], H2_ACT_SCALE, hk, tk, bad);         // as split2h_kernel: bit-identical planes
            h2[k] = hk;
            t2[k] = tk;
          }
          hw[pr] = __builtin_bit_cast(unsigned, h2);
          tw[pr] = __builtin_bit_cast(unsigned, t2);
        }
        const bool odd = lane & 1;
        unsigned r0 = odd ? hw[0] : tw[0], r1 = odd ? hw[1] : tw[1];   // what the neighbour lane needs
        r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)r0, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]: swap with lane ^ 1
        r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)r1, 0xB1, 0xF, 0xF, true);
        // columns 8 (c / 8) .. + 7 of the row: the even lane owns the first four, the odd lane the last four
        const int c8 = c & ~7;
        unsigned short* dst = planes + (((long)(row >> 8) * (C / 32) + (c8 >> 5)) * 256 + (row & 255)) * 32 + (c8 & 31);
        if (!odd) *reinterpret_cast<u4_t*>(dst) = u4_t{hw[0], hw[1], r0, r1};                       // head plane
        else *reinterpret_cast<u4_t*>(dst + plane_stride) = u4_t{r0, r1, tw[0], tw[1]};             // tail plane
      }
    }
  }
  if (bad && range_flag) *range_flag = 1;
}

void launch_layernorm(const float* x, int ldx, float* y, int ldy, int rows, int C, float eps, const float* g,
                      const float* b, const float* aw, const float* ab, hipStream_t s, unsigned short* planes,
                      long plane_stride, int* range_flag) {
  if (rows <= 0) return;
  dim3 grid((rows + 3) / 4), block(256);
  if (C == 1024)
    hipLaunchKernelGGL((layernorm_kernel<1024, 4>), grid, block, 0, s, x, ldx, y, ldy, rows, eps, g, b, aw, ab, planes,
                       plane_stride, range_flag);
  else
    hipLaunchKernelGGL((layernorm_kernel<384, 2>), grid, block, 0, s, x, ldx, y, ldy, rows, eps, g, b, aw, ab, nullptr, 0L,
                       nullptr);
}

// ---------------------------------------------------------------------------------------------
// Embedding rows.  TokenEmbedding (modules/embedding.py:43-47) + language embedding add
// (models/vallex.py:504-505) + SinePositionalEmbedding (modules/embedding.py:93-97:
// x*1.0 + alpha*pe[pos]).  The rounding order of the reference is kept: (tab + lang) first, then
// + (alpha*pe), with no fma contraction.  The PE table is built on the host exactly as the reference does.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_rows_kernel(float* __restrict__ out, const int* __restrict__ dst,
                                                         const float* __restrict__ tabA,
                                                         const int* __restrict__ idA, const float* __restrict__ tabB,
                                                         const int* __restrict__ idB, const float* __restrict__ alpha,
                                                         const float* __restrict__ pe, const int* __restrict__ pos,
                                                         int rows) {
  const int row = blockIdx.x, t = threadIdx.x;     // 256 threads x float4 = 1024
  if (row >= rows) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(tabA + (long)idA[row] * D_MODEL + t * 4);
  if (idB) {
    const int ib = idB[row];
    if (ib >= 0) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(tabB + (long)ib * D_MODEL + t * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(v[e], w[e]);
    }
  }
  const float a = alpha[0];
  const f32x4 p = *reinterpret_cast<const f32x4*>(pe + (long)pos[row] * D_MODEL + t * 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(v[e], __fmul_rn(a, p[e]));
  *reinterpret_cast<f32x4*>(out + (long)(dst ? dst[row] : row) * D_MODEL + t * 4) = v;
}

void launch_embed_rows(float* out, const int* dst, const float* tabA, const int* idA, const float* tabB,
                       const int* idB, const float* alpha, const float* pe, const int* pos, int rows,
                       hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(embed_rows_kernel, dim3(rows), dim3(256), 0, s, out, dst, tabA, idA, tabB, idB, alpha, pe, pos,
                     rows);
}

// NAR acoustic embedding sum (models/vallex.py:605-607,659-662): emb_0[c0] then += emb_j[c_j], j ascending.
__global__ __launch_bounds__(256) void nar_yemb_init_kernel(float* __restrict__ yemb, const float* const* tabs,
                                                            const int* __restrict__ codes,
                                                            const int* __restrict__ nj, int rows) {
  const int row = blockIdx.x, t = threadIdx.x;
  if (row >= rows) return;
  const int n = nj[row];
  f32x4 v = *reinterpret_cast<const f32x4*>(tabs[0] + (long)codes[row * N_Q] * D_MODEL + t * 4);
  for (int j = 1; j < n; ++j) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(tabs[j] + (long)codes[row * N_Q + j] * D_MODEL + t * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(v[e], w[e]);
  }
  *reinterpret_cast<f32x4*>(yemb + (long)row * D_MODEL + t * 4) = v;
}

void launch_nar_yemb_init(float* yemb, const float* const* tabs, const int* codes, const int* nj, int rows,
                          hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(nar_yemb_init_kernel, dim3(rows), dim3(256), 0, s, yemb, tabs, codes, nj, rows);
}

// nar_audio_position + concat with text (models/vallex.py:665-667): out[dst[r]] = yemb[r] + alpha*pe[pos[r]]
__global__ __launch_bounds__(256) void add_pe_scatter_kernel(float* __restrict__ out, const int* __restrict__ dst,
                                                             const float* __restrict__ yemb,
                                                             const float* __restrict__ alpha,
                                                             const float* __restrict__ pe,
                                                             const int* __restrict__ pos, int rows) {
  const int row = blockIdx.x, t = threadIdx.x;
  if (row >= rows) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(yemb + (long)row * D_MODEL + t * 4);
  const f32x4 p = *reinterpret_cast<const f32x4*>(pe + (long)pos[row] * D_MODEL + t * 4);
  const float a = alpha[0];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(v[e], __fmul_rn(a, p[e]));
  *reinterpret_cast<f32x4*>(out + (long)dst[row] * D_MODEL + t * 4) = v;
}

void launch_add_pe_scatter(float* out, const int* dst, const float* yemb, const float* alpha, const float* pe,
                           const int* pos, int rows, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(add_pe_scatter_kernel, dim3(rows), dim3(256), 0, s, out, dst, yemb, alpha, pe, pos, rows);
}

// y_emb[:, prefix_len:] += embedding_layer(samples)  (models/vallex.py:682-683)
__global__ __launch_bounds__(256) void embed_accum_kernel(float* __restrict__ yemb, const int* __restrict__ rowidx,
                                                          const float* __restrict__ tab,
                                                          const int* __restrict__ tok, int n) {
  const int i = blockIdx.x, t = threadIdx.x;
  if (i >= n) return;
  float* y = yemb + (long)rowidx[i] * D_MODEL + t * 4;
  f32x4 v = *reinterpret_cast<f32x4*>(y);
  const f32x4 w = *reinterpret_cast<const f32x4*>(tab + (long)tok[i] * D_MODEL + t * 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(v[e], w[e]);
  *reinterpret_cast<f32x4*>(y) = v;
}

void launch_embed_accum(float* yemb, const int* rowidx, const float* tab, const int* tok, int n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(embed_accum_kernel, dim3(n), dim3(256), 0, s, yemb, rowidx, tab, tok, n);
}

// torch.argmax(logits, dim=-1) (models/vallex.py:679): first index of the maximum.
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int ldx, int rows, int N,
                                                          int* __restrict__ idx) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < N; c += 64) {
    const float v = xr[c];
    if (v > best) { best = v; bi = c; }      // ascending c per lane: first max kept
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  // a row without any comparable value (all NaN: an f16x2 operand left the fp16 range; the phase is re-run in fp32) must still
  // yield a valid table index -- the ids of such a pass are thrown away, a wild gather would fault
  if (lane == 0) idx[row] = bi == 0x7fffffff ? 0 : bi;
}

void launch_argmax_rows(const float* x, int ldx, int rows, int N, int* idx, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, rows, N, idx);
}

// Prefill K/V into the decode cache: present = (k, v) of modules/activation.py:148-157, laid out
// [(b*H + h)][t][64] so a head's keys are one contiguous stream for the decode kernel.
__global__ __launch_bounds__(256) void kv_scatter_kernel(const float* __restrict__ qkv, const int* __restrict__ row_b,
                                                         const int* __restrict__ row_t, int M, float* __restrict__ kc,
                                                         float* __restrict__ vc, int Tmax) {
  const int row = blockIdx.x, t = threadIdx.x;    // 256 threads: float4 index over 1024 = (h, d4)
  if (row >= M) return;
  const int b = row_b[row], tt = row_t[row], h = t >> 4, d = (t & 15) * 4;
  const long dst = ((long)(b * N_HEAD + h) * Tmax + tt) * D_HEAD + d;
  const float* src = qkv + (long)row * (3 * D_MODEL) + t * 4;
  *reinterpret_cast<f32x4*>(kc + dst) = *reinterpret_cast<const f32x4*>(src + D_MODEL);
  *reinterpret_cast<f32x4*>(vc + dst) = *reinterpret_cast<const f32x4*>(src + 2 * D_MODEL);
}

void launch_kv_scatter(const float* qkv, const int* row_b, const int* row_t, int M, float* kc, float* vc, int Tmax,
                       hipStream_t s) {
  if (M <= 0) return;
  hipLaunchKernelGGL(kv_scatter_kernel, dim3(M), dim3(256), 0, s, qkv, row_b, row_t, M, kc, vc, Tmax);
}

// out[r] = W[r].e + b[r]: AdaLN project_layer(stage_embedding) (modules/transformer.py:96-100), run once per
// (stage, norm) at load time -- 7 x 25 GEMVs of 2048x1024.
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ W, const float* __restrict__ e,
                                                   const float* __restrict__ b, float* __restrict__ out, int N,
                                                   int K) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= N) return;
  float s = 0.f;
  for (int c = lane * 4; c < K; c += 256) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(W + (long)row * K + c);
    const f32x4 x = *reinterpret_cast<const f32x4*>(e + c);
    s += w[0] * x[0] + w[1] * x[1] + w[2] * x[2] + w[3] * x[3];
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s + (b ? b[row] : 0.f);
}

void launch_gemv(const float* W, const float* e, const float* b, float* out, int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(gemv_kernel, dim3((N + 3) / 4), dim3(256), 0, s, W, e, b, out, N, K);
}

}  // namespace vx
