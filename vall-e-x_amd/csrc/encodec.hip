// EnCodec 24 kHz SEANet decoder -- the "1-D ConvTranspose stack" of the north star; replaces the reference's legacy
// vocoder path AudioTokenizer.decode -> codec.decode(frames) (data/tokenizer.py:95-96; arithmetic in the pip package
// `encodec`, restated in oracle/encodec_oracle.py and pinned to the transformers port).
//
// Everything dense runs on gemm_f32.hip (fp32 MFMA) through im2col views in channels-last layout:
//   Conv1d(k, causal, reflect pad)      : rows [x[t-k+1] .. x[t]]            -> GEMM  (K = k*Cin)
//   ConvTranspose1d(k = 2r, stride r)   : rows [x[t] | x[t-1]]               -> GEMM  with N = r*Cout, i.e. all r output
//                                         phases of a frame at once; reading the [T][r*Cout] result as [T*r][Cout] IS
//                                         the upsampled sequence (the right-trim of k - r samples is implicit)
//   LSTM                                : input projections for all t as one GEMM; the recurrence h_{t-1}.W_hh^T on the
//                                         skinny MFMA GEMM of decode.hip (batch = MFMA columns) + a fused cell kernel
// This file holds the HBM-bound glue: im2col gathers with the ELU fused in, the LSTM cell, the last 32->1 conv, and -- for the
// ENCODER (prompt enrolment, second half of the file) -- the first 1->32 conv, the padded copy that turns a strided conv into a
// GEMM over overlapping rows, and the residual-VQ select.
#include <algorithm>

#include "vx_common.h"

namespace vx {

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }

// mode 0 (Conv1d, causal, reflect): out[r][tap*C + c] = f(x[seq, t + tap - (k-1)]), negative index j -> x[-j] (or 0 if
//         -j >= T: EncodecConv1d._pad1d zero-extends short inputs before reflecting)
// mode 1 (ConvTranspose1d k=2r):    out[r][tap*C + c] = f(x[seq, t - tap]) for tap in {0,1}, x[-1] = 0
// rows of sequence b: [seq_off[b]*R, (seq_off[b]+seq_len[b])*R); one block row per (row, sequence); float4 over (tap, c)
__global__ __launch_bounds__(256) void im2col_seq_kernel(const float* __restrict__ x, int C, int k, int mode, int elu,
                                                         const int* __restrict__ seq_off,
                                                         const int* __restrict__ seq_len, int R,
                                                         float* __restrict__ out, int ldo) {
  const int b = blockIdx.y;
  const long T = (long)seq_len[b] * R, base = (long)seq_off[b] * R;
  const int per_row = k * C / 4;                                 // float4 per output row
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < T * per_row; i += (long)gridDim.x * 256) {
    const long t = i / per_row;
    const int f = (int)(i - t * per_row), tap = (f * 4) / C, c = f * 4 - tap * C;
    long j = mode == 0 ? t + tap - (k - 1) : t - tap;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    bool ok = j >= 0;
    if (mode == 0 && j < 0) { j = -j; ok = j < T; }
    if (ok) {
      v = *reinterpret_cast<const f32x4*>(x + (base + j) * C + c);
      if (elu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = elu1(v[e]);
      }
    }
    *reinterpret_cast<f32x4*>(out + (base + t) * ldo + tap * C + c) = v;
  }
}

void launch_im2col_seq(const float* x, int C, int k, int mode, int elu, const int* seq_off, const int* seq_len, int R,
                       float* out, int ldo, int batch, long max_rows, hipStream_t s) {
  if (batch <= 0 || max_rows <= 0) return;
  const long work = max_rows * (k * C / 4);
  const int gx = (int)std::min<long>((work + 255) / 256, 4096);
  hipLaunchKernelGGL(im2col_seq_kernel, dim3(gx, batch), dim3(256), 0, s, x, C, k, mode, elu, seq_off, seq_len, R, out,
                     ldo);
}

// ---------------------------------------------------------------------------------------------------------------
// LSTM cell (torch.nn.LSTM gate order i, f, g, o).  gates = xg[row] (input projection + both biases, one big GEMM for
// all t) + sum_ks part[ks][b] (h_{t-1}.W_hh^T from the skinny GEMM).  One block per sequence, one thread per hidden unit.
// Writes h into the packed-x image that feeds the next step's skinny GEMM, and y[row] = h (+ skip[row] on the last layer:
// EncodecLSTM returns lstm(x) + x).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid1(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(512) void lstm_cell_kernel(const float* __restrict__ part, int splitk,
                                                        const float* __restrict__ xg,
                                                        const int* __restrict__ seq_off,
                                                        const int* __restrict__ seq_len, int t,
                                                        float* __restrict__ cstate, float* __restrict__ hp,
                                                        float* __restrict__ y, const float* __restrict__ skip) {
  constexpr int HD = 512, G = 4 * HD;
  const int b = blockIdx.x, j = threadIdx.x;
  if (t >= seq_len[b]) return;
  const long row = (long)seq_off[b] + t;
  float g[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v = part[(long)b * G + q * HD + j];
    for (int ks = 1; ks < splitk; ++ks) v += part[((long)ks * MB + b) * G + q * HD + j];
    g[q] = xg[row * G + q * HD + j] + v;
  }
  const float c = sigmoid1(g[1]) * cstate[b * HD + j] + sigmoid1(g[0]) * tanhf(g[2]);
  const float h = sigmoid1(g[3]) * tanhf(c);
  cstate[b * HD + j] = c;
  hp[(((long)(j >> 3) * 64) + b + 32 * ((j >> 2) & 1)) * 4 + (j & 3)] = h;
  y[row * HD + j] = skip ? h + skip[row * HD + j] : h;
}

void launch_lstm_cell(const float* part, int splitk, const float* xg, const int* seq_off, const int* seq_len, int t,
                      float* cstate, float* hp, float* y, const float* skip, int batch, hipStream_t s) {
  hipLaunchKernelGGL(lstm_cell_kernel, dim3(batch), dim3(512), 0, s, part, splitk, xg, seq_off, seq_len, t, cstate, hp,
                     y, skip);
}

// last layer: ELU -> Conv1d(32, 1, k=7, causal reflect).  One thread per output sample; weights are wave-uniform.
__global__ __launch_bounds__(256) void final_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias,
                                                         const int* __restrict__ seq_off,
                                                         const int* __restrict__ seq_len, int R,
                                                         float* __restrict__ audio, long audio_stride) {
  constexpr int C = 32, K = 7;
  const int b = blockIdx.y;
  const long T = (long)seq_len[b] * R, base = (long)seq_off[b] * R;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  float acc = bias[0];
#pragma unroll
  for (int tap = 0; tap < K; ++tap) {
    long j = t + tap - (K - 1);
    bool ok = true;
    if (j < 0) { j = -j; ok = j < T; }
    if (ok) {
      const float* xr = x + (base + j) * C;
#pragma unroll
      for (int c4 = 0; c4 < C; c4 += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += w[(c4 + e) * K + tap] * elu1(v[e]);     // weight (1, 32, 7): [c][tap]
      }
    }
  }
  audio[(long)b * audio_stride + t] = acc;
}

void launch_final_conv(const float* x, const float* w, const float* bias, const int* seq_off, const int* seq_len, int R,
                       float* audio, long audio_stride, int batch, long max_rows, hipStream_t s) {
  if (batch <= 0 || max_rows <= 0) return;
  hipLaunchKernelGGL(final_conv_kernel, dim3((unsigned)((max_rows + 255) / 256), batch), dim3(256), 0, s, x, w, bias,
                     seq_off, seq_len, R, audio, audio_stride);
}


// ===============================================================================================================
// Encoder side -- prompt enrolment: tokenize_audio -> AudioTokenizer.encode -> codec.encode(wav) (data/tokenizer.py:92-111,
// utils/prompt_making.py:57-84).  SEANet encoder: Conv1d(1,32,k7) -> 4 x [ResnetBlock, ELU,
// Conv1d(C, 2C, k = 2r, stride r)], r = 2,4,5,8 -> LSTM + skip -> ELU -> Conv1d(512,128,k7), then 8 residual VQ steps.
// The dense parts reuse gemm_f32 / the skinny GEMM + lstm_cell_kernel / im2col_seq_kernel above; a strided conv with
// k = 2r needs no im2col at all: in channels-last layout the window of output frame t' is the CONTIGUOUS run of rows
// [(t'-1) r, (t'+1) r) of the (left-padded) input, i.e. a GEMM whose A rows overlap (lda = r C, K = 2 r C).
// ===============================================================================================================

// first layer, one thread per output sample: out[t][c] = bias[c] + sum_tap w[c][tap] * wav[reflect(t + tap - 6)]
__global__ __launch_bounds__(256) void enc_first_conv_kernel(const float* __restrict__ wav, long L,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ out) {
  constexpr int C = 32, K = 7;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= L) return;
  float xs[K];
#pragma unroll
  for (int tap = 0; tap < K; ++tap) {
    long j = t + tap - (K - 1);
    bool ok = true;
    if (j < 0) { j = -j; ok = j < L; }                            // reflect; inputs shorter than the pad are zero-extended
    xs[tap] = ok ? wav[j] : 0.f;
  }
#pragma unroll
  for (int c4 = 0; c4 < C; c4 += 4) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float acc = bias[c4 + e];
#pragma unroll
      for (int tap = 0; tap < K; ++tap) acc += w[(c4 + e) * K + tap] * xs[tap];     // weight (32, 1, 7)
      v[e] = acc;
    }
    *reinterpret_cast<f32x4*>(out + t * C + c4) = v;
  }
}

void launch_enc_first_conv(const float* wav, long L, const float* w, const float* bias, float* out, hipStream_t s) {
  if (L <= 0) return;
  hipLaunchKernelGGL(enc_first_conv_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, s, wav, L, w, bias, out);
}

// ELU + the padding of a causal strided conv (EncodecConv1d: left pad k - stride = r, right pad completes the last frame,
// both reflect): out[j][c] = elu(x'[reflect(j - left)][c]) for j < rows, x' = x zero-extended to Le >= L rows (Le > L only
// for inputs not longer than the pad, EncodecConv1d._pad1d), reflect(u) = u < 0 ? -u : (u >= Le ? 2 (Le - 1) - u : u).
__global__ __launch_bounds__(256) void enc_pad_elu_kernel(const float* __restrict__ x, long L, long Le, int C, int left,
                                                          long rows, float* __restrict__ out) {
  const int per_row = C / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * per_row; i += (long)gridDim.x * 256) {
    const long j = i / per_row;
    const int c4 = (int)(i - j * per_row) * 4;
    long u = j - left;
    if (u < 0) u = -u;
    else if (u >= Le) u = 2 * (Le - 1) - u;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (u >= 0 && u < L) {
      v = *reinterpret_cast<const f32x4*>(x + u * C + c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = elu1(v[e]);
    }
    *reinterpret_cast<f32x4*>(out + j * C + c4) = v;
  }
}

void launch_enc_pad_elu(const float* x, long L, long Le, int C, int left, long rows, float* out, hipStream_t s) {
  if (rows <= 0) return;
  const long work = rows * (C / 4);
  hipLaunchKernelGGL(enc_pad_elu_kernel, dim3((unsigned)std::min<long>((work + 255) / 256, 4096)), dim3(256), 0, s, x, L,
                     Le, C, left, rows, out);
}

// One residual-VQ step (EncodecEuclideanCodebook.quantize + EncodecResidualVectorQuantizer.encode's residual update):
//   code = argmax_c -( |r|^2 - 2 s_c + |e_c|^2 ),  s = r . E^T from the GEMM;  r -= E[code].
// One workgroup per frame; ties -> lowest index.
__global__ __launch_bounds__(256) void rvq_select_kernel(float* __restrict__ resid, const float* __restrict__ scores,
                                                         const float* __restrict__ e2,
                                                         const float* __restrict__ codebook,
                                                         long long* __restrict__ codes, int q) {
  constexpr int D = 128, NC = 1024;
  __shared__ float sh_v[256];
  __shared__ int sh_i[256];
  const long row = blockIdx.x;
  const int tid = threadIdx.x;
  float* r = resid + row * D;
  float sq = tid < D ? r[tid] * r[tid] : 0.f;
  sh_v[tid] = sq;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) sh_v[tid] += sh_v[tid + s];
    __syncthreads();
  }
  const float a = sh_v[0];
  __syncthreads();
  float best = -3.0e38f;
  int bi = NC;
#pragma unroll
  for (int i = 0; i < NC / 256; ++i) {
    const int cidx = tid + 256 * i;
    const float d = -((a - 2.0f * scores[row * NC + cidx]) + e2[cidx]);
    if (d > best) { best = d; bi = cidx; }                         // candidates visited in ascending index order per thread
  }
  sh_v[tid] = best;
  sh_i[tid] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      const float ov = sh_v[tid + s];
      const int oi = sh_i[tid + s];
      if (ov > sh_v[tid] || (ov == sh_v[tid] && oi < sh_i[tid])) { sh_v[tid] = ov; sh_i[tid] = oi; }
    }
    __syncthreads();
  }
  const int code = sh_i[0];
  if (tid == 0) codes[row * 8 + q] = code;
  if (tid < D) r[tid] -= codebook[(long)code * D + tid];
}

void launch_rvq_select(float* resid, const float* scores, const float* e2, const float* codebook, long long* codes, int q,
                       long rows, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(rvq_select_kernel, dim3((unsigned)rows), dim3(256), 0, s, resid, scores, e2, codebook, codes, q);
}

}  // namespace vx
