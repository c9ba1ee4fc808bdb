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
// This file holds the HBM-bound glue: im2col gathers with the ELU fused in, the LSTM cell, the last 32->1 conv.
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

}  // namespace vx
