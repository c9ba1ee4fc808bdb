// Vocos head (pip `vocos`, charactr/vocos-encodec-24khz; reference call sites utils/generation.py:148-150) --
// the HBM-bound pieces.  Activations are channels-last packed frames [sum_b T_b][C]; the dense parts (embed
// conv as im2col GEMM, pwconv1/2, head, inverse DFT) run on gemm_f32.hip.
#include "vx_common.h"

namespace vx {

constexpr int VC_IN = 128, VC_BINS = 641, VC_NFFT = 1280, VC_HOP = 320, VC_PAD = 480;

// codes_to_features: feat[r] = sum_q codebook[q*1024 + code[r][q]]  (q ascending, like .sum(dim=0))
__global__ __launch_bounds__(256) void codebook_sum_kernel(const int* __restrict__ codes,
                                                           const float* __restrict__ cb, float* __restrict__ feat,
                                                           int rows) {
  const int r = blockIdx.x * 8 + (threadIdx.x >> 5), c = (threadIdx.x & 31) * 4;
  if (r >= rows) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(cb + (long)codes[r * N_Q] * VC_IN + c);
#pragma unroll
  for (int q = 1; q < N_Q; ++q) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(cb + ((long)q * 1024 + codes[r * N_Q + q]) * VC_IN + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(v[e], w[e]);
  }
  *reinterpret_cast<f32x4*>(feat + (long)r * VC_IN + c) = v;
}

void launch_codebook_sum(const int* codes, const float* codebook, float* feat, int rows, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(codebook_sum_kernel, dim3((rows + 7) / 8), dim3(256), 0, s, codes, codebook, feat, rows);
}

// im2col for Conv1d(128, 384, k=7, pad=3): out[r][tap*128 + c] = x[r + tap - 3][c] inside the row's own sequence.
__global__ __launch_bounds__(256) void im2col7_kernel(const float* __restrict__ x, const int* __restrict__ row_t,
                                                      const int* __restrict__ row_len, float* __restrict__ out,
                                                      int rows) {
  const int r = blockIdx.x, t = threadIdx.x;
  if (r >= rows || t >= 7 * 32) return;          // 224 float4 per row
  const int tap = t >> 5, c = (t & 31) * 4;
  const int tt = row_t[r] + tap - 3;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (tt >= 0 && tt < row_len[r]) v = *reinterpret_cast<const f32x4*>(x + (long)(r + tap - 3) * VC_IN + c);
  *reinterpret_cast<f32x4*>(out + (long)r * (7 * VC_IN) + tap * VC_IN + c) = v;
}

void launch_im2col7(const float* x, int C, const int* row_t, const int* row_len, float* out, int rows,
                    hipStream_t s) {
  (void)C;
  if (rows <= 0) return;
  hipLaunchKernelGGL(im2col7_kernel, dim3(rows), dim3(256), 0, s, x, row_t, row_len, out, rows);
}

// depthwise Conv1d(C, C, k=7, pad=3, groups=C): out[r][c] = bias[c] + sum_tap w[c][tap] * x[r+tap-3][c]
__global__ __launch_bounds__(128) void dwconv7_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias,
                                                      const int* __restrict__ row_t,
                                                      const int* __restrict__ row_len, float* __restrict__ out,
                                                      int rows, int C) {
  const int r = blockIdx.x, c = threadIdx.x * 4;
  if (r >= rows || c >= C) return;
  const int t0 = row_t[r], len = row_len[r];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tap = 0; tap < 7; ++tap) {
    const int tt = t0 + tap - 3;
    if (tt >= 0 && tt < len) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (long)(r + tap - 3) * C + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += w[(c + e) * 7 + tap] * xv[e];
    }
  }
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] += bv[e];
  *reinterpret_cast<f32x4*>(out + (long)r * C + c) = acc;
}

void launch_dwconv7(const float* x, const float* w, const float* bias, const int* row_t, const int* row_len,
                    float* out, int rows, int C, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(dwconv7_kernel, dim3(rows), dim3(128), 0, s, x, w, bias, row_t, row_len, out, rows, C);
}

// ISTFTHead front: mag = clip(exp(o[:641]), max=100); S = mag * (cos p + i sin p)  -> [re(641) | im(641) | 0-pad]
__global__ __launch_bounds__(256) void istft_prep_kernel(const float* __restrict__ o, int ldo,
                                                         float* __restrict__ reim, int ldr, int rows) {
  const int r = blockIdx.x;
  if (r >= rows) return;
  for (int k = threadIdx.x; k < VC_BINS; k += 256) {
    float mag = expf(o[(long)r * ldo + k]);
    mag = fminf(mag, 100.0f);
    const float p = o[(long)r * ldo + VC_BINS + k];
    reim[(long)r * ldr + k] = mag * cosf(p);
    reim[(long)r * ldr + VC_BINS + k] = mag * sinf(p);
  }
  for (int k = 2 * VC_BINS + threadIdx.x; k < ldr; k += 256) reim[(long)r * ldr + k] = 0.f;
}

void launch_istft_prep(const float* o, int ldo, float* reim, int ldr, int rows, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(istft_prep_kernel, dim3(rows), dim3(256), 0, s, o, ldo, reim, ldr, rows);
}

// ISTFT tail, padding="same": overlap-add of the windowed frames (kernel 1280, stride 320), trim 480 each side,
// divide by the overlap-added hann^2 envelope.  frames already carry the window (folded into the DFT basis).
// win2 = hann(1280)^2.  One thread per output sample.
__global__ __launch_bounds__(256) void overlap_add_kernel(const float* __restrict__ frames, int ldf,
                                                          const int* __restrict__ seq_off,
                                                          const int* __restrict__ seq_len,
                                                          const float* __restrict__ win2,
                                                          float* __restrict__ audio, long audio_stride) {
  const int b = blockIdx.y;
  const int T = seq_len[b];
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= T * VC_HOP) return;
  const int i = s + VC_PAD;
  int f0 = (i - (VC_NFFT - 1) + VC_HOP - 1) / VC_HOP;
  if (i - (VC_NFFT - 1) < 0) f0 = 0;
  int f1 = i / VC_HOP;
  if (f1 > T - 1) f1 = T - 1;
  float acc = 0.f, env = 0.f;
  const long base = seq_off[b];
  for (int f = f0; f <= f1; ++f) {
    const int n = i - f * VC_HOP;
    acc += frames[(base + f) * (long)ldf + n];
    env += win2[n];
  }
  // audio_stride 0: the sequences' audio is packed like their frames (sample offset seq_off * hop)
  audio[(audio_stride ? (long)b * audio_stride : base * VC_HOP) + s] = acc / env;
}

void launch_overlap_add(const float* frames, int ldf, const int* seq_off, const int* seq_len, const float* win2,
                        float* audio, long audio_stride, int batch, int max_T, hipStream_t s) {
  if (batch <= 0 || max_T <= 0) return;
  hipLaunchKernelGGL(overlap_add_kernel, dim3((max_T * VC_HOP + 255) / 256, batch), dim3(256), 0, s, frames, ldf,
                     seq_off, seq_len, win2, audio, audio_stride);
}

}  // namespace vx
