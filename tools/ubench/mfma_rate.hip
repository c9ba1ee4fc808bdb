// Micro-benchmark (kernel-development aid): issue rate of v_mfma_f32_32x32x16_bf16 with 1, 2 and 4 independent accumulator
// chains per wave, 512 workgroups x 4 waves (2 waves per SIMD on 256 CUs).  Reports s_memtime ticks per MFMA seen by
// wave 0, the shader clock (s_memtime / s_memrealtime) and the aggregate rate from wall time.
// RND = 1 feeds pseudo-random, per-lane, per-instruction changing operands (4 register sets): constant operands toggle few
// bits and draw little power, real GEMM data does not -- if the sustained clock (and with it the achievable MFMA rate) drops
// under random data, that is the ceiling the bf16x3 GEMMs run into.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ITER = 4000;

template <int CH, int RND>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
  bf16x8 ra[4], rb[4];                                             // random operand sets in [-1, 1)
  unsigned st = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 8; ++e) {
      st = st * 1664525u + 1013904223u;
      ra[j][e] = (__bf16)((float)(int)(st >> 8) * (1.0f / 8388608.0f) - 1.0f);
      st = st * 1664525u + 1013904223u;
      rb[j][e] = (__bf16)((float)(int)(st >> 8) * (1.0f / 8388608.0f) - 1.0f);
    }
  f32x16 c[4];
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) c[j][e] = 0.f;
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < CH; ++j)
      c[j] = RND ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[(it + j) & 3], rb[(it + 2 * j + 1) & 3], c[j], 0, 0, 0)
                 : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[j], 0, 0, 0);
  }
  float acc = 0;
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc += c[j][e];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = __builtin_readcyclecounter() - c0;
    clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

template <int CH, int RND>
void run(float* out, unsigned long long* clk, int blocks) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CH, RND>), dim3(blocks), dim3(256), 0, 0, out, ITER, clk);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<CH, RND>), dim3(blocks), dim3(256), 0, 0, out, ITER, clk);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double mfmas = (double)blocks * 4 * ITER * CH;
  printf("%s chains=%d blocks=%4d : wave 0 %.1f ticks/MFMA, shader clock %.0f MHz | wall %.1f us => %.1f TFLOP/s bf16 dense aggregate (%.2f MFMA/cycle/SIMD-slot at 1024 SIMDs)\n",
         RND ? "random-data" : "const-data ", CH, blocks, (double)h[0] / (ITER * CH), (double)h[0] / ((double)h[1] / 100.0), ms * 1e3,
         mfmas * 32768.0 / (ms * 1e-3) / 1e12, mfmas / 1024.0 / (ms * 1e-3 * ((double)h[0] / ((double)h[1] / 100.0)) * 1e6));
}

int main() {
  float* out; unsigned long long* clk;
  (void)hipMalloc((void**)&out, 16384 * 256 * 4);
  (void)hipMalloc((void**)&clk, 16);
  run<1, 0>(out, clk, 512);
  for (int blocks : {256, 512, 2048}) {
    run<1, 0>(out, clk, blocks);
    run<2, 0>(out, clk, blocks);
    run<4, 0>(out, clk, blocks);
    run<4, 1>(out, clk, blocks);
  }
  // long runs (~30 ms each): does the clock hold once the power management reacts?
  for (int rep = 0; rep < 3; ++rep) {
    run<4, 0>(out, clk, 16384);
    run<4, 1>(out, clk, 16384);
  }
  (void)hipDeviceSynchronize();
  return 0;
}
