// Micro-benchmark (kernel-development aid): does other work hide under a SATURATED matrix pipe?  Each wave loops over
// [4 independent v_mfma_f32_32x32x16_bf16 | N extra instructions behind each]; 512 workgroups x 4 waves = 2 waves per
// SIMD, so the MFMAs alone need 2 x 4 x 32 = 256 cycles per iteration.  Extra = v_fma_f32 (independent registers) or
// conflict-free ds_read_b128 whose results are only waited for at the top of the next iteration.  Reports s_memtime
// ticks per iteration as seen by wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 3000;

template <int N, int OP, bool MFMA>   // OP 0: v_fma_f32, 1: ds_read_b128
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
  for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<float*>(lds)[i] = 0.001f * i;
  __syncthreads();
  const unsigned laddr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
  f32x16 c[4];
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) c[j][e] = 0.f;
  float v[16];
  for (int e = 0; e < 16; ++e) v[e] = threadIdx.x * 0.01f + e;
  f32x4 ld[8];
  for (int e = 0; e < 8; ++e) ld[e] = f32x4{0, 0, 0, 0};
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (OP == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ld[0]), "+v"(ld[1]), "+v"(ld[2]), "+v"(ld[3]), "+v"(ld[4]), "+v"(ld[5]), "+v"(ld[6]), "+v"(ld[7]));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MFMA) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[j], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < N; ++n) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[(j * N + n) & 15]));
        if (OP == 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[(j * N + n) & 7]) : "v"(laddr), "n"(((j * N + n) & 7) * 1024));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float acc = 0;
  for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc += c[j][e];
  for (int e = 0; e < 16; ++e) acc += v[e];
  for (int e = 0; e < 8; ++e) acc += ld[e][0];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = __builtin_readcyclecounter() - c0;
    clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

template <int N, int OP, bool MFMA>
void run(const char* name, float* out, unsigned long long* clk) {
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<N, OP, MFMA>), dim3(512), dim3(256), 0, 0, out, ITER, clk);
  (void)hipDeviceSynchronize();
  unsigned long long h[2];
  (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("%-14s per MFMA: %d, mfma=%d : %7.1f ticks / iteration of 4 MFMAs (MFMA-only floor at 2 waves/SIMD: 256), clock %.0f MHz\n", name, N,
         (int)MFMA, (double)h[0] / ITER, (double)h[0] / ((double)h[1] / 100.0));
}

int main() {
  float* out; unsigned long long* clk;
  (void)hipMalloc((void**)&out, 512 * 256 * 4);
  (void)hipMalloc((void**)&clk, 16);
  run<0, 0, true>("nothing", out, clk);
  run<0, 0, true>("nothing", out, clk);
  run<2, 0, true>("v_fma_f32", out, clk);
  run<4, 0, true>("v_fma_f32", out, clk);
  run<8, 0, true>("v_fma_f32", out, clk);
  run<8, 0, false>("v_fma_f32", out, clk);
  run<1, 1, true>("ds_read_b128", out, clk);
  run<2, 1, true>("ds_read_b128", out, clk);
  run<2, 1, false>("ds_read_b128", out, clk);
  return 0;
}
