// Micro-benchmark (kernel-development aid): how do the matrix pipe and the VALU of one gfx950 SIMD overlap?
// Each wave runs ITER iterations of [1 x v_mfma_f32_32x32x16_bf16 | N x VALU op]; reports cycles per iteration (wall
// time x nominal 2.4 GHz) for 1 and 2 waves per SIMD, several N and several VALU ops.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int ITER = 4000;

// OP: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_cvt_pk_bf16_f32, 3 v_exp_f32, 4 v_cndmask (cmp+cndmask pair), 5 none,
//     6 ds_read_b128 (conflict-free, results unused but waited for every iteration), 7 ds_read_b128 feeding the MFMA's A operand
template <int N, int OP, int CHAINS, bool MFMA>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
  const unsigned long long c_start = __builtin_readcyclecounter(), r_start = __builtin_amdgcn_s_memrealtime();
  __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
  for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<float*>(lds)[i] = 0.001f * i;
  __syncthreads();
  const unsigned laddr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 ld[4];
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
  f32x16 c0 = {0}, c1 = {0};
  float v[16];
  for (int e = 0; e < 16; ++e) v[e] = threadIdx.x * 0.01f + e;
  f32x2 pv[8];
  for (int e = 0; e < 8; ++e) pv[e] = f32x2{v[e], v[e + 8]};
  unsigned w[16];
  for (int e = 0; e < 16; ++e) w[e] = threadIdx.x + e;
  for (int it = 0; it < iters; ++it) {
    if (MFMA) {
      if (CHAINS == 1 || !(it & 1)) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      else c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[n & 15]));
      if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(pv[n & 7]));
      if (OP == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[n & 15]) : "v"(v[n & 15]), "v"(v[(n + 1) & 15]));
      if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[n & 15]));
      if (OP == 6) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[n & 3]) : "v"(laddr), "n"((n & 7) * 1024));
      if (OP == 7) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a) : "v"(laddr), "n"((n & 7) * 1024));
      if (OP == 4) asm volatile("v_cmp_lt_i32 vcc, %1, %2\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(w[n & 15]) : "v"(w[(n + 1) & 15]), "v"(w[(n + 2) & 15]) : "vcc");
    }
    if (OP == 6) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ld[0]), "+v"(ld[1]), "+v"(ld[2]), "+v"(ld[3]));
    if (OP == 7) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a));
    __builtin_amdgcn_sched_barrier(0);
  }
  float acc = 0;
  for (int e = 0; e < 16; ++e) acc += c0[e] + c1[e] + v[e] + (float)w[e];
  for (int e = 0; e < 8; ++e) acc += pv[e][0] + pv[e][1];
  for (int e = 0; e < 4; ++e) acc += ld[e][0] + (float)a[e];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) {   // shader-clock cycles and 100 MHz reference ticks spent in this wave
    clk[0] = __builtin_readcyclecounter() - c_start;
    clk[1] = __builtin_amdgcn_s_memrealtime() - r_start;
  }
}

template <int N, int OP, int CHAINS, bool MFMA>
void run(const char* name, float* out) {
  static unsigned long long* clk = nullptr;
  if (!clk) hipMalloc((void**)&clk, 16);
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * wps;                                // 4 waves per block -> wps waves per SIMD on 256 CUs
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<N, OP, CHAINS, MFMA>), dim3(blocks), dim3(256), 0, 0, out, ITER, clk);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<N, OP, CHAINS, MFMA>), dim3(blocks), dim3(256), 0, 0, out, ITER, clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%-28s N=%2d mfma=%d waves/SIMD=%d : %7.1f nominal cycles/iter (2.4 GHz x wall) | wave 0: %.1f s_memtime ticks/iter, %.3f us/iter by s_memrealtime => s_memtime rate %.0f MHz\n",
           name, N, (int)MFMA, wps, ms * 1e-3 * 2.4e9 / ITER, (double)h[0] / ITER, (double)h[1] / 100.0 / ITER,
           h[1] ? (double)h[0] / ((double)h[1] / 100.0) : 0.0);
  }
}

int main() {
  float* out;
  hipMalloc((void**)&out, 512 * 256 * 4);
  run<0, 5, 1, true>("warmup", out);
  run<0, 5, 1, true>("mfma only", out);
  run<4, 0, 1, true>("v_fma_f32", out);
  run<8, 0, 1, true>("v_fma_f32", out);
  run<16, 0, 1, false>("v_fma_f32", out);
  run<2, 6, 1, true>("ds_read_b128 (unused)", out);
  hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
