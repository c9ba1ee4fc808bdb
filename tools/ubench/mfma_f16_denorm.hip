// Does v_mfma_f32_32x32x16_f16 honour fp16 DENORMAL inputs on gfx950, or flush them to zero?
// A = 2^-20 (an fp16 subnormal), B = 1024: every output must be 16 * 2^-20 * 1024 = 2^-6 = 0.015625 (0 if flushed).
// Also the packed conversion used by the split kernels: (_Float16)x for a subnormal-range x must not flush either.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float a_val, float b_val) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)a_val; b[e] = (_Float16)b_val; }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; }
}
int main() {
  float* d;
  hipMalloc(&d, 8);
  const float vals[3] = {9.5367431640625e-07f /* 2^-20 */, 3.0e-5f, 6.1035e-5f /* just above min normal */};
  for (float v : vals) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, v, 1024.0f);
    float h[2];
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a = %.6e: cvt f16 -> %.6e | mfma 16 * a * 1024 = %.6e (exact %.6e) %s\n", v, h[1], h[0], 16.0 * h[1] * 1024.0,
           h[0] == 16.0f * h[1] * 1024.0f ? "DENORMALS HONOURED" : "FLUSHED / WRONG");
  }
  return 0;
}
