// Micro-benchmark (kernel-development aid): LDS throughput of the access patterns the kernels use, to check the
// bank-conflict model behind their layouts.  Every CU runs 8 waves (2 workgroups x 4) issuing the same ds instruction
// with pattern-specific lane addresses; reports LDS cycles per wave instruction per CU (nominal 2.4 GHz).
// Ideal: ds_read_b128 / ds_write_b128 = 8 (1 KiB at 128 B/clk), b64 = 4, b32 = 2.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 2000, UNR = 16;

__device__ __forceinline__ int vt_row(int d) { return d * 64 + (d >> 2) * 16; }

__device__ int pattern_addr(int pat, int tid) {
  const int lane = tid & 63, wid = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  switch (pat) {
    case 0: return lane * 16;                                                      // linear b128
    case 1: { const int row = wid * 32 + l31; return row * 64 + ((hi ^ ((row >> 2) & 3)) * 16); }       // bf16x3 GEMM fragment
    case 2: return l31 * 144 + hi * 16;                                            // attention K planes
    case 3: return vt_row(l31) + hi * 16;                                          // attention V^T planes
    case 4: return (l31 * 36 + hi * 4) * 4;                                        // fp32 GEMM fragment (36-float rows)
    case 5: return lane * 256;                                                     // worst case: one bank
    case 6: { const int idx = tid, row = idx >> 2, ch = idx & 3; return row * 64 + ((ch ^ ((row >> 2) & 3)) * 16); }  // GEMM staging write b128
    case 7: return (tid >> 4) * 144 + (tid & 15) * 8;                              // attention K staging write b64
    case 8: { const int c4 = (tid & 15) * 4, kp2 = (tid >> 4) * 2;                 // attention V^T staging write b32 (dim e = 0)
              const int vpos = ((kp2 >> 4) * 2 + ((kp2 >> 2) & 1)) * 8 + (kp2 & 3) + 4 * ((kp2 >> 3) & 1);
              return vt_row(c4) + vpos * 2; }
    case 9: return l31 * 128 + hi * 16;                                            // unpadded 128-B rows (32-way?)
    default: return 0;
  }
}

template <int KIND>   // 0 read b128, 1 write b128, 2 write b64, 3 write b32
__global__ __launch_bounds__(256) void k(int pat, float* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int addr = pattern_addr(pat, threadIdx.x) & 0xfff0 & ~(KIND == 3 ? 0 : 0);
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(lds)[i] = (float)i;
  __syncthreads();
  f32x4 acc = {0, 0, 0, 0};
  f32x4 val = {1.f, 2.f, 3.f, 4.f};
  const unsigned base = (unsigned)(size_t)lds + (KIND == 3 ? pattern_addr(pat, threadIdx.x) : addr);
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (KIND == 0) { f32x4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(base)); asm volatile("s_waitcnt lgkmcnt(8)"); acc += t; }
      if (KIND == 1) asm volatile("ds_write_b128 %0, %1" ::"v"(base), "v"(val));
      if (KIND == 2) asm volatile("ds_write_b64 %0, %1" ::"v"(base), "v"(f32x2{val[0], val[1]}));
      if (KIND == 3) asm volatile("ds_write_b32 %0, %1" ::"v"(base), "v"(val[0]));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int KIND>
void run(const char* name, int pat, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(512), dim3(256), 0, 0, pat, out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
  }
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  // per CU: 8 waves x ITER x UNR instructions
  printf("%-44s : %6.2f cycles / wave-instruction / CU\n", name, ms * 1e-3 * 2.4e9 / (8.0 * ITER * UNR));
}

int main() {
  float* out;
  (void)hipMalloc((void**)&out, 512 * 256 * 4);
  run<0>("warmup", 0, out);
  run<0>("read b128 linear (ideal 8)", 0, out);
  run<0>("read b128 bf16x3 GEMM fragment", 1, out);
  run<0>("read b128 attention K planes (144-B rows)", 2, out);
  run<0>("read b128 attention V^T planes", 3, out);
  run<0>("read b128 fp32 GEMM fragment (36-float rows)", 4, out);
  run<0>("read b128 unpadded 128-B rows", 9, out);
  run<0>("read b128 one bank (worst)", 5, out);
  run<1>("write b128 linear (ideal 8)", 0, out);
  run<1>("write b128 GEMM staging (swizzled)", 6, out);
  run<2>("write b64 linear (ideal 4)", 0, out);
  run<2>("write b64 attention K staging", 7, out);
  run<3>("write b32 linear (ideal 2)", 0, out);
  run<3>("write b32 attention V^T staging", 8, out);
  (void)hipDeviceSynchronize();
  return 0;
}
