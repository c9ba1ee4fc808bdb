// dispatch_spread.hip -- how long does the chip take to START the workgroups of one launch, and what does it depend on?
// Kernel-development aid (round 6); never on the product path.  Motivation: profiles/r06_step_timeline.log -- the 256 workgroups of the
// fused dec_attn (1024 threads, 128 VGPRs, 16 kernel-argument dwords preloaded into SGPRs) start over 1.65 us, the 256 workgroups of
// linear1 (512 threads) over 0.29 us; a workgroup that starts late ends late, and the launch ends with its last workgroup.
//
// Every workgroup stamps the 100 MHz wall clock in its first instruction; reported per configuration: last start - first start and the
// 50 % / 90 % quantiles, median over `reps` launches (each launch behind a stream sync, so the chip is idle when it begins).
// Configurations: block size 256 / 512 / 1024 x kernel-argument dwords 4 / 16 x VGPR budget (small / 128 forced) x LDS 0 / 64 KiB, at 256
// workgroups (one per CU) and, for the 256- and 512-thread blocks, at the grid with the same number of WAVES as 256 x 1024.
//     hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 tools/ubench/dispatch_spread.hip -o /tmp/dispatch_spread
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Args16 { unsigned long long* out; int a[14]; };      // 16 dwords
struct Args4 { unsigned long long* out; int a[2]; };        // 4 dwords

template <int BLOCK, bool BIGV, int LDS_KB, typename A>
__global__ __launch_bounds__(BLOCK) void stamp_kernel(A g) {
  const unsigned long long t = wall_clock64();
  __shared__ float lds[LDS_KB > 0 ? LDS_KB * 256 : 1];
  if (LDS_KB > 0) lds[threadIdx.x] = (float)t;
  if (BIGV) asm volatile("v_mov_b32 v127, 0" ::: "v127");    // the kernel is allocated 128 VGPRs per lane
  if (threadIdx.x == 0) g.out[blockIdx.x] = t;
  if (LDS_KB > 0 && lds[(threadIdx.x + 1) % BLOCK] == -1.f) g.out[0] = 0;
  int s = 0;
  for (int i = 0; i < (int)(sizeof(g.a) / sizeof(int)); ++i) s += g.a[i];
  if (s == 12345) g.out[1] = 0;                               // every argument is live
}

template <int BLOCK, bool BIGV, int LDS_KB, typename A>
void run(const char* name, int grid, unsigned long long* d_out, int reps) {
  A g{};
  g.out = d_out;
  std::vector<unsigned long long> h(grid);
  std::vector<double> last, q50, q90;
  for (int r = 0; r < reps + 2; ++r) {
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL((stamp_kernel<BLOCK, BIGV, LDS_KB, A>), dim3(grid), dim3(BLOCK), 0, 0, g);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), d_out, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (r < 2) continue;
    std::sort(h.begin(), h.end());
    last.push_back((h[grid - 1] - h[0]) / 100.0);
    q50.push_back((h[grid / 2] - h[0]) / 100.0);
    q90.push_back((h[grid * 9 / 10] - h[0]) / 100.0);
  }
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  printf("%-46s grid %5d  waves %5d | start spread us: 50%% %5.2f  90%% %5.2f  last %5.2f\n", name, grid, grid * BLOCK / 64, med(q50), med(q90),
         med(last));
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 21;
  unsigned long long* d_out;
  CHECK(hipMalloc(&d_out, 8192 * sizeof(unsigned long long)));
  printf("== 256 workgroups (one per CU)\n");
  run<256, false, 0, Args4>("256 thr,  4 arg dwords, few VGPRs", 256, d_out, reps);
  run<256, false, 0, Args16>("256 thr, 16 arg dwords, few VGPRs", 256, d_out, reps);
  run<512, false, 0, Args4>("512 thr,  4 arg dwords, few VGPRs", 256, d_out, reps);
  run<512, false, 0, Args16>("512 thr, 16 arg dwords, few VGPRs", 256, d_out, reps);
  run<512, true, 0, Args16>("512 thr, 16 arg dwords, 128 VGPRs", 256, d_out, reps);
  run<1024, false, 0, Args4>("1024 thr,  4 arg dwords, few VGPRs", 256, d_out, reps);
  run<1024, false, 0, Args16>("1024 thr, 16 arg dwords, few VGPRs", 256, d_out, reps);
  run<1024, true, 0, Args4>("1024 thr,  4 arg dwords, 128 VGPRs", 256, d_out, reps);
  run<1024, true, 0, Args16>("1024 thr, 16 arg dwords, 128 VGPRs (dec_attn)", 256, d_out, reps);
  run<1024, true, 64, Args16>("1024 thr, 16 arg dwords, 128 VGPRs, 64 KiB LDS", 256, d_out, reps);
  printf("== 4096 waves in smaller workgroups\n");
  run<512, true, 0, Args16>("512 thr, 16 arg dwords, 128 VGPRs", 512, d_out, reps);
  run<512, true, 0, Args4>("512 thr,  4 arg dwords, 128 VGPRs", 512, d_out, reps);
  run<256, true, 0, Args16>("256 thr, 16 arg dwords, 128 VGPRs", 1024, d_out, reps);
  run<256, true, 0, Args4>("256 thr,  4 arg dwords, 128 VGPRs", 1024, d_out, reps);
  printf("== the decode GEMM grids\n");
  run<256, false, 16, Args4>("256 thr,  4 arg dwords, 16 KiB LDS (in_proj)", 512, d_out, reps);
  run<512, false, 16, Args4>("512 thr,  4 arg dwords, 16 KiB LDS (linear1)", 256, d_out, reps);
  return 0;
}
