// prefetch_l2.hip -- can the HBM-idle tail of dec_attn carry the NEXT weight-streaming GEMM's weights into the L2 of the XCD that will
// read them?  Kernel-development aid (round 6), pricing before building; never on the product path.
//
// profiles/r06_step_timeline.log: the K/V stream of the fused dec_attn ends ~2.8 us before the kernel does (combine + out_proj), then come a
// boundary, the reduce + LN launch and another boundary before linear1 asks HBM for its 16.8 MB -- and pays a cold start (~2 us before the
// first byte).  A decode GEMM on weights the previous launch just read runs 4.5 instead of 6.5 us (round 2, tools/l2_retention.py: clean
// lines survive a kernel boundary in the XCD's L2).  Emulated here, one hipGraph chain of `layers` x [A | L | G]:
//   A  256 workgroups x 1024 threads stream `kv_mb` MB with non-temporal loads (dec_attn's stream); with PF each workgroup then loads
//      -- temporally -- the 64 KiB weight tile that workgroup `same linear id` of G will read (same id => same XCD), and waits for it;
//   L  32 workgroups, a few KB (the reduce + LN launch in between);
//   G  256 workgroups x 512 threads, 64 KiB of weights each, non-temporal, all requests up front (linear1).
// Every layer has its own weight buffer and the K/V buffers rotate over 2 GB, so nothing is reused across layers.
// Reported: us per layer without prefetch, with prefetch, and with a prefetch of the WRONG buffer (the cost alone).
//     hipcc --offload-arch=gfx950 -O3 tools/ubench/prefetch_l2.hip -o tools/ubench/prefetch_l2.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int PF>
__global__ __launch_bounds__(1024) void a_kernel(const float* __restrict__ kv, long f4_per_wg, const float* __restrict__ wpf, float* __restrict__ sink,
                                                 const float* __restrict__ wo_heads, int tail) {
  const int wg = blockIdx.x + gridDim.x * blockIdx.y;
  const f32x4* p = reinterpret_cast<const f32x4*>(kv) + (long)wg * f4_per_wg + threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 a[4], b[4];
  const long n = f4_per_wg / 1024;                        // float4 per thread
#pragma unroll
  for (int u = 0; u < 4; ++u) a[u] = __builtin_nontemporal_load(p + (long)u * 1024);
  for (long i = 4; i < n + 4; i += 8) {
#pragma unroll
    for (int u = 0; u < 4; ++u) b[u] = __builtin_nontemporal_load(p + (long)std::min(i + u, n - 1) * 1024);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += a[u];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = __builtin_nontemporal_load(p + (long)std::min(i + 4 + u, n - 1) * 1024);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += b[u];
  }
  // dec_attn's tail (TAIL, argv[4]): the head's 256 KiB W_o slice out of L2 (16 float4 per thread, shared by the 16 workgroups of a head) and a
  // little arithmetic; the prefetch is requested BEHIND the W_o loads (loads return in order: a cold prefetch in front would hold them up)
  f32x4 wo[16];
  if (tail) {
    const f32x4* q = reinterpret_cast<const f32x4*>(wo_heads) + (long)blockIdx.x * 16 * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 16; ++u) wo[u] = q[u * 1024];
  }
  f32x4 pf[4];
  if (PF) {                                               // 64 KiB = 4 float4 per thread, temporal
    const f32x4* w = reinterpret_cast<const f32x4*>(wpf) + (long)wg * 4096 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) pf[u] = w[u * 1024];
  }
  if (tail) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += wo[u] * acc;
    __syncthreads();
    if (acc[0] == 12345.678f) sink[wg + 1024] = acc[1];
    else if (threadIdx.x < 64) sink[2048 + (wg * 64 + threadIdx.x) % 1024] = acc[2];      // the result store of the epilogue
  }
  if (PF) {
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += pf[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[wg] = acc[0];
}

__global__ __launch_bounds__(256) void l_kernel(const float* __restrict__ in, float* __restrict__ out) {
  const f32x4 v = reinterpret_cast<const f32x4*>(in)[blockIdx.x * 256 + threadIdx.x];
  reinterpret_cast<f32x4*>(out)[blockIdx.x * 256 + threadIdx.x] = v + 1.0f;
}

__global__ __launch_bounds__(512) void g_kernel(const float* __restrict__ w, const float* __restrict__ x, float* __restrict__ sink) {
  const f32x4* p = reinterpret_cast<const f32x4*>(w) + (long)blockIdx.x * 4096 + threadIdx.x;
  f32x4 v[8];
  const f32x4 xv = reinterpret_cast<const f32x4*>(x)[threadIdx.x];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + u * 512);
  f32x4 acc = xv;
#pragma unroll
  for (int u = 0; u < 8; ++u) acc += v[u];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[blockIdx.x] = acc[0];
}

int main(int argc, char** argv) {
  const int layers = argc > 1 ? atoi(argv[1]) : 48, reps = argc > 2 ? atoi(argv[2]) : 11, kv_mb = argc > 3 ? atoi(argv[3]) : 184, tail = argc > 4 ? atoi(argv[4]) : 1;
  const int NW = 12, NKV = 11;
  const size_t wbytes = (size_t)256 * 65536;              // 16.8 MB
  const long f4_per_wg = ((long)kv_mb * 1000000 / 256 / 16) / 8192 * 8192;
  std::vector<float*> w(NW), kv(NKV);
  for (auto& q : w) { CHECK(hipMalloc(&q, wbytes)); CHECK(hipMemset(q, 0, wbytes)); }
  for (auto& q : kv) { CHECK(hipMalloc(&q, (size_t)f4_per_wg * 256 * 16)); CHECK(hipMemset(q, 0, (size_t)f4_per_wg * 256 * 16)); }
  float *sink, *lbuf, *wo;
  CHECK(hipMalloc(&sink, 4096 * 4));
  CHECK(hipMalloc(&wo, (size_t)16 * 16 * 1024 * 16));
  CHECK(hipMemset(wo, 0, (size_t)16 * 16 * 1024 * 16));
  CHECK(hipMalloc(&lbuf, 2 * 32 * 256 * 16));
  CHECK(hipMemset(lbuf, 0, 2 * 32 * 256 * 16));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  auto run = [&](int mode) {                              // 0 no prefetch, 1 prefetch of the tile G reads, 2 prefetch of another layer's buffer
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < layers; ++l) {
      const float* wl = w[l % NW];
      const float* wp = mode == 2 ? w[(l + 5) % NW] : wl;
      if (mode == 0) hipLaunchKernelGGL(a_kernel<0>, dim3(16, 16), dim3(1024), 0, s, kv[l % NKV], f4_per_wg, wp, sink, wo, tail);
      else hipLaunchKernelGGL(a_kernel<1>, dim3(16, 16), dim3(1024), 0, s, kv[l % NKV], f4_per_wg, wp, sink, wo, tail);
      hipLaunchKernelGGL(l_kernel, dim3(32), dim3(256), 0, s, lbuf, lbuf + 32 * 256 * 4);
      hipLaunchKernelGGL(g_kernel, dim3(256), dim3(512), 0, s, wl, lbuf + 32 * 256 * 4, sink);
    }
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<double> us;
    for (int r = 0; r < reps + 2; ++r) {
      CHECK(hipEventRecord(e0, s));
      CHECK(hipGraphLaunch(ge, s));
      CHECK(hipEventRecord(e1, s));
      CHECK(hipStreamSynchronize(s));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) us.push_back(ms * 1000.0 / layers);
    }
    std::sort(us.begin(), us.end());
    CHECK(hipGraphExecDestroy(ge));
    CHECK(hipGraphDestroy(g));
    return us[us.size() / 2];
  };
  printf("A streams %.1f MB per launch (%s); us per [A | L | G] layer, median of %d replays of a %d-layer chain\n", f4_per_wg * 256 * 16 / 1e6,
         tail ? "with the out_proj tail" : "no tail", reps, layers);
  for (int round = 0; round < 2; ++round)
    printf("no prefetch %7.2f   prefetch of G's tiles %7.2f   prefetch of another buffer (cost alone) %7.2f\n", run(0), run(1), run(2));
  return 0;
}
