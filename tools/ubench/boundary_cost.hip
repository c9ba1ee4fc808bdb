// boundary_cost.hip -- what does the boundary between two DEPENDENT kernels of a hipGraph chain cost, and does it depend on how the
// producer stored its output?  Kernel-development aid (round 6); never on the product path.
//
// The decode step is a chain of 74 dependent launches; profiles/r06_step_timeline.log puts 1.3-1.7 us between the last instruction of
// one kernel and the first of the next.  Part of a kernel's end is the release of its writes (L2 write-back: the 8 XCDs' L2s are not
// coherent with each other).  Question: is the boundary shorter when the producer leaves NO dirty line behind -- stores that write
// through to memory (sc0 sc1) -- or when it writes less?
//
// A chain of `links` pairs [W: G workgroups x 256 threads store one float4 each | R: 256 workgroups, every thread loads one float4 that a
// workgroup on another XCD wrote and thread 0 stores 4 bytes] is captured into one hipGraph and replayed; reported: microseconds per pair,
// for W's store flavour in {plain, nt, sc1, sc0 sc1} and G in {32 (128 KiB, a reduce + LN launch), 256 (1 MiB, a split-K GEMM's slabs), 1024 (4 MiB)}.
// The chain of W alone and of R alone (no data dependence, same barriers) is the baseline.
//     hipcc --offload-arch=gfx950 -O3 tools/ubench/boundary_cost.hip -o tools/ubench/boundary_cost.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void w_kernel(float* __restrict__ out, float seed) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  f32x4 v = {seed, seed + (float)i, 1.0f, -seed};
  f32x4* p = reinterpret_cast<f32x4*>(out) + i;
  if (MODE == 0) *p = v;
  else if (MODE == 1) __builtin_nontemporal_store(v, p);
  else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

// LMODE: how the consumer loads (0 plain, 1 nt, 2 sc1, 3 sc0 sc1) -- second table of main()
template <int LMODE = 0>
__global__ __launch_bounds__(256) void r_kernel(const float* __restrict__ in, long n4, float* __restrict__ sink) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 257 % n4;      // a line some other workgroup (another XCD) wrote
  const f32x4* p = reinterpret_cast<const f32x4*>(in) + i;
  f32x4 v;
  if (LMODE == 0) v = *p;
  else if (LMODE == 1) v = __builtin_nontemporal_load(p);
  else if (LMODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (threadIdx.x == 0) sink[blockIdx.x] = v[0] + v[3];
}

template <int MODE, int LMODE = 0>
double chain(float* buf, float* sink, int G, int links, int reps, int what /* 0 pairs, 1 W only, 2 R only */) {
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  const long n4 = (long)G * 256;
  hipGraph_t g;
  hipGraphExec_t ge;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int l = 0; l < links; ++l) {
    if (what != 2) hipLaunchKernelGGL(w_kernel<MODE>, dim3(G), dim3(256), 0, s, buf, (float)l);
    if (what != 1) hipLaunchKernelGGL(r_kernel<LMODE>, dim3(256), dim3(256), 0, s, buf, n4, sink);
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
    if (r >= 2) us.push_back(ms * 1000.0 / links);
  }
  std::sort(us.begin(), us.end());
  CHECK(hipGraphExecDestroy(ge));
  CHECK(hipGraphDestroy(g));
  CHECK(hipStreamDestroy(s));
  return us[us.size() / 2];
}

int main(int argc, char** argv) {
  const int links = argc > 1 ? atoi(argv[1]) : 200, reps = argc > 2 ? atoi(argv[2]) : 11;
  float *buf, *sink;
  CHECK(hipMalloc(&buf, (size_t)1024 * 256 * 16));
  CHECK(hipMalloc(&sink, 4096));
  CHECK(hipMemset(buf, 0, (size_t)1024 * 256 * 16));
  printf("us per link of a %d-link graph chain (median of %d replays)\n", links, reps);
  printf("%-28s %10s %10s %10s %10s\n", "", "plain", "nt", "sc1", "sc0 sc1");
  for (int G : {32, 256, 1024}) {
    char name[64];
    snprintf(name, sizeof name, "W(%4d KiB) -> R   pairs", G * 4);
    printf("%-28s %10.2f %10.2f %10.2f %10.2f\n", name, chain<0>(buf, sink, G, links, reps, 0), chain<1>(buf, sink, G, links, reps, 0),
           chain<2>(buf, sink, G, links, reps, 0), chain<3>(buf, sink, G, links, reps, 0));
    snprintf(name, sizeof name, "W(%4d KiB) alone", G * 4);
    printf("%-28s %10.2f %10.2f %10.2f %10.2f\n", name, chain<0>(buf, sink, G, links, reps, 1), chain<1>(buf, sink, G, links, reps, 1),
           chain<2>(buf, sink, G, links, reps, 1), chain<3>(buf, sink, G, links, reps, 1));
  }
  printf("%-28s %10.2f\n", "R alone (1 MiB source)", chain<0>(buf, sink, 256, links, reps, 2));
  printf("\nconsumer's load flavour (producer: sc1 stores, 1 MiB)\n%-28s %10s %10s %10s %10s\n", "", "plain", "nt", "sc1", "sc0 sc1");
  printf("%-28s %10.2f %10.2f %10.2f %10.2f\n", "W(1024 KiB) -> R   pairs", chain<2, 0>(buf, sink, 256, links, reps, 0), chain<2, 1>(buf, sink, 256, links, reps, 0),
         chain<2, 2>(buf, sink, 256, links, reps, 0), chain<2, 3>(buf, sink, 256, links, reps, 0));
  printf("%-28s %10.2f %10.2f %10.2f %10.2f\n", "R alone", chain<0, 0>(buf, sink, 256, links, reps, 2), chain<0, 1>(buf, sink, 256, links, reps, 2),
         chain<0, 2>(buf, sink, 256, links, reps, 2), chain<0, 3>(buf, sink, 256, links, reps, 2));
  return 0;
}
