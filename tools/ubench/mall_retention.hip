// mall_retention.hip -- does a block of "weights" survive in the 256 MiB Infinity Cache while a multi-GB stream with a
// different cache policy passes through?  (The decode step reads 609 MB of weights + ~2.2 GB of K/V per step; if K/V read with
// policy P do not displace lines that were read with policy Q, the weights of a few layers could be served from the Infinity
// Cache on every step.)  Kernel-development aid; never on the product path.
//
//   for W in {48, 96, 192} MB, for (Wpol, Spol) in policy pairs:
//     read W (Wpol) twice; time a third read            -> "hot"   (nothing in between)
//     read S = 2.4 GB (Spol), timed                      -> stream bandwidth of that policy
//     time a read of W (Wpol)                            -> "after" (what survived the stream)
//
// policies: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0 sc1 nt   (global_load_dwordx4 cache-policy bits of gfx940+)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int POL>
__device__ __forceinline__ void ld8(u32x4 (&v)[8], const u32x4* p, size_t stride) {
#define LD(U, MOD) asm volatile("global_load_dwordx4 %0, %1, off " MOD : "=&v"(v[U]) : "v"(p + (U) * stride) : "memory")
#define LD8(MOD) LD(0, MOD); LD(1, MOD); LD(2, MOD); LD(3, MOD); LD(4, MOD); LD(5, MOD); LD(6, MOD); LD(7, MOD)
  if (POL == 0) { LD8(""); }
  if (POL == 1) { LD8("nt"); }
  if (POL == 2) { LD8("sc1"); }
  if (POL == 3) { LD8("sc0 sc1"); }
  if (POL == 4) { LD8("sc1 nt"); }
  if (POL == 5) { LD8("sc0 sc1 nt"); }
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
#undef LD8
#undef LD
}

// n16 = number of 16-byte words, a multiple of gridDim.x * 256 * 8
template <int POL>
__global__ __launch_bounds__(256) void rd_kernel(const u32x4* __restrict__ p, size_t n16, u32x4* sink) {
  const size_t nthr = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (; i < n16; i += nthr * 8) {
    u32x4 v[8];
    ld8<POL>(v, p + i, nthr);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u];
  }
  if (acc.x == 0x9e3779b9u && acc.y == 0x7f4a7c15u && acc.z == 1u && acc.w == 2u) sink[0] = acc;   // never (keeps the loads)
}

static void launch(int pol, const u32x4* p, size_t n16, u32x4* sink, hipStream_t st) {
  const int grid = 2048;
  switch (pol) {
    case 0: rd_kernel<0><<<grid, 256, 0, st>>>(p, n16, sink); break;
    case 1: rd_kernel<1><<<grid, 256, 0, st>>>(p, n16, sink); break;
    case 2: rd_kernel<2><<<grid, 256, 0, st>>>(p, n16, sink); break;
    case 3: rd_kernel<3><<<grid, 256, 0, st>>>(p, n16, sink); break;
    case 4: rd_kernel<4><<<grid, 256, 0, st>>>(p, n16, sink); break;
    default: rd_kernel<5><<<grid, 256, 0, st>>>(p, n16, sink); break;
  }
}

static float timed(int pol, const u32x4* p, size_t n16, u32x4* sink, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  CHECK(hipEventRecord(e0, st));
  launch(pol, p, n16, sink, st);
  CHECK(hipEventRecord(e1, st));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main() {
  const char* pn[6] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt"};
  const size_t quantum = (size_t)2048 * 256 * 8 * 16;                 // bytes one grid pass covers (64 MiB)
  const size_t s_bytes = quantum * 36;                                  // 2.4 GB stream
  const size_t w_max = quantum * 12;                                    // 12 'layers' of 64 MiB (first part: up to 192 MiB)
  u32x4 *S, *W, *sink;
  CHECK(hipMalloc(&S, s_bytes));
  CHECK(hipMalloc(&W, w_max));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(S, 1, s_bytes));
  CHECK(hipMemset(W, 2, w_max));
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) timed(0, S, s_bytes / 16, sink, st, e0, e1);      // clocks up
  printf("stream bandwidth by policy (2.4 GB, second of two runs)\n");
  for (int pol = 0; pol < 6; ++pol) {
    timed(pol, S, s_bytes / 16, sink, st, e0, e1);
    const float ms = timed(pol, S, s_bytes / 16, sink, st, e0, e1);
    printf("  %-11s %7.3f ms  %6.0f GB/s\n", pn[pol], ms, s_bytes / ms / 1e6);
  }
  // W is read in 3/4-MiB-interleaved fashion by the same kernel (one quantum = 64 MiB per grid pass), sizes 64 / 128 / 192 MiB
  const int pairs[][2] = {{0, 0}, {0, 1}, {0, 4}, {0, 5}, {0, 3}, {1, 1}, {1, 0}, {2, 1}, {3, 1}};   // {Wpol, Spol}
  for (int wq = 1; wq <= 3; ++wq) {
    const size_t w_bytes = quantum * wq;
    printf("W = %zu MiB\n", w_bytes >> 20);
    for (auto& pr : pairs) {
      const int wp = pr[0], sp = pr[1];
      timed(wp, W, w_bytes / 16, sink, st, e0, e1);
      timed(wp, W, w_bytes / 16, sink, st, e0, e1);
      const float hot = timed(wp, W, w_bytes / 16, sink, st, e0, e1);
      const float sm = timed(sp, S, s_bytes / 16, sink, st, e0, e1);
      const float after = timed(wp, W, w_bytes / 16, sink, st, e0, e1);
      printf("  W %-11s S %-11s  hot %6.1f us (%6.0f GB/s)   stream %6.0f GB/s   after %6.1f us (%6.0f GB/s)\n", pn[wp], pn[sp],
             hot * 1e3, w_bytes / hot / 1e6, s_bytes / sm / 1e6, after * 1e3, w_bytes / after / 1e6);
    }
  }
  // partial streams: how many MB of a plain / nt stream does it take to push a 64 MiB plain-read W out?
  printf("W = 64 MiB (plain), stream length sweep\n");
  for (int sp = 0; sp <= 1; ++sp)
    for (int sq : {1, 2, 4, 8, 16, 36}) {
      const size_t w_bytes = quantum;
      timed(0, W, w_bytes / 16, sink, st, e0, e1);
      timed(0, W, w_bytes / 16, sink, st, e0, e1);
      timed(sp, S, quantum * sq / 16, sink, st, e0, e1);
      const float after = timed(0, W, w_bytes / 16, sink, st, e0, e1);
      printf("  S %-5s %5zu MiB   after %6.1f us (%6.0f GB/s)\n", pn[sp], (quantum * sq) >> 20, after * 1e3, w_bytes / after / 1e6);
    }
  // the decode step's access pattern: 12 x [64 MiB of weights | 192 MiB of K/V], every step the same addresses.  The weights of the
  // first k layers are read with the temporal policy `wp`, all other weights nt, the K/V with `sp`: does a step get shorter
  // because k x 64 MiB stay in the Infinity Cache from one step to the next?
  printf("step emulation: 12 x [64 MiB W | 192 MiB KV], 20 steps, ms per step (3 GiB per step)\n");
  const int wpols[] = {0, 2, 3};
  for (int sp : {1, 0, 4, 5})
    for (int wi = 0; wi < 3; ++wi)
      for (int k : {0, 1, 2, 3}) {
        if (k == 0 && wi > 0) continue;
        const int wp = wpols[wi];
        auto one_step = [&]() {
          for (int l = 0; l < 12; ++l) {
            launch(l < k ? wp : 1, W + (size_t)l * (quantum / 16), quantum / 16, sink, st);
            launch(sp, S + (size_t)l * (3 * quantum / 16), 3 * quantum / 16, sink, st);
          }
        };
        for (int i = 0; i < 3; ++i) one_step();
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < 20; ++i) one_step();
        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  KV %-11s  first %d layers' W %-8s  %7.3f ms per step  (%5.0f GB/s algorithmic)\n", pn[sp], k, k ? pn[wp] : "-", ms / 20,
               48.0 * quantum / (ms / 20) / 1e6);
      }
  return 0;
}
