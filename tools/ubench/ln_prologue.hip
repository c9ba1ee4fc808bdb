// ln_prologue.hip -- pricing of a design that was NOT built (round-5 verdict, task 3): "linear2 / out_proj produce COMPLETE rows and the
// consumer GEMM computes the LayerNorm statistics in its prologue from the 128 KB h image it already reads" -- i.e. every one of the
// consumer's workgroups pulls all 32 x 1024 fp32 of h through L2 and reduces 32 rows before its first MFMA.  Kill criterion of the
// verdict: the prologue must stay under 3 us.  Kernel-development aid; never on the product path.
//
// A decode-step-shaped chain is replayed from a hipGraph, `reps` times:   W (writes h, 32 workgroups: the producer)  ->  C (consumer)
// with C one of
//   S   256-thread workgroups that stream their 64 KB share of a 16.8 MB weight matrix (non-temporal, all requests up front) -- what a
//       decode GEMM's memory side does today (no prologue);
//   P   the prologue alone: read 128 KB of h (L2 / Infinity-Cache resident, written by W a moment ago), row sums and sums of squares
//       of 32 rows (wave DPP-free shuffles + LDS across the 4 waves), normalised slice to LDS;
//   PS  prologue with the weight requests issued FIRST (the weights are in flight while h is reduced), then the same consumption as S.
// Reported: microseconds per (W, C) pair and per C alone (pair minus the W-only chain), for 256 / 384 / 512 consumer workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int ROWS = 32, D = 1024;

__global__ __launch_bounds__(256) void write_h(float* __restrict__ h, float seed) {
  const int r = blockIdx.x;                                   // one row per workgroup, like dec_reduce_ln_pack
  f32x4 v = {seed + r, seed * 0.5f + threadIdx.x, 1.0f, -seed};
  *reinterpret_cast<f32x4*>(h + (long)r * D + threadIdx.x * 4) = v;
}

template <bool PROLOGUE, bool STREAM>
__global__ __launch_bounds__(256) void consumer(const float* __restrict__ h, const float* __restrict__ w, long w_per_wg, float* __restrict__ out) {
  __shared__ float red[2][4][ROWS];
  __shared__ float xs[ROWS * 32];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  f32x4 wv[16];
  if (STREAM) {                                               // 64 KB per workgroup = 16 float4 per thread, requested before anything else
    const f32x4* wp = reinterpret_cast<const f32x4*>(w + (long)blockIdx.x * w_per_wg) + t;
#pragma unroll
    for (int u = 0; u < 16; ++u) wv[u] = __builtin_nontemporal_load(wp + u * 256);
  }
  float acc = 0.f;
  if (PROLOGUE) {
    // thread t holds column chunk 4 t .. 4 t + 3 of all 32 rows: 32 float4 loads, one round trip
    f32x4 hv[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) hv[r] = *reinterpret_cast<const f32x4*>(h + (long)r * D + t * 4);
    float s[ROWS], q[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      s[r] = (hv[r][0] + hv[r][1]) + (hv[r][2] + hv[r][3]);
      q[r] = (hv[r][0] * hv[r][0] + hv[r][1] * hv[r][1]) + (hv[r][2] * hv[r][2] + hv[r][3] * hv[r][3]);
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { s[r] += __shfl_xor(s[r], o, 64); q[r] += __shfl_xor(q[r], o, 64); }
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < ROWS; ++r) { red[0][wid][r] = s[r]; red[1][wid][r] = q[r]; }
    }
    __syncthreads();
    // every thread normalises its chunk of the rows of the K slice this workgroup would contract (32 columns: a 1/32 slice)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const float sum = (red[0][0][r] + red[0][1][r]) + (red[0][2][r] + red[0][3][r]);
      const float sq = (red[1][0][r] + red[1][1][r]) + (red[1][2][r] + red[1][3][r]);
      const float mean = sum * (1.0f / D), var = sq * (1.0f / D) - mean * mean;
      const float rstd = rsqrtf(var + 1e-5f);
      if (t < 8) {
        const f32x4 x = hv[r];
        xs[r * 32 + t * 4 + 0] = (x[0] - mean) * rstd; xs[r * 32 + t * 4 + 1] = (x[1] - mean) * rstd;
        xs[r * 32 + t * 4 + 2] = (x[2] - mean) * rstd; xs[r * 32 + t * 4 + 3] = (x[3] - mean) * rstd;
      }
    }
    __syncthreads();
    acc = xs[(t * 7) & (ROWS * 32 - 1)];
  }
  if (STREAM) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += (wv[u][0] + wv[u][1]) + (wv[u][2] + wv[u][3]);
  }
  if (acc == 123.456f) out[blockIdx.x] = acc;                 // never (keeps the work)
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 400;
  CHECK(hipSetDevice(0));
  float *h, *out;
  CHECK(hipMalloc(&h, ROWS * D * 4));
  CHECK(hipMalloc(&out, 4096));
  // 12 different weight matrices of 512 x 64 KB so that a replay never re-reads a matrix that is still cache resident
  const long w_per_wg = 64 * 1024 / 4;
  const int NW = 12;
  std::vector<float*> w(NW);
  for (auto& p : w) { CHECK(hipMalloc(&p, 512 * 64 * 1024)); CHECK(hipMemset(p, 0, 512 * 64 * 1024)); }
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  auto chain = [&](int mode, int grid) {                      // capture NW x (W, C) pairs into one graph, replay reps / NW times
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < NW; ++i) {
      write_h<<<ROWS, 256, 0, st>>>(h, 1.0f + i);
      if (mode == 1) consumer<false, true><<<grid, 256, 0, st>>>(h, w[i], w_per_wg, out);
      if (mode == 2) consumer<true, false><<<grid, 256, 0, st>>>(h, w[i], w_per_wg, out);
      if (mode == 3) consumer<true, true><<<grid, 256, 0, st>>>(h, w[i], w_per_wg, out);
    }
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CHECK(hipGraphLaunch(ge, st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, st));
    const int n = reps / NW + 1;
    for (int i = 0; i < n; ++i) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
    return (double)ms * 1e3 / (n * NW);
  };
  const double base = chain(0, 0);
  printf("W only (producer + its launch boundary): %.2f us per launch\n", base);
  printf("%-28s %8s %8s %8s\n", "consumer (us per launch)", "256 WG", "384 WG", "512 WG");
  const char* names[4] = {"", "S  stream 64 KB / WG", "P  LN prologue alone", "PS prologue + stream"};
  for (int mode = 1; mode <= 3; ++mode) {
    printf("%-28s", names[mode]);
    for (int grid : {256, 384, 512}) printf(" %8.2f", chain(mode, grid) - base);
    printf("\n");
  }
  printf("kill criterion (verdict r04, task 3): PS - S  must stay under 3 us\n");
  return 0;
}
