// bf16x3 GEMM (arithmetic: gemm_bf16x3.hip), third generation: big tile, k-step stages, three-deep async LDS ring.
//
//   C[m][n] = resid[m][n] + colscale[n] * act( sum_k A[m][k] W[n][k] + bias[n] ),  A, W as 3 bf16 planes
//
// What the two earlier kernels showed (tools/gemm_bench.py, tools/ubench): the register-staged 128x128 kernel spends
// ~28 % of its time in the [barrier | ds_write_b128 x 12 | barrier] section; the two-stage DMA kernel
// (gemm_bf16x3_dma.hip) removes that section but then waits at every K tile for its global_load_lds data, which takes
// longer than one tile of MFMAs to arrive.  This kernel gives the DMA two full k-steps of lead and cuts the operand
// traffic per flop by a third:
//   * tile 256 (M) x 256 (N), 512 threads = 8 waves as 4 x 2, wave tile 64 x 128 = 2 x 4 MFMA tiles (128 accumulator
//     VGPRs); one workgroup per CU, 2 waves per SIMD;
//   * a stage is ONE k-step (16 k): 3 planes x (256 + 256) rows x 32 B = 48 KiB; three stages = 144 KiB of LDS;
//   * operand planes are k-step-major, [K/16][rows][16] bf16 (split3_k16_kernel): a stage is 48 wave-level
//     global_load_lds_dwordx4 of 1 KiB, contiguous in memory and in LDS (6 per wave), and a fragment read (lane = row
//     l%32, 8 k of half l/32) is a permutation of one contiguous KiB: conflict-free without padding or swizzle;
//   * per k-step: s_waitcnt vmcnt(6) (everything but the newest stage has landed) | barrier | DMA for k-step j+2 |
//     18 ds_read_b128 + 48 MFMAs (four accumulators interleaved, so no MFMA waits on its predecessor).
#include <algorithm>

#include "vx_common.h"

namespace vx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int RM = 256, RN = 256, RK = 16;
constexpr int PL = 256 * 32;                                     // one plane of one operand in a stage: 8 KiB
constexpr int RSTAGE = 6 * PL;                                   // A planes 0-2, W planes 0-2: 48 KiB
constexpr int RDMA = 6;                                          // DMA instructions per wave and stage (8 waves)
constexpr int RDMA4 = 12;                                        // ... with 4 waves

__device__ __forceinline__ float gelu_erf4(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

}  // namespace

// x[rows][K] fp32 (row r read at gather ? gather[r] : r) -> planes[p][K/16][rows][16] bf16, p = 0..2
__global__ __launch_bounds__(256) void split3_k16_kernel(const float* __restrict__ x, int ldx, long rows, int K,
                                                         const int* __restrict__ gather,
                                                         unsigned short* __restrict__ planes, long plane_stride) {
  const long total = (long)(K / 16) * rows * 2;                  // work item = (k-step, row, 16-byte half), half fastest
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i & 1);
    const long rr = i >> 1;
    const long ks = rr / rows, r = rr - ks * rows;
    const long src = gather ? gather[r] : r;
    const float* xp = x + src * ldx + ks * 16 + ch * 8;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(xp);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(xp + 4);
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    bf16x8 p1, p2, p3;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const __bf16 a1 = (__bf16)v[e];                            // RNE
      const float r1 = v[e] - (float)a1;                         // exact
      const __bf16 a2 = (__bf16)r1;
      const float r2 = r1 - (float)a2;                           // exact
      p1[e] = a1; p2[e] = a2; p3[e] = (__bf16)r2;
    }
    unsigned short* o = planes + i * 8;                          // == ((ks * rows + r) * 16 + ch * 8)
    *reinterpret_cast<bf16x8*>(o) = p1;
    *reinterpret_cast<bf16x8*>(o + plane_stride) = p2;
    *reinterpret_cast<bf16x8*>(o + 2 * plane_stride) = p3;
  }
}

void launch_split3_k16(const float* x, int ldx, long rows, int K, const int* gather, unsigned short* planes,
                       long plane_stride, hipStream_t s) {
  if (rows <= 0) return;
  const long total = rows * (K / 8);
  const int grid = (int)std::min<long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(split3_k16_kernel, dim3(grid), dim3(256), 0, s, x, ldx, rows, K, gather, planes, plane_stride);
}

__global__ __launch_bounds__(512, 1) void gemm_bf16x3_ring_kernel(GemmX3Args g) {
  __shared__ __attribute__((aligned(1024))) unsigned char ring[3 * RSTAGE];
  unsigned char* const st0 = ring;
  unsigned char* const st1 = ring + RSTAGE;
  unsigned char* const st2 = ring + 2 * RSTAGE;

  // rasterisation as in gemm_f32.hip: XCD-contiguous runs of the tile order, GM M-tiles deep groups
  constexpr int GM = 8;
  const int tiles_m = (g.M + RM - 1) / RM, tiles_n = (g.N + RN - 1) / RN;
  const int nwg = tiles_m * tiles_n;
  int wg = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int per_group = GM * tiles_n;
  const int grp = wg / per_group, in_grp = wg - grp * per_group;
  const int gm0 = grp * GM;
  const int gm_rows = (tiles_m - gm0 < GM) ? tiles_m - gm0 : GM;
  const int tm = gm0 + in_grp % gm_rows, tn = in_grp / gm_rows;
  const int m0 = tm * RM, n0 = tn * RN;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1, hi = lane >> 5, l31 = lane & 31;

  // DMA plan: instruction q = wid * 6 + j of a stage covers rows 32 (q % 8) .. + 31 of operand plane q / 8 (0-2 A, 3-5 W);
  // lane -> (row l / 2 of the 32, 16-byte half l % 2): 1 KiB contiguous on both sides
  const unsigned short* src[RDMA];                               // per-lane global address at k-step 0
  long kstride[RDMA];                                            // elements per k-step (rows * 16)
  int lds_off[RDMA];                                             // wave-uniform byte offset inside a stage
#pragma unroll
  for (int j = 0; j < RDMA; ++j) {
    const int q = wid * RDMA + j, op = q >> 3, r32 = q & 7;
    const bool isA = op < 3;
    const int p = isA ? op : op - 3;
    const int lim = isA ? g.M : g.N;
    int grow = (isA ? m0 : n0) + r32 * 32 + (lane >> 1);
    grow = grow < lim ? grow : lim - 1;                          // rows past the edge: clamped, never stored
    src[j] = (isA ? g.A + p * g.a_plane : g.W + p * g.w_plane) + (long)grow * RK + (lane & 1) * 8;
    kstride[j] = (long)lim * RK;
    lds_off[j] = op * PL + r32 * 1024;
  }
  auto dma = [&](unsigned char* stage, int ks) {
#pragma unroll
    for (int j = 0; j < RDMA; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + ks * kstride[j]), (lptr_t)(stage + lds_off[j]), 16, 0, 0);
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Fragment reads are inline asm: for a ds_read it can see, the compiler's waitcnt pass assumes it may alias ANY
  // LDS-DMA still in flight and inserts s_waitcnt vmcnt(0) in front of it, which would throw away the ring's lead.
  // Which stage has landed is tracked by hand (vmcnt(6) + barrier below), so are the lgkmcnt waits of these reads.
  const unsigned lds0 = (unsigned)(size_t)ring;                  // low 32 bits of a flat LDS address = LDS offset
  const unsigned a_addr = lds0 + (wm * 64 + l31) * 32 + hi * 16;              // + stage + plane * PL + i * 1024
  const unsigned w_addr = lds0 + 3 * PL + (wn * 128 + l31) * 32 + hi * 16;    // + stage + plane * PL + jn * 1024
#define VX_LDS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  auto compute = [&](int stage_off) {
    bf16x8 w[3][4], a[2][3];
    const unsigned wa = w_addr + stage_off, aa = a_addr + stage_off;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      VX_LDS_READ(w[p][0], wa, p * PL);
      VX_LDS_READ(w[p][1], wa, p * PL + 1024);
      VX_LDS_READ(w[p][2], wa, p * PL + 2048);
      VX_LDS_READ(w[p][3], wa, p * PL + 3072);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) VX_LDS_READ(a[0][p], aa, p * PL);
#pragma unroll
    for (int p = 0; p < 3; ++p) VX_LDS_READ(a[1][p], aa, p * PL + 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // LDS returns in order: the w and a[0] reads are done when at most 3 are outstanding.  The waits name the
      // registers they release as in/out operands, otherwise the scheduler is free to hoist the MFMAs above them.
      if (i == 0)
        asm volatile("s_waitcnt lgkmcnt(3)"
                     : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[0][2]), "+v"(w[0][3]), "+v"(w[1][0]), "+v"(w[1][1]),
                       "+v"(w[1][2]), "+v"(w[1][3]), "+v"(w[2][0]), "+v"(w[2][1]), "+v"(w[2][2]), "+v"(w[2][3]),
                       "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2])
                     :
                     : "memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]) : : "memory");
      // transposed product (A operand = W rows); per accumulator the six terms keep the order of gemm_bf16x3.hip (small
      // terms first), the four accumulators of an i are interleaved
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const int pw = t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0;   // w3 a1, w1 a3, w2 a2, w2 a1, w1 a2, w1 a1
        const int pa = t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0;
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[pw][jn], a[i][pa], acc[i][jn], 0, 0, 0);
      }
    }
  };
#undef VX_LDS_READ
  // one k-step: everything but the newest DMA group has landed -> barrier -> refill the stage freed two steps ago
  auto kstep = [&](int ks, int nks, int cur_off, unsigned char* refill) {
    // (a bare s_barrier: __syncthreads() is a fence and makes the compiler wait for vmcnt(0), i.e. for the newest DMA
    // group as well; LDS-DMA data is in LDS once vmcnt has counted it, and every ds_read of the stage being refilled
    // was consumed by MFMAs issued before this point)
    if (ks + 1 < nks) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ks + 2 < nks) dma(refill, ks + 2);
    compute(cur_off);
  };

  const int nks = g.K / RK;
  dma(st0, 0);
  if (1 < nks) dma(st1, 1);
  for (int ks = 0; ks < nks; ks += 3) {
    kstep(ks, nks, 0, st2);
    if (ks + 1 < nks) kstep(ks + 1, nks, RSTAGE, st0);
    if (ks + 2 < nks) kstep(ks + 2, nks, 2 * RSTAGE, st1);
  }

  // epilogue: acc[i][jn][4*g4 + e] = C[m0 + wm*64 + i*32 + l31][n0 + wn*128 + jn*32 + 8*g4 + 4*hi + e]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + l31;
    if (m >= g.M) continue;
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int n = n0 + wn * 128 + jn * 32 + 8 * g4 + 4 * hi;
        if (n >= g.N) continue;
        f32x4 v = {acc[i][jn][4 * g4], acc[i][jn][4 * g4 + 1], acc[i][jn][4 * g4 + 2], acc[i][jn][4 * g4 + 3]};
        if (g.bias) {
          const f32x4 bi = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bi[e];
        }
        if (g.act == ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (g.act == ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf4(v[e]);
        }
        if (g.colscale) {
          const f32x4 cs = *reinterpret_cast<const f32x4*>(g.colscale + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] * cs[e];
        }
        if (g.resid) {
          const f32x4 rr = *reinterpret_cast<const f32x4*>(g.resid + (long)m * g.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rr[e] + v[e];
        }
        *reinterpret_cast<f32x4*>(g.C + (long)m * g.ldc + n) = v;
      }
    }
  }
}

// Variant with FOUR waves of 128 x 128 (4 x 4 MFMA tiles, 256 accumulator registers -> AGPRs, one wave per SIMD):
// 24 fragment reads per 96 MFMAs instead of 18 per 48.  LDS reads and MFMAs do not overlap on this part
// (tools/ubench/mfma_valu: their times add), so reads per MFMA is what is left to cut.
__global__ __launch_bounds__(256, 1) void gemm_bf16x3_ring4_kernel(GemmX3Args g) {
  __shared__ __attribute__((aligned(1024))) unsigned char ring[3 * RSTAGE];
  unsigned char* const st0 = ring;
  unsigned char* const st1 = ring + RSTAGE;
  unsigned char* const st2 = ring + 2 * RSTAGE;

  // rasterisation as in gemm_f32.hip: XCD-contiguous runs of the tile order, GM M-tiles deep groups
  constexpr int GM = 8;
  const int tiles_m = (g.M + RM - 1) / RM, tiles_n = (g.N + RN - 1) / RN;
  const int nwg = tiles_m * tiles_n;
  int wg = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int per_group = GM * tiles_n;
  const int grp = wg / per_group, in_grp = wg - grp * per_group;
  const int gm0 = grp * GM;
  const int gm_rows = (tiles_m - gm0 < GM) ? tiles_m - gm0 : GM;
  const int tm = gm0 + in_grp % gm_rows, tn = in_grp / gm_rows;
  const int m0 = tm * RM, n0 = tn * RN;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1, hi = lane >> 5, l31 = lane & 31;   // 2 x 2 waves

  // DMA plan: instruction q = wid * 6 + j of a stage covers rows 32 (q % 8) .. + 31 of operand plane q / 8 (0-2 A, 3-5 W);
  // lane -> (row l / 2 of the 32, 16-byte half l % 2): 1 KiB contiguous on both sides
  const unsigned short* src[RDMA4];                               // per-lane global address at k-step 0
  long kstride[RDMA4];                                            // elements per k-step (rows * 16)
  int lds_off[RDMA4];                                             // wave-uniform byte offset inside a stage
#pragma unroll
  for (int j = 0; j < RDMA4; ++j) {
    const int q = wid * RDMA4 + j, op = q >> 3, r32 = q & 7;
    const bool isA = op < 3;
    const int p = isA ? op : op - 3;
    const int lim = isA ? g.M : g.N;
    int grow = (isA ? m0 : n0) + r32 * 32 + (lane >> 1);
    grow = grow < lim ? grow : lim - 1;                          // rows past the edge: clamped, never stored
    src[j] = (isA ? g.A + p * g.a_plane : g.W + p * g.w_plane) + (long)grow * RK + (lane & 1) * 8;
    kstride[j] = (long)lim * RK;
    lds_off[j] = op * PL + r32 * 1024;
  }
  auto dma = [&](unsigned char* stage, int ks) {
#pragma unroll
    for (int j = 0; j < RDMA4; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + ks * kstride[j]), (lptr_t)(stage + lds_off[j]), 16, 0, 0);
  };

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Fragment reads are inline asm: for a ds_read it can see, the compiler's waitcnt pass assumes it may alias ANY
  // LDS-DMA still in flight and inserts s_waitcnt vmcnt(0) in front of it, which would throw away the ring's lead.
  // Which stage has landed is tracked by hand (vmcnt(6) + barrier below), so are the lgkmcnt waits of these reads.
  const unsigned lds0 = (unsigned)(size_t)ring;                  // low 32 bits of a flat LDS address = LDS offset
  const unsigned a_addr = lds0 + (wm * 128 + l31) * 32 + hi * 16;              // + stage + plane * PL + i * 1024
  const unsigned w_addr = lds0 + 3 * PL + (wn * 128 + l31) * 32 + hi * 16;    // + stage + plane * PL + jn * 1024
#define VX_LDS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  auto compute = [&](int stage_off) {
    bf16x8 w[3][4], a[4][3];
    const unsigned wa = w_addr + stage_off, aa = a_addr + stage_off;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      VX_LDS_READ(w[p][0], wa, p * PL);
      VX_LDS_READ(w[p][1], wa, p * PL + 1024);
      VX_LDS_READ(w[p][2], wa, p * PL + 2048);
      VX_LDS_READ(w[p][3], wa, p * PL + 3072);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) VX_LDS_READ(a[0][p], aa, p * PL);
#pragma unroll
    for (int p = 0; p < 3; ++p) VX_LDS_READ(a[1][p], aa, p * PL + 1024);
#pragma unroll
    for (int p = 0; p < 3; ++p) VX_LDS_READ(a[2][p], aa, p * PL + 2048);
#pragma unroll
    for (int p = 0; p < 3; ++p) VX_LDS_READ(a[3][p], aa, p * PL + 3072);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // LDS returns in order; the waits name the registers they release (see the 8-wave kernel)
      if (i == 0)
        asm volatile("s_waitcnt lgkmcnt(9)"
                     : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[0][2]), "+v"(w[0][3]), "+v"(w[1][0]), "+v"(w[1][1]),
                       "+v"(w[1][2]), "+v"(w[1][3]), "+v"(w[2][0]), "+v"(w[2][1]), "+v"(w[2][2]), "+v"(w[2][3]),
                       "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2])
                     :
                     : "memory");
      else if (i == 1) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]) : : "memory");
      else if (i == 2) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[2][2]) : : "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[3][0]), "+v"(a[3][1]), "+v"(a[3][2]) : : "memory");
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const int pw = t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0;   // w3 a1, w1 a3, w2 a2, w2 a1, w1 a2, w1 a1
        const int pa = t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0;
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[pw][jn], a[i][pa], acc[i][jn], 0, 0, 0);
      }
    }
  };
#undef VX_LDS_READ
  // one k-step: everything but the newest DMA group has landed -> barrier -> refill the stage freed two steps ago
  auto kstep = [&](int ks, int nks, int cur_off, unsigned char* refill) {
    // (a bare s_barrier: __syncthreads() is a fence and makes the compiler wait for vmcnt(0), i.e. for the newest DMA
    // group as well; LDS-DMA data is in LDS once vmcnt has counted it, and every ds_read of the stage being refilled
    // was consumed by MFMAs issued before this point)
    if (ks + 1 < nks) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ks + 2 < nks) dma(refill, ks + 2);
    compute(cur_off);
  };

  const int nks = g.K / RK;
  dma(st0, 0);
  if (1 < nks) dma(st1, 1);
  for (int ks = 0; ks < nks; ks += 3) {
    kstep(ks, nks, 0, st2);
    if (ks + 1 < nks) kstep(ks + 1, nks, RSTAGE, st0);
    if (ks + 2 < nks) kstep(ks + 2, nks, 2 * RSTAGE, st1);
  }

  // epilogue: acc[i][jn][4*g4 + e] = C[m0 + wm*64 + i*32 + l31][n0 + wn*128 + jn*32 + 8*g4 + 4*hi + e]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 128 + i * 32 + l31;
    if (m >= g.M) continue;
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int n = n0 + wn * 128 + jn * 32 + 8 * g4 + 4 * hi;
        if (n >= g.N) continue;
        f32x4 v = {acc[i][jn][4 * g4], acc[i][jn][4 * g4 + 1], acc[i][jn][4 * g4 + 2], acc[i][jn][4 * g4 + 3]};
        if (g.bias) {
          const f32x4 bi = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bi[e];
        }
        if (g.act == ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (g.act == ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf4(v[e]);
        }
        if (g.colscale) {
          const f32x4 cs = *reinterpret_cast<const f32x4*>(g.colscale + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] * cs[e];
        }
        if (g.resid) {
          const f32x4 rr = *reinterpret_cast<const f32x4*>(g.resid + (long)m * g.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rr[e] + v[e];
        }
        *reinterpret_cast<f32x4*>(g.C + (long)m * g.ldc + n) = v;
      }
    }
  }
}

void launch_gemm_bf16x3_ring(const GemmX3Args& g, hipStream_t s) {
  const int tiles = ((g.M + RM - 1) / RM) * ((g.N + RN - 1) / RN);
  if (tiles <= 0) return;
  hipLaunchKernelGGL(gemm_bf16x3_ring_kernel, dim3(tiles), dim3(512), 0, s, g);
}

}  // namespace vx

namespace vx {
void launch_gemm_bf16x3_ring4(const GemmX3Args& g, hipStream_t s) {
  const int tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256);
  if (tiles <= 0) return;
  hipLaunchKernelGGL(gemm_bf16x3_ring4_kernel, dim3(tiles), dim3(256), 0, s, g);
}
}  // namespace vx
