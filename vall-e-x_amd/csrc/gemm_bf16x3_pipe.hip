// bf16x3 GEMM (arithmetic: gemm_bf16x3.hip), fourth generation: the ring kernel's async k-step stages, plus fragment
// prefetch ACROSS the per-stage rendezvous so that no wave ever waits for LDS data with the matrix pipe idle.
//
// tools/ubench/mfma_overlap shows that ds_read_b128 and VALU work hide completely under a saturated matrix pipe; what the
// earlier kernels lose (they sit at ~50 % of the sustained MFMA rate with < 1 other instruction per MFMA) is the
// rendezvous per tile: wait for the DMA, barrier, then every wave of the CU waits for its first fragments at once.
// Here iteration j (one k-step of 16) runs
//     12 MFMAs of k-step j                              (fragments already in registers)
//     s_waitcnt vmcnt(5) | s_barrier                    stage j+1 has landed for everybody (its DMA left two steps ago)
//     DMA for k-step j+3 into the stage freed by j-1    (4-deep ring, 36 KiB per stage)
//     18 ds_read_b128: fragments of k-step j+1          (second register set)
//     12 MFMAs of k-step j                              (cover the LDS latency of those reads)
// Tile 256 (M) x 128 (N), 512 threads = 8 waves as 4 x 2, wave tile 64 x 64; planes k-step-major [K/16][rows][16]
// (split3_k16_kernel); a stage is 36 wave-level global_load_lds_dwordx4 of 1 KiB -- every wave issues 5 (the last four
// are duplicates of the first four, same bytes to the same place) so that one vmcnt immediate fits all waves.
#include "vx_common.h"

namespace vx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int PM = 256, PN = 128, PK = 16;
constexpr int APL = PM * 32, WPL = PN * 32;                      // one plane of a stage: 8 KiB / 4 KiB
constexpr int PSTAGE = 3 * APL + 3 * WPL;                        // 36 KiB
constexpr int PDMA = 5;

__device__ __forceinline__ float gelu_erf5(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

}  // namespace

__global__ __launch_bounds__(512, 1) void gemm_bf16x3_pipe_kernel(GemmX3Args g) {
  __shared__ __attribute__((aligned(1024))) unsigned char ring[4 * PSTAGE];

  // rasterisation as in gemm_f32.hip: XCD-contiguous runs of the tile order, GM M-tiles deep groups
  constexpr int GM = 8;
  const int tiles_m = (g.M + PM - 1) / PM, tiles_n = (g.N + PN - 1) / PN;
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
  const int m0 = tm * PM, n0 = tn * PN;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1, hi = lane >> 5, l31 = lane & 31;

  // DMA plan: instruction q of a stage (q < 36) covers 32 rows x 32 B = 1 KiB: q < 24: A plane q / 8, rows 32 (q % 8) ..;
  // else W plane (q - 24) / 4, rows 32 ((q - 24) % 4) ...  Wave w issues q = w + 8 j, j = 0..4, taken modulo 36.
  const unsigned short* src[PDMA];
  long kstride[PDMA];
  int lds_off[PDMA];
#pragma unroll
  for (int j = 0; j < PDMA; ++j) {
    int q = wid + 8 * j;
    q = q < 36 ? q : q - 36;                                     // the four duplicates
    const bool isA = q < 24;
    const int qq = isA ? q : q - 24;
    const int p = isA ? qq >> 3 : qq >> 2, r32 = isA ? qq & 7 : qq & 3;
    const int lim = isA ? g.M : g.N;
    int grow = (isA ? m0 : n0) + r32 * 32 + (lane >> 1);
    grow = grow < lim ? grow : lim - 1;                          // rows past the edge: clamped, never stored
    src[j] = (isA ? g.A + p * g.a_plane : g.W + p * g.w_plane) + (long)grow * PK + (lane & 1) * 8;
    kstride[j] = (long)lim * PK;
    lds_off[j] = (isA ? p * APL : 3 * APL + p * WPL) + r32 * 1024;
  }
  auto dma = [&](int buf, int ks) {
#pragma unroll
    for (int j = 0; j < PDMA; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + ks * kstride[j]), (lptr_t)(ring + buf * PSTAGE + lds_off[j]), 16, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment reads are inline asm (see gemm_bf16x3_ring.hip: a visible ds_read makes the compiler wait for vmcnt(0))
  const unsigned lds0 = (unsigned)(size_t)ring;
  const unsigned a_addr = lds0 + (wm * 64 + l31) * 32 + hi * 16;             // + stage + plane * APL + i * 1024
  const unsigned w_addr = lds0 + 3 * APL + (wn * 64 + l31) * 32 + hi * 16;   // + stage + plane * WPL + jn * 1024
#define VX_LDS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  struct Frag { bf16x8 w[3][2], a[2][3]; };
  auto read_frags = [&](Frag& f, int buf) {
    const unsigned wa = w_addr + buf * PSTAGE, aa = a_addr + buf * PSTAGE;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      VX_LDS_READ(f.w[p][0], wa, p * WPL);
      VX_LDS_READ(f.w[p][1], wa, p * WPL + 1024);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      VX_LDS_READ(f.a[0][p], aa, p * APL);
      VX_LDS_READ(f.a[1][p], aa, p * APL + 1024);
    }
  };
#undef VX_LDS_READ
  // all 12 fragment registers are released by this wait (in/out operands keep consumers behind it)
  auto wait_frags = [&](Frag& f) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.w[0][0]), "+v"(f.w[0][1]), "+v"(f.w[1][0]), "+v"(f.w[1][1]), "+v"(f.w[2][0]), "+v"(f.w[2][1]),
                   "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[0][2]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.a[1][2])
                 :
                 : "memory");
  };
  auto mfmas = [&](const Frag& f, int i) {
    // transposed product (A operand = W rows); six terms per accumulator in the order of gemm_bf16x3.hip, two accumulators interleaved
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const int pw = t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0;     // w3 a1, w1 a3, w2 a2, w2 a1, w1 a2, w1 a1
      const int pa = t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0;
#pragma unroll
      for (int jn = 0; jn < 2; ++jn)
        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.w[pw][jn], f.a[i][pa], acc[i][jn], 0, 0, 0);
    }
  };

  const int nks = g.K / PK;
  // one k-step; `cur` holds its fragments, `nxt` receives those of k-step ks + 1 (stage (ks + 1) % 4)
  auto kstep = [&](int ks, Frag& cur, Frag& nxt) {
    mfmas(cur, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < nks) {
      // stage ks+1 was requested two iterations ago; only the newest stage (ks+2, 5 instructions per wave) may be in flight
      if (ks + 2 < nks) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                              // everybody's part of stage ks+1 is in LDS, and everybody
      asm volatile("" ::: "memory");                             // has its fragments of stage ks-1 (issued an iteration ago)
      if (ks + 3 < nks) dma((ks + 3) & 3, ks + 3);
      read_frags(nxt, (ks + 1) & 3);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfmas(cur, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < nks) wait_frags(nxt);                           // long since landed: 12 MFMAs were issued in between
  };

  Frag f0, f1;
  dma(0, 0);
  if (1 < nks) dma(1, 1);
  if (2 < nks) dma(2, 2);
  if (2 < nks) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if (1 < nks) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_frags(f0, 0);
  wait_frags(f0);
  for (int ks = 0; ks < nks; ks += 2) {
    kstep(ks, f0, f1);
    if (ks + 1 < nks) kstep(ks + 1, f1, f0);
  }

  // epilogue: acc[i][jn][4*g4 + e] = C[m0 + wm*64 + i*32 + l31][n0 + wn*64 + jn*32 + 8*g4 + 4*hi + e]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + l31;
    if (m >= g.M) continue;
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int n = n0 + wn * 64 + jn * 32 + 8 * g4 + 4 * hi;
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
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf5(v[e]);
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

void launch_gemm_bf16x3_pipe(const GemmX3Args& g, hipStream_t s) {
  const int tiles = ((g.M + PM - 1) / PM) * ((g.N + PN - 1) / PN);
  if (tiles <= 0) return;
  hipLaunchKernelGGL(gemm_bf16x3_pipe_kernel, dim3(tiles), dim3(512), 0, s, g);
}

}  // namespace vx
