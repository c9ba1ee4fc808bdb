// bf16x3 GEMM (see gemm_bf16x3.hip for the arithmetic) with the operand tiles brought into LDS by the DMA path:
// global_load_lds_dwordx4 writes 16 B per lane straight into LDS (destination = M0 + 16 * lane), so a tile neither
// passes through VGPRs nor costs ds_write issue slots, and -- with two LDS stages -- lands while the previous tile's
// MFMAs run.  The register-staged kernel loses ~28 % of its time to the [barrier | 12 x ds_write_b128 | barrier] section
// of every K tile (tools/gemm_bench.py probes; ds_write_b128 sustains only ~75 B/clk per CU, tools/ubench/lds_patterns).
//
//   C[m][n] = resid[m][n] + colscale[n] * act( sum_k A[m][k] W[n][k] + bias[n] ),  A, W as 3 bf16 planes, K-tile-major
//
// Workgroup: 512 threads = 8 waves as 4 (M) x 2 (N), tile 256 x 128 x 32, wave tile 64 x 64 (2 x 2 MFMA tiles), one
// workgroup per CU (2 waves per SIMD).  LDS: 2 stages x (3 planes x (256 + 128) rows x 64 B) = 144 KiB.
// A stage is 72 wave-level DMA instructions of 1 KiB (16 rows x 64 B, contiguous in the K-tile-major planes AND in LDS),
// 9 per wave.  LDS rows are 64 B with the 16-B chunk index XOR-swizzled by (row/4)%4 exactly as in gemm_bf16x3.hip
// (conflict-free ds_read_b128 fragments); the swizzle is applied on the GLOBAL side: lane l of a DMA instruction fills
// LDS slot (row l/4, chunk l%4) and therefore fetches global chunk (l%4) ^ ((l/16)%4) of that row.
//
// Per K tile: s_waitcnt vmcnt(0) (my DMA landed) | barrier (everybody's landed, everybody finished the other stage) |
// issue the next tile's DMA into the other stage | 2 k-steps x (18 ds_read_b128 + 24 MFMAs) per wave.
#include <algorithm>

#include "vx_common.h"

namespace vx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int DM = 256, DN = 128, DK = 32, DLD = 64;            // tile; LDS row stride in bytes
constexpr int A_PL = DM * DLD, W_PL = DN * DLD;                  // one plane of a stage: 16 KiB / 8 KiB
constexpr int STAGE = 3 * A_PL + 3 * W_PL;                       // 72 KiB
constexpr int NDMA = 9;                                          // DMA instructions per wave and stage

__device__ __forceinline__ float gelu_erf3(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

}  // namespace

// V = 0 product kernel.  Timing probes (tools/gemm_bench.py, results meaningless): V = 1 no DMA after the first two tiles
// (compute + fragment reads + barriers only); V = 2 no MFMAs; V = 3 fragments read once (MFMAs on stale registers);
// V = 4 no barriers / vmcnt waits (racy).
template <int V>
__global__ __launch_bounds__(512, 1) void gemm_bf16x3_dma_kernel(GemmX3Args g) {
  __shared__ __attribute__((aligned(1024))) unsigned char stage0[STAGE];
  __shared__ __attribute__((aligned(1024))) unsigned char stage1[STAGE];

  // rasterisation as in gemm_f32.hip: XCD-contiguous runs of the tile order, GM M-tiles deep groups
  constexpr int GM = 8;
  const int tiles_m = (g.M + DM - 1) / DM, tiles_n = (g.N + DN - 1) / DN;
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
  const int m0 = tm * DM, n0 = tn * DN;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1, hi = lane >> 5, l31 = lane & 31;

  // DMA plan: instruction q = wid * 9 + j of a stage; q < 48: A plane q / 16, rows 16 (q % 16) ..; else W plane
  // (q - 48) / 8, rows 16 ((q - 48) % 8) ...  Lane -> (row l / 4 of the 16, LDS chunk slot l % 4).
  const unsigned short* src[NDMA];                               // per-lane global address at K tile 0
  int lds_off[NDMA];                                             // wave-uniform LDS byte offset inside a stage
  long kstep[NDMA];                                              // elements to advance per K tile (rows * 32)
#pragma unroll
  for (int j = 0; j < NDMA; ++j) {
    const int q = wid * NDMA + j;
    const bool isA = q < 48;
    const int qq = isA ? q : q - 48;
    const int p = isA ? qq >> 4 : qq >> 3, r16 = isA ? qq & 15 : qq & 7;
    const int row = r16 * 16 + (lane >> 2);
    const int ch = (lane & 3) ^ ((lane >> 4) & 3);               // global chunk that belongs in this lane's LDS slot
    int grow = (isA ? m0 : n0) + row;
    const int lim = isA ? g.M : g.N;
    grow = grow < lim ? grow : lim - 1;                          // rows past the edge: clamped, never stored
    src[j] = (isA ? g.A + p * g.a_plane : g.W + p * g.w_plane) + (long)grow * DK + ch * 8;
    kstep[j] = (long)lim * DK;
    lds_off[j] = (isA ? p * A_PL : 3 * A_PL + p * W_PL) + r16 * 1024;
  }
  auto dma = [&](unsigned char* stage, int kt) {
#pragma unroll
    for (int j = 0; j < NDMA; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + kt * kstep[j]), (lptr_t)(stage + lds_off[j]), 16, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment offsets inside a plane: row * 64 + ((2 s + hi) ^ ((row / 4) % 4)) * 16; rows are multiples of 32 + l31
  const int sw = (l31 >> 2) & 3;
  const int a_row = (wm * 64 + l31) * DLD, w_row = (wn * 64 + l31) * DLD;
  // fragments of k-step s: w[plane][jn], a[i][plane]; the next k-step's 18 reads are issued before this one's MFMAs
  auto frags = [&](const unsigned char* stage, int s, bf16x8 (&w)[3][2], bf16x8 (&a)[2][3]) {
    const int coff = ((2 * s + hi) ^ sw) * 16;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int jn = 0; jn < 2; ++jn)
        w[p][jn] = *reinterpret_cast<const bf16x8*>(stage + 3 * A_PL + p * W_PL + w_row + jn * 32 * DLD + coff);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) a[i][p] = *reinterpret_cast<const bf16x8*>(stage + p * A_PL + a_row + i * 32 * DLD + coff);
  };
  auto mfmas = [&](const bf16x8 (&w)[3][2], const bf16x8 (&a)[2][3]) {
    // transposed product (A operand = W rows); per accumulator the six terms keep the order of gemm_bf16x3.hip (small
    // terms first), but the two accumulators of an i are interleaved so no MFMA waits on the one before it
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const int pw = t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0;   // w3 a1, w1 a3, w2 a2, w2 a1, w1 a2, w1 a1
        const int pa = t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0;
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
          if (V == 2) acc[i][jn][0] += (float)w[pw][jn][0] * (float)a[i][pa][0];
          else acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[pw][jn], a[i][pa], acc[i][jn], 0, 0, 0);
        }
      }
    }
  };
  // One K tile.  The next tile's DMA goes out FIRST: its landing time, not the LDS latency of the first fragments, is
  // what the barrier at the top of the next tile waits for (issuing it behind the fragment reads cost 6 %).
  bf16x8 w0[3][2], a0[2][3], w1[3][2], a1[2][3];
  auto ktile = [&](const unsigned char* stage, unsigned char* other, int kt_next, bool more, bool first) {
    if (more && (V != 1 || kt_next < 2)) dma(other, kt_next);
    if (V != 3 || first) {
      frags(stage, 0, w0, a0);
      frags(stage, 1, w1, a1);
    }
    mfmas(w0, a0);
    mfmas(w1, a1);
  };
  auto rendezvous = [&]() {
    if (V == 4) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  const int nk = g.K / DK;
  dma(stage0, 0);
  for (int kt = 0; kt < nk; kt += 2) {
    rendezvous();
    ktile(stage0, stage1, kt + 1, kt + 1 < nk, kt == 0);
    if (kt + 1 < nk) {
      rendezvous();
      ktile(stage1, stage0, kt + 2, kt + 2 < nk, false);
    }
  }
  if (V == 4) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }

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
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf3(v[e]);
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

void launch_gemm_bf16x3_dma(const GemmX3Args& g, hipStream_t s) {
  const int tiles = ((g.M + DM - 1) / DM) * ((g.N + DN - 1) / DN);
  if (tiles <= 0) return;
  hipLaunchKernelGGL(gemm_bf16x3_dma_kernel<0>, dim3(tiles), dim3(512), 0, s, g);
}

#ifdef VX_DEV_PROBES   // timing probes: tools-only build (vall-e-x_amd/_build.py --dev), never in the product library
void launch_gemm_bf16x3_dma_probe(const GemmX3Args& g, int variant, hipStream_t s) {
  const int tiles = ((g.M + DM - 1) / DM) * ((g.N + DN - 1) / DN);
  if (tiles <= 0) return;
  const dim3 grid(tiles), block(512);
  if (variant == 1) hipLaunchKernelGGL(gemm_bf16x3_dma_kernel<1>, grid, block, 0, s, g);
  else if (variant == 2) hipLaunchKernelGGL(gemm_bf16x3_dma_kernel<2>, grid, block, 0, s, g);
  else if (variant == 3) hipLaunchKernelGGL(gemm_bf16x3_dma_kernel<3>, grid, block, 0, s, g);
  else hipLaunchKernelGGL(gemm_bf16x3_dma_kernel<4>, grid, block, 0, s, g);
}

#endif

}  // namespace vx
