// fp32-in / fp32-accumulate MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fma chain).
//
// Used for every dense projection on the full-sequence paths: AR prefill, the 7 NAR stages, the Vocos
// backbone/head and the ISTFT-as-DFT product.  Replaces F.linear / torch._C._nn.linear
// (modules/activation.py:144,166; modules/transformer.py:371-373; models/vallex.py:568,677).
//
//   C[m][n] = resid[m][n] + colscale[n] * act( sum_k A[m][k] * W[n][k] + bias[n] )
//
// Tile: 128(M) x 128(N) x 32(K) per 256-thread workgroup; 4 waves as 2x2, each wave 64x64 = 2x2 MFMA
// 32x32 tiles (64 accumulator VGPRs).  Operands are staged global -> registers -> LDS with the next K-tile's
// global loads in flight under the current tile's MFMAs.  LDS rows are padded to 36 floats so the
// ds_read_b128 fragment reads (16-lane groups, 64-dword bank row) are conflict-free.
//
// k-permutation: one ds_read_b128 gives a lane 4 consecutive k of its row; lanes 0-31 take k0..k0+3 and lanes
// 32-63 take k0+4..k0+7.  MFMA step j then contracts k = k0 + 4*(lane>>5) + j for BOTH operands, which is a
// bijection of the 8 k's onto (step, half) -- the sum is the same set of products (order differs only).
#include "vx_common.h"

namespace vx {

constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = 36;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Ws[BN * LDS_LD];

  // Tile rasterisation for L2 reuse.  (1) XCD-aware: consecutive workgroup ids land on different XCDs (id % 8), each
  // with a private 4 MiB L2, so ids are remapped to give every XCD one contiguous run of the tile order.
  // (2) Inside that order tiles are grouped GM M-tiles deep: m runs fastest inside a group, then n, then the next
  // group.  The ~96 workgroups an XCD keeps resident then cover a GM x 6 patch of the output: GM A-panels and 6
  // W-panels are shared through L2 instead of 96 A-panels + 1 W-panel (4x fewer L2 fills per K step).
  constexpr int GM = 16;
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  int wg = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int per_group = GM * tiles_n;
  const int grp = wg / per_group, in_grp = wg - grp * per_group;
  const int gm0 = grp * GM;
  const int gm_rows = (tiles_m - gm0 < GM) ? tiles_m - gm0 : GM;   // last group may be short
  const int tm = gm0 + in_grp % gm_rows, tn = in_grp / gm_rows;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, hi = lane >> 5, l31 = lane & 31;

  // staging map: thread -> rows (tid>>3) + 32*i, float4 column (tid&7)
  const int srow = tid >> 3, scol = (tid & 7) * 4;
  const float* aptr[4];
  const float* wptr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + srow + 32 * i;
    m = m < g.M ? m : g.M - 1;
    const long arow = g.row_gather ? g.row_gather[m] : m;
    aptr[i] = g.A + arow * (long)g.lda + scol;
    int wn_row = n0 + srow + 32 * i;                               // N need not fill the last tile: rows past N are
    wn_row = wn_row < g.N ? wn_row : g.N - 1;                      // clamped here and never stored (epilogue guard)
    wptr[i] = g.W + (long)wn_row * g.ldw + scol;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[4], rw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ra[i] = *reinterpret_cast<const f32x4*>(aptr[i]);
    rw[i] = *reinterpret_cast<const f32x4*>(wptr[i]);
  }

  const int nk = g.K / BK;
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();   // previous tile's fragment reads are done
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f32x4*>(&As[(srow + 32 * i) * LDS_LD + scol]) = ra[i];
      *reinterpret_cast<f32x4*>(&Ws[(srow + 32 * i) * LDS_LD + scol]) = rw[i];
    }
    __syncthreads();
    if (kt + 1 < nk) {   // next tile's loads fly under this tile's 64 MFMAs
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const f32x4*>(aptr[i] + (kt + 1) * BK);
        rw[i] = *reinterpret_cast<const f32x4*>(wptr[i] + (kt + 1) * BK);
      }
    }
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      f32x4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const f32x4*>(&As[(wm * 64 + i * 32 + l31) * LDS_LD + kb * 8 + hi * 4]);
        b[i] = *reinterpret_cast<const f32x4*>(&Ws[(wn * 64 + i * 32 + l31) * LDS_LD + kb * 8 + hi * 4]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jn = 0; jn < 2; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[jn][j], a[i][j], acc[i][jn], 0, 0, 0);
    }
  }

  // The MFMA computes the TRANSPOSED tile (A operand = W rows, B operand = activation rows), so a lane owns one output
  // row m and 4-element runs of consecutive n:
  //   acc[i][jn][4*g4 + e] = C[m0 + wm*64 + i*32 + l31][n0 + wn*64 + jn*32 + 8*g4 + 4*hi + e]
  // -> 16 float4 stores per lane instead of 64 dword stores (the epilogue is store-issue bound), bias / column scale /
  // residual as float4 too.
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + l31;
    if (m >= g.M) continue;
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int n = n0 + wn * 64 + jn * 32 + 8 * g4 + 4 * hi;
        if (n >= g.N) continue;                                      // N % 4 == 0: a float4 is all in or all out
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
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        } else if (g.act == ACT_ELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : expm1f(v[e]);
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

void launch_gemm_f32(const GemmArgs& g, hipStream_t s) {
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  if (tiles <= 0) return;
  hipLaunchKernelGGL(gemm_f32_kernel, dim3(tiles), dim3(256), 0, s, g);
}

}  // namespace vx
