// fp32-accurate GEMM on the bf16 matrix cores: every fp32 operand is split into three bf16 terms (x = x1 + x2 + x3,
// 8 + 8 + 8 mantissa bits, exact to 2^-27 relative) and the six products that matter are accumulated in fp32:
//
//     a.b ~= a1b1 + (a1b2 + a2b1) + (a2b2 + a1b3 + a3b1)            dropped terms are <= 2^-27 |a.b|
//
// Each bf16 x bf16 product is exact in fp32, the accumulation is the MFMA's fp32 accumulator, so the result carries the
// same error class as an fp32 fma chain (fewer roundings, in fact: 6 accumulations per 16 k instead of 16) -- bit-exact
// token parity with the fp32 reference path is kept (tests/test_gpu_parity.py).  Cost: 6 x v_mfma_f32_32x32x16_bf16
// (32 cycles each, 16 k) = 192 cycles per 32x32x16 block of work, against 8 x 64 = 512 cycles on v_mfma_f32_32x32x2_f32:
// 2.67x more matrix throughput for the NAR / prefill projections, which are 45 % of end-to-end time.
//
// Operands arrive pre-split as three bf16 planes in K-tile-major order [3][K/32][rows][32] (weights once at load,
// activations by split3_kernel), so a staged 128 x 32 tile is one contiguous 8 KiB run of full cache lines.  Tile 128 x 128 x 32, 4 waves as 2 x 2, 2 x 2 MFMA tiles per wave; LDS rows of 64 B with the 16-B chunk index
// XOR-swizzled by (row/4)%4 so both the staging ds_write_b128 and the fragment ds_read_b128 are conflict-free; the transposed product (A operand = W) gives each lane one output row
// and float4 runs of n for the epilogue, as in gemm_f32.hip.
#include <algorithm>

#include "vx_common.h"

namespace vx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int XB_M = 128, XB_N = 128, XB_K = 32, XB_LD = 64;     // LDS row stride in bytes: unpadded, XOR-swizzled chunks

__device__ __forceinline__ float gelu_erf2(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// ---------------------------------------------------------------------------------------------------------------
// x[rows][K] fp32 (row r read at gather ? gather[r] : r) -> planes[p][rows][K] bf16, p = 0..2
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, int ldx, long rows, int K,
                                                     const int* __restrict__ gather, unsigned short* __restrict__ planes,
                                                     long plane_stride) {
  // Output layout is K-TILE-MAJOR: plane[kt][row][32]  (kt = k / 32).  The GEMM stages 128 rows x 32 k per plane and
  // K tile; with this layout that is ONE contiguous 8 KiB run (full 128-B lines, 1 KiB per wave load) instead of 128
  // half-used lines 2*K bytes apart -- the row-major image made the kernel L2-bandwidth bound (each line was fetched
  // twice, once per K tile).  Work item = (kt, row, 16-byte chunk), chunk fastest, so writes are contiguous too.
  const long total = (long)(K / 32) * rows * 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i & 3);
    const long rr = i >> 2;
    const long kt = rr / rows, r = rr - kt * rows;
    const long src = gather ? gather[r] : r;
    const float* xp = x + src * ldx + kt * 32 + ch * 8;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(xp);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(xp + 4);
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    bf16x8 p1, p2, p3;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const __bf16 a1 = (__bf16)v[e];                          // RNE (v_cvt_pk_bf16_f32)
      const float r1 = v[e] - (float)a1;                       // exact
      const __bf16 a2 = (__bf16)r1;
      const float r2 = r1 - (float)a2;                         // exact
      p1[e] = a1; p2[e] = a2; p3[e] = (__bf16)r2;
    }
    unsigned short* o = planes + i * 8;                        // == ((kt * rows + r) * 32 + ch * 8)
    *reinterpret_cast<bf16x8*>(o) = p1;
    *reinterpret_cast<bf16x8*>(o + plane_stride) = p2;
    *reinterpret_cast<bf16x8*>(o + 2 * plane_stride) = p3;
  }
}

void launch_split3(const float* x, int ldx, long rows, int K, const int* gather, unsigned short* planes,
                   long plane_stride, hipStream_t s) {
  if (rows <= 0) return;
  const long total = rows * (K / 8);
  const int grid = (int)std::min<long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(split3_kernel, dim3(grid), dim3(256), 0, s, x, ldx, rows, K, gather, planes, plane_stride);
}

// ---------------------------------------------------------------------------------------------------------------
// C[m][n] = resid[m][n] + colscale[n] * act( sum_k A[m][k] W[n][k] + bias[n] ),  A, W given as 3 bf16 planes
// ---------------------------------------------------------------------------------------------------------------
// V = 0 product kernel.  Timing probes for tools/gemm_bench.py (results are meaningless): V = 1 no global loads / LDS
// writes after the first tile; V = 2 no MFMAs; V = 3 no fragment reads after the first tile (MFMAs on stale registers).
template <int V>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(GemmX3Args g) {
  __shared__ __attribute__((aligned(16))) unsigned char As[3][XB_M * XB_LD];
  __shared__ __attribute__((aligned(16))) unsigned char Ws[3][XB_N * XB_LD];

  // same rasterisation as gemm_f32.hip: XCD-contiguous runs, GM M-tiles deep groups
  constexpr int GM = 16;
  const int tiles_m = (g.M + XB_M - 1) / XB_M, tiles_n = (g.N + XB_N - 1) / XB_N;
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
  const int m0 = tm * XB_M, n0 = tn * XB_N;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1, hi = lane >> 5, l31 = lane & 31;

  // staging: per plane 128 rows x 4 chunks of 16 B; thread handles (row = idx>>2, chunk = idx&3) for idx = tid, tid+256
  const unsigned short* aptr[2];
  const unsigned short* wptr[2];
  int lds_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i, row = idx >> 2, ch = idx & 3;
    int m = m0 + row;
    m = m < g.M ? m : g.M - 1;
    int n = n0 + row;
    n = n < g.N ? n : g.N - 1;
    aptr[i] = g.A + (long)m * XB_K + ch * 8;                    // k-tile-major planes: [kt][row][32]
    wptr[i] = g.W + (long)n * XB_K + ch * 8;
    lds_off[i] = row * XB_LD + ((ch ^ ((row >> 2) & 3)) * 16);   // 16-B chunk index XOR (row/4)%4: conflict-free for
                                                                  // the ds_write_b128 here AND the ds_read_b128 below
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 ra[3][2], rw[3][2];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ra[p][i] = *reinterpret_cast<const u32x4*>(aptr[i] + p * g.a_plane);
      rw[p][i] = *reinterpret_cast<const u32x4*>(wptr[i] + p * g.w_plane);
    }

  const int nk = g.K / XB_K;
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    if (V != 1 || kt == 0) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          *reinterpret_cast<u32x4*>(&As[p][lds_off[i]]) = ra[p][i];
          *reinterpret_cast<u32x4*>(&Ws[p][lds_off[i]]) = rw[p][i];
        }
    }
    __syncthreads();
    if (V != 1 && kt + 1 < nk) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ra[p][i] = *reinterpret_cast<const u32x4*>(aptr[i] + p * g.a_plane + (long)(kt + 1) * g.M * XB_K);
          rw[p][i] = *reinterpret_cast<const u32x4*>(wptr[i] + p * g.w_plane + (long)(kt + 1) * g.N * XB_K);
        }
    }
#pragma unroll
    for (int s = 0; s < XB_K / 16; ++s) {
      // fragment of k-step s: lane supplies row (l31) and k = 16 s + 8 hi + 0..7  -> 16-byte chunk 2 s + hi of the row
      bf16x8 w[3][2];
      if (V != 3 || kt == 0)
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
          w[p][jn] = *reinterpret_cast<const bf16x8*>(&Ws[p][(wn * 64 + jn * 32 + l31) * XB_LD + (((2 * s + hi) ^ ((l31 >> 2) & 3)) * 16)]);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        bf16x8 a[3];                                             // only one A row-tile live at a time (VGPR budget)
        if (V != 3 || kt == 0)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          a[p] = *reinterpret_cast<const bf16x8*>(&As[p][(wm * 64 + i * 32 + l31) * XB_LD + (((2 * s + hi) ^ ((l31 >> 2) & 3)) * 16)]);
        // transposed product (A operand = W rows): small terms first, then the leading one
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
          f32x16 c = acc[i][jn];
          if (V == 2) {
            c[0] += (float)(a[0][0] + a[1][1] + a[2][2]) + (float)(w[0][jn][0] + w[1][jn][1] + w[2][jn][2]);
            acc[i][jn] = c;
            continue;
          }
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2][jn], a[0], c, 0, 0, 0);   // w3 a1
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0][jn], a[2], c, 0, 0, 0);   // w1 a3
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1][jn], a[1], c, 0, 0, 0);   // w2 a2
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1][jn], a[0], c, 0, 0, 0);   // w2 a1
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0][jn], a[1], c, 0, 0, 0);   // w1 a2
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0][jn], a[0], c, 0, 0, 0);   // w1 a1
          acc[i][jn] = c;
        }
      }
    }
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
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf2(v[e]);
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

void launch_gemm_bf16x3(const GemmX3Args& g, hipStream_t s) {
  const int tiles = ((g.M + XB_M - 1) / XB_M) * ((g.N + XB_N - 1) / XB_N);
  if (tiles <= 0) return;
  hipLaunchKernelGGL(gemm_bf16x3_kernel<0>, dim3(tiles), dim3(256), 0, s, g);
}

#ifdef VX_DEV_PROBES   // timing probes: tools-only build (vall-e-x_amd/_build.py --dev), never in the product library
void launch_gemm_bf16x3_probe(const GemmX3Args& g, int variant, hipStream_t s) {
  const int tiles = ((g.M + XB_M - 1) / XB_M) * ((g.N + XB_N - 1) / XB_N);
  if (variant == 1) hipLaunchKernelGGL(gemm_bf16x3_kernel<1>, dim3(tiles), dim3(256), 0, s, g);
  else if (variant == 2) hipLaunchKernelGGL(gemm_bf16x3_kernel<2>, dim3(tiles), dim3(256), 0, s, g);
  else hipLaunchKernelGGL(gemm_bf16x3_kernel<3>, dim3(tiles), dim3(256), 0, s, g);
}

#endif

}  // namespace vx
