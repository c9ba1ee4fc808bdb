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
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

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
  int tm, tn;
  tile_walk(blockIdx.x, tiles_m, tiles_n, GM, g.walk, tm, tn);
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
          const f32x4 rr = *reinterpret_cast<const f32x4*>(g.resid + (long)(g.resid_rows ? g.resid_rows[m] : m) * g.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rr[e] + v[e];
        }
        *reinterpret_cast<f32x4*>(g.C + (long)m * g.ldc + n) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same GEMM for the long row sets of the reference-arithmetic mode (vx_config.arith = f32: every transformer projection of the
// AR prefill and of the 7 NAR stages, M = 12 288 .. 31 616 packed rows, N a multiple of 128): operand tiles go global -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass) into TWO LDS stages, ONE rendezvous per K tile, fragments of k-block
// kb + 1 read while the 16 MFMAs of kb run.  Tile 256 (M) x 128 (N) x 32 (K), 8 waves 4 x 2, wave tile 64 x 64 as in gemm_f32_kernel
// -- per output element the SAME sequence of v_mfma_f32_32x32x2_f32 with the same operands (k-blocks of 8 in order, the k-permutation
// of the header comment), so the two kernels agree bit for bit.
// LDS image: a row is the 32 floats of the K tile = 128 B = eight 16-B chunks; chunk c of row r sits in slot c ^ ((r >> 1) & 7): the 16
// lanes a ds_read_b128 serves together (rows r .. r + 15, one chunk index) land in 16 different 16-B slots of the 256-B bank row.  The
// swizzle is applied on the GLOBAL side of the DMA (an instruction writes 1 KiB = 8 rows linearly: lane -> row l >> 3, slot l & 7, and
// fetches chunk slot ^ swizzle of its row; every row is still one full 128-B line).
namespace {
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;
}  // namespace

template <int TM, int TN>
__global__ __launch_bounds__(TM * 2, TM == 256 ? 1 : 2) void gemm_f32_dma_kernel(GemmArgs g) {
  constexpr int NWAVE = TM / 32;                                  // 8 (4 x 2) or 4 (2 x 2) waves
  constexpr int NJ = TN / 64;                                     // 32-column blocks of a wave: 2 (wave tile 64 x 64) or 4 (64 x 128)
  constexpr int A_BYTES = TM * 128, W_BYTES = TN * 128, STAGE = A_BYTES + W_BYTES;     // 64 / 48 / 32 KiB per stage
  constexpr int NDMA = STAGE / (NWAVE * 1024);                    // 1 KiB DMA instructions per wave and stage: 8 / 6 / 8
  constexpr int NA = TM / 8;                                      // ... the first NA of a stage fetch A (8 rows each)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

  constexpr int GM = TM == 256 ? 8 : 16;
  const int tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + TN - 1) / TN;
  int tm, tn;
  tile_walk(blockIdx.x, tiles_m, tiles_n, GM, g.walk, tm, tn);
  const int m0 = tm * TM, n0 = tn * TN;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1, hi = lane >> 5, l31 = lane & 31;

  // DMA plan: instruction q = wid * NDMA + j of a stage covers rows 8 q' .. 8 q' + 7 of A (q < NA) or of W
  const float* src[NDMA];
  int lds_off[NDMA];
#pragma unroll
  for (int j = 0; j < NDMA; ++j) {
    const int q = wid * NDMA + j;
    const bool isA = q < NA;
    const int row = (isA ? q : q - NA) * 8 + (lane >> 3);
    const int ch = (lane & 7) ^ ((row >> 1) & 7);
    if (isA) {
      int m = m0 + row;
      m = m < g.M ? m : g.M - 1;                                  // rows past M: clamped, never stored
      const long arow = g.row_gather ? g.row_gather[m] : m;
      src[j] = g.A + arow * (long)g.lda + ch * 4;
    } else {
      int n = n0 + row;
      n = n < g.N ? n : g.N - 1;
      src[j] = g.W + (long)n * g.ldw + ch * 4;
    }
    lds_off[j] = (isA ? 0 : A_BYTES) + (isA ? q : q - NA) * 1024;
  }
  auto dma = [&](int stage, int kt) {
#pragma unroll
    for (int j = 0; j < NDMA; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + kt * BK), (lptr_t)(lds + stage * STAGE + lds_off[j]), 16, 0, 0);
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: row (wm * 64 + i * 32 + l31) of A, (wn * TN / 2 + i * 32 + l31) of W; chunk 2 kb + hi, swizzled by (l31 >> 1) & 7
  const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)lds;
  const unsigned swz = (unsigned)((l31 >> 1) & 7);
  const unsigned a_base = lds0 + (unsigned)((wm * 64 + l31) * 128), w_base = lds0 + (unsigned)(A_BYTES + (wn * (TN / 2) + l31) * 128);
  // inline-asm reads: for a C++ LDS load the compiler cannot tell the stage being filled from the stage being read and waits for the
  // DMA (vmcnt(0)) in front of every fragment read; the hazards are ordered by the rendezvous below
  auto frags = [&](int stage, int kb, f32x4 (&a)[2], f32x4 (&b)[NJ]) {
    const unsigned co = (unsigned)(stage * STAGE) + ((((unsigned)(2 * kb + hi)) ^ swz) << 4);
    asm volatile("ds_read_b128 %0, %1" : "=v"(a[0]) : "v"(a_base + co));
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(a[1]) : "v"(a_base + co));
    asm volatile("ds_read_b128 %0, %1" : "=v"(b[0]) : "v"(w_base + co));
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(b[1]) : "v"(w_base + co));
    if constexpr (NJ == 4) {
      asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(b[2]) : "v"(w_base + co));
      asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(b[3]) : "v"(w_base + co));
    }
  };
  auto mfmas = [&](const f32x4 (&a)[2], const f32x4 (&b)[NJ]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[jn][j], a[i][j], acc[i][jn], 0, 0, 0);
  };
  // wait until the OLDER of two fragment sets has arrived (the NFR reads of the younger one stay in flight); the set is named as
  // in / out operands so that neither the compiler nor the scheduler moves its consumers above the wait
  auto wait_set = [&](f32x4 (&a)[2], f32x4 (&b)[NJ], bool last) {
    if constexpr (NJ == 4) {
      if (last) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
      else asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    } else {
      if (last) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
      else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  const int nk = g.K / BK;
  f32x4 a0[2], b0[NJ], a1[2], b1[NJ];
  dma(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int st = kt & 1;
    // rendezvous: this wave's requests of tile kt have landed (vmcnt) and -- behind the barrier -- everybody's; every wave is past its
    // fragment reads of tile kt - 1 (their data fed MFMAs that were issued before the barrier), so the other stage may be refilled
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    frags(st, 0, a0, b0);
    if (kt + 1 < nk) dma(st ^ 1, kt + 1);
    frags(st, 1, a1, b1);
    wait_set(a0, b0, false);
    mfmas(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    frags(st, 2, a0, b0);
    wait_set(a1, b1, false);
    mfmas(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    frags(st, 3, a1, b1);
    wait_set(a0, b0, false);
    mfmas(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    wait_set(a1, b1, true);
    mfmas(a1, b1);
  }

  // epilogue: gemm_f32_kernel's arithmetic (a lane owns one output row and 4-element runs of consecutive n), but no load -> wait ->
  // use chain per 4 columns (two waves per SIMD hide little of it; the f16x2 kernels taught the lesson, profiles/
  // r05_gemm_w128_workload_ab.log): the tile's bias / column-scale values go through the now idle LDS once, and the residual vectors
  // of block (i, jn + 1) are requested before block (i, jn) is processed.
  const bool has_bias = g.bias != nullptr, has_cs = g.colscale != nullptr, has_res = g.resid != nullptr;
  float* const lbias = reinterpret_cast<float*>(lds);
  float* const lcs = lbias + TN;
  __syncthreads();                                   // every wave is past its last fragment reads: the stages are free
  if (tid < TN / 4) {
    const int n = n0 + tid * 4;                      // N % 4 == 0: a float4 is all in or all out
    if (has_bias && n < g.N) *reinterpret_cast<f32x4*>(lbias + tid * 4) = *reinterpret_cast<const f32x4*>(g.bias + n);
    if (has_cs && n < g.N) *reinterpret_cast<f32x4*>(lcs + tid * 4) = *reinterpret_cast<const f32x4*>(g.colscale + n);
  }
  __syncthreads();
  // residual row of this lane's two output rows (i = 0, 1): the row itself, or through the row map of a compacted row set
  // (resid_rows, the trimmed last NAR layer) -- looked up ONCE here, not in front of every residual vector
  long rrow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0 + wm * 64 + i * 32 + l31;
    m = m < g.M ? m : g.M - 1;
    rrow[i] = (has_res && g.resid_rows) ? g.resid_rows[m] : m;
  }
  auto load_rr = [&](int t, f32x4 (&rr)[4]) {       // block t = NJ i + jn
    const int i = t / NJ, jn = t % NJ;
    const float* rp = g.resid + rrow[i] * g.ldr + n0 + wn * (TN / 2) + jn * 32 + 4 * hi;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int n = n0 + wn * (TN / 2) + jn * 32 + 8 * g4 + 4 * hi;
      rr[g4] = n < g.N ? *reinterpret_cast<const f32x4*>(rp + 8 * g4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  f32x4 rrA[4], rrB[4];
  if (has_res) load_rr(0, rrA);
#pragma unroll
  for (int t = 0; t < 2 * NJ; ++t) {
    const int i = t / NJ, jn = t % NJ;
    const int m = m0 + wm * 64 + i * 32 + l31;
    f32x4 (&rr)[4] = (t & 1) ? rrB : rrA;
    if (has_res && t + 1 < 2 * NJ) load_rr(t + 1, (t & 1) ? rrA : rrB);
    if (m >= g.M) continue;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int nl = wn * (TN / 2) + jn * 32 + 8 * g4 + 4 * hi, n = n0 + nl;
      if (n >= g.N) continue;
      f32x4 v = {acc[i][jn][4 * g4], acc[i][jn][4 * g4 + 1], acc[i][jn][4 * g4 + 2], acc[i][jn][4 * g4 + 3]};
      if (has_bias) {
        const f32x4 bi = *reinterpret_cast<const f32x4*>(lbias + nl);
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
      if (has_cs) {
        const f32x4 cs = *reinterpret_cast<const f32x4*>(lcs + nl);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * cs[e];
      }
      if (has_res) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rr[g4][e] + v[e];
      }
      *reinterpret_cast<f32x4*>(g.C + (long)m * g.ldc + n) = v;
    }
  }
}

// variant 0: the product's choice -- the LDS-DMA kernel for long row sets whose operands it can address (16-B aligned rows), the
// register-staged kernel otherwise (Vocos / EnCodec: short row sets, N = 1282 / 1025 ...);  1 / 2 / 3 / 4: register-staged / DMA 256 x 128 /
// DMA 128 x 128 / DMA 256 x 256 forced (A/B; VX_GEMM_F32_VARIANT in the environment forces one for a whole process)
int gemm_walk_env() {
  static const int walk = [] {
    const char* e = getenv("VX_GEMM_WALK");
    if (!e || !*e) return 0;
    const int gm = atoi(e);
    if (gm < 1 || gm > 255) return 0;
    const char* comma = strchr(e, ',');
    return gm | ((comma && comma[1] == 'c') ? 256 : 0);
  }();
  return walk;
}

void launch_gemm_f32(const GemmArgs& g_in, hipStream_t s, int variant) {
  GemmArgs g = g_in;
  if (!g.walk) g.walk = gemm_walk_env();
  if (g.M <= 0 || g.N <= 0) return;
  if (variant == 0) {
    static const int env = [] { const char* e = getenv("VX_GEMM_F32_VARIANT"); return e ? atoi(e) : 0; }();
    variant = env;
  }
  const bool dma_ok = g.K % BK == 0 && g.lda % 4 == 0 && g.ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(g.A) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(g.W) & 15) == 0;
  // measured on MI355X (profiles/r05_gemm_f32_ab.log: the four NAR shapes at M = 31 616 and the prefill's M = 12 288, interleaved,
  // random operands): against the register-staged kernel the 256 x 256 DMA tile is -7.5 / -6.6 / -6.5 / -7.7 % at M = 31 616 (the
  // 256 x 128 one -4.4 / -4.8 / -2.1 / -2.6 %) and +13 % at M = 12 288 (48 x 12 tiles = 2.25 rounds on 256 CUs), where the 128 x 128 DMA
  // tile (two workgroups per CU) is -1.3 %; all four kernels agree bit for bit
  if (variant == 0) {
    variant = !(dma_ok && g.N % BN == 0 && g.M >= 2048) ? 1 : (g.M < 16384 ? 3 : (g.N % 256 == 0 ? 4 : 2));
    static const bool tile_model = [] { const char* e = getenv("VX_GEMM_F32_TILES"); return !(e && e[0] == '0'); }();   // =0: A/B
    if (variant == 4 && tile_model) {
      // round 6: rounds of tiles on the 256 CUs.  The trimmed last layers of the NAR stages (19 200 rows x N = 1024: 300 tiles of
      // 256 x 256 = 1.17 rounds, paid as 2) run 3 rounds of 256 x 128 tiles instead (a round of those is ~0.52 of a 256 x 256 round,
      // profiles/r05_gemm_f32_ab.log); the full-length shapes keep 256 x 256 (6 vs 6.2, 2 vs 2.07, 8 vs 8.3 rounds)
      const long rm = (g.M + 255) / 256;
      const double c256 = (double)((rm * (g.N / 256) + 255) / 256);
      const double c128 = (double)((rm * (g.N / 128) + 255) / 256) * 0.5175;
      if (c128 < c256) variant = 2;
    }
  }
  if (variant != 1 && !dma_ok) variant = 1;
  if (variant == 2) {
    const int tiles = ((g.M + 255) / 256) * ((g.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_f32_dma_kernel<256, 128>), dim3(tiles), dim3(512), 0, s, g);
  } else if (variant == 3) {
    const int tiles = ((g.M + 127) / 128) * ((g.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_f32_dma_kernel<128, 128>), dim3(tiles), dim3(256), 0, s, g);
  } else if (variant == 4) {
    const int tiles = ((g.M + 255) / 256) * ((g.N + 255) / 256);
    hipLaunchKernelGGL((gemm_f32_dma_kernel<256, 256>), dim3(tiles), dim3(512), 0, s, g);
  } else {
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(tiles), dim3(256), 0, s, g);
  }
}

}  // namespace vx
