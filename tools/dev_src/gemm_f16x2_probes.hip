// TOOLS-ONLY translation unit (vall-e-x_amd/_build.py --dev; never part of libvallex_hip.so of the product): the research
// template of the f16x2 GEMM with its timing probes V = 1 .. 18 (tools/gemm_bench.py, tools/gemm_timeline.py).  V = 0 here is the
// kernel as it was before the DMA instructions were spread between the MFMAs (kept for A/B); the product kernel lives in
// vall-e-x_amd/csrc/gemm_f16x2.hip and is V = 15 of this file (TN = 256) / V = 0 (TN = 128), without the probe branches.
#include <algorithm>
#include <type_traits>

#include "../../vall-e-x_amd/csrc/vx_common.h"

namespace vx {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int HM = 256, HK = 32, HLD = 64;                       // TN (columns of a tile) = 128 or 256: template parameter
constexpr int HA_PL = HM * HLD;                                  // 16 KiB per A plane and stage
constexpr int WTR = H2_TILE_W;                                   // rows of a W plane tile (256): a TN = 128 tile is half of one

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

}  // namespace

#ifdef VX_DEV_PROBES
// development aid (tools/gemm_timeline.py): shader-clock stamps of wave 0 of the first 256 workgroups around two consecutive
// k-steps in steady state: [0] before the rendezvous, [1] after it, [2] DMA of the next stage issued, [3] fragment reads issued,
// [4] MFMAs issued, then the same five for the second k-step at [5..9]; [10] loop done, [11] epilogue done, [12] entry
__device__ unsigned long long vx_gstamps[256 * 16];
#define VX_GSTAMP(COND, SLOT)                                                                     \
  do {                                                                                            \
    if ((COND) && threadIdx.x == 0 && blockIdx.x < 256) vx_gstamps[blockIdx.x * 16 + (SLOT)] = __builtin_readcyclecounter(); \
  } while (0)
void dev_read_gemm_stamps(unsigned long long* out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(vx_gstamps), sizeof(unsigned long long) * 256 * 16);
}
#else
#define VX_GSTAMP(COND, SLOT)
#endif

// V = 0 product kernel.  Timing probes (VX_DEV_PROBES builds, tools/gemm_bench.py; results meaningless): V = 1 no DMA after the
// first two tiles; V = 2 no MFMAs; V = 3 fragments read once; V = 4 no barriers / vmcnt waits (racy); V = 11 / 12 operands staged
// through registers instead of LDS-DMA (12: without the MFMAs).
// V = 13: probe 1 (no DMA) with ONE aliased LDS stage and two workgroups per CU -- does the compute path (fragment reads,
// MFMAs, barriers) run faster when a second, independent workgroup shares the CU?
// V = 14 / 15 (TN = 256 only; real kernels: same sums as V = 0): the eight LDS-DMA instructions of the next stage are not issued
// back to back right behind the rendezvous -- by all eight waves of the CU at once, with the matrix pipe idle meanwhile -- but one
// at a time between the MFMAs of this K tile (ISA of V = 0: barrier | 8 x [2 v_lshl_add_u64, s_mov m0, global_load_lds] | 6 ds_read |
// 8 MFMA | 6 ds_read | 16 MFMA | 6 ds_read | 8 MFMA | 6 ds_read | 16 MFMA).  14: behind each of the FIRST eight MFMAs (the
// data still has almost the whole K tile to land: with two stages the landing time is what the next rendezvous waits for);
// 15: behind every sixth MFMA (even spread; the last requests are issued late).
// TN = 256: a 256 x 256 tile (wave tile 64 x 128, 128 accumulator registers -- possible since the single accumulator): a third
// less operand traffic per flop and half the barriers; its fragments are read one k16 step at a time (48 registers).
template <int V, int TN>
__global__ __launch_bounds__(512, V == 13 ? 2 : 1) void gemm_f16x2_probe_kernel(GemmX3Args g) {
  constexpr int HN = TN, HW_PL = TN * HLD, HSTAGE = 2 * HA_PL + 2 * HW_PL;     // 48 / 64 KiB per stage
  constexpr int HNDMA = HSTAGE / (8 * 1024);                                    // 1 KiB DMA instructions per wave and stage: 6 / 8
  constexpr int NJ = TN / 64;                                                   // 32-column blocks of a wave: 2 / 4
  __shared__ __attribute__((aligned(1024))) unsigned char stage0[HSTAGE];
  __shared__ __attribute__((aligned(1024))) unsigned char stage1_[V == 13 ? 16 : HSTAGE];
  unsigned char* const stage1 = V == 13 ? stage0 : stage1_;

  constexpr int GM = V == 5 ? 4 : V == 6 ? 16 : V == 7 ? 2 : 8;     // probes 5-7: other XCD-wave shapes (GM x 32/GM tiles)
  const int tiles_m = (g.M + HM - 1) / HM, tiles_n = (g.N + HN - 1) / HN;
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
  const int m0 = tm * HM, n0 = tn * HN;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1, hi = lane >> 5, l31 = lane & 31;

  // DMA plan: instruction q = wid * HNDMA + j of a stage; q < 32: A plane q / 16, rows 16 (q % 16) ..; else W plane
  // (q - 32) / (TN / 16), rows 16 ((q - 32) % (TN / 16)) ...  Lane -> (row l / 4 of the 16, LDS chunk slot l % 4), swizzle on the
  // global side.  W planes are tiled in 256 rows: a TN = 128 tile is the upper or lower half of one.
  const unsigned short* src[HNDMA];
  int lds_off[HNDMA];
  long kstep[HNDMA];
#pragma unroll
  for (int j = 0; j < HNDMA; ++j) {
    const int q = wid * HNDMA + j;
    const bool isA = q < 32;
    const int qq = isA ? q : q - 32;
    const int p = isA ? qq >> 4 : qq / (TN / 16), r16 = isA ? qq & 15 : qq % (TN / 16);
    const int row = r16 * 16 + (lane >> 2) + (isA ? 0 : (tn * TN) % WTR);
    const int ch = (lane & 3) ^ ((lane >> 4) & 3);
    // tile-major planes: a panel is K/32 blocks of 256 rows x 32 (this workgroup's rows of each block are one contiguous run)
    const long tile0 = (V == 8 || V == 9) ? 0 : (long)(isA ? tm : (tn * TN) / WTR) * (g.K / HK) * ((isA ? HM : WTR) * HK);   // probes 8/9: every workgroup streams tile 0 (all L2 hits)
    src[j] = (isA ? g.A + p * g.a_plane : g.W + p * g.w_plane) + tile0 + (long)row * HK + ch * 8;
    kstep[j] = (long)(isA ? HM : WTR) * HK;
    lds_off[j] = (isA ? p * HA_PL : 2 * HA_PL + p * HW_PL) + r16 * 1024;
  }
  auto dma = [&](unsigned char* stage, int kt) {
#pragma unroll
    for (int j = 0; j < (V == 10 ? 4 : HNDMA); ++j)            // probe 10: two thirds of the operand bytes (results meaningless)
      __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + kt * kstep[j]), (lptr_t)(stage + lds_off[j]), 16, 0, 0);
  };

  f32x16 acc[2][NJ];                                             // ONE accumulator per 32 x 32 block: tail and head products share a scale
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int sw = (l31 >> 2) & 3;
  const int a_row = (wm * 64 + l31) * HLD, w_row = (wn * (TN / 2) + l31) * HLD;
  auto frags = [&](const unsigned char* stage, int s, f16x8 (&w)[2][NJ], f16x8 (&a)[2][2]) {
    const int coff = ((2 * s + hi) ^ sw) * 16;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int jn = 0; jn < NJ; ++jn)
        w[p][jn] = *reinterpret_cast<const f16x8*>(stage + 2 * HA_PL + p * HW_PL + w_row + jn * 32 * HLD + coff);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p) a[i][p] = *reinterpret_cast<const f16x8*>(stage + p * HA_PL + a_row + i * 32 * HLD + coff);
  };
  auto mfmas = [&](const f16x8 (&w)[2][NJ], const f16x8 (&a)[2][2]) {
    // transposed product (A operand = W rows); the three terms of a block go into the same accumulator, small ones first; the
    // four blocks of the wave take turns so that no MFMA waits for the one before it on the same accumulator
    if (V == 2 || V == 9 || V == 12) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn)
          acc[i][jn][0] += (float)w[1][jn][0] * (float)a[i][0][0] + (float)w[0][jn][1] * (float)a[i][1][1] + (float)w[0][jn][0] * (float)a[i][0][0];
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jn = 0; jn < NJ; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1][jn], a[i][0], acc[i][jn], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jn = 0; jn < NJ; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0][jn], a[i][1], acc[i][jn], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jn = 0; jn < NJ; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0][jn], a[i][0], acc[i][jn], 0, 0, 0);
  };
  f16x8 rg[HNDMA];
  auto gload = [&](int kt) {
#pragma unroll
    for (int j = 0; j < HNDMA; ++j) rg[j] = *reinterpret_cast<const f16x8*>(src[j] + kt * kstep[j]);
  };
  auto lwrite = [&](unsigned char* stage) {
#pragma unroll
    for (int j = 0; j < HNDMA; ++j) *reinterpret_cast<f16x8*>(stage + lds_off[j] + lane * 16) = rg[j];
  };
  f16x8 w0[2][NJ], a0[2][2], w1[2][(TN == 128 || V == 16) ? NJ : 1], a1[2][2];
  // probes 14 / 15: one k16 step (fragments + its 24 MFMAs) with the next stage's DMA instructions threaded between the MFMAs:
  // 14: instruction j behind MFMA j + 1 of step 0 (j = 0..7); 15: instruction 4 s + 0..3 behind MFMAs 3, 9, 15, 21 of step s
  auto kstep_spread = [&](const unsigned char* stage, unsigned char* other, int kt_next, bool more, int s, f16x8 (&w)[2][NJ],
                          f16x8 (&a)[2][2], bool read_frags = true) {
    if (read_frags) frags(stage, s, w, a);
    int n = 0;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int wp = p == 0 ? 1 : 0, ap = p == 1 ? 1 : 0;          // tail.head, head.tail, head.head: the order of mfmas()
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) {
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[wp][jn], a[i][ap], acc[i][jn], 0, 0, 0);
          ++n;
          // 15 / 16: behind every sixth MFMA of the whole K tile; 20: behind every third MFMA of k16 step 0 (all eight requests in
          // the FIRST half of the tile: each has >= 24 MFMA times to land before the next rendezvous); 21: every fourth MFMA, six in
          // step 0 and two at the start of step 1; 22: every second MFMA of step 0 from the fourth on (first third of the tile)
          const bool here = V == 14 ? (s == 0 && n <= HNDMA)
                          : V == 20 ? (s == 0 && n % 3 == 0)
                          : V == 21 ? (n % 4 == 0 && (s == 0 || n <= 8))
                          : V == 22 ? (s == 0 && n >= 4 && n % 2 == 0 && n <= 18)
                                    : (n % 6 == 3);
          if (here) {
            const int j = V == 14 ? n - 1 : V == 20 ? n / 3 - 1 : V == 21 ? (s == 0 ? n / 4 - 1 : 5 + n / 4) : V == 22 ? n / 2 - 2
                                  : (4 * s + n / 6) % HNDMA;
            __builtin_amdgcn_sched_barrier(0);
            if (more)
              __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + kt_next * kstep[j]), (lptr_t)(other + lds_off[j]), 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    }
  };
  auto ktile = [&](const unsigned char* stage, unsigned char* other, int kt_next, bool more, bool first, int sb) {
    (void)sb;
    if (V != 14 && V != 15 && V != 16 && V != 20 && V != 21 && V != 22 && more && ((V != 1 && V != 13) || kt_next < 2)) dma(other, kt_next);
    VX_GSTAMP(sb >= 0, sb + 2);
    if constexpr (TN == 128) {
      if (V != 3 || first) {
        frags(stage, 0, w0, a0);
        frags(stage, 1, w1, a1);
      }
      VX_GSTAMP(sb >= 0, sb + 3);
      mfmas(w0, a0);
      mfmas(w1, a1);
    } else if constexpr (V == 14 || V == 15 || V == 20 || V == 21 || V == 22) {
      kstep_spread(stage, other, kt_next, more, 0, w0, a0);
      kstep_spread(stage, other, kt_next, more, 1, w0, a0);
    } else if constexpr (V == 16) {                 // 15 + both k16 steps' fragments requested up front (96 fragment registers)
      frags(stage, 0, w0, a0);
      frags(stage, 1, w1, a1);
      kstep_spread(stage, other, kt_next, more, 0, w0, a0, false);
      kstep_spread(stage, other, kt_next, more, 1, w1, a1, false);
    } else {                                        // 48 fragment registers: one k16 step at a time
      frags(stage, 0, w0, a0);
      mfmas(w0, a0);
      frags(stage, 1, w0, a0);
      mfmas(w0, a0);
    }
    VX_GSTAMP(sb >= 0, sb + 4);
  };
  auto rendezvous = [&]() {
    if (V == 4) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  const int nk = g.K / HK;
  VX_GSTAMP(true, 12);
  if constexpr ((V == 11 || V == 12) && TN == 128) {
    gload(0);
    lwrite(stage0);
    if (nk > 1) gload(1);
    for (int kt = 0; kt < nk; kt += 2) {
      __syncthreads();
      if (kt + 1 < nk) { lwrite(stage1); if (kt + 2 < nk) gload(kt + 2); }
      frags(stage0, 0, w0, a0); frags(stage0, 1, w1, a1);
      mfmas(w0, a0); mfmas(w1, a1);
      if (kt + 1 < nk) {
        __syncthreads();
        if (kt + 2 < nk) { lwrite(stage0); if (kt + 3 < nk) gload(kt + 3); }
        frags(stage1, 0, w0, a0); frags(stage1, 1, w1, a1);
        mfmas(w0, a0); mfmas(w1, a1);
      }
    }
  } else {
  dma(stage0, 0);
  for (int kt = 0; kt < nk; kt += 2) {
    const bool st = V == 0 && kt == 8;              // dev builds: stamp k-steps 8 and 9
    (void)st;
    VX_GSTAMP(st, 0);
    rendezvous();
    VX_GSTAMP(st, 1);
    ktile(stage0, stage1, kt + 1, kt + 1 < nk, kt == 0, st ? 0 : -1);
    if (kt + 1 < nk) {
      VX_GSTAMP(st, 5);
      rendezvous();
      VX_GSTAMP(st, 6);
      ktile(stage1, stage0, kt + 2, kt + 2 < nk, false, st ? 5 : -1);
    }
  }
  }
  VX_GSTAMP(true, 10);
  if (V == 4) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }

  // epilogue: C = accumulator * 2^-(sa + sw) (+ bias, activation, residual as in the bf16x3 kernels).
  // With g.out_planes set, the result is NOT written as fp32 rows: it is split on the spot into the f16x2 planes of the NEXT
  // GEMM's A operand (tile-major, K = this N), so linear1 -> linear2 needs neither an fp32 round trip of the [M][4096] hidden
  // activations nor a split pass.  The 32 columns of a (jn) block are exactly one K tile of the consumer; lanes l and l ^ 32 hold
  // complementary 4-column halves of each 8-column group, so they trade halves (one ds_bpermute per word) and every lane stores
  // 16 contiguous bytes per plane.  Same conversions as split2h_kernel => bit-identical planes.
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + l31;
    if (m >= g.M) continue;
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) {
      unsigned hw[4][2], tw[4][2];                               // [g4][pair]: packed fp16 heads / scaled tails (planes mode)
      bool bad = false;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int n = n0 + wn * (TN / 2) + jn * 32 + 8 * g4 + 4 * hi;
        if (n >= g.N) continue;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][jn][4 * g4 + e] * g.descale;
        if (g.bias) {
          const f32x4 bi = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bi[e];
        }
        if (g.act == ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (g.resid) {
          const f32x4 rr = *reinterpret_cast<const f32x4*>(g.resid + (long)m * g.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rr[e] + v[e];
        }
        if (!g.out_planes) {
          *reinterpret_cast<f32x4*>(g.C + (long)m * g.ldc + n) = v;
        } else {
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
            f16x2v h2, t2;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              _Float16 hq, tq;
              h2_split(v[2 * pr + q], H2_ACT_SCALE, hq, tq, bad);   // as split2h_kernel: bit-identical planes
              h2[q] = hq;
              t2[q] = tq;
            }
            hw[g4][pr] = __builtin_bit_cast(unsigned, h2);
            tw[g4][pr] = __builtin_bit_cast(unsigned, t2);
          }
        }
      }
      if (g.out_planes) {
        // consumer plane element (m, k = n): ((m / 256) * (N / 32) + n / 32) * 256 * 32 + (m % 256) * 32 + n % 32
        const long blk = ((long)tm * (g.N / HK) + (n0 + wn * (TN / 2) + jn * 32) / HK) * (HM * HK) + (long)(wm * 64 + i * 32 + l31) * HK;
#pragma unroll
        for (int j = 0; j < 2; ++j) {                            // columns 16 j .. 16 j + 15 of the K tile
          // lane hi = 0 keeps its g4 = 2j words and wants the partner's g4 = 2j words; lane hi = 1 keeps g4 = 2j + 1
          const int keep = 2 * j + hi, give = 2 * j + 1 - hi;
          unsigned rh[2], rt[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            rh[q] = (unsigned)__shfl_xor((int)hw[give][q], 32, 64);
            rt[q] = (unsigned)__shfl_xor((int)tw[give][q], 32, 64);
          }
          typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
          const u32x4v oh = hi ? u32x4v{rh[0], rh[1], hw[keep][0], hw[keep][1]} : u32x4v{hw[keep][0], hw[keep][1], rh[0], rh[1]};
          const u32x4v ot = hi ? u32x4v{rt[0], rt[1], tw[keep][0], tw[keep][1]} : u32x4v{tw[keep][0], tw[keep][1], rt[0], rt[1]};
          unsigned short* o = g.out_planes + blk + 16 * j + 8 * hi;
          *reinterpret_cast<u32x4v*>(o) = oh;
          *reinterpret_cast<u32x4v*>(o + g.out_plane) = ot;
        }
        if (bad && g.range_flag) *g.range_flag = 1;
      }
    }
  }
  VX_GSTAMP(true, 11);
}


// ------------------------------------------------------------------------------------------------------------------------------
// Probe 23 (round 5): the verdict's prescription for a power-bound kernel, built to be measured -- "a 128 x 128 per-wave register
// tile (0.33 ds_read_b128 per MFMA instead of 0.5)".  Same 256 x 256 x 32 workgroup tile, same operand planes, same two 64 KiB LDS
// stages and the same per-element accumulation order as the product kernel (=> same sums), but FOUR waves of 128 x 128 (16
// accumulator blocks = 256 accumulator registers, one wave per SIMD) and, because a lone wave per SIMD has nobody to hide its waits,
// a complete software pipeline:  two fragment sets (the 16 fragments of the NEXT k16 step are read, one behind every third MFMA,
// while the 48 MFMAs of this step run), the ONE rendezvous of a K tile sits between its two k16 steps, and the 16 LDS-DMA
// requests of tile kt + 2 are issued -- one behind every third MFMA -- during the second step of tile kt, so that they have a whole
// k16 step (>= 1 536 matrix-pipe cycles) left to land before the rendezvous of tile kt + 1 asks for them.  Inline-asm LDS reads
// with counted waits (a C++ LDS load behind an LDS-DMA makes the compiler wait for the DMA).
__global__ __launch_bounds__(256, 1) void gemm_f16x2_w128_probe_kernel(GemmX3Args g) {
  constexpr int TN = 256, HW_PL = TN * HLD, HSTAGE = 2 * HA_PL + 2 * HW_PL;     // 64 KiB per stage
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * HSTAGE];
  constexpr int GM = 8;
  const int tiles_m = (g.M + HM - 1) / HM, tiles_n = g.N / TN;
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
  const int m0 = tm * HM, n0 = tn * TN;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1, hi = lane >> 5, l31 = lane & 31;

  // DMA plan: wave w fetches ONE plane of a stage (0 / 1: A head / tail, 2 / 3: W head / tail): 16 instructions of 1 KiB = 16 rows
  // each; in the tile-major planes the 256 rows x 32 of a K tile are one contiguous 16 KiB run, so instruction j is base + j KiB
  const bool isA = wid < 2;
  const int pl = wid & 1;
  const unsigned short* dbase = (isA ? g.A + pl * g.a_plane + (long)tm * (g.K / HK) * (HM * HK)
                                     : g.W + pl * g.w_plane + (long)tn * (g.K / HK) * (WTR * HK)) +
                                (long)(lane >> 2) * HK + (((lane & 3) ^ ((lane >> 4) & 3)) * 8);
  const int dlds = (isA ? pl * HA_PL : 2 * HA_PL + pl * HW_PL);
  auto dma1 = [&](int stage, int kt, int j) {
    __builtin_amdgcn_global_load_lds((gptr_t)(dbase + (long)kt * (HM * HK) + j * 512), (lptr_t)(lds + stage * HSTAGE + dlds + j * 1024), 16, 0, 0);
  };

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)lds;
  const unsigned sw = (unsigned)((l31 >> 2) & 3);
  const unsigned a_base = lds0 + (unsigned)((wm * 128 + l31) * HLD), w_base = lds0 + (unsigned)(2 * HA_PL + (wn * 128 + l31) * HLD);
  // fragment f of a set: f < 8: W plane f / 4, block f % 4;  f >= 8: A block (f - 8) / 2, plane (f - 8) % 2
  f16x8 F0[16], F1[16];
#define VX_RD(DST, BASE, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(BASE), "n"(OFF))
  auto rd1 = [&](f16x8 (&F)[16], unsigned co, int f) {
    const unsigned wa = w_base + co, aa = a_base + co;
    switch (f) {
      case 0: VX_RD(F[0], wa, 0 * HW_PL + 0 * 2048); break;   case 1: VX_RD(F[1], wa, 0 * HW_PL + 1 * 2048); break;
      case 2: VX_RD(F[2], wa, 0 * HW_PL + 2 * 2048); break;   case 3: VX_RD(F[3], wa, 0 * HW_PL + 3 * 2048); break;
      case 4: VX_RD(F[4], wa, 1 * HW_PL + 0 * 2048); break;   case 5: VX_RD(F[5], wa, 1 * HW_PL + 1 * 2048); break;
      case 6: VX_RD(F[6], wa, 1 * HW_PL + 2 * 2048); break;   case 7: VX_RD(F[7], wa, 1 * HW_PL + 3 * 2048); break;
      case 8: VX_RD(F[8], aa, 0 * HA_PL + 0 * 2048); break;   case 9: VX_RD(F[9], aa, 1 * HA_PL + 0 * 2048); break;
      case 10: VX_RD(F[10], aa, 0 * HA_PL + 1 * 2048); break; case 11: VX_RD(F[11], aa, 1 * HA_PL + 1 * 2048); break;
      case 12: VX_RD(F[12], aa, 0 * HA_PL + 2 * 2048); break; case 13: VX_RD(F[13], aa, 1 * HA_PL + 2 * 2048); break;
      case 14: VX_RD(F[14], aa, 0 * HA_PL + 3 * 2048); break; default: VX_RD(F[15], aa, 1 * HA_PL + 3 * 2048); break;
    }
  };
  auto coff = [&](int stage, int s) { return (unsigned)(stage * HSTAGE) + (((unsigned)(2 * s + hi) ^ sw) << 4); };
  auto wait_set = [&](f16x8 (&F)[16]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7]), "+v"(F[8]), "+v"(F[9]),
                   "+v"(F[10]), "+v"(F[11]), "+v"(F[12]), "+v"(F[13]), "+v"(F[14]), "+v"(F[15]));
    __builtin_amdgcn_sched_barrier(0);
  };
  // one k16 step on set F: 48 MFMAs in the product's order (per block: tail.head, head.tail, head.head).  Behind every third MFMA:
  // one fragment read of the other set (if rd) and one LDS-DMA request (if dm).
  auto step = [&](const f16x8 (&F)[16], f16x8 (&G)[16], auto rd, unsigned rco, auto dm, int dstage, int dkt) {
    int n = 0;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int wp = p == 0 ? 1 : 0, ap = p == 1 ? 1 : 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[wp * 4 + jn], F[8 + 2 * i + ap], acc[i][jn], 0, 0, 0);
          ++n;
          if (n % 3 == 2 && (decltype(rd)::value || decltype(dm)::value)) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (decltype(rd)::value) rd1(G, rco, n / 3);
            if constexpr (decltype(dm)::value) dma1(dstage, dkt, n / 3);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    }
  };
  // one K tile: k16 step 0 on F0 (reading F1 <- the tile's second step), the rendezvous, k16 step 1 on F1 (RD: reading F0 <- the next
  // tile's first step from the other stage; DM: requesting tile kt + 2 into this tile's stage)
  auto ktile = [&](int kt, auto rd, auto dm) {
    const int A = kt & 1, B = A ^ 1;
    wait_set(F0);
    step(F0, F1, std::true_type{}, coff(A, 1), std::false_type{}, 0, 0);
    // the ONE rendezvous of the tile: this wave's requests of tile kt + 1 have landed (issued a whole k16 step ago or more), its
    // reads of stage A are complete; behind the barrier the same holds for every wave: stage B may be read, stage A refilled
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(F1[0]), "+v"(F1[1]), "+v"(F1[2]), "+v"(F1[3]), "+v"(F1[4]), "+v"(F1[5]), "+v"(F1[6]), "+v"(F1[7]), "+v"(F1[8]),
                   "+v"(F1[9]), "+v"(F1[10]), "+v"(F1[11]), "+v"(F1[12]), "+v"(F1[13]), "+v"(F1[14]), "+v"(F1[15])
                 :: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    step(F1, F0, rd, coff(B, 0), dm, A, kt + 2);
  };

  const int nk = g.K / HK;                                        // >= 2 (the launcher checks K % 64 == 0)
#pragma unroll
  for (int j = 0; j < 16; ++j) dma1(0, 0, j);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    const unsigned co = coff(0, 0);
#pragma unroll
    for (int f = 0; f < 16; ++f) rd1(F0, co, f);
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) dma1(1, 1, j);
  int kt = 0;
  for (; kt + 2 < nk; ++kt) ktile(kt, std::true_type{}, std::true_type{});
  ktile(kt, std::true_type{}, std::false_type{});
  ktile(kt + 1, std::false_type{}, std::false_type{});
#undef VX_RD

  // epilogue: fp32 rows (descale, bias, activation, residual) -- the planes-out mode of the product kernel is not needed for the A/B
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 128 + i * 32 + l31;
    if (m >= g.M) continue;
#pragma unroll
    for (int jn = 0; jn < 4; ++jn)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int n = n0 + wn * 128 + jn * 32 + 8 * g4 + 4 * hi;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][jn][4 * g4 + e] * g.descale;
        if (g.bias) {
          const f32x4 bi = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bi[e];
        }
        if (g.act == ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
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

#ifdef VX_DEV_PROBES   // timing probes: tools-only build (vall-e-x_amd/_build.py --dev), never in the product library
void launch_gemm_f16x2_probe(const GemmX3Args& g, int variant, hipStream_t s) {
  const int tiles = ((g.M + HM - 1) / HM) * ((g.N + 127) / 128);
  if (tiles <= 0) return;
  const dim3 grid(tiles), block(512);
  if (variant == 1) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<1, 128>), grid, block, 0, s, g);
  else if (variant == 2) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<2, 128>), grid, block, 0, s, g);
  else if (variant == 3) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<3, 128>), grid, block, 0, s, g);
  else if (variant == 5) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<5, 128>), grid, block, 0, s, g);
  else if (variant == 6) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<6, 128>), grid, block, 0, s, g);
  else if (variant == 7) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<7, 128>), grid, block, 0, s, g);
  else if (variant == 8) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<8, 128>), grid, block, 0, s, g);
  else if (variant == 9) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<9, 128>), grid, block, 0, s, g);
  else if (variant == 10) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<10, 128>), grid, block, 0, s, g);
  else if (variant == 11) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<11, 128>), grid, block, 0, s, g);
  else if (variant == 12) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<12, 128>), grid, block, 0, s, g);
  else if (variant == 13) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<13, 128>), grid, block, 0, s, g);
  else if (variant == 14 || variant == 15) {                       // 256 x 256 tile kernels: their own grid; N must be a multiple of 256
    const dim3 grid256(((g.M + HM - 1) / HM) * (g.N / 256));
    if (g.N % 256 != 0) return;
    if (variant == 14) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<14, 256>), grid256, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_f16x2_probe_kernel<15, 256>), grid256, block, 0, s, g);
  }
  else if (variant == 23) {                                        // round 5: 128 x 128 per-wave tiles, 4 waves, full software pipeline
    if (g.N % 256 != 0 || g.K % 64 != 0) return;
    hipLaunchKernelGGL(gemm_f16x2_w128_probe_kernel, dim3(((g.M + HM - 1) / HM) * (g.N / 256)), dim3(256), 0, s, g);
  }
  else if (variant >= 16 && variant <= 22) {    // 19: the 256 x 256 kernel with the DMA instructions back to back (before round 3)    // 16: see the kernel; 17 / 18: probes 1 / 2 (no DMA / no MFMAs) on the 256 x 256 tile
    const dim3 grid256(((g.M + HM - 1) / HM) * (g.N / 256));
    if (g.N % 256 != 0) return;
    if (variant == 16) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<16, 256>), grid256, block, 0, s, g);
    else if (variant == 19) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<0, 256>), grid256, block, 0, s, g);
    else if (variant == 20) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<20, 256>), grid256, block, 0, s, g);
    else if (variant == 21) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<21, 256>), grid256, block, 0, s, g);
    else if (variant == 22) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<22, 256>), grid256, block, 0, s, g);
    else if (variant == 17) hipLaunchKernelGGL((gemm_f16x2_probe_kernel<1, 256>), grid256, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_f16x2_probe_kernel<2, 256>), grid256, block, 0, s, g);
  }
  else hipLaunchKernelGGL((gemm_f16x2_probe_kernel<4, 128>), grid, block, 0, s, g);
}
#endif

}  // namespace vx
