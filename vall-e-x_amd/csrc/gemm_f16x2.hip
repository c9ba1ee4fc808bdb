// "f16x2" GEMM -- the default arithmetic of every transformer projection on the full-sequence paths (prefill, NAR).
// Every fp32 operand is scaled by a power of two and split into an fp16 head and an fp16 tail AT THE SAME SCALE,
//     X = x * 2^s,   h = fp16(X),   t = fp16(X - h)          (22 significant bits; s: vx_common.h, H2_ACT_SHIFT / per weight)
// and   A.B ~= ta.hb + ha.tb + ha.hb      accumulated in ONE fp32 accumulator (smallest terms first), descaled by 2^-(sa + sb)
// in the epilogue -- THREE v_mfma_f32_32x32x16_f16 per 32x32x16 block instead of the six bf16 MFMAs of bf16x3 (and 4 instead of 6 operand
// bytes per element); every f16 x f16 product is exact in the fp32 accumulator.  Error per product ~2^-22 relative (the ta.tb
// term and the tail rounding are dropped): measured on MI355X against the fp32-MFMA kernel on the four NAR shapes (M = 31616,
// uniform [-1, 1) operands) the max |difference| is 6.5e-5 .. 2.8e-4 -- the SAME as bf16x3's (8.0e-5 .. 4.4e-4): both sit inside
// the reassociation noise of an fp32 accumulation over K = 1024 .. 4096, which is what separates any two fp32 GEMMs
// (profiles/r02_gemm_ab.log).  Every live-reference golden (short, sharp-attention, full-length 600 x 8 ids) stays bit-exact.
// Speed: 270-320 fp32-equivalent TF vs 172-186 for bf16x3 on the same shapes (x 1.6-1.7).
// fp16 range: |X| must stay below 65504, i.e. |activation| < 2047 (LayerNorm outputs, ReLU'd FFN activations, attention outputs
// of this model are far inside; a device flag makes the engine re-run the phase in exact fp32 otherwise); weights are scaled
// from their own max.  Tails below 2^-14 are fp16 subnormals (honoured by the matrix cores): <= 2^-25 * 2^-s absolute.
//
// Structure: tile 256 x 256 x 32 (8 waves 4 x 2, wave tile 64 x 128 = 8 accumulator blocks, 128 accumulator registers, fragments
// read one k16 step at a time; one workgroup per CU) for the long row sets, tile 128 x 128 x 32 (4 waves 2 x 2, wave tile 64 x 64,
// TWO workgroups per CU) for the short ones -- launch_gemm_f16x2 picks by a measured cost model; a 256 x 128 instantiation is kept
// for A/B.  Operand tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR round trip) into two LDS stages of 64 / 32 KiB
// (2 planes x (TM + TN) rows x 64 B), the XOR swizzle of the 64-B LDS rows applied on the global side.  256-wide tile: the eight DMA instructions a wave issues for the NEXT
// stage are threaded between the MFMAs of this K tile (one behind every sixth MFMA) instead of back to back behind the rendezvous
// with the matrix pipe idle: -2 .. -7 % on the four NAR shapes (profiles/r03_gemm_bench.log, columns h2-256x256 vs -dma-spread),
// same sums.  The timing probes this kernel grew up with live in tools/dev_src/gemm_f16x2_probes.hip (tools-only build).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "vx_common.h"

namespace vx {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int HM = 256, HK = 32, HLD = 64;                       // rows of an A plane BLOCK (H2_TILE_A); tile rows / columns: template parameters
constexpr int WTR = H2_TILE_W;                                   // rows of a W plane tile (256): a TN = 128 tile is half of one

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

}  // namespace

// x[rows][K] fp32 (row r read at gather ? gather[r] : r) -> planes[p][rows/TR][K/32][TR][32] fp16 of X = x * scale, p = 0 head, 1 tail.
// TILE-major: the TR = 256 (activations) or 128 (weights) rows x K panel that ONE workgroup of the GEMM streams is one
// contiguous run (TR * K * 2 B per plane), walked linearly by its K loop -- instead of 16 KiB pieces 64 B * rows apart
// (K-tile-major, the bf16x3 layout), i.e. one DRAM page / TLB entry per K tile and workgroup.  Rows past `rows` in the last
// tile are never written (the GEMM reads them as garbage into accumulator rows it never stores).
// range_flag (optional): set to 1 if any |x| does not fit fp16 (>= 65504 or non-finite) -- the engine then re-runs the phase on
// the exact-fp32 kernels instead of letting an inf head poison the GEMM silently.
__global__ __launch_bounds__(256) void split2h_kernel(const float* __restrict__ x, int ldx, long rows, int K,
                                                      const int* __restrict__ gather,
                                                      unsigned short* __restrict__ planes, long plane_stride, int tile_rows,
                                                      int* __restrict__ range_flag, float scale) {
  const long total = (long)(K / 32) * rows * 4;
  const int nkt = K / 32;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i & 3);
    const long rr = i >> 2;
    const long kt = rr / rows, r = rr - kt * rows;
    const long src = gather ? gather[r] : r;
    const float* xp = x + src * ldx + kt * 32 + ch * 8;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(xp);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(xp + 4);
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    f16x8 h, t;
    bool bad = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      _Float16 he, te;
      h2_split(v[e], scale, he, te, bad);
      h[e] = he;
      t[e] = te;
    }
    const long tile = r / tile_rows, rin = r - tile * tile_rows;
    unsigned short* o = planes + ((tile * nkt + kt) * tile_rows + rin) * 32 + ch * 8;
    *reinterpret_cast<f16x8*>(o) = h;
    *reinterpret_cast<f16x8*>(o + plane_stride) = t;
    if (bad && range_flag) *range_flag = 1;
  }
}

// plane_stride must be >= roundup(rows, tile_rows) * K elements
void launch_split2h(const float* x, int ldx, long rows, int K, const int* gather, unsigned short* planes, long plane_stride,
                    int tile_rows, int* range_flag, float scale, hipStream_t s) {
  if (rows <= 0) return;
  const long total = rows * (K / 8);
  hipLaunchKernelGGL(split2h_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, s, x, ldx, rows, K,
                     gather, planes, plane_stride, tile_rows, range_flag, scale);
}

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out_bits) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __builtin_bit_cast(unsigned, m));     // non-negative floats order like their bits
}
void launch_absmax(const float* x, long n, unsigned* out_bits, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 2048)), dim3(256), 0, s, x, n, out_bits);
}

// TM = rows of a tile: 256 (8 waves, one workgroup per CU) or 128 (4 waves, 32 KiB stages, TWO workgroups per CU) -- the 128-row tile
// is for row sets too short to fill the chip with 256-row tiles (one utterance: M = 983 gives 4 x N/128 workgroups on 256 CUs); its
// A panel is the upper or lower half of a 256-row plane block.  Per output element the sequence of accumulations is the same for
// every tile shape, so the result does not depend on which one runs.
// NST = LDS stages: 2, or 4 for the 128 x 128 tile when there is at most ONE workgroup per CU anyway (one utterance: the K loop of a
// lone workgroup is a chain of LDS-DMA round trips, ~0.8 us per K tile against 0.2 us of MFMAs; with the requests three tiles ahead
// -- counted s_waitcnt vmcnt(16 / 8 / 0) before the barrier -- the round trips overlap).
// PRIO: wave priority (guide T5).  0 none;  1 static: the second-dispatched half of the workgroup's waves (wid >= NWAVE / 2) runs at
// priority 1 throughout -- with two waves per SIMD the younger wave otherwise loses every arbitration to the older one;  2
// s_setprio(1) around the 24 MFMAs of every k16 step.  The product instantiates VX_GEMM_PRIO; tools builds all three.
#ifndef VX_GEMM_PRIO
#define VX_GEMM_PRIO 0
#endif
template <int TN, int TM, int NST = 2, int PRIO = 0>
__global__ __launch_bounds__(TM * 2, (TM == 256 || NST == 4) ? 1 : 2) void gemm_f16x2_kernel(GemmX3Args g) {
  constexpr int NWAVE = TM / 32;                                                 // 8 / 4 waves: TM / 64 along M x 2 along N
  constexpr int HA_PL = TM * HLD;                                                // 16 / 8 KiB per A plane and stage
  constexpr int HN = TN, HW_PL = TN * HLD, HSTAGE = 2 * HA_PL + 2 * HW_PL;     // 64 / 48 / 32 KiB per stage
  constexpr int HNDMA = HSTAGE / (NWAVE * 1024);                                // 1 KiB DMA instructions per wave and stage: 8 / 6 / 8
  constexpr int NA = 2 * TM / 16;                                                // ... of which the first NA (both planes) fetch A
  constexpr int NJ = TN / 64;                                                   // 32-column blocks of a wave: 2 / 4
  __shared__ __attribute__((aligned(1024))) unsigned char stage0[HSTAGE];
  __shared__ __attribute__((aligned(1024))) unsigned char stage1[HSTAGE];
  __shared__ __attribute__((aligned(1024))) unsigned char stage2[NST == 4 ? HSTAGE : 16];
  __shared__ __attribute__((aligned(1024))) unsigned char stage3[NST == 4 ? HSTAGE : 16];

  // XCD-aware order: consecutive block ids land on different XCDs (round robin), so XCD x walks its own contiguous range of
  // tiles, in groups of GM row tiles x all column tiles (the A panels of a group stay in that XCD's L2 while W streams)
  constexpr int GM = 8;
  const int tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + HN - 1) / HN;
  int tm, tn;
  tile_walk(blockIdx.x, tiles_m, tiles_n, GM, g.walk, tm, tn);
  const int m0 = tm * TM, n0 = tn * HN;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1, hi = lane >> 5, l31 = lane & 31;
  if constexpr (PRIO == 1) {
    if (wid >= NWAVE / 2) __builtin_amdgcn_s_setprio(1);         // wid is scalar (readfirstlane): s_setprio ignores EXEC
  }

  // DMA plan: instruction q = wid * HNDMA + j of a stage; q < 32: A plane q / 16, rows 16 (q % 16) ..; else W plane
  // (q - 32) / (TN / 16), rows 16 ((q - 32) % (TN / 16)) ...  Lane -> (row l / 4 of the 16, LDS chunk slot l % 4), swizzle on the
  // global side.  W planes are tiled in 256 rows: a TN = 128 tile is the upper or lower half of one.
  const unsigned short* src[HNDMA];
  int lds_off[HNDMA];
  long kstep[HNDMA];
#pragma unroll
  for (int j = 0; j < HNDMA; ++j) {
    const int q = wid * HNDMA + j;
    const bool isA = q < NA;
    const int qq = isA ? q : q - NA;
    const int p = isA ? qq / (TM / 16) : qq / (TN / 16), r16 = isA ? qq % (TM / 16) : qq % (TN / 16);
    const int row = r16 * 16 + (lane >> 2) + (isA ? m0 % HM : (tn * TN) % WTR);
    const int ch = (lane & 3) ^ ((lane >> 4) & 3);
    // tile-major planes: a panel is K/32 blocks of 256 rows x 32 (this workgroup's rows of each block are one contiguous run)
    const long tile0 = (long)(isA ? m0 / HM : (tn * TN) / WTR) * (g.K / HK) * ((isA ? HM : WTR) * HK);
    src[j] = (isA ? g.A + p * g.a_plane : g.W + p * g.w_plane) + tile0 + (long)row * HK + ch * 8;
    kstep[j] = (long)(isA ? HM : WTR) * HK;
    lds_off[j] = (isA ? p * HA_PL : 2 * HA_PL + p * HW_PL) + r16 * 1024;
  }
  auto dma1 = [&](unsigned char* stage, int kt, int j) {
    __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + kt * kstep[j]), (lptr_t)(stage + lds_off[j]), 16, 0, 0);
  };
  auto dma = [&](unsigned char* stage, int kt) {
#pragma unroll
    for (int j = 0; j < HNDMA; ++j) dma1(stage, kt, j);
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
  auto frags_asm = [&](const unsigned char* stage, int s, f16x8 (&w)[2][NJ], f16x8 (&a)[2][2]) {
    const unsigned base = (unsigned)(unsigned long long)(lptr_t)stage + (unsigned)(((2 * s + hi) ^ sw) * 16);
    const unsigned wa = base + (unsigned)w_row, aa = base + (unsigned)a_row;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int jn = 0; jn < NJ; ++jn)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[p][jn]) : "v"(wa), "n"(2 * HA_PL + p * HW_PL + jn * 32 * HLD));
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[i][p]) : "v"(aa), "n"(p * HA_PL + i * 32 * HLD));
  };
  // One k16 step: transposed product (A operand = W rows); the three terms of a block go into the same accumulator, small ones
  // first -- tail.head, head.tail, head.head -- and the blocks of the wave take turns, so no MFMA waits for the one before it
  // on the same accumulator.  SPREAD (256-wide tile): DMA instruction 4 s + 0..3 of the next stage behind MFMAs 3, 9, 15, 21.
  auto kstep16 = [&](const f16x8 (&w)[2][NJ], const f16x8 (&a)[2][2], unsigned char* other, int kt_next, bool more, int s) {
    int n = 0;
    if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int wp = p == 0 ? 1 : 0, ap = p == 1 ? 1 : 0;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) {
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[wp][jn], a[i][ap], acc[i][jn], 0, 0, 0);
          ++n;
          if (TN == 256 && n % 6 == 3) {
            __builtin_amdgcn_sched_barrier(0);
            if (more) dma1(other, kt_next, (4 * s + n / 6) % HNDMA);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    }
    if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);
  };
  f16x8 w0[2][NJ], a0[2][2], w1[2][TN == 128 ? NJ : 1], a1[2][2];
  auto ktile = [&](const unsigned char* stage, unsigned char* other, int kt_next, bool more) {
    if constexpr (TN == 128) {                     // both k16 steps' fragments up front, DMA of the next stage behind the rendezvous
      if (more) dma(other, kt_next);
      frags(stage, 0, w0, a0);
      frags(stage, 1, w1, a1);
      kstep16(w0, a0, other, kt_next, more, 0);
      kstep16(w1, a1, other, kt_next, more, 1);
    } else {                                       // 48 fragment registers: one k16 step at a time
      frags(stage, 0, w0, a0);
      kstep16(w0, a0, other, kt_next, more, 0);
      frags(stage, 1, w0, a0);
      kstep16(w0, a0, other, kt_next, more, 1);
    }
  };
  auto rendezvous = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  const int nk = g.K / HK;
  if constexpr (NST == 4) {
    static_assert(TN == 128 && TM == 128, "the four-stage ring is the short-row-set kernel");
    // tile t lives in stage t % 4; its requests are issued three tiles ahead, right behind the barrier of tile t - 3 (every wave
    // is past its reads of tile t - 4, the previous tenant of the stage).  Before the barrier of tile t a wave waits for ITS OWN
    // requests of tile t only: the (up to two) younger groups of 8 stay in flight.
    unsigned char* const st[4] = {stage0, stage1, stage2, stage3};
    dma(stage0, 0);
    if (nk > 1) dma(stage1, 1);
    if (nk > 2) dma(stage2, 2);
    auto step = [&](int t, int u) {
      const int younger = nk - 1 - t;                                       // request groups issued after tile t's: min(2, younger)
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // bare barrier: __syncthreads() carries a workgroup-scope fence, which the compiler implements as s_waitcnt vmcnt(0) -- it
      // would wait for the younger request groups as well.  What has to be ordered here is LDS traffic inside one CU: this wave's
      // fragment reads of the previous tile are complete (their data fed its MFMAs), its requests of tile t have landed.
      __builtin_amdgcn_s_barrier();
      if (t + 3 < nk) dma(st[(u + 3) & 3], t + 3);
      // fragment reads as inline asm: for a C++ LDS load the compiler's LDS-DMA tracking cannot tell the stages apart and inserts
      // s_waitcnt vmcnt(0) right behind the requests just issued; the hazards are ordered by the counted wait + barrier above
      frags_asm(st[u], 0, w0, a0);
      frags_asm(st[u], 1, w1, a1);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w0[0][0]), "+v"(w0[0][1]), "+v"(w0[1][0]), "+v"(w0[1][1]), "+v"(a0[0][0]), "+v"(a0[0][1]),
                   "+v"(a0[1][0]), "+v"(a0[1][1]));
      asm volatile("" : "+v"(w1[0][0]), "+v"(w1[0][1]), "+v"(w1[1][0]), "+v"(w1[1][1]), "+v"(a1[0][0]), "+v"(a1[0][1]), "+v"(a1[1][0]),
                   "+v"(a1[1][1]));
      kstep16(w0, a0, nullptr, 0, false, 0);
      kstep16(w1, a1, nullptr, 0, false, 1);
    };
    for (int kt = 0; kt < nk; kt += 4) {
      step(kt, 0);
      if (kt + 1 < nk) step(kt + 1, 1);
      if (kt + 2 < nk) step(kt + 2, 2);
      if (kt + 3 < nk) step(kt + 3, 3);
    }
  } else {
  dma(stage0, 0);
  for (int kt = 0; kt < nk; kt += 2) {
    rendezvous();
    ktile(stage0, stage1, kt + 1, kt + 1 < nk);
    if (kt + 1 < nk) {
      rendezvous();
      ktile(stage1, stage0, kt + 2, kt + 2 < nk);
    }
  }
  }
  // epilogue: C = accumulator * 2^-(sa + sw) (+ bias, activation, residual as in the bf16x3 kernels).
  // With g.out_planes set, the result is NOT written as fp32 rows: it is split on the spot into the f16x2 planes of the NEXT
  // GEMM's A operand (tile-major, K = this N), so linear1 -> linear2 needs neither an fp32 round trip of the [M][4096] hidden
  // activations nor a split pass.  The 32 columns of a (jn) block are exactly one K tile of the consumer; lanes l and l ^ 32 hold
  // complementary 4-column halves of each 8-column group, so they trade halves (one ds_bpermute per word) and every lane stores
  // 16 contiguous bytes per plane.  Same conversions as split2h_kernel => bit-identical planes.
  // Round 6: no dependent load -> wait -> use chain per 4 columns any more (the lesson of the four-wave kernel, profiles/
  // r05_gemm_w128_workload_ab.log, carried to this one: it still serves the 128 x 128 tiles of the prefill and of the trimmed last
  // layers): the four bias and the four residual vectors of block (i, jn + 1) are requested before block (i, jn) is processed, and
  // the residual row of a lane's two output rows (through resid_rows for compacted row sets) is looked up once.  Same values, same
  // operations per element: bit-identical output.
  const bool has_bias = g.bias != nullptr, has_res = g.resid != nullptr;
  long mrow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0 + wm * 64 + i * 32 + l31;
    m = m < g.M ? m : g.M - 1;
    mrow[i] = (has_res && g.resid_rows) ? g.resid_rows[m] : m;
  }
  auto load_blk = [&](int t, f32x4 (&bb)[4], f32x4 (&rr)[4]) {       // block t = NJ i + jn
    const int i = t / NJ, jn = t % NJ;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int n = n0 + wn * (TN / 2) + jn * 32 + 8 * g4 + 4 * hi;
      const bool in = n < g.N;                                       // N % 4 == 0: a float4 is all in or all out
      bb[g4] = (has_bias && in) ? *reinterpret_cast<const f32x4*>(g.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      rr[g4] = (has_res && in) ? *reinterpret_cast<const f32x4*>(g.resid + mrow[i] * g.ldr + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  f32x4 bbA[4], rrA[4], bbB[4], rrB[4];
  load_blk(0, bbA, rrA);
#pragma unroll
  for (int t = 0; t < 2 * NJ; ++t) {
    const int i = t / NJ, jn = t % NJ;
    const int m = m0 + wm * 64 + i * 32 + l31;
    f32x4 (&bbc)[4] = (t & 1) ? bbB : bbA;
    f32x4 (&rrc)[4] = (t & 1) ? rrB : rrA;
    if (t + 1 < 2 * NJ) load_blk(t + 1, (t & 1) ? bbA : bbB, (t & 1) ? rrA : rrB);
    if (m >= g.M) continue;
    {
      unsigned hw[4][2], tw[4][2];                               // [g4][pair]: packed fp16 heads / scaled tails (planes mode)
      bool bad = false;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int n = n0 + wn * (TN / 2) + jn * 32 + 8 * g4 + 4 * hi;
        if (n >= g.N) continue;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][jn][4 * g4 + e] * g.descale;
        if (has_bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bbc[g4][e];
        }
        if (g.act == ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (has_res) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rrc[g4][e] + v[e];
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
        const long blk = ((long)(m0 / HM) * (g.N / HK) + (n0 + wn * (TN / 2) + jn * 32) / HK) * (HM * HK) +
                         (long)(m0 % HM + wm * 64 + i * 32 + l31) * HK;
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
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 5: the 256 x 256 workgroup tile on FOUR waves of 128 x 128 (16 accumulator blocks = 256 accumulator registers in AGPRs, one
// wave per SIMD): 16 fragment reads per 48 MFMAs (0.33 ds_read_b128 per MFMA instead of 0.5) -- what the operand-pattern test of
// this round asked for (profiles/r05_gemm_operand_patterns.txt: the 8-wave kernel is power / clock-limited on real operands, so
// only less data movement per MFMA pays).  A lone wave per SIMD has nobody to hide its waits, hence a complete software pipeline:
//   * two fragment sets: the 16 fragments of the NEXT k16 step are read, one behind every third MFMA, while the 48 MFMAs of this
//     step run (across the K-tile boundary too);
//   * the ONE rendezvous of a K tile sits BETWEEN its two k16 steps;
//   * the 16 LDS-DMA requests of tile kt + 2 are issued, one behind every third MFMA, during the second step of tile kt: they have a
//     whole k16 step (>= 1 536 matrix-pipe cycles) left to land before the rendezvous of tile kt + 1 asks for them.
// Inline-asm LDS reads with counted waits (a C++ LDS load behind an LDS-DMA makes the compiler wait for the DMA).  Same operand
// planes, same LDS image and per output element the same MFMA sequence as gemm_f16x2_kernel => bit-identical sums.  Measured
// against it on MI355X (profiles/r05_gemm_w128_ab.log, M = 31 616, interleaved, random operands): QKV 604 -> 518 us (-14 %), linear1
// 841 -> 761 (-9.5 %), linear2 774 -> 718 (-7.3 %), out_proj 176 -> 173; on all-zero operands 0.62-0.71 of the 833 TF ceiling.
__global__ __launch_bounds__(256, 1) void gemm_f16x2_w128_kernel(GemmX3Args g) {
  constexpr int TN = 256, HA_PL = HM * HLD, HW_PL = TN * HLD, HSTAGE = 2 * HA_PL + 2 * HW_PL;     // 64 KiB per stage
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * HSTAGE];
  constexpr int GM = 8;
  const int tiles_m = (g.M + HM - 1) / HM, tiles_n = g.N / TN;
  int tm, tn;
  tile_walk(blockIdx.x, tiles_m, tiles_n, GM, g.walk, tm, tn);
  const int m0 = tm * HM, n0 = tn * TN;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1, hi = lane >> 5, l31 = lane & 31;

  // DMA plan: wave w fetches ONE plane of a stage (0 / 1: A head / tail, 2 / 3: W head / tail): 16 instructions of 1 KiB = 16 rows
  // each; in the tile-major planes the 256 rows x 32 of a K tile are one contiguous 16 KiB run, so instruction j is base + j KiB.
  // Lane -> (row l / 4 of the 16, LDS chunk slot l % 4), swizzle on the global side (as in gemm_f16x2_kernel).
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
  // one k16 step on set F: 48 MFMAs in gemm_f16x2_kernel's order (per block: tail.head, head.tail, head.head).  Behind every third
  // MFMA: one fragment read of the other set (RD) and one LDS-DMA request (DM).
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

  const int nk = g.K / HK;                                        // >= 2 (launch_gemm_f16x2 checks)
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

  // epilogue: the arithmetic of gemm_f16x2_kernel's (fp32 rows, or the NEXT GEMM's operand planes), 4 x 4 blocks of 32 x 32 per wave.
  // One wave per SIMD: nobody hides a load's round trip, so nothing is loaded one dependent vector at a time (measured inside the
  // workload with gemm_f16x2_kernel's load -> wait -> use form: linear1 +8 %, QKV +3 % against the 8-wave kernel although the main
  // loop is 10-14 % faster): the tile's 256 bias values go through the (now idle) LDS once, and the four residual vectors of block
  // (i, jn + 1) are requested before block (i, jn) is processed.
  const bool has_bias = g.bias != nullptr, has_res = g.resid != nullptr, relu = g.act == ACT_RELU;
  float* const lbias = reinterpret_cast<float*>(lds);
  __syncthreads();                                   // every wave is past its last fragment reads: the stages are free
  if (has_bias && tid < 64) *reinterpret_cast<f32x4*>(lbias + tid * 4) = *reinterpret_cast<const f32x4*>(g.bias + n0 + tid * 4);
  __syncthreads();
  int mrow[4];                                       // residual row per row block (rows past M: clamped, never stored)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 128 + i * 32 + l31;
    const int mc = m < g.M ? m : g.M - 1;
    mrow[i] = (has_res && g.resid_rows) ? g.resid_rows[mc] : mc;
  }
  auto load_rr = [&](int t, f32x4 (&rr)[4]) {       // block t = 4 i + jn
    const float* rp = g.resid + (long)mrow[t >> 2] * g.ldr + n0 + wn * 128 + (t & 3) * 32 + 4 * hi;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) rr[g4] = *reinterpret_cast<const f32x4*>(rp + 8 * g4);
  };
  f32x4 rrA[4], rrB[4];
  if (has_res) load_rr(0, rrA);
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int i = t >> 2, jn = t & 3;
    const int m = m0 + wm * 128 + i * 32 + l31;
    const bool live = m < g.M;
    f32x4 (&rr)[4] = (t & 1) ? rrB : rrA;
    if (has_res && t + 1 < 16) load_rr(t + 1, (t & 1) ? rrA : rrB);
    unsigned hw[4][2], tw[4][2];
    bool bad = false;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int n = n0 + wn * 128 + jn * 32 + 8 * g4 + 4 * hi;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[i][jn][4 * g4 + e] * g.descale;
      if (has_bias) {
        const f32x4 bi = *reinterpret_cast<const f32x4*>(lbias + wn * 128 + jn * 32 + 8 * g4 + 4 * hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bi[e];
      }
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (has_res) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rr[g4][e] + v[e];
      }
      if (!g.out_planes) {
        if (live) *reinterpret_cast<f32x4*>(g.C + (long)m * g.ldc + n) = v;
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
      const long blk = ((long)(m0 / HM) * (g.N / HK) + (n0 + wn * 128 + jn * 32) / HK) * (HM * HK) + (long)(wm * 128 + i * 32 + l31) * HK;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
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
        if (live) {
          unsigned short* o = g.out_planes + blk + 16 * j + 8 * hi;
          *reinterpret_cast<u32x4v*>(o) = oh;
          *reinterpret_cast<u32x4v*>(o + g.out_plane) = ot;
        }
      }
      if (bad && live && g.range_flag) *g.range_flag = 1;
    }
  }
}

// Tile choice (tn = 0).  Two kernels cover every row count: 256 x 256 tiles (one 8-wave workgroup per CU; ~15 % fewer operand bytes
// and barriers per flop) and 128 x 128 tiles (two 4-wave workgroups per CU) -- measured on MI355X over M = 384 .. 31 616 on the four
// NAR shapes (tools/gemm_short_rows.py, profiles/r03_gemm_tiles.log) the 128 x 128 tile is never slower than 256 x 128, and which of
// the two wins is a matter of how the tile count quantises onto 256 CUs.  Cost model in units u of one 128 x 128 tile's work on a CU
// (fits every measured point within a few percent): a round of 256 x 256 tiles costs 4 u; the 128-row kernel runs two tiles per CU
// in 2.3 u, a lone one in 1.3 u.  All shapes give bit-identical sums (the per-element accumulation order does not depend on the tile).
// tn = 128 / 256 / -128: forced (256 x 128 / 256 x 256 on the 8-wave kernel / 128 x 128; benchmarks); 257: 256 x 256 on the 4-wave kernel.
void launch_gemm_f16x2(const GemmX3Args& g_in, hipStream_t s, int tn) {
  GemmX3Args g = g_in;
  if (!g.walk) g.walk = gemm_walk_env();
  int tm = HM;
  const int tn_in = tn == 257 ? 0 : tn;          // 257: 256 x 256 tiles forced ON THE FOUR-WAVE KERNEL (benchmarks)
  if (tn == 257) tn = 256;
  if (tn == 0) {
    const long mt256 = (g.M + 255) / 256, mt128 = (g.M + 127) / 128;
    const long t128 = mt128 * ((g.N + 127) / 128);
    const long rem = t128 % 512;
    const double c128 = (double)(t128 / 512) * 2.3 + (rem == 0 ? 0.0 : rem <= 256 ? 1.3 : 2.3);
    const double c256 = g.N % 256 == 0 ? (double)((mt256 * (g.N / 256) + 255) / 256) * 4.0 : 1e30;
    if (c256 < c128) tn = 256;
    else { tn = 128; tm = 128; }
  } else if (tn == -128 || tn == -129) { tm = tn == -129 ? 129 : 128; tn = 128; }      // -129: 128 x 128 tiles, two stages forced
  const bool two_stage = tm == 129;
  if (two_stage) tm = 128;
  const int tiles = ((g.M + tm - 1) / tm) * ((g.N + tn - 1) / tn);
  if (tiles <= 0) return;
#ifdef VX_DEV_PROBES
  static const int env_prio = [] { const char* e = getenv("VX_GEMM_PRIO_RT"); return e ? atoi(e) : -1; }();
  const int prio = g.dev_variant > 0 ? g.dev_variant : (env_prio >= 0 ? env_prio : VX_GEMM_PRIO);
  if (tn == 256 && prio == 1) { hipLaunchKernelGGL((gemm_f16x2_kernel<256, 256, 2, 1>), dim3(tiles), dim3(512), 0, s, g); return; }
  if (tn == 256 && prio == 2) { hipLaunchKernelGGL((gemm_f16x2_kernel<256, 256, 2, 2>), dim3(tiles), dim3(512), 0, s, g); return; }
  if (tn == 256 && prio == 0) { hipLaunchKernelGGL((gemm_f16x2_kernel<256, 256, 2, 0>), dim3(tiles), dim3(512), 0, s, g); return; }
#endif
  // 256 x 256 tiles: four waves of 128 x 128 (round 5) unless forced back (tn = 256 from a benchmark, VX_GEMM_W128=0) or K is one tile
  static const bool w128 = [] { const char* e = getenv("VX_GEMM_W128"); return !(e && e[0] == '0'); }();
  if (tn == 256 && tn_in != 256 && w128 && g.K >= 2 * HK && g.N % 256 == 0)
    hipLaunchKernelGGL(gemm_f16x2_w128_kernel, dim3(tiles), dim3(256), 0, s, g);
  else if (tn == 256) hipLaunchKernelGGL((gemm_f16x2_kernel<256, 256, 2, VX_GEMM_PRIO>), dim3(tiles), dim3(512), 0, s, g);
  else if (tm == 128 && tiles <= 256 && !two_stage)      // at most one workgroup per CU: four LDS stages, requests three K tiles ahead
    hipLaunchKernelGGL((gemm_f16x2_kernel<128, 128, 4>), dim3(tiles), dim3(256), 0, s, g);
  else if (tm == 128) hipLaunchKernelGGL((gemm_f16x2_kernel<128, 128>), dim3(tiles), dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_f16x2_kernel<128, 256>), dim3(tiles), dim3(512), 0, s, g);
}

}  // namespace vx
