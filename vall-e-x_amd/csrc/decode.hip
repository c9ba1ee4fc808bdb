// AR decode step for gfx950: the HBM-bound core of VALLE.inference (models/vallex.py:528-598).
//
// One step = one new codec frame for every active row of the micro-batch (<= 32 rows).  Per layer the reference
// runs norm1 -> QKV linear -> cat(past_kv) -> q.k^T -> softmax -> .v -> out_proj -> +res -> norm2 -> linear1 -> ReLU
// -> linear2 -> +res on ONE row per sequence (modules/transformer.py:337-347, modules/activation.py:142-167), copying
// the whole KV cache with torch.cat every layer.  Here:
//
//   * the KV cache is an in-place arena  [(b*16 + head)][t][64] fp32 per layer: a head's keys/values are ONE
//     contiguous stream; a step appends one 256-B row per head and reads 512*ctx bytes per (row, head);
//   * dec_attn streams that cache with 16 lanes per 256-B row (float4 per lane, 4 rows = 1 KiB per wave load,
//     16 KiB in flight per wave), DPP row reductions for q.k, online softmax per 16-lane group, and combines
//     groups -> waves -> (optionally) ctx-splits;
//   * the projections are weight-streaming skinny GEMMs on the f32 MFMA: the batch (padded to 32 rows) is the
//     32-wide MFMA column block, weights are pre-packed at load into the lane-linear image the MFMA A operand
//     wants, so every wave load is 1 KiB contiguous and each weight byte is read exactly once per step
//     (non-temporal).  Split-K partial slabs are summed in the consumer's prologue -- no atomics, bit-stable.
//   * per-row lengths / positions / EOS flags live on the device, so the whole step is a fixed launch sequence
//     (hipGraph-capturable) and the host only polls `active` every few steps.
#include <vector>

#include "vx_common.h"

namespace vx {

// ------------------------------------------------------------------------------------------------------------
// packed images
//   weight : Wp[((nt*KB + kb)*64 + lane)*4 + j] = W[nt*32 + (lane&31)][kb*8 + 4*(lane>>5) + j]   (KB = K/8)
//   x      : xp[(kb*64 + b + 32*hi)*4 + j]      = x[b][kb*8 + 4*hi + j]                            (b < 32)
// MFMA step j of k-block kb contracts k = kb*8 + 4*(lane>>5) + j on both operands (a bijection of the 8 k's).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ W, int N, int K,
                                                          float* __restrict__ Wp, int Npad) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;         // float4 index
  const int KB = K / 8;
  const long total = (long)(Npad / 32) * KB * 64;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  const long t = i >> 6;
  const int kb = (int)(t % KB), nt = (int)(t / KB);
  const int n = nt * 32 + (lane & 31), k = kb * 8 + 4 * (lane >> 5);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (n < N) v = *reinterpret_cast<const f32x4*>(W + (long)n * K + k);
  *reinterpret_cast<f32x4*>(Wp + i * 4) = v;
}

void launch_pack_weight(const float* W, int N, int K, float* Wp, int Npad, hipStream_t s) {
  const long total = (long)(Npad / 32) * (K / 8) * 64;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, N, K, Wp, Npad);
}

// ------------------------------------------------------------------------------------------------------------
// skinny GEMM: partial[ks][b][n] = sum_{k in slice ks} x[b][k] * W[n][k]
// grid = (Npad/32, splitk); 4 waves split the block's K slice and reduce through LDS.
// ------------------------------------------------------------------------------------------------------------
#ifdef VX_DEV_PROBES
// Development timeline (tools/step_timeline.py, dev library only): thread 0 of every workgroup of the decode kernels stores
// the 100 MHz wall clock at a few points; one shared clock, so the gaps BETWEEN kernels show up as well.
// vx_stamps[type][workgroup < 512][slot < 8]; types: 0 QKV, 1 linear2, 2 predict, 3 linear1, 4 reduce+LN<16>, 5 <8>, 6 dec_attn, 7 sampler
__device__ unsigned long long vx_stamps[8 * 512 * 8];
#define VX_STAMP(TYPE, SLOT)                                                                              \
  do {                                                                                                    \
    if (threadIdx.x == 0) {                                                                               \
      const int bl_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                      \
      if (bl_ < 512) vx_stamps[((TYPE) * 512 + bl_) * 8 + (SLOT)] = wall_clock64();                         \
    }                                                                                                     \
  } while (0)
void dev_read_stamps(unsigned long long* out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(vx_stamps), sizeof(unsigned long long) * 8 * 512 * 8);
}
void dev_clear_stamps() {
  static std::vector<unsigned long long> z(8 * 512 * 8, 0ull);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(vx_stamps), z.data(), z.size() * sizeof(unsigned long long));
}
#else
#define VX_STAMP(TYPE, SLOT)
#endif

// Results of the kernels of the 5 .. 32-row decode chain are stored WRITE-THROUGH at agent scope (`sc1`).  The step is a chain of
// dependent launches and a kernel ends with the release of its writes -- the write-back of whatever its XCD's L2 still holds dirty
// (the 8 L2s are not coherent with each other); a result that went through to memory when it was stored leaves nothing to write
// back.  tools/ubench/boundary_cost.hip: 0.2-0.3 us less per link of a graph chain (profiles/r06_boundary_cost.txt); the 32-row
// step: AR 454.7 -> 446.9 ms per batch, same ids digest (profiles/r06_wt_stores_ab.log).  Same values, bit-identical.  NOT for the
// <= 4-row chain: its consumers rebuild their input rows in EVERY workgroup and find a plainly stored result in the L2 of the XCD
// that wrote it (1 row: AR +5.3 ms, 4 rows: +9.7 ms with write-through, same log), and not for the rows a step appends to the K / V
// cache or the sampler's row state (+5 ms at 32 rows).  The consumer of a result is always a LATER launch, so the asm needs no
// ordering against this kernel's own loads.  -DVX_DEC_WT=0: plain stores everywhere (A/B builds).
#ifndef VX_DEC_WT
#define VX_DEC_WT 1
#endif
__device__ __forceinline__ void store_result(float* p, const f32x4& v, bool wt = true) {
  if (VX_DEC_WT != 0 && wt) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  else *reinterpret_cast<f32x4*>(p) = v;
}
__device__ __forceinline__ void store_result(float* p, float v, bool wt = true) {
  if (VX_DEC_WT != 0 && wt) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  else *p = v;
}

// output: raw split-K partial slab [ks][b][n] (row-major per batch row; consumers: dec_attn, dec_reduce_ln_pack, dec_sample)
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const float* __restrict__ Wp, const float* __restrict__ xp,
                                                          float* __restrict__ out, int Npad, int K, int splitk, int wt) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  const int stype_ = Npad == 3 * D_MODEL ? 0 : (Npad == D_MODEL ? 1 : 2);
  (void)stype_;
  VX_STAMP(stype_, 0);
  // (no "every row has finished" early exit here: reading that flag -- written by the previous step's sampler -- costs a
  // memory round trip before the first weight load of EVERY launch; steps after the last EOS are bounded by sync_every)
  const int nt = blockIdx.x, ks = blockIdx.y;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int KB = K / 8;
  const int kb_per_wave = KB / (splitk * 4);
  const int kb0 = (ks * 4 + wid) * kb_per_wave;
  const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + ((long)nt * KB + kb0) * 64 + lane;
  const f32x4* xq = reinterpret_cast<const f32x4*>(xp) + (long)kb0 * 64 + lane;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  f32x4 w[8], x[8], w2[8], x2[8];
  // x first: it is L2-resident and returns early, so the MFMAs can start as soon as the first weight tile lands.  Two register
  // buffers: the loads of round r + 1 are requested BEFORE the MFMAs of round r (linear2, K = 4096, has two rounds per wave; a
  // single buffer left HBM idle for the length of a round's 32 MFMAs).  Same MFMA sequence either way.
#define VX_SK_LOAD(WW, XX, I)                                                                      \
  _Pragma("unroll") for (int u = 0; u < 8; ++u) XX[u] = xq[(long)((I) + u) * 64];                   \
  _Pragma("unroll") for (int u = 0; u < 8; ++u) WW[u] = __builtin_nontemporal_load(wp + (long)((I) + u) * 64);
#define VX_SK_MFMA(WW, XX)                                                                         \
  _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                     \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(WW[u][j], XX[u][j], acc, 0, 0, 0);
  VX_SK_LOAD(w, x, 0)
  VX_STAMP(stype_, 1);
  for (int i = 0;;) {
    if (i + 8 < kb_per_wave) { VX_SK_LOAD(w2, x2, i + 8) }
    VX_SK_MFMA(w, x)
    i += 8;
    if (i >= kb_per_wave) break;
    if (i + 8 < kb_per_wave) { VX_SK_LOAD(w, x, i + 8) }
    VX_SK_MFMA(w2, x2)
    i += 8;
    if (i >= kb_per_wave) break;
  }
#undef VX_SK_LOAD
#undef VX_SK_MFMA
  VX_STAMP(stype_, 2);

  // acc[r] = out[b = lane&31][n = nt*32 + (r&3) + 8*(r>>2) + 4*(lane>>5)].  The four waves' partial sums are combined
  // in parallel: wave w finishes registers 4w..4w+3 (one 16-byte store per lane), summing the waves in ascending order.
#pragma unroll
  for (int r = 0; r < 16; ++r) red[((wid * 16 + r) * 64) + lane] = acc[r];
  __syncthreads();
  VX_STAMP(stype_, 3);
  f32x4 t;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float* rp = red + ((wid * 4 + j) * 64) + lane;
    t[j] = ((rp[0] + rp[16 * 64]) + rp[2 * 16 * 64]) + rp[3 * 16 * 64];
  }
  float* dst = out + ((long)ks * MB + (lane & 31)) * Npad + nt * 32 + 4 * (lane >> 5);
  store_result(dst + wid * 8, t, wt != 0);
  VX_STAMP(stype_, 4);
}

// wt: write-through result stores (store_result above) -- the 5 .. 32-row decode chain only
void launch_skinny_gemm(const float* Wp, const float* xp, float* partial, int Npad, int K, int splitk,
                        hipStream_t s, bool wt) {
  hipLaunchKernelGGL(skinny_gemm_kernel, dim3(Npad / 32, splitk), dim3(256), 0, s, Wp, xp, partial, Npad, K, splitk, wt ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------------------
// DEFAULT in_proj of the 32-row decode chain since round 5 (VX_QKV_BALANCED=0 reverts to skinny_gemm_kernel with 4 K slices; the two
// sum q in a different order, so their logits are not bit-identical to each other -- every golden is bit-exact under both, tools/
// gpu_call.sh `switches`): the in_proj GEMM of the decode step on a grid that divides the chip.  skinny_gemm_kernel runs it as
// 96 column tiles x 4 K slices = 384 workgroups on 256 CUs: half the CUs stream two 32 KB tiles, the other half one, and the launch
// ends with the loaded half (profiles/r02_step_timeline.log: average workgroup done at 3.7 us, last at 5.3 us; linear2 with its 256
// workgroups 4.8 / 5.6).  Here the 32 column tiles of q are cut into EIGHT K slices (256 workgroups x 16 KB) and the 64 tiles of k, v
// stay at four (256 workgroups x 32 KB): 512 workgroups, dispatched in block-id order two per CU, one of each kind = 48 KB on every CU.
// k and v: the arithmetic of skinny_gemm_kernel with splitk = 4, operation for operation (same slabs).  q: eight slabs instead of four
// (a different summation order: dec_attn_kernel<*, SK_QKV_BALANCED> sums them, and the goldens decide).  Slab layout unchanged,
// out[(ks * MB + b) * 3072 + n]; the k / v columns of slabs 4..7 are never written or read.
__global__ __launch_bounds__(256, 2) void skinny_qkv_bal_kernel(const float* __restrict__ Wp, const float* __restrict__ xp,
                                                             float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  constexpr int Npad = 3 * D_MODEL, KB = D_MODEL / 8;
  const int id = blockIdx.x;
  const bool isq = id < 256;                                       // uniform over the workgroup
  const int nt = isq ? (id & 31) : 32 + ((id - 256) & 63);
  const int ks = isq ? (id >> 5) : ((id - 256) >> 6);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int kb0 = (ks * 4 + wid) * (isq ? 4 : 8);                  // k-blocks of 8 per wave: 4 (q, 8 slices) or 8 (k / v, 4 slices)
  const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + ((long)nt * KB + kb0) * 64 + lane;
  const f32x4* xq = reinterpret_cast<const f32x4*>(xp) + (long)kb0 * 64 + lane;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f32x4 w[8], x[8];
  if (isq) {
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = xq[(long)u * 64];
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = __builtin_nontemporal_load(wp + (long)u * 64);
    __builtin_amdgcn_sched_barrier(0);      // every request before the first MFMA (left alone the scheduler sinks the loads to their uses)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[u][j], x[u][j], acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = xq[(long)u * 64];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = __builtin_nontemporal_load(wp + (long)u * 64);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[u][j], x[u][j], acc, 0, 0, 0);
  }
  // the four waves' partial sums, combined exactly as in skinny_gemm_kernel (ascending wave order)
#pragma unroll
  for (int r = 0; r < 16; ++r) red[((wid * 16 + r) * 64) + lane] = acc[r];
  __syncthreads();
  f32x4 t;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float* rp = red + ((wid * 4 + j) * 64) + lane;
    t[j] = ((rp[0] + rp[16 * 64]) + rp[2 * 16 * 64]) + rp[3 * 16 * 64];
  }
  float* dst = out + ((long)ks * MB + (lane & 31)) * Npad + nt * 32 + 4 * (lane >> 5);
  store_result(dst + wid * 8, t);
}

void launch_skinny_qkv_balanced(const float* Wp, const float* xp, float* partial, hipStream_t s) {
  hipLaunchKernelGGL(skinny_qkv_bal_kernel, dim3(512), dim3(256), 0, s, Wp, xp, partial);
}

// ------------------------------------------------------------------------------------------------------------
// linear1 (N = 4096) with its epilogue fused and NO split-K: 16-row weight tiles on v_mfma_f32_16x16x4_f32 give
// 256 workgroups (one per CU), 8 waves each splitting K; the 32 batch rows are two 16-wide MFMA column blocks that
// share one weight fragment.  out = relu(x.W^T + bias) is written straight into linear2's packed-x image
// (modules/transformer.py:371-373).
//   weight image: W16[((nt*KB16 + kb)*64 + lane)*4 + j] = W[16nt + (lane&15)][16kb + 4(lane>>4) + j]   (KB16 = K/16)
//   x fragment  : read from the 32-row packed image: x[b = 16half + (lane&15)][16kb + 4(lane>>4) + j]
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weight16_kernel(const float* __restrict__ W, int N, int K,
                                                            float* __restrict__ Wp) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;         // float4 index
  const int KB = K / 16;
  const long total = (long)(N / 16) * KB * 64;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  const long t = i >> 6;
  const int kb = (int)(t % KB), nt = (int)(t / KB);
  const int n = nt * 16 + (lane & 15), k = kb * 16 + 4 * (lane >> 4);
  *reinterpret_cast<f32x4*>(Wp + i * 4) = *reinterpret_cast<const f32x4*>(W + (long)n * K + k);
}

void launch_pack_weight16(const float* W, int N, int K, float* Wp, hipStream_t s) {
  const long total = (long)(N / 16) * (K / 16) * 64;
  hipLaunchKernelGGL(pack_weight16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, N, K, Wp);
}

constexpr int S16_WAVES = 8;

__global__ __launch_bounds__(S16_WAVES * 64) void skinny16_relu_pack_kernel(const float* __restrict__ W16,
                                                                            const float* __restrict__ xp,
                                                                            const float* __restrict__ bias,
                                                                            float* __restrict__ xp_out, int K) {
  __shared__ __attribute__((aligned(16))) float red[S16_WAVES * 8 * 64];
  const int nt = blockIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int KB = K / 16, per = KB / S16_WAVES;        // K = 1024: 64 k-blocks, 8 per wave
  const int kb0 = wid * per;
  const int kg = lane >> 4, bl = lane & 15;
  const f32x4* wp = reinterpret_cast<const f32x4*>(W16) + ((long)nt * KB + kb0) * 64 + lane;
  // float4 index of x[b = bl (+16)][16 kb + 4 kg ..] in the 32-row image: (2kb + (kg>>1))*64 + b + 32*(kg&1)
  const f32x4* xq = reinterpret_cast<const f32x4*>(xp) + (long)(2 * kb0 + (kg >> 1)) * 64 + bl + 32 * (kg & 1);

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  f32x4 w[8], x0[8], x1[8];
  // requested with the first weights: a load in the epilogue would put one more memory round trip behind the MFMAs
  const f32x4 bi = *reinterpret_cast<const f32x4*>(bias + nt * 16 + 4 * kg);
#define VX_S16_LOAD(I)                                                                                                   \
  _Pragma("unroll") for (int u = 0; u < 8; ++u) { x0[u] = xq[(long)((I) + u) * 128]; x1[u] = xq[(long)((I) + u) * 128 + 16]; } \
  _Pragma("unroll") for (int u = 0; u < 8; ++u) w[u] = __builtin_nontemporal_load(wp + (long)((I) + u) * 64);
  VX_STAMP(3, 0);
  VX_S16_LOAD(0)
  VX_STAMP(3, 1);
  for (int i = 0;;) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u][j], x0[u][j], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u][j], x1[u][j], acc1, 0, 0, 0);
      }
    i += 8;
    if (i >= per) break;
    VX_S16_LOAD(i)
  }
#undef VX_S16_LOAD
  VX_STAMP(3, 2);
  // acc{h}[r] = out[b = 16h + (lane&15)][n = 16nt + 4(lane>>4) + r].  Wave 0 finishes acc0 and wave 1 acc1 (each sums the
  // eight waves' partials in ascending order, adds the bias, applies ReLU and stores its half of the packed image).
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[(wid * 8 + r) * 64 + lane] = acc0[r];
    red[(wid * 8 + 4 + r) * 64 + lane] = acc1[r];
  }
  __syncthreads();
  if (wid < 2) {
    f32x4 a4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sum = red[(wid * 4 + r) * 64 + lane];
#pragma unroll
      for (int w2 = 1; w2 < S16_WAVES; ++w2) sum += red[(w2 * 8 + wid * 4 + r) * 64 + lane];
      a4[r] = sum;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) a4[r] = fmaxf(a4[r] + bi[r], 0.f);
    // n = 16nt + 4kg + r  ->  linear2's k: kb = 2nt + (kg>>1), hi = kg&1, j = r
    float* o = xp_out + (((long)(2 * nt + (kg >> 1)) * 64) + bl + 32 * (kg & 1)) * 4;
    store_result(o + wid * 16 * 4, a4);
  }
  VX_STAMP(3, 4);
}

void launch_skinny16_relu_pack(const float* W16, const float* xp, const float* bias, float* xp_out, int N, int K,
                               hipStream_t s) {
  hipLaunchKernelGGL(skinny16_relu_pack_kernel, dim3(N / 16), dim3(S16_WAVES * 64), 0, s, W16, xp, bias, xp_out, K);
}

// ------------------------------------------------------------------------------------------------------------
// row kernels of the step: one 256-thread block per batch row, thread t owns columns 4t..4t+3 of the 1024.
// ------------------------------------------------------------------------------------------------------------
// Sum over the 64 lanes, the same value in every lane, without the LDS crossbar: four DPP steps inside each 16-lane row
// (every lane of a row ends up with the row total), row_bcast15 / row_bcast31 carry the totals of rows 0-2 up into row 3,
// lane 63 holds the wave total and is broadcast through an SGPR.  (Six ds_bpermute round trips, ~400 cycles, become ~50;
// the decode step runs 50 of these reductions back to back on its latency chain.)
__device__ __forceinline__ float wave_sum64(float v) {
  int x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));   // row_half_mirror
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));   // row_mirror
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false));  // row_bcast15 -> rows 1, 3
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false));  // row_bcast31 -> rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// One 64-lane wave per batch row: lane l owns float4 columns c4 = l + 64 i (i < 4) of the 1024 -- no LDS, no barriers.
// LayerNorm (two-pass, registers) written in the packed-x image: float4 column c4 -> kb = c4>>1, hi = c4&1.
__device__ __forceinline__ void ln_pack_row(const f32x4 (&v)[4], int b, const f32x4 (&gg)[4], const f32x4 (&be)[4],
                                            float* __restrict__ xp, bool wt = false) {
  const int lane = threadIdx.x;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  const float mean = wave_sum64(s) * (1.0f / D_MODEL);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum64(q) * (1.0f / D_MODEL) + LN_EPS);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c4 = lane + 64 * i;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * gg[i][e] + be[i][e];
    store_result(xp + (((long)(c4 >> 1) * 64) + b + 32 * (c4 & 1)) * 4, o, wt);
  }
}

// h[b] = resid[b] + (sum_ks partial[ks][b] + bias);  xp = pack(LN(h)).  partial may be null (splitk = 0).
// modules/transformer.py:345-346 (x = x + attn_out ; x = x + ff(norm2(x))) fused with the next norm.
// One 256-thread block per batch row, thread t owns float4 column t of the 1024: SK + 4 independent 16-byte loads per thread
// (a single wave per row had to pull 36-68 KB by itself: 64 loads per lane), slabs summed in ascending ks order, LayerNorm
// statistics (two-pass) through two 4-wave LDS exchanges.
template <int SK>
__global__ __launch_bounds__(256) void dec_reduce_ln_pack_kernel(const float* __restrict__ partial, int npad,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ resid,
                                                                 float* __restrict__ h, const float* __restrict__ g,
                                                                 const float* __restrict__ bb,
                                                                 float* __restrict__ xp) {
  __shared__ float red[2][4];
  const int b = blockIdx.x, t = threadIdx.x, wid = t >> 6, c = t * 4;
  VX_STAMP(SK == 16 ? 4 : 5, 0);
  f32x4 p[SK > 0 ? SK : 1];
#pragma unroll
  for (int ks = 0; ks < SK; ++ks) p[ks] = *reinterpret_cast<const f32x4*>(partial + ((long)ks * MB + b) * npad + c);
  const f32x4 gg = *reinterpret_cast<const f32x4*>(g + c), be = *reinterpret_cast<const f32x4*>(bb + c);
  f32x4 r = {0.f, 0.f, 0.f, 0.f}, bi = {0.f, 0.f, 0.f, 0.f};
  if (resid) r = *reinterpret_cast<const f32x4*>(resid + (long)b * D_MODEL + c);
  if (SK > 0 && bias) bi = *reinterpret_cast<const f32x4*>(bias + c);
  f32x4 v = SK > 0 ? p[0] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 1; ks < SK; ++ks)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += p[ks][e];
  if (SK > 0 && bias)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += bi[e];
  if (resid)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = r[e] + v[e];
  if (h) store_result(h + (long)b * D_MODEL + c, v);
  VX_STAMP(SK == 16 ? 4 : 5, 2);
  // LayerNorm (F.layer_norm, eps 1e-5): mean, then the centred second moment
  float s1 = wave_sum64((v[0] + v[1]) + (v[2] + v[3]));
  if ((t & 63) == 0) red[0][wid] = s1;
  __syncthreads();
  const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * (1.0f / D_MODEL);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
  q = wave_sum64(q);
  if ((t & 63) == 0) red[1][wid] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * (1.0f / D_MODEL) + LN_EPS);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * gg[e] + be[e];
  // packed-x image: float4 column c4 = t -> kb = c4 >> 1, hi = c4 & 1
  store_result(xp + (((long)(t >> 1) * 64) + b + 32 * (t & 1)) * 4, o);
  VX_STAMP(SK == 16 ? 4 : 5, 4);
}

void launch_dec_reduce_ln_pack(const float* partial, int splitk, int npad, const float* bias, const float* resid,
                               float* h, const float* g, const float* b, float* xp, int batch, hipStream_t s) {
#define VX_RLP(SKV) hipLaunchKernelGGL((dec_reduce_ln_pack_kernel<SKV>), dim3(batch), dim3(256), 0, s, partial, npad, bias, resid, h, g, b, xp)
  if (splitk == 0) VX_RLP(0);
  else if (splitk == 4) VX_RLP(4);
  else if (splitk == 8) VX_RLP(8);
  else VX_RLP(16);
#undef VX_RLP
}

// Consumer of the fused out_proj WITH context splits (dec_attn_kernel<true, *, true>): row b's attention output through out_proj is
//   sum_h sum_s w(h, s) * slab[h * NS + s][b][:],  w(h, s) = e^(m_s - M_h) / sum_s' e^(m_s' - M_h) l_s'   (the combine of
// dec_attn_combine_kernel applied BEHIND the head's W_o slice: out_proj is linear), summed in ascending (h, s) order; then bias,
// residual, LayerNorm and pack exactly as dec_reduce_ln_pack_kernel.  One 256-thread block per row.
template <int NS>
__global__ __launch_bounds__(256) void dec_reduce_ln_split_kernel(const float* __restrict__ slabs, const float* __restrict__ part_ml,
                                                                  const float* __restrict__ bias, const float* __restrict__ resid,
                                                                  float* __restrict__ h, const float* __restrict__ g,
                                                                  const float* __restrict__ bb, float* __restrict__ xp) {
  __shared__ float red[2][4];
  __shared__ float wsh[N_HEAD * NS];
  const int b = blockIdx.x, t = threadIdx.x, wid = t >> 6, c = t * 4;
  constexpr int HC = 4;                              // heads per chunk: HC * NS 16-byte loads in flight per thread
  f32x4 p[HC * NS];
  const float* sl = slabs + (long)b * D_MODEL + c;
#pragma unroll
  for (int i = 0; i < HC * NS; ++i) p[i] = *reinterpret_cast<const f32x4*>(sl + (long)i * MB * D_MODEL);
  if (t < N_HEAD) {                                  // thread t: the weights of head t
    const float* ml = part_ml + (long)(b * N_HEAD + t) * NS * 2;
    float m[NS], l[NS], mt = -1e30f;
#pragma unroll
    for (int s = 0; s < NS; ++s) { m[s] = ml[2 * s]; l[s] = ml[2 * s + 1]; mt = fmaxf(mt, m[s]); }
    float a[NS], lt = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) { a[s] = expf(m[s] - mt); lt += l[s] * a[s]; }
#pragma unroll
    for (int s = 0; s < NS; ++s) wsh[t * NS + s] = a[s] / lt;
  }
  const f32x4 gg = *reinterpret_cast<const f32x4*>(g + c), be = *reinterpret_cast<const f32x4*>(bb + c);
  const f32x4 r = *reinterpret_cast<const f32x4*>(resid + (long)b * D_MODEL + c);
  const f32x4 bi = *reinterpret_cast<const f32x4*>(bias + c);
  __syncthreads();
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int hc = 0; hc < N_HEAD / HC; ++hc) {
    f32x4 q[HC * NS];
    if (hc + 1 < N_HEAD / HC) {
#pragma unroll
      for (int i = 0; i < HC * NS; ++i) q[i] = *reinterpret_cast<const f32x4*>(sl + (long)((hc + 1) * HC * NS + i) * MB * D_MODEL);
    }
#pragma unroll
    for (int i = 0; i < HC * NS; ++i) {
      const float w = wsh[hc * HC * NS + i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(w, p[i][e], v[e]);
    }
#pragma unroll
    for (int i = 0; i < HC * NS; ++i) p[i] = q[i];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = r[e] + (v[e] + bi[e]);
  store_result(h + (long)b * D_MODEL + c, v);
  float s1 = wave_sum64((v[0] + v[1]) + (v[2] + v[3]));
  if ((t & 63) == 0) red[0][wid] = s1;
  __syncthreads();
  const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * (1.0f / D_MODEL);
  float q2 = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q2 += d * d; }
  q2 = wave_sum64(q2);
  if ((t & 63) == 0) red[1][wid] = q2;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * (1.0f / D_MODEL) + LN_EPS);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * gg[e] + be[e];
  store_result(xp + (((long)(t >> 1) * 64) + b + 32 * (t & 1)) * 4, o);
}

// false = this split count is not instantiated (nothing launched)
bool launch_dec_reduce_ln_split(const float* slabs, const float* part_ml, int nsplit, const float* bias, const float* resid, float* h,
                                const float* g, const float* b, float* xp, int batch, hipStream_t s) {
#define VX_RLS(NSV) hipLaunchKernelGGL((dec_reduce_ln_split_kernel<NSV>), dim3(batch), dim3(256), 0, s, slabs, part_ml, bias, resid, h, g, b, xp)
  if (nsplit == 2) VX_RLS(2);
  else if (nsplit == 3) VX_RLS(3);
  else if (nsplit == 4) VX_RLS(4);
  else return false;
#undef VX_RLS
  return true;
}

// ------------------------------------------------------------------------------------------------------------
// Small batches (<= SB_MAX rows; BASELINE config 2 is ONE utterance): the step is a pure chain of launch latencies, so the tiny
// kernels between the weight-streaming GEMMs are folded into the GEMM that consumes their output.  EVERY workgroup of the
// consumer recomputes the few rows it needs (a row is 4 KB; SK + 1 slabs of it per row come out of L2) while its first weight
// tile is already in flight:
//   * reduce + residual + LayerNorm (dec_reduce_ln_pack)  -> prologue of the QKV / linear1 / predict GEMM,
//   * the context-split combine of dec_attn (dec_attn_combine) -> prologue of the out_proj GEMM.
// The arithmetic of a row is the code of the stand-alone kernels, operation for operation, so a row's ids do not depend on the
// batch size it ran in.  The residual stream h ping-pongs between two buffers (workgroup 0 writes the new h while the others
// still read the old one).  12 x 5 + 2 launches per step instead of 12 x 8 + 2.
// ------------------------------------------------------------------------------------------------------------
// sum over the 16 lanes of a DPP row (every lane of the row gets the total)
__device__ __forceinline__ float dpp_sum16(float x) {
  int v = __builtin_bit_cast(int, x);
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v = __builtin_bit_cast(int, x);
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v = __builtin_bit_cast(int, x);
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));   // row_half_mirror
  v = __builtin_bit_cast(int, x);
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));   // row_mirror
  return x;
}

constexpr int SB_MAX = SB_ROWS;
constexpr float NEG_BIG = -1e30f;

// one row by a team of 256 threads (thread tt owns float4 column tt); `red` = this team's [2][4] LDS words.  Contains two
// __syncthreads(): every thread of the block must call it the same number of times (live = false: barriers only).
template <int SK>
__device__ __forceinline__ void sb_reduce_ln_row(bool live, const float* __restrict__ partial, int npad, const float* __restrict__ bias,
                                                 const float* __restrict__ resid, float* __restrict__ h_out,
                                                 const float* __restrict__ g, const float* __restrict__ bb, int m, int tt,
                                                 float (*red)[4], float* __restrict__ xs_row) {
  const int wid = tt >> 6, c = tt * 4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f}, gg = v, be = v;
  if (live) {
    f32x4 p[SK];
#pragma unroll
    for (int ks = 0; ks < SK; ++ks) p[ks] = *reinterpret_cast<const f32x4*>(partial + ((long)ks * MB + m) * npad + c);
    gg = *reinterpret_cast<const f32x4*>(g + c);
    be = *reinterpret_cast<const f32x4*>(bb + c);
    const f32x4 r = *reinterpret_cast<const f32x4*>(resid + (long)m * D_MODEL + c);
    f32x4 bi = {0.f, 0.f, 0.f, 0.f};
    if (bias) bi = *reinterpret_cast<const f32x4*>(bias + c);
    v = p[0];
#pragma unroll
    for (int ks = 1; ks < SK; ++ks)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += p[ks][e];
    if (bias)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += bi[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = r[e] + v[e];
    if (h_out) *reinterpret_cast<f32x4*>(h_out + (long)m * D_MODEL + c) = v;
  }
  const float s1 = wave_sum64((v[0] + v[1]) + (v[2] + v[3]));
  if ((tt & 63) == 0) red[0][wid] = s1;
  __syncthreads();
  const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * (1.0f / D_MODEL);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
  q = wave_sum64(q);
  if ((tt & 63) == 0) red[1][wid] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * (1.0f / D_MODEL) + LN_EPS);
  if (live) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * gg[e] + be[e];
    *reinterpret_cast<f32x4*>(xs_row + c) = o;
  }
}

// attention output of row m from the context-split partials: the body of dec_attn_combine_kernel (thread tt -> head tt >> 4,
// float4 chunk tt & 15; splits in ascending order, ot / lt per element)
// NS (the number of context splits) is a compile-time constant so that all 2 NS loads are requested together: a runtime loop
// walked the splits one memory round trip at a time, twice (10.2 us per out_proj launch at batch 1, profiles/r03_b1_kernel_stats_v1.csv)
// qk_new != null (the producer was dec_attn_qkv_kernel): the LAST partial is the new token's own -- (m = q . k_new, l = 1, o = v_new);
// its m is formed here from qk_new[(row, head)][0] = q / 8 and [1] = k_new (the 16 threads of a head are one DPP row)
template <int NS>
__device__ __forceinline__ void sb_combine_row(const float* __restrict__ part_o, const float* __restrict__ part_ml, int m,
                                               int tt, float* __restrict__ xs_row, const float* __restrict__ qk_new) {
  const int h = tt >> 4, c = tt & 15;
  const long pi = (long)(m * N_HEAD + h) * NS;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 ml[NS];
  f32x4 po[NS];
  f32x4 qn = {0.f, 0.f, 0.f, 0.f}, kn = qn;
  if (qk_new) {
    qn = *reinterpret_cast<const f32x4*>(qk_new + ((long)(m * N_HEAD + h) * 2) * D_HEAD + c * 4);
    kn = *reinterpret_cast<const f32x4*>(qk_new + ((long)(m * N_HEAD + h) * 2 + 1) * D_HEAD + c * 4);
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (!(qk_new && s == NS - 1)) ml[s] = *reinterpret_cast<const f32x2*>(part_ml + (pi + s) * 2);
    po[s] = *reinterpret_cast<const f32x4*>(part_o + (pi + s) * D_HEAD + c * 4);
  }
  {
    float d = qn[0] * kn[0] + qn[1] * kn[1] + qn[2] * kn[2] + qn[3] * kn[3];
    d = dpp_sum16(d);                                          // executed by every thread (full DPP rows)
    if (qk_new) ml[NS - 1] = f32x2{d, 1.0f};
  }
  float mt = NEG_BIG;
#pragma unroll
  for (int s = 0; s < NS; ++s) mt = fmaxf(mt, ml[s][0]);
  float lt = 0.f;
  f32x4 ot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const float a = expf(ml[s][0] - mt);
    lt += ml[s][1] * a;
#pragma unroll
    for (int e = 0; e < 4; ++e) ot[e] += po[s][e] * a;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) ot[e] = ot[e] / lt;
  *reinterpret_cast<f32x4*>(xs_row + h * D_HEAD + c * 4) = ot;
}

// skinny_gemm_kernel (K = 1024) whose x operand is computed by the prologue.  MODE 0: rows = LN(resid + sum of SK slabs + bias);
// MODE 1: rows = combine of the dec_attn partials (SK = the number of context splits).  The first weight tile is requested BEFORE
// the prologue.
template <int MODE, int SK>
__global__ __launch_bounds__(256) void skinny_gemm_sb_kernel(const float* __restrict__ Wp, float* __restrict__ out, int Npad, int splitk,
                                                             const float* __restrict__ partial, int pnpad, const float* __restrict__ bias,
                                                             const float* __restrict__ resid, float* __restrict__ h_out,
                                                             const float* __restrict__ g, const float* __restrict__ bb,
                                                             const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                             int nsplit, int M, const float* __restrict__ qk_new) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  __shared__ __attribute__((aligned(16))) float xs[SB_MAX][D_MODEL];
  __shared__ float st[2][4];
  constexpr int K = D_MODEL;
  const int nt = blockIdx.x, ks = blockIdx.y;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  constexpr int KB = K / 8;
  const int kb_per_wave = KB / (splitk * 4);
  const int kb0 = (ks * 4 + wid) * kb_per_wave;
  const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + ((long)nt * KB + kb0) * 64 + lane;
  f32x4 w[8], x[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) w[u] = __builtin_nontemporal_load(wp + (long)u * 64);
  const bool first = blockIdx.x == 0 && blockIdx.y == 0;
  for (int m = 0; m < M; ++m) {
    if (MODE == 0) sb_reduce_ln_row<SK>(true, partial, pnpad, bias, resid, first ? h_out : nullptr, g, bb, m, threadIdx.x, st, xs[m]);
    else sb_combine_row<SK>(part_o, part_ml, m, threadIdx.x, xs[m], qk_new);
  }
  __syncthreads();
  // x fragment of lane (b = lane & 31, hi = lane >> 5) for k-block kb: x[b][8 kb + 4 hi ..]; rows >= M are zero columns of the MFMA
  const int b = lane & 31, hi = lane >> 5;
  const float* xrow = xs[b < M ? b : 0] + 4 * hi;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int i = 0;;) {
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = b < M ? *reinterpret_cast<const f32x4*>(xrow + (long)(kb0 + i + u) * 8) : zero;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[u][j], x[u][j], acc, 0, 0, 0);
    i += 8;
    if (i >= kb_per_wave) break;
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = __builtin_nontemporal_load(wp + (long)(i + u) * 64);
  }
  // epilogue of skinny_gemm_kernel
#pragma unroll
  for (int r = 0; r < 16; ++r) red[((wid * 16 + r) * 64) + lane] = acc[r];
  __syncthreads();
  f32x4 t;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float* rp = red + ((wid * 4 + j) * 64) + lane;
    t[j] = ((rp[0] + rp[16 * 64]) + rp[2 * 16 * 64]) + rp[3 * 16 * 64];
  }
  float* dst = out + ((long)ks * MB + (lane & 31)) * Npad + nt * 32 + 4 * (lane >> 5);
  *reinterpret_cast<f32x4*>(dst + wid * 8) = t;
}

// The small-batch launchers compile the producer's split count into the kernel; a configuration that is not instantiated is
// reported to the caller (false, nothing launched) -- the engine asks sb_chain_supported() first and otherwise runs the general
// chain, and a launcher that still refuses turns into VX_EINVAL from the ABI call, never into a dead host process.
bool sb_chain_supported(int sk_l2, int sk_out, int nsplit, int batch) {
  return (sk_l2 == 8 || sk_l2 == 4) && sk_out == 4 && (nsplit == 16 || nsplit == 8) && batch <= SB_MAX;
}
// ... with norm1 + QKV folded into the attention launch (dec_attn_qkv_kernel): nsplit context splits + 1 partial for the new token
bool sb_qkv_chain_supported(int sk_l2, int sk_out, int nsplit, int batch) {
  return sk_l2 == 8 && sk_out == 4 && (nsplit == 16 || nsplit == 8 || nsplit == 4) && batch <= SB_MAX;
}

bool launch_skinny_gemm_sb_ln(const float* Wp, float* partial_out, int Npad, int splitk, const float* partial_in, int sk_in,
                              const float* bias, const float* resid, float* h_out, const float* g, const float* b, int batch,
                              hipStream_t s) {
  if (sk_in == 8)
    hipLaunchKernelGGL((skinny_gemm_sb_kernel<0, 8>), dim3(Npad / 32, splitk), dim3(256), 0, s, Wp, partial_out, Npad, splitk, partial_in,
                       D_MODEL, bias, resid, h_out, g, b, nullptr, nullptr, 0, batch, nullptr);
  else if (sk_in == 4)
    hipLaunchKernelGGL((skinny_gemm_sb_kernel<0, 4>), dim3(Npad / 32, splitk), dim3(256), 0, s, Wp, partial_out, Npad, splitk, partial_in,
                       D_MODEL, bias, resid, h_out, g, b, nullptr, nullptr, 0, batch, nullptr);
  else return false;                       // split-K factor of the producer not compiled in
  return true;
}

bool launch_skinny_gemm_sb_combine(const float* Wp, float* partial_out, int Npad, int splitk, const float* part_o, const float* part_ml,
                                   int nsplit, int batch, hipStream_t s, const float* qk_new) {
  // nsplit = number of partials per (row, head).  qk_new != null: the last one is the new token's (dec_attn_qkv_kernel).
  // (tried for 5 .. 8 rows as well -- BASELINE config 5 decodes 8 -- with 4 splits: 134.7 vs 138.5 audio-s/s, the 8-row prologue
  // costs more than the combine launch it removes; DESIGN.md dead-end table)
  if (batch > SB_MAX) return false;
#define VX_SBC(NSV)                                                                                                              \
  else if (nsplit == NSV)                                                                                                        \
    hipLaunchKernelGGL((skinny_gemm_sb_kernel<1, NSV>), dim3(Npad / 32, splitk), dim3(256), 0, s, Wp, partial_out, Npad, splitk, nullptr, 0, \
                       nullptr, nullptr, nullptr, nullptr, nullptr, part_o, part_ml, nsplit, batch, qk_new);
  if (false) {}
  VX_SBC(16) VX_SBC(8) VX_SBC(17) VX_SBC(9) VX_SBC(5)
#undef VX_SBC
  else return false;                       // this many context splits x rows are not compiled in
  return true;
}

// skinny16_relu_pack_kernel (linear1) with the reduce + residual + LayerNorm of its input rows as prologue; rows 16..31 of the
// MFMA column blocks do not exist at these batch sizes, so the second accumulator is dropped.
template <int SK>
__global__ __launch_bounds__(S16_WAVES * 64) void skinny16_sb_kernel(const float* __restrict__ W16, const float* __restrict__ bias,
                                                                     float* __restrict__ xp_out, const float* __restrict__ partial,
                                                                     const float* __restrict__ pbias, const float* __restrict__ resid,
                                                                     float* __restrict__ h_out, const float* __restrict__ g,
                                                                     const float* __restrict__ bb, int M) {
  __shared__ __attribute__((aligned(16))) float red[S16_WAVES * 4 * 64];
  __shared__ __attribute__((aligned(16))) float xs[SB_MAX][D_MODEL];
  __shared__ float st[2][2][4];
  constexpr int K = D_MODEL;
  const int nt = blockIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  constexpr int KB = K / 16, per = KB / S16_WAVES;
  const int kb0 = wid * per;
  const int kg = lane >> 4, bl = lane & 15;
  const f32x4* wp = reinterpret_cast<const f32x4*>(W16) + ((long)nt * KB + kb0) * 64 + lane;
  f32x4 w[8], x0[8];
  const f32x4 bi = *reinterpret_cast<const f32x4*>(bias + nt * 16 + 4 * kg);
#pragma unroll
  for (int u = 0; u < 8; ++u) w[u] = __builtin_nontemporal_load(wp + (long)u * 64);
  // two teams of 256 threads: team 0 rows 0, 2; team 1 rows 1, 3
  const int team = threadIdx.x >> 8, tt = threadIdx.x & 255;
  const bool first = blockIdx.x == 0;
  for (int r0 = 0; r0 < M; r0 += 2) {
    const int m = r0 + team;
    sb_reduce_ln_row<SK>(m < M, partial, D_MODEL, pbias, resid, first ? h_out : nullptr, g, bb, m < M ? m : 0, tt, st[team], xs[m < M ? m : 0]);
  }
  __syncthreads();
  const float* xrow = xs[bl < M ? bl : 0] + 4 * kg;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
  static_assert(per == 8, "one round of 8 k-blocks per wave at K = 1024");
#pragma unroll
  for (int u = 0; u < 8; ++u) x0[u] = bl < M ? *reinterpret_cast<const f32x4*>(xrow + (long)(kb0 + u) * 16) : zero;
#pragma unroll
  for (int u = 0; u < 8; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u][j], x0[u][j], acc0, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wid * 4 + r) * 64 + lane] = acc0[r];
  __syncthreads();
  if (wid == 0) {
    f32x4 a4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sum = red[r * 64 + lane];
#pragma unroll
      for (int w2 = 1; w2 < S16_WAVES; ++w2) sum += red[(w2 * 4 + r) * 64 + lane];
      a4[r] = sum;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) a4[r] = fmaxf(a4[r] + bi[r], 0.f);
    float* o = xp_out + (((long)(2 * nt + (kg >> 1)) * 64) + bl + 32 * (kg & 1)) * 4;
    *reinterpret_cast<f32x4*>(o) = a4;
  }
}

bool launch_skinny16_sb_ln(const float* W16, const float* bias, float* xp_out, int N, const float* partial_in, int sk_in,
                           const float* pbias, const float* resid, float* h_out, const float* g, const float* b, int batch,
                           hipStream_t s) {
  if (sk_in != 4) return false;            // split-K factor of the producer (out_proj) not compiled in
  hipLaunchKernelGGL((skinny16_sb_kernel<4>), dim3(N / 16), dim3(S16_WAVES * 64), 0, s, W16, bias, xp_out, partial_in, pbias, resid, h_out,
                     g, b, batch);
  return true;
}

// Start of a step: embed the newest token of each row at its audio position (the reference re-embeds all of y and
// keeps the last row, models/vallex.py:529-531,552-553), then norm1 of layer 0.
__global__ __launch_bounds__(64) void dec_embed_ln_pack_kernel(const int* __restrict__ tok,
                                                               const int* __restrict__ pos,
                                                               const float* __restrict__ tab,
                                                               const float* __restrict__ alpha,
                                                               const float* __restrict__ pe, float* __restrict__ h,
                                                               const float* __restrict__ g,
                                                               const float* __restrict__ bb,
                                                               float* __restrict__ xp) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const long trow = (long)tok[b] * D_MODEL, prow = (long)pos[b] * D_MODEL;
  const float a = alpha[0];
  f32x4 v[4], gg[4], be[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (lane + 64 * i) * 4;
    gg[i] = *reinterpret_cast<const f32x4*>(g + c);
    be[i] = *reinterpret_cast<const f32x4*>(bb + c);
    v[i] = *reinterpret_cast<const f32x4*>(tab + trow + c);
    const f32x4 p = *reinterpret_cast<const f32x4*>(pe + prow + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[i][e] = __fadd_rn(v[i][e], __fmul_rn(a, p[e]));
    *reinterpret_cast<f32x4*>(h + (long)b * D_MODEL + c) = v[i];
  }
  ln_pack_row(v, b, gg, be, xp);
}

void launch_dec_embed_ln_pack(const int* tok, const int* pos, const float* tab, const float* alpha, const float* pe,
                              float* h, const float* g, const float* b, float* xp, int batch, hipStream_t s) {
  hipLaunchKernelGGL(dec_embed_ln_pack_kernel, dim3(batch), dim3(64), 0, s, tok, pos, tab, alpha, pe, h, g, b, xp);
}

// ------------------------------------------------------------------------------------------------------------
// dec_attn: softmax(q.K^T/8).V over the cache for one new token per row  (modules/activation.py:148-165 with T=1).
// grid = (head, row, split).  Lane = (g = lane>>4 : row slot, c = lane&15 : float4 chunk of the 64-float row).
// ------------------------------------------------------------------------------------------------------------

// Workgroup barrier that orders LDS traffic only: this wave's LDS writes are complete (lgkmcnt), then s_barrier.  __syncthreads()
// carries a workgroup-scope fence that the compiler implements as s_waitcnt vmcnt(0): in dec_attn_qkv_kernel every barrier of the
// LayerNorm prologue would wait for the 256 KB of weight rows requested before it.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

constexpr int ATT_U = 4;              // rows per lane-group per buffer; two buffers in flight (2 x 8 KiB per wave)
constexpr int ATT_WAVES = 8;          // 512-thread workgroup
constexpr int ATT_STRIDE = ATT_WAVES * 4 * ATT_U;   // rows consumed per block iteration
static_assert(ATT_STRIDE == DEC_ATTN_TILE, "the engine's Tmax guard for the fused variant");

// FUSE_OUT (only launched with nsplit == 1): out_proj (modules/activation.py:166) is folded into the epilogue.  The
// workgroup of (head h, row b) multiplies its normalised 64-float head output with the head's 64 columns of W_o and writes
// the per-head partial  out_h[b][n] = sum_d W_o[n][64h + d] o[d]  (n < 1024); the 16 head slabs are summed in head order by
// the next kernel's prologue (dec_reduce_ln_pack<16>), exactly like split-K slabs.  W_o is stored head-major,
// wo_heads[((h * 16 + d/4) * 1024 + n) * 4 + d%4], so a wave reads 1 KiB runs; a head's 256 KiB slice is shared by the 32
// row-workgroups of that head, which all run on XCDs h % 8 (block id % 8) and find it in that XCD's L2.  This removes one
// weight-streaming launch + kernel boundary per layer from the latency chain of the step.
// The fused variant runs TWO rows of the same head per workgroup (16 waves: waves 0-7 stream launch slot y, waves 8-15 slot
// y + ceil(batch/2) -- with the engine's row order a long and a short context), so the head's 256 KiB W_o slice, whose
// L2 -> CU read (64 B/clk per CU) is what the epilogue costs, is read once per two rows.
// SPLIT (FUSE_OUT only, round 6: 5 .. 16 rows): the fused out_proj WITH context splits.  out_proj is linear, so the workgroup of split
// sp multiplies its UNNORMALISED partial output (o_s = sum p v, p = exp(s - m_s)) with the head's W_o slice and writes slab h * nsplit + sp;
// (m_s, l_s) go to part_ml as in the unfused kernel and dec_reduce_ln_split_kernel weighs the slabs: out = sum_h sum_s (e^(m_s - M_h) / L_h) W_o,h o_s.
// Two launches per layer fewer than dec_attn | dec_attn_combine | out_proj GEMM.
template <bool FUSE_OUT, int SK, bool SPLIT = false>
__global__ __launch_bounds__(ATT_WAVES * 64 * (FUSE_OUT ? 2 : 1), 4) void dec_attn_kernel(
    // the first 16 dwords arrive preloaded in SGPRs (-amdgpu-kernarg-preload-count=16, _build.py): everything the requests at the
    // head of the kernel are built from -- the K / V arena, the slot records, Tmax, batch -- sits there; an argument behind them
    // costs an s_load round trip (the grid's y extent too: it is derived from `batch` instead of read from the dispatch packet)
    float* __restrict__ kc, float* __restrict__ vc, const int* __restrict__ slot_meta, int Tmax, int batch, int nsplit, int,
    const float* __restrict__ qkv_partial, const float* __restrict__ qkv_bias, float* __restrict__ xp_out,
    float* __restrict__ part_o, float* __restrict__ part_ml, const float* __restrict__ wo_heads, float* __restrict__ out_heads) {
  constexpr int NR = FUSE_OUT ? 2 : 1;        // rows per workgroup
  __shared__ __attribute__((aligned(16))) float sh_o[NR][ATT_WAVES][64];
  __shared__ __attribute__((aligned(16))) float sh_ot[NR][64];
  __shared__ float sh_m[NR][ATT_WAVES], sh_l[NR][ATT_WAVES];
  // slot_meta[slot] = {batch row, cached rows incl. the new token, row still active, -}: ONE 16-byte load tells the
  // workgroup everything it needs before its first K/V load (a slot -> row -> ctx_len / active chain would be two
  // dependent round trips at the head of every launch).  The engine orders the slots so that the long contexts sit in
  // the first half and slot y + ceil(batch/2) holds a short one: unfused, the two share a CU (workgroups are dispatched
  // in block-id order, two per CU); fused, they share a workgroup.  Either way every CU streams about the same KV bytes.
  VX_STAMP(6, 0);
  const int r = FUSE_OUT ? (int)(threadIdx.x >> 9) : 0;
  const int gy = (batch + 1) >> 1;            // == gridDim.y of the fused launch
  const int slot = FUSE_OUT ? (int)blockIdx.y + r * gy : (int)blockIdx.y;
  const bool valid = slot < batch;
  const int h = blockIdx.x, sp = blockIdx.z;
  const int lane = threadIdx.x & 63, wid = (threadIdx.x >> 6) & (ATT_WAVES - 1), g = lane >> 4, c = lane & 15;
  // The KV arena is indexed by launch SLOT (the prefill scatters a sequence's K / V into the arena row of its slot, engine.hip):
  // the address of this workgroup's stream follows from the block id alone.
  const long head_base = ((long)((valid ? slot : 0) * N_HEAD + h) * Tmax) * D_HEAD;
  const f32x4* kp = reinterpret_cast<const f32x4*>(kc + head_base) + c;
  const f32x4* vp = reinterpret_cast<const f32x4*>(vc + head_base) + c;
  constexpr int RS = ATT_WAVES * 4;           // row stride between a lane-group's consecutive rows
  f32x4 kA[ATT_U], vA[ATT_U], kB[ATT_U], vB[ATT_U];
  // One context split (the 32-row chain): the first tile -- rows 0 .. ATT_STRIDE-1 of the stream -- is requested HERE, before the
  // slot record below has arrived: the record is a dependent memory round trip (~1.5 us: written by the previous step's sampler,
  // it comes from memory in every launch) during which nothing streamed.  Every context of at least a tile holds real rows there.
  // (The record is requested FIRST: loads return in order, so waiting for a record requested behind the tile would wait for the
  // tile.)
#if defined(VX_DEC_ATTN_LATE_TILE)             // A/B builds: the first tile behind the record, as until round 4
  constexpr bool early = false;
#else
  constexpr bool early = FUSE_OUT && !SPLIT;  // the engine only fuses when Tmax >= DEC_ATTN_TILE (weights.hip); a split's rows start at t0: behind the record
#endif
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 meta = *reinterpret_cast<const i32x4*>(slot_meta + 4 * (valid ? slot : 0));
  __builtin_amdgcn_sched_barrier(0);
  if (early) {
#pragma unroll
    for (int u = 0; u < ATT_U; ++u) kA[u] = __builtin_nontemporal_load(kp + (long)(wid * 4 + g + RS * u) * 16);
#pragma unroll
    for (int u = 0; u < ATT_U; ++u) vA[u] = __builtin_nontemporal_load(vp + (long)(wid * 4 + g + RS * u) * 16);
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" : "+v"(meta));     // all four words stay allocated up to here: no register of the record is reused (= waited for) above
  const int b = meta[0];
  const bool live = valid && meta[2] != 0;
  bool live_rr[NR];                                       // fused: the state of BOTH halves, the same in every thread
  if (!FUSE_OUT) {
    if (!live) return;
    live_rr[0] = true;
  } else {
    const int s1 = (int)blockIdx.y + gy;
    live_rr[0] = slot_meta[4 * (int)blockIdx.y + 2] != 0;
    live_rr[NR - 1] = s1 < batch && slot_meta[4 * (s1 < batch ? s1 : 0) + 2] != 0;
    if (!live_rr[0] && !live_rr[NR - 1]) return;          // uniform over the workgroup
  }
  const int ctx = live ? meta[1] : 1;         // cached rows INCLUDING the new token (at ctx-1); a dead half streams nothing
  const int npast = ctx - 1;

  // this block's slice of the past rows; this lane-group's rows are first, first + 16*WAVES/4.. (stride per u)
  const int chunk = ((npast + nsplit - 1) / nsplit + 15) & ~15;
  const int t0 = sp * chunk;
  const int t1 = (t0 + chunk < npast) ? t0 + chunk : npast;
  int base = t0 + wid * 4 + g;

#define ATT_LOAD(KK, VV, BASE)                                             \
  _Pragma("unroll") for (int u = 0; u < ATT_U; ++u) {                       \
    int t = (BASE) + RS * u;                                               \
    t = t < t1 ? t : t1 - 1;                                               \
    KK[u] = __builtin_nontemporal_load(kp + (long)t * 16);                 \
  }                                                                        \
  _Pragma("unroll") for (int u = 0; u < ATT_U; ++u) {                       \
    int t = (BASE) + RS * u;                                               \
    t = t < t1 ? t : t1 - 1;                                               \
    VV[u] = __builtin_nontemporal_load(vp + (long)t * 16);                 \
  }
  // first tile in flight BEFORE the q/k/v reduction below (it does not depend on q).  One split: it was requested above, ahead of
  // the record; a context shorter than a tile (rows past it may never have been written) requests it again, clamped.
  if (early ? t1 < ATT_STRIDE : true) {
    if (base < t1) { ATT_LOAD(kA, vA, base) }
  }

  // q / k_new / v_new of this head: reduce the QKV split-K partials + bias (in_proj, modules/activation.py:144).  SK is a
  // compile-time constant so that all 3 SK + 3 loads are issued together: a runtime loop waits for every slab in turn,
  // SK + 1 dependent memory round trips at the head of the launch with only the first K/V tile in flight.
  f32x4 q4, k4, v4;
  {
    // SK > 10 (skinny_qkv_bal_kernel, the default in_proj of the 32-row chain since round 5): SK / 10 slabs of q, SK % 10 slabs of k and v
    constexpr int SKQ = SK > 10 ? SK / 10 : SK, SKV = SK > 10 ? SK % 10 : SK;
    const int NP = 3 * D_MODEL;
    const float* p = qkv_partial + (long)b * NP + h * D_HEAD + c * 4;
    f32x4 pq[SKQ], pk[SKV], pv[SKV];
#pragma unroll
    for (int ks = 0; ks < SKV; ++ks) {                 // (the product's request order for SK <= 10)
      const float* ps = p + (long)ks * MB * NP;
      pq[ks] = *reinterpret_cast<const f32x4*>(ps);
      pk[ks] = *reinterpret_cast<const f32x4*>(ps + D_MODEL);
      pv[ks] = *reinterpret_cast<const f32x4*>(ps + 2 * D_MODEL);
    }
#pragma unroll
    for (int ks = SKV; ks < SKQ; ++ks) pq[ks] = *reinterpret_cast<const f32x4*>(p + (long)ks * MB * NP);
    const float* bp = qkv_bias + h * D_HEAD + c * 4;
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(bp + D_MODEL);
    const f32x4 b2 = *reinterpret_cast<const f32x4*>(bp + 2 * D_MODEL);
    q4 = pq[0]; k4 = pk[0]; v4 = pv[0];
#pragma unroll
    for (int ks = 1; ks < SKV; ++ks)
#pragma unroll
      for (int e = 0; e < 4; ++e) { q4[e] += pq[ks][e]; k4[e] += pk[ks][e]; v4[e] += pv[ks][e]; }
#pragma unroll
    for (int ks = SKV; ks < SKQ; ++ks)
#pragma unroll
      for (int e = 0; e < 4; ++e) q4[e] += pq[ks][e];
#pragma unroll
    for (int e = 0; e < 4; ++e) { q4[e] = (q4[e] + b0[e]) * 0.125f; k4[e] += b1[e]; v4[e] += b2[e]; }
  }
  if (live && sp == 0 && wid == 0 && g == 0) {        // in-place append: present = (k, v) (modules/activation.py:151-157)
    *reinterpret_cast<f32x4*>(kc + head_base + (long)npast * D_HEAD + c * 4) = k4;
    *reinterpret_cast<f32x4*>(vc + head_base + (long)npast * D_HEAD + c * 4) = v4;
  }

  VX_STAMP(6, 1);
  float m = NEG_BIG, l = 0.f;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
#define ATT_CONSUME(KK, VV, BASE)                                                                  \
  {                                                                                                \
    float sc[ATT_U];                                                                               \
    float m_new = m;                                                                               \
    _Pragma("unroll") for (int u = 0; u < ATT_U; ++u) {                                             \
      float d = q4[0] * KK[u][0] + q4[1] * KK[u][1] + q4[2] * KK[u][2] + q4[3] * KK[u][3];         \
      d = dpp_sum16(d);                                                                            \
      sc[u] = ((BASE) + RS * u < t1) ? d : NEG_BIG;                                                \
      m_new = fmaxf(m_new, sc[u]);                                                                 \
    }                                                                                              \
    const float alpha = expf(m - m_new);                                                           \
    l *= alpha;                                                                                    \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) o[e] *= alpha;                                    \
    _Pragma("unroll") for (int u = 0; u < ATT_U; ++u) {                                             \
      const float p = ((BASE) + RS * u < t1) ? expf(sc[u] - m_new) : 0.f;                          \
      l += p;                                                                                      \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) o[e] += p * VV[u][e];                           \
    }                                                                                              \
    m = m_new;                                                                                     \
  }
  // double-buffered stream: the next tile's 8 loads are in flight while the current tile is reduced
  while (base < t1) {
    int nb = base + ATT_STRIDE;
    if (nb < t1) { ATT_LOAD(kB, vB, nb) }
    ATT_CONSUME(kA, vA, base)
    base = nb;
    if (base >= t1) break;
    nb = base + ATT_STRIDE;
    if (nb < t1) { ATT_LOAD(kA, vA, nb) }
    ATT_CONSUME(kB, vB, base)
    base = nb;
  }
#undef ATT_LOAD
#undef ATT_CONSUME
  VX_STAMP(6, 2);
  // fused out_proj: the first half of this thread's W_o values (n = tid, d < 32 of the head) is requested now -- it does
  // not depend on the attention result and its L2 latency hides under the group / wave combine below; the second half
  // is requested after the combine, when the streaming registers are free
  f32x4 wo[8], wo2[8];
  const f32x4* wh = reinterpret_cast<const f32x4*>(wo_heads) + (long)h * 16 * D_MODEL + threadIdx.x;   // n = tid (< 1024)
  if (FUSE_OUT) {
#pragma unroll
    for (int dg = 0; dg < 8; ++dg) wo[dg] = wh[dg * D_MODEL];
  }
  // the new token itself (always visible: last mask row is all False, models/vallex.py:535-549)
  if (sp == nsplit - 1 && wid == 0 && g == 0) {
    float d = q4[0] * k4[0] + q4[1] * k4[1] + q4[2] * k4[2] + q4[3] * k4[3];
    d = dpp_sum16(d);
    const float m_new = fmaxf(m, d);
    const float alpha = expf(m - m_new), p = expf(d - m_new);
    l = l * alpha + p;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = o[e] * alpha + p * v4[e];
    m = m_new;
  }

  // combine the 4 lane-groups of the wave (xor 16, 32), then the waves through LDS
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64);
    f32x4 o2;
#pragma unroll
    for (int e = 0; e < 4; ++e) o2[e] = __shfl_xor(o[e], off, 64);
    const float mn = fmaxf(m, m2), a1 = expf(m - mn), a2 = expf(m2 - mn);
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = o[e] * a1 + o2[e] * a2;
    m = mn;
  }
  if (g == 0) {
    *reinterpret_cast<f32x4*>(&sh_o[r][wid][c * 4]) = o;
    if (c == 0) { sh_m[r][wid] = m; sh_l[r][wid] = l; }
  }
  lds_barrier();      // LDS traffic only (measured equal to the fenced barrier, profiles/r04_dec_attn_barrier_ab.log)
  if (wid == 0 && g == 0) {
    float mt = NEG_BIG;
#pragma unroll
    for (int w = 0; w < ATT_WAVES; ++w) mt = fmaxf(mt, sh_m[r][w]);
    float lt = 0.f;
    f32x4 ot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < ATT_WAVES; ++w) {
      const float a = expf(sh_m[r][w] - mt);
      lt += sh_l[r][w] * a;
      const f32x4 ow = *reinterpret_cast<const f32x4*>(&sh_o[r][w][c * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) ot[e] += ow[e] * a;
    }
    if constexpr (FUSE_OUT && SPLIT) {
      *reinterpret_cast<f32x4*>(&sh_ot[r][c * 4]) = ot;        // unnormalised: the consumer holds every split's (m, l)
      if (c == 0 && valid) {
        const long pi = ((long)(b * N_HEAD + h) * nsplit + sp);
        store_result(part_ml + pi * 2, mt);
        store_result(part_ml + pi * 2 + 1, lt);
      }
    } else if (nsplit == 1) {
      const float inv = 1.0f / lt;
#pragma unroll
      for (int e = 0; e < 4; ++e) ot[e] *= inv;
      if (FUSE_OUT) *reinterpret_cast<f32x4*>(&sh_ot[r][c * 4]) = ot;
      // column k = h*64 + 4c: kb = h*8 + (c>>1), hi = c&1
      else store_result(xp_out + (((long)(h * 8 + (c >> 1)) * 64) + b + 32 * (c & 1)) * 4, ot);
    } else {
      const long pi = ((long)(b * N_HEAD + h) * nsplit + sp);
      store_result(part_o + pi * D_HEAD + c * 4, ot);
      if (c == 0) { store_result(part_ml + pi * 2, mt); store_result(part_ml + pi * 2 + 1, lt); }
    }
  }
  VX_STAMP(6, 3);
  if (FUSE_OUT) {
#pragma unroll
    for (int dg = 0; dg < 8; ++dg) wo2[dg] = wh[(8 + dg) * D_MODEL];
    lds_barrier();
    float acc[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) acc[rr] = 0.f;
#pragma unroll
    for (int dg = 0; dg < 16; ++dg) {
      const f32x4 w4 = dg < 8 ? wo[dg & 7] : wo2[dg & 7];
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) {
        const f32x4 o4 = *reinterpret_cast<const f32x4*>(&sh_ot[rr][dg * 4]);      // LDS broadcast
        acc[rr] = fmaf(w4[0], o4[0], acc[rr]);
        acc[rr] = fmaf(w4[1], o4[1], acc[rr]);
        acc[rr] = fmaf(w4[2], o4[2], acc[rr]);
        acc[rr] = fmaf(w4[3], o4[3], acc[rr]);
      }
    }
    // pin: LLVM's Sink pass would otherwise move the FMA chains into the conditional stores below and leave the 32 LDS
    // reads (128 VGPRs) live above them
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) asm volatile("" : "+v"(acc[rr]));
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
      if (!live_rr[rr]) continue;
      const int br = slot_meta[4 * ((int)blockIdx.y + rr * gy)];
      store_result(out_heads + ((long)(SPLIT ? h * nsplit + sp : h) * MB + br) * D_MODEL + threadIdx.x, acc[rr]);
    }
  }
  VX_STAMP(6, 4);
}

// ------------------------------------------------------------------------------------------------------------
// Small batches, one launch fewer per layer (round 4): norm1 + the QKV projection + the context-split attention in ONE kernel.
// At <= 4 rows the step is a chain of launch latencies (QKV GEMM 7.5 us + dec_attn 5.6 us per layer at batch 1,
// profiles/r04_gaps_b1a.csv); every split workgroup of dec_attn already rebuilt q from the QKV slabs.  Here workgroup
// (head h, slot y, split z) computes what IT needs of in_proj (modules/activation.py:144) itself, on the VALU, from the raw fp32
// weight rows (a head's q slice is 64 rows x 4 KB, L2-resident after the first workgroup of the head touched it):
//   * q = W_q[h] x + b (64 rows, 8 per wave; a row is 4 float4 loads per lane, k = 4 lane + 256 i, 16 FMAs in a fixed order, one DPP
//     wave sum) -- all 32 loads of a lane are requested before the LayerNorm prologue;
//   * its SHARE of the new token's key and value: rows [z R, (z + 1) R) of W_k[h] and of W_v[h], R = 64 / NSPL -- requested right
//     behind the q rows' FMAs, so their L2 latency runs under the K/V stream, and finished after it: appended to the cache, k_new
//     also into qk_new (with q, for the consumer), v_new as the output of the extra partial;
//   * then the split's share of the cached keys / values exactly as dec_attn_kernel.
// The new token's own partial is (m = q . k_new, l = 1, o = v_new): its m needs all 64 k_new values, which no single workgroup has --
// the consumer (prologue of the out_proj GEMM, sb_combine_row) forms the dot product from qk_new.  So the combine sees NSPL + 1
// partials.  x = norm1(h) comes from the prologue the QKV GEMM of the small-batch chain used to run: LayerNorm of (resid + sum of
// the SKP linear2 slabs + bias) for THIS row (one 256-thread team, the body of dec_reduce_ln_pack), or -- layer 0, SKP = 0 -- the
// packed image the sampler left.  (First version, measured in profiles/r04_sb_qkv_ab.log: two extra workgroups per (head, row) for
// k_new / v_new; the one with q AND k_new to contract set the kernel's duration, 10.8 us at one row.)
// ------------------------------------------------------------------------------------------------------------
template <int SKP, int NSPL>
__global__ __launch_bounds__(ATT_WAVES * 64, 2) void dec_attn_qkv_kernel(
    const float* __restrict__ in_w, const float* __restrict__ in_b, float* __restrict__ kc, float* __restrict__ vc, int Tmax,
    const int* __restrict__ slot_meta, float* __restrict__ part_o, float* __restrict__ part_ml, float* __restrict__ qk_new,
    const float* __restrict__ partial_in, const float* __restrict__ pbias, const float* __restrict__ resid,
    float* __restrict__ h_out, const float* __restrict__ g, const float* __restrict__ bb, const float* __restrict__ xp) {
  constexpr int R = D_HEAD / NSPL;                             // rows of k_new and of v_new this workgroup contracts
  constexpr int E = 2 * R / ATT_WAVES;                         // ... per wave (NSPL 16 / 8 / 4 / 2 -> 1 / 2 / 4 / 8)
  static_assert(R * NSPL == D_HEAD && E * ATT_WAVES == 2 * R && E >= 1, "NSPL must be 2, 4, 8 or 16");
  __shared__ __attribute__((aligned(16))) float xs[D_MODEL];
  __shared__ __attribute__((aligned(16))) float sh_q[D_HEAD];
  __shared__ __attribute__((aligned(16))) float sh_o[ATT_WAVES][64];
  __shared__ float sh_m[ATT_WAVES], sh_l[ATT_WAVES];
  __shared__ float st[2][4];
  VX_STAMP(6, 0);
  // On this chain launch slot == batch row (the engine keeps the launch order of <= 4 rows in batch order), so NOTHING the prologue
  // requests depends on the slot record: the record (context length, active flag), the LayerNorm inputs of the row and the weight
  // rows of the head all go out at once.  (With b = meta[0] the record was one full memory round trip in front of everything:
  // profiles/r04_timeline_b1_before.log, x in LDS 3.4 us and q ready 5.6 us after the start.)
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const int h = blockIdx.x, b = blockIdx.y, z = blockIdx.z;
  const i32x4 meta = *reinterpret_cast<const i32x4*>(slot_meta + 4 * b);
  constexpr int NS1 = NSPL + 1;                               // partials per (row, head): the context splits + the new token
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, gq = lane >> 4, c = lane & 15;
  // Order of the requests = order of need.  A CU's vector-memory path serves its waves' requests in order (~64 B per clock): the 12
  // small loads of the LayerNorm prologue go FIRST, then the 256 KB of q rows (which the prologue's latency then hides), and the first
  // K/V tile only behind the prologue -- it is needed after q.  (With the weight rows in front, the prologue's loads waited ~2 us in
  // the queue behind them.)
  const int tt = threadIdx.x;
  const bool team = tt < 256;                                 // the LayerNorm team (thread tt owns float4 column tt)
  f32x4 p[SKP > 0 ? SKP : 1], gg = {0.f, 0.f, 0.f, 0.f}, be = gg, rr = gg, bi = gg, x0 = gg;
  if (SKP > 0) {
    if (team) {
      const int cc = tt * 4;
#pragma unroll
      for (int ks = 0; ks < SKP; ++ks) p[ks] = *reinterpret_cast<const f32x4*>(partial_in + ((long)ks * MB + b) * D_MODEL + cc);
      rr = *reinterpret_cast<const f32x4*>(resid + (long)b * D_MODEL + cc);
      bi = *reinterpret_cast<const f32x4*>(pbias + cc);
      gg = *reinterpret_cast<const f32x4*>(g + cc);
      be = *reinterpret_cast<const f32x4*>(bb + cc);
    }
  } else if (team) {
    // layer 0: the sampler (or dec_embed_ln_pack) left norm1(h) in the packed image: float4 column c4 -> (c4 >> 1) * 64 + b + 32 (c4 & 1)
    x0 = *(reinterpret_cast<const f32x4*>(xp) + ((long)(tt >> 1) * 64 + b + 32 * (tt & 1)));
  }
  asm volatile("" ::: "memory");                               // compiler: keep the request order
  const f32x4* w4 = reinterpret_cast<const f32x4*>(in_w) + lane;              // float4 column `lane` of row 0
  f32x4 wv[8][4];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) wv[r][i] = w4[(long)(h * D_HEAD + wid * 8 + r) * (D_MODEL / 4) + 64 * i];
  asm volatile("" ::: "memory");
  if (meta[0] != b || meta[2] == 0) return;                   // row finished (or not this chain's slot order): uniform over the workgroup
  const int ctx = meta[1], npast = ctx - 1;
  const long head_base = ((long)(b * N_HEAD + h) * Tmax) * D_HEAD;

  // ---- x = norm1(h) of row b into LDS ----
  if (SKP > 0) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (team) {
      const int cc = tt * 4;
      v = p[0];
#pragma unroll
      for (int ks = 1; ks < SKP; ++ks)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += p[ks][e];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += bi[e];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = rr[e] + v[e];
      if (h_out && h == 0 && z == 0) *reinterpret_cast<f32x4*>(h_out + (long)b * D_MODEL + cc) = v;
    }
    const float s1 = wave_sum64((v[0] + v[1]) + (v[2] + v[3]));
    if (team && (tt & 63) == 0) st[0][tt >> 6] = s1;
    lds_barrier();
    const float mean = ((st[0][0] + st[0][1]) + (st[0][2] + st[0][3])) * (1.0f / D_MODEL);
    float q2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q2 += d * d; }
    q2 = wave_sum64(q2);
    if (team && (tt & 63) == 0) st[1][tt >> 6] = q2;
    lds_barrier();
    const float rstd = 1.0f / sqrtf(((st[1][0] + st[1][1]) + (st[1][2] + st[1][3])) * (1.0f / D_MODEL) + LN_EPS);
    if (team) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * gg[e] + be[e];
      *reinterpret_cast<f32x4*>(&xs[tt * 4]) = o;
    }
  } else if (team) {
    *reinterpret_cast<f32x4*>(&xs[tt * 4]) = x0;
  }

  // this split's slice of the cached rows: the first K/V tile is requested here -- it does not depend on q, and the q rows' FMAs
  // and the barrier below run under its latency
  const f32x4* kp = reinterpret_cast<const f32x4*>(kc + head_base) + c;
  const f32x4* vp = reinterpret_cast<const f32x4*>(vc + head_base) + c;
  const int chunk = ((npast + NSPL - 1) / NSPL + 15) & ~15;
  const int t0 = z * chunk;
  const int t1 = (t0 + chunk < npast) ? t0 + chunk : npast;
  constexpr int RS = ATT_WAVES * 4;
  int base = t0 + wid * 4 + gq;
  f32x4 kA[ATT_U], vA[ATT_U], kB[ATT_U], vB[ATT_U];
#define ATT_LOAD(KK, VV, BASE)                                             \
  _Pragma("unroll") for (int u = 0; u < ATT_U; ++u) {                       \
    int t = (BASE) + RS * u;                                               \
    t = t < t1 ? t : t1 - 1;                                               \
    KK[u] = __builtin_nontemporal_load(kp + (long)t * 16);                 \
  }                                                                        \
  _Pragma("unroll") for (int u = 0; u < ATT_U; ++u) {                       \
    int t = (BASE) + RS * u;                                               \
    t = t < t1 ? t : t1 - 1;                                               \
    VV[u] = __builtin_nontemporal_load(vp + (long)t * 16);                 \
  }
  if (base < t1) { ATT_LOAD(kA, vA, base) }
  lds_barrier();
  VX_STAMP(6, 1);

  // ---- q: the wave's 8 rows ----
  f32x4 xv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const f32x4*>(&xs[(lane + 64 * i) * 4]);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = fmaf(wv[r][i][e], xv[i][e], acc);
    acc = wave_sum64(acc);
    if (lane == 0) sh_q[wid * 8 + r] = (acc + in_b[h * D_HEAD + wid * 8 + r]) * 0.125f;
  }
  // ---- this wave's E rows of the workgroup's k_new / v_new share: requested now, contracted behind the stream ----
  // extra row e = wid E + r of the 2R: e < R is k_new[z R + e], else v_new[z R + e - R]
  f32x4 we[E][4];
  int erow[E];
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int e = wid * E + r;
    erow[r] = (e < R ? D_MODEL + h * D_HEAD + z * R + e : 2 * D_MODEL + h * D_HEAD + z * R + (e - R));
#pragma unroll
    for (int i = 0; i < 4; ++i) we[r][i] = w4[(long)erow[r] * (D_MODEL / 4) + 64 * i];
  }
  lds_barrier();
  const f32x4 q4 = *reinterpret_cast<const f32x4*>(&sh_q[c * 4]);                   // already scaled by 1/8
  VX_STAMP(6, 2);
  if (z == 0 && threadIdx.x < 16) *reinterpret_cast<f32x4*>(qk_new + ((long)(b * N_HEAD + h) * 2) * D_HEAD + c * 4) = q4;

  float m = NEG_BIG, l = 0.f;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
#define ATT_CONSUME(KK, VV, BASE)                                                                  \
  {                                                                                                \
    float sc[ATT_U];                                                                               \
    float m_new = m;                                                                               \
    _Pragma("unroll") for (int u = 0; u < ATT_U; ++u) {                                             \
      float d = q4[0] * KK[u][0] + q4[1] * KK[u][1] + q4[2] * KK[u][2] + q4[3] * KK[u][3];         \
      d = dpp_sum16(d);                                                                            \
      sc[u] = ((BASE) + RS * u < t1) ? d : NEG_BIG;                                                \
      m_new = fmaxf(m_new, sc[u]);                                                                 \
    }                                                                                              \
    const float alpha = expf(m - m_new);                                                           \
    l *= alpha;                                                                                    \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) o[e] *= alpha;                                    \
    _Pragma("unroll") for (int u = 0; u < ATT_U; ++u) {                                             \
      const float p = ((BASE) + RS * u < t1) ? expf(sc[u] - m_new) : 0.f;                          \
      l += p;                                                                                      \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) o[e] += p * VV[u][e];                           \
    }                                                                                              \
    m = m_new;                                                                                     \
  }
  while (base < t1) {
    int nb = base + ATT_STRIDE;
    if (nb < t1) { ATT_LOAD(kB, vB, nb) }
    ATT_CONSUME(kA, vA, base)
    base = nb;
    if (base >= t1) break;
    nb = base + ATT_STRIDE;
    if (nb < t1) { ATT_LOAD(kA, vA, nb) }
    ATT_CONSUME(kB, vB, base)
    base = nb;
  }
#undef ATT_LOAD
#undef ATT_CONSUME
  VX_STAMP(6, 3);
  // ---- the share of k_new / v_new (weights long arrived) ----
  const long pn = (long)(b * N_HEAD + h) * NS1 + NSPL;                                // the new token's partial
#pragma unroll
  for (int r = 0; r < E; ++r) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = fmaf(we[r][i][e], xv[i][e], acc);
    acc = wave_sum64(acc);
    if (lane == 0) {
      const int e = wid * E + r;
      const float val = acc + in_b[erow[r]];
      if (e < R) {                                             // k_new[d], d = z R + e: cache append + for the consumer's q . k_new
        const int d = z * R + e;
        kc[head_base + (long)npast * D_HEAD + d] = val;
        qk_new[((long)(b * N_HEAD + h) * 2 + 1) * D_HEAD + d] = val;
      } else {                                                 // v_new[d]: cache append + the output of the new token's partial (p = 1)
        const int d = z * R + (e - R);
        vc[head_base + (long)npast * D_HEAD + d] = val;
        part_o[pn * D_HEAD + d] = val;
      }
    }
  }
  // combine the 4 lane-groups of the wave, then the waves through LDS (as dec_attn_kernel)
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64);
    f32x4 o2;
#pragma unroll
    for (int e = 0; e < 4; ++e) o2[e] = __shfl_xor(o[e], off, 64);
    const float mn = fmaxf(m, m2), a1 = expf(m - mn), a2 = expf(m2 - mn);
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = o[e] * a1 + o2[e] * a2;
    m = mn;
  }
  if (gq == 0) {
    *reinterpret_cast<f32x4*>(&sh_o[wid][c * 4]) = o;
    if (c == 0) { sh_m[wid] = m; sh_l[wid] = l; }
  }
  __syncthreads();
  if (wid == 0 && gq == 0) {
    float mt = NEG_BIG;
#pragma unroll
    for (int w = 0; w < ATT_WAVES; ++w) mt = fmaxf(mt, sh_m[w]);
    float lt = 0.f;
    f32x4 ot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < ATT_WAVES; ++w) {
      const float a = expf(sh_m[w] - mt);
      lt += sh_l[w] * a;
      const f32x4 ow = *reinterpret_cast<const f32x4*>(&sh_o[w][c * 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) ot[e] += ow[e] * a;
    }
    const long pi = (long)(b * N_HEAD + h) * NS1 + z;
    *reinterpret_cast<f32x4*>(part_o + pi * D_HEAD + c * 4) = ot;
    if (c == 0) { part_ml[pi * 2] = mt; part_ml[pi * 2 + 1] = lt; }
  }
  VX_STAMP(6, 4);
}

// false = configuration not instantiated (skp: slabs of the previous layer's linear2, 0 for layer 0; nsplit 4 / 8 / 16)
bool launch_dec_attn_qkv(const float* in_w, const float* in_b, float* kc, float* vc, int Tmax, const int* slot_meta, float* part_o,
                         float* part_ml, float* qk_new, int nsplit, int batch, const float* partial_in, int skp, const float* pbias,
                         const float* resid, float* h_out, const float* g, const float* b, const float* xp, hipStream_t s) {
  const dim3 grid(N_HEAD, batch, nsplit), block(ATT_WAVES * 64);
#define VX_AQ(SKPV, NSV)                                                                                                          \
  hipLaunchKernelGGL((dec_attn_qkv_kernel<SKPV, NSV>), grid, block, 0, s, in_w, in_b, kc, vc, Tmax, slot_meta, part_o, part_ml, qk_new, \
                     partial_in, pbias, resid, h_out, g, b, xp)
  if (skp == 8) {
    if (nsplit == 16) VX_AQ(8, 16); else if (nsplit == 8) VX_AQ(8, 8); else if (nsplit == 4) VX_AQ(8, 4);
    else return false;                   // (2 splits = 8 extra weight rows per wave in flight: spills)
  } else if (skp == 0) {
    if (nsplit == 16) VX_AQ(0, 16); else if (nsplit == 8) VX_AQ(0, 8); else if (nsplit == 4) VX_AQ(0, 4);
    else return false;
  } else return false;
#undef VX_AQ
  return true;
}

bool launch_dec_attn(const float* qkv_partial, int splitk, const float* qkv_bias, float* kc, float* vc, int Tmax,
                     const int* slot_meta, float* xp_out, float* part_o, float* part_ml, int nsplit, int batch,
                     const float* wo_heads, float* out_heads, hipStream_t s) {
  if (splitk == SK_QKV_BALANCED) {         // 8 slabs of q, 4 of k / v (skinny_qkv_bal_kernel)
    if (wo_heads && nsplit > 1)           // fused out_proj with context splits (5 .. 16 rows): slabs h * nsplit + sp, weighed by dec_reduce_ln_split
      hipLaunchKernelGGL((dec_attn_kernel<true, SK_QKV_BALANCED, true>), dim3(N_HEAD, (batch + 1) / 2, nsplit), dim3(ATT_WAVES * 64 * 2), 0, s, kc, vc,
                         slot_meta, Tmax, batch, nsplit, 0, qkv_partial, qkv_bias, xp_out, part_o, part_ml, wo_heads, out_heads);
    else if (wo_heads && nsplit == 1)
      hipLaunchKernelGGL((dec_attn_kernel<true, SK_QKV_BALANCED>), dim3(N_HEAD, (batch + 1) / 2, 1), dim3(ATT_WAVES * 64 * 2), 0, s, kc, vc, slot_meta,
                         Tmax, batch, 1, 0, qkv_partial, qkv_bias, xp_out, part_o, part_ml, wo_heads, out_heads);
    else
      hipLaunchKernelGGL((dec_attn_kernel<false, SK_QKV_BALANCED>), dim3(N_HEAD, batch, nsplit), dim3(ATT_WAVES * 64), 0, s, kc, vc, slot_meta, Tmax,
                         batch, nsplit, 0, qkv_partial, qkv_bias, xp_out, part_o, part_ml, (const float*)nullptr, (float*)nullptr);
    return true;
  }
  if (splitk != 4) return false;           // the QKV split-K factor is compiled in
  if (wo_heads && nsplit > 1)
    hipLaunchKernelGGL((dec_attn_kernel<true, 4, true>), dim3(N_HEAD, (batch + 1) / 2, nsplit), dim3(ATT_WAVES * 64 * 2), 0, s, kc, vc, slot_meta,
                       Tmax, batch, nsplit, 0, qkv_partial, qkv_bias, xp_out, part_o, part_ml, wo_heads, out_heads);
  else if (wo_heads && nsplit == 1)
    hipLaunchKernelGGL((dec_attn_kernel<true, 4>), dim3(N_HEAD, (batch + 1) / 2, 1), dim3(ATT_WAVES * 64 * 2), 0, s, kc, vc, slot_meta,
                       Tmax, batch, 1, 0, qkv_partial, qkv_bias, xp_out, part_o, part_ml, wo_heads, out_heads);
  else
    hipLaunchKernelGGL((dec_attn_kernel<false, 4>), dim3(N_HEAD, batch, nsplit), dim3(ATT_WAVES * 64), 0, s, kc, vc, slot_meta, Tmax,
                       batch, nsplit, 0, qkv_partial, qkv_bias, xp_out, part_o, part_ml, (const float*)nullptr, (float*)nullptr);
  return true;
}

// W_o [1024][1024] -> head-major image for the fused out_proj: out[((h*16 + d/4)*1024 + n)*4 + d%4] = W_o[n][64h + d]
__global__ __launch_bounds__(256) void pack_wo_heads_kernel(const float* __restrict__ W, float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;             // float4 index: (h*16 + dg)*1024 + n
  if (i >= (long)N_HEAD * 16 * D_MODEL) return;
  const int n = (int)(i % D_MODEL), hd = (int)(i / D_MODEL);       // hd = h*16 + dg -> k = 4 hd
  *reinterpret_cast<f32x4*>(out + i * 4) = *reinterpret_cast<const f32x4*>(W + (long)n * D_MODEL + hd * 4);
}

void launch_pack_wo_heads(const float* W, float* out, hipStream_t s) {
  hipLaunchKernelGGL(pack_wo_heads_kernel, dim3(N_HEAD * 16 * D_MODEL / 256), dim3(256), 0, s, W, out);
}

__global__ __launch_bounds__(64) void dec_attn_combine_kernel(const float* __restrict__ part_o,
                                                              const float* __restrict__ part_ml, int nsplit,
                                                              const int* __restrict__ active,
                                                              float* __restrict__ xp_out) {
  const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  if (!active[b]) return;
  const long pi = (long)(b * N_HEAD + h) * nsplit;
  float mt = NEG_BIG;
  for (int s = 0; s < nsplit; ++s) mt = fmaxf(mt, part_ml[(pi + s) * 2]);
  float lt = 0.f, ot = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float a = expf(part_ml[(pi + s) * 2] - mt);
    lt += part_ml[(pi + s) * 2 + 1] * a;
    ot += part_o[(pi + s) * D_HEAD + d] * a;
  }
  const int k = h * D_HEAD + d;
  xp_out[(((long)(k >> 3) * 64) + b + 32 * ((k >> 2) & 1)) * 4 + (k & 3)] = ot / lt;
}

void launch_dec_attn_combine(const float* part_o, const float* part_ml, int nsplit, const int* active, float* xp_out,
                             int batch, hipStream_t s) {
  hipLaunchKernelGGL(dec_attn_combine_kernel, dim3(N_HEAD, batch), dim3(64), 0, s, part_o, part_ml, nsplit, active,
                     xp_out);
}

// ------------------------------------------------------------------------------------------------------------
// dec_sample: ar_predict_layer logits (split-K partials) -> topk_sampling (models/vallex.py:791-853) -> EOS / cap
// bookkeeping (models/vallex.py:572-598), one block per row, everything stays on the device.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// 64-lane min / max of ints, the same value in every lane: DPP row steps, row_bcast15 / 31 into row 3, lane 63 broadcast
// (see wave_sum64; `old` = the lane's own value, so lanes the row masks leave out keep it)
template <bool MAX>
__device__ __forceinline__ int wave_minmax64i(int v) {
#define VX_MM(CTRL, RM) { const int t = __builtin_amdgcn_update_dpp(v, v, CTRL, RM, 0xF, false); v = MAX ? max(v, t) : min(v, t); }
  VX_MM(0xB1, 0xF) VX_MM(0x4E, 0xF) VX_MM(0x141, 0xF) VX_MM(0x140, 0xF) VX_MM(0x142, 0xA) VX_MM(0x143, 0xC)
#undef VX_MM
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min64i(int v) { return wave_minmax64i<false>(v); }
__device__ __forceinline__ int wave_max64i(int v) { return wave_minmax64i<true>(v); }

// One wave per row.  The 1025 logits live in registers, lane l owning the CONTIGUOUS indices [17l, 17l+17) so the
// inverse-CDF running sum follows index order (sequential inside a lane, Hillis-Steele across lanes: a fixed,
// run-to-run deterministic order).
constexpr int SPL = 17;   // 64 * 17 = 1088 >= 1025

// wave-wide reductions: 4 DPP steps inside the 16-lane rows + two cross-row exchanges; every lane gets the result
__device__ __forceinline__ float dpp_max16(float x) {
  int v = __builtin_bit_cast(int, x);
  x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false)));    // quad_perm [1,0,3,2]
  v = __builtin_bit_cast(int, x);
  x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false)));    // quad_perm [2,3,0,1]
  v = __builtin_bit_cast(int, x);
  x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false)));   // row_half_mirror
  v = __builtin_bit_cast(int, x);
  x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false)));   // row_mirror
  return x;
}
__device__ __forceinline__ float wave_max64f(float x) {
  x = dpp_max16(x);
  int v = __builtin_bit_cast(int, x);
  x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xA, 0xF, false)));     // row_bcast15
  v = __builtin_bit_cast(int, x);
  x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xC, 0xF, false)));     // row_bcast31
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
__device__ __forceinline__ int dpp_sum16i(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true);
  return x;
}
__device__ __forceinline__ int wave_sum64i_fast(int x) {
  x = dpp_sum16i(x);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);     // row_bcast15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);     // row_bcast31 -> rows 2, 3
  return __builtin_amdgcn_readlane(x, 63);
}

// topk_sampling (models/vallex.py:836-853) + the stop rule (:572-598) + -- when a.emb_tab is set -- the start of the NEXT
// decode step for the row: embedding of the committed token at its position and norm1 of layer 0 in the packed-x image
// (what dec_embed_ln_pack does as a separate launch).  One launch + one kernel boundary less per step.
template <int SK>
__global__ __launch_bounds__(64) void dec_sample_kernel(SampleArgs a) {
  __shared__ float lg[64 * SPL];
  const int b = blockIdx.x, lane = threadIdx.x;
  VX_STAMP(7, 0);
  // row state: every scalar the kernel needs, requested up front (independent loads)
  const bool act = a.active[b] != 0;
  if (!act && !a.logits_out) return;
  const int ngen = a.n_gen[b], pos = a.cur_pos[b], ctx = a.ctx_len[b], tlen = a.text_len[b], slot = a.slot_of[b];
  const unsigned long long seed = a.uniforms ? 0ull : a.seed_dev[0];      // independent load, issued with the row state

  {
    // all SPL x SK loads are independent and issued together (SK is a compile-time constant and the tail index is clamped,
    // not branched around: a conditional load per slab makes the compiler wait for each of the 17 groups in turn)
    float pp[SPL][SK];
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
      const int n = lane + 64 * i, nc = n < AR_LOGITS ? n : AR_LOGITS - 1;
#pragma unroll
      for (int ks = 0; ks < SK; ++ks) pp[i][ks] = a.partial[((long)ks * MB + b) * a.npad + nc];
    }
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
      const int n = lane + 64 * i;
      float t = pp[i][0];
      if (SK == 2) t = t + pp[i][1];
      if (SK == 4) t = ((t + pp[i][1]) + pp[i][2]) + pp[i][3];
      if (n >= AR_LOGITS) t = -INFINITY;
      lg[n] = t;
      if (a.logits_out && n < AR_LOGITS) a.logits_out[(long)b * AR_LOGITS + n] = t;
    }
  }
  __syncthreads();
  VX_STAMP(7, 1);
  if (!act || !a.commit) return;

  // the draw and the fixed operands of the fused embedding do not depend on the logits: request them now
  float u;
  if (a.uniforms) u = a.uniforms[(long)ngen * a.uniforms_stride + b];
  else   // counter-based: seed, row and step each pass through their own mixing round (no (seed, row) aliasing)
    u = (float)(splitmix64(splitmix64(splitmix64(seed) + (unsigned long long)b) + (unsigned long long)ngen) >> 40) *
        (1.0f / 16777216.0f);
  f32x4 pe4[4], gg[4], be[4];
  float alpha = 0.f;
  if (a.emb_tab) {
    alpha = a.emb_alpha[0];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = (lane + 64 * i) * 4;
      pe4[i] = *reinterpret_cast<const f32x4*>(a.pe + (long)(pos + 1) * D_MODEL + c);
      gg[i] = *reinterpret_cast<const f32x4*>(a.ln_g + c);
      be[i] = *reinterpret_cast<const f32x4*>(a.ln_b + c);
    }
  }

  float v[SPL];
#pragma unroll
  for (int j = 0; j < SPL; ++j) v[j] = lg[lane * SPL + j];        // stride 17 floats: conflict-free
  if (a.temperature != 1.0f) {                                     // models/vallex.py:845-846
#pragma unroll
    for (int j = 0; j < SPL; ++j) v[j] = v[j] / a.temperature;
  }
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < SPL; ++j) mx = fmaxf(mx, v[j]);
  mx = wave_max64f(mx);

  if (a.top_k > 0 && a.top_k < AR_LOGITS) {                        // :803-809, ties with the k-th value are kept
    // fast path: walk down top_k DISTINCT values (one wave reduction each); if exactly top_k logits are >= the last one
    // there were no ties on the way and it is the k-th largest with multiplicity
    float thr = mx;
    for (int it = 1; it < a.top_k && thr != -INFINITY; ++it) {
      float cur = -INFINITY;
#pragma unroll
      for (int j = 0; j < SPL; ++j) if (v[j] < thr) cur = fmaxf(cur, v[j]);
      thr = wave_max64f(cur);
    }
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < SPL; ++j) cnt += (v[j] >= thr);
    if (wave_sum64i_fast(cnt) != a.top_k) {                        // ties (or fewer than k finite logits): exact walk
      float prev = INFINITY;
      int count = 0;
      thr = mx;
      while (true) {
        float cur = -INFINITY;
#pragma unroll
        for (int j = 0; j < SPL; ++j) if (v[j] < prev) cur = fmaxf(cur, v[j]);
        cur = wave_max64f(cur);
        int c2 = 0;
#pragma unroll
        for (int j = 0; j < SPL; ++j) c2 += (v[j] == cur);
        count += wave_sum64i_fast(c2);
        thr = cur;
        if (count >= a.top_k || cur == -INFINITY) break;
        prev = cur;
      }
    }
#pragma unroll
    for (int j = 0; j < SPL; ++j) if (v[j] < thr) v[j] = -INFINITY;
  }
  // softmax numerators (F.softmax = exp(x - max) / sum; the common 1/sum cancels in the inverse CDF)
  float e[SPL], loc = 0.f;
#pragma unroll
  for (int j = 0; j < SPL; ++j) { e[j] = expf(v[j] - mx); loc += e[j]; }
  float incl = loc;                                                // inclusive scan over lanes
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  const float total = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, incl), 63));
  const float thresh = u * total;
  float c = incl - loc;
  int cand = 0x7fffffff, lastnz = -1;
#pragma unroll
  for (int j = 0; j < SPL; ++j) {
    if (e[j] > 0.f) {
      c += e[j];
      lastnz = lane * SPL + j;
      if (c > thresh && cand == 0x7fffffff) cand = lane * SPL + j;
    }
  }
  cand = wave_min64i(cand);
  const int last_all = wave_max64i(lastnz);
  int tok = cand == 0x7fffffff ? last_all : cand;
  // non-finite logits (an f16x2 operand of the prefill left the fp16 range: the engine re-runs the prefill in fp32 and samples
  // again) leave no candidate at all: keep the index inside the tables, the token of such a pass is never used
  if ((unsigned)tok > (unsigned)EOS_ID) tok = EOS_ID;

  // log-prob of the pick under the filtered distribution (F.log_softmax, models/vallex.py:851-852), accumulated per
  // row for best-of-N beam selection (models/vallex.py:572); the owning lane adds it.
  if (a.sum_logp && tok / SPL == lane) {
    float vt = 0.f;
#pragma unroll
    for (int j = 0; j < SPL; ++j) if (tok - lane * SPL == j) vt = v[j];
    a.sum_logp[b] += (vt - mx) - logf(total);
  }

  // wave-uniform from here: every lane holds the same tok / state
  if (a.force_eos_at >= 0 && ngen >= a.force_eos_at) tok = EOS_ID;
  // stop test: EOS, or (y_len - prompt_len) > 16 * text_len  (models/vallex.py:575-578; y has BOS: 1 + ngen); the arena
  // cap (ngen >= gen_stride) is reported to the host as a truncation (vx_last_truncated)
  const bool stop = tok == EOS_ID || (1 + ngen) > 16 * tlen || ngen >= a.gen_stride;
  if (lane == 0) {
    if (stop) {
      a.active[b] = 0;
      a.slot_meta[4 * slot + 2] = 0;
      if (a.n_active) atomicSub(a.n_active, 1);
    } else {
      a.gen[(long)b * a.gen_stride + ngen] = tok;
      a.n_gen[b] = ngen + 1;
      a.cur_tok[b] = tok;
      a.cur_pos[b] = pos + 1;
      a.ctx_len[b] = ctx + 1;
      a.slot_meta[4 * slot + 1] = ctx + 1;
    }
  }
  VX_STAMP(7, 2);
  if (stop || !a.emb_tab) return;
  // start of the next step for this row: h = emb[tok] + alpha * pe[pos + 1]; xp = pack(LN(h))   (dec_embed_ln_pack)
  f32x4 hv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cc = (lane + 64 * i) * 4;
    hv[i] = *reinterpret_cast<const f32x4*>(a.emb_tab + (long)tok * D_MODEL + cc);
#pragma unroll
    for (int q = 0; q < 4; ++q) hv[i][q] = __fadd_rn(hv[i][q], __fmul_rn(alpha, pe4[i][q]));
    store_result(a.emb_h + (long)b * D_MODEL + cc, hv[i], a.wt != 0);
  }
  ln_pack_row(hv, b, gg, be, a.emb_xp, a.wt != 0);
  VX_STAMP(7, 4);
}

bool launch_dec_sample(const SampleArgs& a, hipStream_t s) {
  if (a.splitk == 4) hipLaunchKernelGGL(dec_sample_kernel<4>, dim3(a.batch), dim3(64), 0, s, a);
  else if (a.splitk == 2) hipLaunchKernelGGL(dec_sample_kernel<2>, dim3(a.batch), dim3(64), 0, s, a);
  else if (a.splitk == 1) hipLaunchKernelGGL(dec_sample_kernel<1>, dim3(a.batch), dim3(64), 0, s, a);
  else return false;                       // split-K factor of the predict layer not compiled in
  return true;
}

// best_of beams (models/vallex.py:525-527 repeats the prompt N times and prefills N times): the prefill ran ONCE, on row 0;
// copy its L cached K/V rows of every (layer, head) to rows 1 .. beams-1.  grid = (head, beam-1, 2 * layers), 256 threads.
__global__ __launch_bounds__(256) void beam_kv_broadcast_kernel(float* __restrict__ kc, float* __restrict__ vc, long cache_layer,
                                                                int Tmax, int L) {
  const int h = blockIdx.x, beam = blockIdx.y + 1, l = blockIdx.z >> 1;
  float* base = ((blockIdx.z & 1) ? vc : kc) + (long)l * cache_layer;
  const f32x4* src = reinterpret_cast<const f32x4*>(base + (long)h * Tmax * D_HEAD);
  f32x4* dst = reinterpret_cast<f32x4*>(base + ((long)beam * N_HEAD + h) * Tmax * D_HEAD);
  for (int i = threadIdx.x; i < L * (D_HEAD / 4); i += 256) dst[i] = src[i];
}

void launch_beam_kv_broadcast(float* kc, float* vc, long cache_layer, int layers, int Tmax, int L, int beams, hipStream_t s) {
  if (beams <= 1 || L <= 0) return;
  hipLaunchKernelGGL(beam_kv_broadcast_kernel, dim3(N_HEAD, beams - 1, 2 * layers), dim3(256), 0, s, kc, vc, cache_layer, Tmax, L);
}

// teacher forcing (tests): commit a caller-chosen token exactly like dec_sample would
__global__ void dec_force_token_kernel(const int* __restrict__ tok, int* cur_tok, int* cur_pos, int* ctx_len,
                                       int* n_gen, int* gen, int gen_stride, const int* active, int batch,
                                       int* slot_meta, const int* __restrict__ slot_of) {
  const int b = threadIdx.x;
  if (b >= batch || !active[b]) return;
  const int ngen = n_gen[b];
  if (ngen >= gen_stride) return;
  gen[(long)b * gen_stride + ngen] = tok[b];
  n_gen[b] = ngen + 1;
  cur_tok[b] = tok[b];
  cur_pos[b] += 1;
  ctx_len[b] += 1;
  slot_meta[4 * slot_of[b] + 1] = ctx_len[b];
}

void launch_dec_force_token(const int* tok, int* cur_tok, int* cur_pos, int* ctx_len, int* n_gen, int* gen,
                            int gen_stride, const int* active, int batch, int* slot_meta, const int* slot_of,
                            hipStream_t s) {
  hipLaunchKernelGGL(dec_force_token_kernel, dim3(1), dim3(64), 0, s, tok, cur_tok, cur_pos, ctx_len, n_gen, gen,
                     gen_stride, active, batch, slot_meta, slot_of);
}

}  // namespace vx
