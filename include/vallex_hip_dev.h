/* vallex_hip_dev.h -- measurement and kernel-development entries of libvallex_hip.so.
 *
 * NOT part of the drop-in boundary (include/vallex_hip.h): nothing here has a counterpart in the reference and the Python
 * mirrors of utils/generation.py / models/vallex.py never call these.  bench.py (roofline leg), tools/ and a few GPU tests do.
 * Kept in the same library so that the kernels that are timed are the kernels that ship.
 */
#ifndef VALLEX_HIP_DEV_H
#define VALLEX_HIP_DEV_H

#include "vallex_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-class event timing --------------------------------------------------------------------------------
 * HIP-event timing of kernel classes on the context's own stream (bench.py roofline leg).
 * which: 0 = dec_attn (KV streaming), 1 = skinny GEMMs, 2 = transformer projections (full-sequence GEMMs), 3 = full-seq
 * attention, 4 = the fp32 GEMMs of the Vocos / EnCodec heads, 5 = the LSTM recurrences of the EnCodec decoder (one event pair
 * per layer around its T dependent steps; `algo_bytes` of this class counts the steps).
 * vx_prof_enable(1) makes the AR step run un-graphed with an event pair around each launch of every class. */
int vx_prof_enable(vx_ctx* ctx, int32_t on);
int vx_prof_get(vx_ctx* ctx, int32_t which, double* total_ms, int64_t* launches, double* algo_bytes);
int vx_prof_reset(vx_ctx* ctx);
/* GPU-bound micro-replay of one decode kernel on the state the last AR run left behind: `reps` back-to-back launches
 * between ONE event pair (eager per-launch events pick up host launch gaps; events recorded inside a hipGraph cannot be
 * timed on ROCm 7.2).  which 0: dec_attn with every row at context prefill_len + gen_offset; which 1: the four
 * weight-streaming GEMMs of a layer.  avg_us = per launch; algo_bytes = algorithmic bytes per launch. */
int vx_bench_kernel(vx_ctx* ctx, int32_t which, int32_t reps, int32_t gen_offset, double* avg_us, double* algo_bytes);
/* kernel-development aid: time one full-sequence GEMM kernel (0 fp32 MFMA, 1 bf16x3) on scratch data and
 * report its max abs difference to the fp32-MFMA kernel.  Not used by the product path. */
int vx_bench_gemm(vx_ctx* ctx, int32_t M, int32_t N, int32_t K, int32_t kernel, int32_t reps, double* avg_us,
                  double* max_abs_diff);
int vx_bench_attn(vx_ctx* ctx, int32_t batch, int32_t len, int32_t causal, int32_t variant, int32_t reps, double* avg_us,
                  double* max_diff);
/* vx_bench_gemm + the shader clock the chip HOLDS while that kernel runs: a one-wave side kernel on a second stream counts
 * shader-clock ticks (s_memtime) over 2 ms of the constant 100 MHz counter (s_memrealtime) while the timed launches execute.
 * The MFMA peaks of the guide assume the 2.4 GHz boost clock; under the f16 matrix kernels this board holds less (power), and
 * bench.py reports the value measured in ITS run next to every MFMA-bound roofline fraction.  clock_mhz = 0 if the probe saw
 * no overlap (kernel too short). */
int vx_bench_gemm_clock(vx_ctx* ctx, int32_t M, int32_t N, int32_t K, int32_t kernel, int32_t reps, double* avg_us,
                        double* max_abs_diff, double* clock_mhz);

/* Epilogue cross-check of the two f16x2 GEMM kernels on the same operand planes (four waves of 128 x 128 against eight waves of
 * 64 x 128; bit-identical by construction): mode 0 = bias + ReLU + out_planes, 1 = bias + residual through resid_rows (ragged M),
 * 2 = bias + residual in place; mode + 10: the eight-wave template on 128 x 128 tiles instead of 256 x 256.  differing / compared =
 * 32-bit words of C (16-bit words of the planes in mode 0).  N % 256 == 0. */
int vx_bench_gemm_epilogue(vx_ctx* ctx, int32_t M, int32_t N, int32_t K, int32_t mode, int64_t* differing, int64_t* compared);

#ifdef __cplusplus
}
#endif
#endif /* VALLEX_HIP_DEV_H */
