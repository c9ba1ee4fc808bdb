/* c_gemm.c -- kernel-development aid (NOT an example of the product boundary: it calls the measurement entries of
 * include/vallex_hip_dev.h): time one full-sequence GEMM kernel of libvallex_hip.so on scratch operands from a plain C process, with
 * the shader clock the chip holds while it runs, for each operand pattern VX_BENCH_GEMM_DATA knows (bench_harness.hip).
 *
 *   gcc -std=c99 -O2 -Wall -Wextra -Werror -pedantic -Iinclude tools/c_gemm.c -Lvall-e-x_amd/csrc -lvallex_hip \
 *       -Wl,-rpath,$PWD/vall-e-x_amd/csrc -o /tmp/c_gemm
 *   /tmp/c_gemm [M N K [kernel [reps]]]        kernel: 0 fp32 MFMA (3 / 4 / 5 / 14: register-staged / LDS-DMA 256 x 128 / LDS-DMA
 *                                              128 x 128 / LDS-DMA 256 x 256 forced), 1 bf16x3, 2 bf16x3 + LDS-DMA, 6 f16x2 (product), 8 f16x2 256 x 256
 *
 * The question it answers ("where round 5 starts"): is gemm_f16x2 bound by its schedule or by the board's power
 * limit?  Same launches, same bytes moved, quieter operands (zero fp16 tails / all zero): faster and a higher clock = power. */
#define _POSIX_C_SOURCE 200112L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vallex_hip_dev.h"

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 31616, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 1024;
  const int kernel = argc > 4 ? atoi(argv[4]) : 6, reps = argc > 5 ? atoi(argv[5]) : 20;
  vx_config cfg;
  vx_ctx* ctx = NULL;
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = (uint32_t)sizeof cfg;
  cfg.num_layers = 1; cfg.max_batch = 1; cfg.max_text = 8; cfg.max_prompt = 8; cfg.max_new = 8;
  int rc = vx_create(0, &cfg, &ctx);
  if (rc != VX_OK) { fprintf(stderr, "vx_create -> %d: %s\n", rc, vx_last_error(NULL)); return 10; }
  const char* modes[4] = {"random", "zero_tail", "zero", "const"};
  const double mfma_per_product = (kernel == 0 || (kernel >= 3 && kernel <= 5) || kernel == 14) ? 1.0 : (kernel == 1 || kernel == 2 ? 6.0 : 3.0);   /* 16-bit MFMAs per fp32 block */
  printf("kernel %d  M %d  N %d  K %d  reps %d\n%-10s %10s %12s %12s %10s %12s\n", kernel, M, N, K, reps, "operands", "us", "fp32-eq TF",
         "MFMA PF", "clock MHz", "max|diff|");
  const char* only = getenv("VX_C_GEMM_MODES");        /* e.g. "random": one pattern, one pass (A/B of kernels) */
  for (int pass = 0; pass < (only ? 1 : 2); ++pass)    /* two interleaved passes over the patterns */
    for (int m = 0; m < 4; ++m) {
      if (only && strcmp(only, modes[m]) != 0) continue;
      setenv("VX_BENCH_GEMM_DATA", modes[m], 1);
      double us = 0, diff = 0, mhz = 0;
      rc = vx_bench_gemm_clock(ctx, M, N, K, kernel, reps, &us, &diff, &mhz);
      if (rc != VX_OK) { fprintf(stderr, "vx_bench_gemm_clock -> %d: %s\n", rc, vx_last_error(ctx)); vx_destroy(ctx); return 11; }
      const double tf = 2.0 * M * (double)N * K / (us * 1e-6) * 1e-12;
      printf("%-10s %10.1f %12.1f %12.3f %10.0f %12.3g\n", modes[m], us, tf, (kernel == 0 || (kernel >= 3 && kernel <= 5) || kernel == 14) ? 0.0 : tf * mfma_per_product * 1e-3, mhz, diff);
    }
  vx_destroy(ctx);
  return 0;
}
