// Measurement and kernel-development entries (include/vallex_hip_dev.h): per-class HIP-event profiling, back-to-back kernel
// replays on the live decode state, stand-alone GEMM / attention micro-benchmarks.  Never on the product path.
#include "../../include/vallex_hip_dev.h"
#include "engine_ctx.h"

namespace {

// Kernel-development aid (VX_BENCH_CLOCK=1 in vx_bench_gemm): one wave that sits next to the kernel under test for `ref_ticks` of
// the constant 100 MHz counter and reports how many shader-clock ticks (s_memtime) went by -> the clock the chip actually
// holds under that load.  Bounded by the real-time counter, so it always terminates.
__global__ void clock_probe_kernel(unsigned long long* out, unsigned long long ref_ticks) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
  while (__builtin_amdgcn_s_memrealtime() - r0 < ref_ticks) __builtin_amdgcn_s_sleep(16);
  if (threadIdx.x == 0) {
    out[0] = __builtin_readcyclecounter() - c0;
    out[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}


}  // namespace

extern "C" {

int vx_prof_enable(vx_ctx* c, int32_t on) {
  if (!c) return VX_EINVAL;
  c->prof_on = on;
  return VX_OK;
}

int vx_prof_reset(vx_ctx* c) {
  if (!c) return VX_EINVAL;
  for (auto& p : c->prof) { p.used = 0; p.bytes = 0; }
  return VX_OK;
}

int vx_prof_get(vx_ctx* c, int32_t which, double* total_ms, int64_t* launches, double* algo_bytes) {
  if (!c || which < 0 || which > 5) return VX_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  SYNC();
  ProfClass& p = c->prof[which];
  double tot = 0;
  for (size_t i = 0; i + 1 < p.used; i += 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]));
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = (int64_t)(p.used / 2);
  if (algo_bytes) *algo_bytes = p.bytes;
  return VX_OK;
}

// Back-to-back replays of ONE decode kernel on the live state of the last AR run, bracketed by a single HIP event
// pair on the engine stream (GPU-bound: no host gaps inside the interval).  which 0: dec_attn of layer 0 with every
// row's context set to prefill_len + gen_offset; which 1: the five weight-streaming GEMMs of a step's layer 0
// (+ predict layer), reported per launch.
int vx_bench_kernel(vx_ctx* c, int32_t which, int32_t reps, int32_t gen_offset, double* avg_us, double* algo_bytes) {
  if (!c || reps <= 0 || !avg_us || !algo_bytes) return VX_EINVAL;
  if (c->cur_batch <= 0) FAIL(VX_ESTATE, "no AR run to replay");
  HIPCHK(hipSetDevice(c->dev));
  const int nb = c->cur_batch;
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  double bytes = 0;
  int launches = 0;
  if (which == 0) {
    std::vector<int> ctx(nb), one(nb, 1);
    for (int i = 0; i < nb; ++i) {
      ctx[i] = std::min(c->h_L[i] + std::max(gen_offset, 1), c->Tmax - 1);
      bytes += (double)ctx[i] * 2.0 * D_MODEL * 4.0;
    }
    if (c->fuse_out && c->nsplit == 1) bytes += (double)D_MODEL * D_MODEL * 4.0;     // + W_o, streamed once (fused out_proj)
    // the replay's contexts go into the per-slot view dec_attn reads (the row order of the last prefill is kept)
    std::vector<int> meta(4 * nb);
    D2H(meta.data(), c->slot_meta, meta.size() * sizeof(int));
    SYNC();
    for (int y = 0; y < nb; ++y) { meta[4 * y + 1] = ctx[meta[4 * y]]; meta[4 * y + 2] = 1; }
    H2D(c->slot_meta, meta.data(), meta.size() * sizeof(int));
    H2D(c->ctx_len, ctx.data(), nb * sizeof(int));
    H2D(c->active, one.data(), nb * sizeof(int));
    SYNC();
    // rotate over the layers' KV arenas like the real step does: the working set (NL x ~178 MB at batch 32) is far beyond
    // the 256 MiB Infinity Cache, so no launch is served from it
    const size_t cache_layer = (size_t)c->mbr * N_HEAD * c->Tmax * D_HEAD;
    auto attn_l = [&](int r) {
      const int l = r % c->NL;
      const bool fused = c->fuse_out && (c->nsplit == 1 || c->split_fused);
      // slab counts of the product's in_proj (84 = 8 slabs of q, 4 of k / v: skinny_qkv_bal_kernel)
      launch_dec_attn(c->p_qkv, c->qkv_bal ? 84 : SK_QKV, c->ar[l].in_b, c->kc + l * cache_layer, c->vc + l * cache_layer, c->Tmax,
                      c->slot_meta, c->xp_att, c->part_o, c->part_ml, c->nsplit, nb, fused ? c->ar[l].out_wh : nullptr, c->p_oh,
                      c->stream);
    };
    for (int w = 0; w < 3; ++w) attn_l(w);
    HIPCHK(hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps; ++r) attn_l(r);
    HIPCHK(hipEventRecord(e1, c->stream));
    launches = reps;
  } else if (which == 1) {
    const LayerW& L = c->ar[0];
    auto seq = [&]() {
      if (c->qkv_bal) launch_skinny_qkv_balanced(L.in_wp, c->xp, c->p_qkv, c->stream);      // the in_proj the decode step runs
      // (write-through result stores, as in the 32-row step: decode.hip store_result)
      else launch_skinny_gemm(L.in_wp, c->xp, c->p_qkv, 3 * D_MODEL, D_MODEL, SK_QKV, c->stream, true);
      launch_skinny_gemm(L.out_wp, c->xp_att, c->p_o, D_MODEL, D_MODEL, SK_OUT, c->stream, true);
      launch_skinny16_relu_pack(L.l1_wp, c->xp, L.l1_b, c->xp4, D_FF, D_MODEL, c->stream);
      launch_skinny_gemm(L.l2_wp, c->xp4, c->p_o, D_MODEL, D_FF, SK_L2, c->stream, true);
    };
    seq();
    HIPCHK(hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps; ++r) seq();
    HIPCHK(hipEventRecord(e1, c->stream));
    launches = reps * 4;
    bytes = 12.0 * D_MODEL * D_MODEL * 4.0 / 4.0;     // per launch: a layer's 12 d^2 weights over its 4 GEMMs
  } else if (which == 2) {
    // cache-retention probe: the SAME weight-streaming GEMM (layer 0 QKV, 12.6 MB) back to back -- what a launch costs when
    // its weights were read a moment ago (memory-side cache hits) instead of coming cold from HBM (which 1)
    const LayerW& L = c->ar[0];
    auto one = [&]() { launch_skinny_gemm(L.in_wp, c->xp, c->p_qkv, 3 * D_MODEL, D_MODEL, SK_QKV, c->stream); };
    one();
    HIPCHK(hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps; ++r) one();
    HIPCHK(hipEventRecord(e1, c->stream));
    launches = reps;
    bytes = 3.0 * D_MODEL * D_MODEL * 4.0;
#ifdef VX_DEV_PROBES
  } else if (which == 3) {
    // development timeline (tools/step_timeline.py): `reps` graph replays of a ONE-layer decode step (QKV | attention |
    // reduce+LN | linear1 | linear2 | reduce+LN | predict | sampler) on the live state; the kernels stamp the wall clock
    // (decode.hip) and the caller fetches the stamps of the last replay with vx_dev_stamps.
    vx_sampling sp{};
    sp.struct_size = sizeof(vx_sampling); sp.top_k = 10; sp.temperature = 1.0f; sp.seed = 1; sp.force_eos_at = -1; sp.best_of = 1;
    SampleArgs sa = make_sample_args(c, &sp, 1, nullptr);
    const int nl_keep = c->NL;
    c->NL = 1;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    ar_step_launches(c, &sa);
    HIPCHK(hipStreamEndCapture(c->stream, &g));
    c->NL = nl_keep;
    HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    for (int w = 0; w < 3; ++w) HIPCHK(hipGraphLaunch(ge, c->stream));
    SYNC();
    dev_clear_stamps();
    HIPCHK(hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps; ++r) HIPCHK(hipGraphLaunch(ge, c->stream));
    HIPCHK(hipEventRecord(e1, c->stream));
    SYNC();
    (void)hipGraphExecDestroy(ge);
    launches = reps;
    bytes = 0;
#endif
  } else {
    FAIL(VX_EINVAL, "which must be 0, 1 or 2");
  }
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_us = (double)ms * 1e3 / launches;
  *algo_bytes = bytes;
  HIPCHK(hipGetLastError());
  return VX_OK;
}

#ifdef VX_DEV_PROBES
extern "C" int vx_dev_stamps(unsigned long long* out) { dev_read_stamps(out); return VX_OK; }
extern "C" int vx_dev_gemm_stamps(unsigned long long* out) { dev_read_gemm_stamps(out); return VX_OK; }
#endif

// Stand-alone GEMM micro-benchmark on scratch buffers (kernel development aid; never on the product path):
// kernel 0 = gemm_f32 (3 / 4 / 5: its register-staged / LDS-DMA 256 x 128 / LDS-DMA 128 x 128 kernel forced), 1 = gemm_bf16x3,
// 2 = gemm_bf16x3_dma, 6 = gemm_f16x2 (the default of the model path);
// 11-13 / 21-24 = timing probes of the bf16x3 kernels (VX_DEV_PROBES builds only).  Reports the average launch time and the max abs
// difference of the first and last 256 output rows against the fp32-MFMA kernel.
static int bench_gemm_impl(vx_ctx* c, int32_t M, int32_t N, int32_t K, int32_t kernel, int32_t reps, double* avg_us,
                           double* max_abs_diff, double* clock_mhz) {
  if (!c || M <= 0 || N <= 0 || K <= 0 || K % 32 || N % 4 || reps <= 0 || !avg_us || !max_abs_diff) return VX_EINVAL;
#ifndef VX_DEV_PROBES
  if (kernel < 0 || (kernel > 10 && kernel != 14 && kernel != 15))
    FAIL(VX_EINVAL, "kernel must be 0 .. 10, 14 or 15 (probes and the priority variants 12 / 13 need a VX_DEV_PROBES build)");
#endif
  HIPCHK(hipSetDevice(c->dev));
  float *A = nullptr, *Wt = nullptr, *C0 = nullptr, *C1 = nullptr;
  unsigned short *A3 = nullptr, *W3 = nullptr;
  auto cleanup = [&]() { for (void* p : {(void*)A, (void*)Wt, (void*)C0, (void*)C1, (void*)A3, (void*)W3}) if (p) (void)hipFree(p); };
  hipError_t he;
#define TRY(x) if ((he = (x)) != hipSuccess) { cleanup(); c->err = std::string(#x) + ": " + hipGetErrorString(he); return VX_EHIP; }
  TRY(hipMalloc((void**)&A, (size_t)M * K * 4));
  TRY(hipMalloc((void**)&Wt, (size_t)N * K * 4));
  TRY(hipMalloc((void**)&C0, (size_t)M * N * 4));
  TRY(hipMalloc((void**)&C1, (size_t)M * N * 4));
  TRY(hipMalloc((void**)&A3, (size_t)3 * h2_plane(M, K, H2_TILE_A) * 2));
  TRY(hipMalloc((void**)&W3, (size_t)3 * h2_plane(N, K, H2_TILE_W) * 2));
  TRY(hipMemset(A3, 0, (size_t)3 * h2_plane(M, K, H2_TILE_A) * 2));
  TRY(hipMemset(W3, 0, (size_t)3 * h2_plane(N, K, H2_TILE_W) * 2));
  {
    std::vector<float> h((size_t)std::max(M, N) * K);
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    // VX_BENCH_GEMM_DATA (kernel-development aid): what the operands look like to the matrix pipe.  unset / "random": uniform in
    // [-1, 1) (heads and tails both busy);  "zero_tail": the same values rounded to 10 significant bits -> every fp16 tail plane is
    // exactly zero (the tail x head products toggle nothing);  "zero": all operands zero;  "const": every element 0.5.  Same
    // launches, same bytes moved: a kernel that gets faster on quieter operands is bound by power, not by its schedule.
    const char* dm = getenv("VX_BENCH_GEMM_DATA");
    const int data_mode = !dm ? 0 : (!strcmp(dm, "zero_tail") ? 1 : (!strcmp(dm, "zero") ? 2 : (!strcmp(dm, "const") ? 3 : 0)));
    auto fill = [&](size_t n) {
      for (size_t i = 0; i < n; ++i) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        float v = (float)((st >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f;
        if (data_mode == 1) { int ex; const float mant = frexpf(v, &ex); v = ldexpf(rintf(mant * 1024.0f) / 1024.0f, ex); }
        else if (data_mode == 2) v = 0.0f;
        else if (data_mode == 3) v = 0.5f;
        h[i] = v;
      }
    };
    fill((size_t)M * K);
    H2D(A, h.data(), (size_t)M * K * 4);
    fill((size_t)N * K);
    H2D(Wt, h.data(), (size_t)N * K * 4);
  }
  GemmArgs g0{};
  g0.A = A; g0.lda = K; g0.W = Wt; g0.ldw = K; g0.C = C0; g0.ldc = N; g0.M = M; g0.N = N; g0.K = K; g0.act = ACT_NONE;
  launch_gemm_f32(g0, c->stream, 1);                                     // the comparison baseline: the register-staged fp32 kernel
  if ((kernel >= 6 && kernel <= 13) || kernel == 15 || kernel >= 61) {   // fp16 head / tail planes
    launch_split2h(A, K, M, K, nullptr, A3, h2_plane(M, K, H2_TILE_A), H2_TILE_A, nullptr, H2_ACT_SCALE, c->stream);
    launch_split2h(Wt, K, N, K, nullptr, W3, h2_plane(N, K, H2_TILE_W), H2_TILE_W, nullptr, 16384.0f, c->stream);   // |w| < 1
  } else {
    launch_split3(A, K, M, K, nullptr, A3, (long)M * K, c->stream);
    launch_split3(Wt, K, N, K, nullptr, W3, (long)N * K, c->stream);
  }
  GemmX3Args gx{};
  const bool h2 = (kernel >= 6 && kernel <= 13) || kernel == 15 || kernel >= 61;
  gx.A = A3; gx.a_plane = h2 ? h2_plane(M, K, H2_TILE_A) : (long)M * K; gx.W = W3; gx.w_plane = h2 ? h2_plane(N, K, H2_TILE_W) : (long)N * K; gx.C = C1; gx.ldc = N; gx.M = M; gx.N = N; gx.K = K;
  gx.act = ACT_NONE;
  gx.descale = ldexpf(1.0f, -(H2_ACT_SHIFT + 14));
  GemmArgs g1 = g0;
  g1.C = C1;
  // VX_BENCH_GEMM_WALK="<gm>[,c]" (read at EVERY call of the harness, unlike the launchers' VX_GEMM_WALK): tile-order A/B inside one
  // process, interleaved (tools/gemm_walk_sweep.py ab); -1 = the order the kernels had before round 6 (8 row tiles deep, row-fastest)
  if (const char* we = getenv("VX_BENCH_GEMM_WALK")) {
    const int gm = atoi(we);
    const char* comma = strchr(we, ',');
    const int wk = gm == -1 ? 8 : ((gm >= 1 && gm <= 255) ? (gm | ((comma && comma[1] == 'c') ? 256 : 0)) : 0);
    gx.walk = wk;
    g1.walk = wk;
  }
  auto run = [&]() {
    if (kernel == 0) launch_gemm_f32(g1, c->stream);                       // fp32 MFMA, the product's choice of kernel
    else if (kernel >= 3 && kernel <= 5) launch_gemm_f32(g1, c->stream, kernel - 2);   // 3 register-staged / 4 LDS-DMA 256 x 128 / 5 LDS-DMA 128 x 128
    else if (kernel == 14) launch_gemm_f32(g1, c->stream, 4);                          // LDS-DMA 256 x 256
    else if (kernel == 1) launch_gemm_bf16x3(gx, c->stream);
    else if (kernel == 2) launch_gemm_bf16x3_dma(gx, c->stream);
    else if (kernel == 6) launch_gemm_f16x2(gx, c->stream);              // the product's choice of tile
    else if (kernel == 7) launch_gemm_f16x2(gx, c->stream, 128);
    else if (kernel == 8) launch_gemm_f16x2(gx, c->stream, 256);          // 256 x 256 tiles on the 8-wave kernel (64 x 128 per wave)
    else if (kernel == 15) launch_gemm_f16x2(gx, c->stream, 257);         // 256 x 256 tiles on the 4-wave kernel (128 x 128 per wave)
    else if (kernel == 9) launch_gemm_f16x2(gx, c->stream, -128);         // 128 x 128 tiles (the short-row-set kernel) forced
    else if (kernel == 10) launch_gemm_f16x2(gx, c->stream, -129);        // ... with two LDS stages forced (A/B of the four-stage ring)
#ifdef VX_DEV_PROBES
    else if (kernel == 12 || kernel == 13) {                            // 256 x 256 tiles with wave priority: static for waves 4-7 / per k16 step
      GemmX3Args gp = gx;
      gp.dev_variant = kernel - 11;
      launch_gemm_f16x2(gp, c->stream, 256);
    }
    else if (kernel >= 61) launch_gemm_f16x2_probe(gx, kernel - 60, c->stream);         // 61-64: probes of the f16x2 kernel
    else if (kernel >= 21) launch_gemm_bf16x3_dma_probe(gx, kernel - 20, c->stream);   // 21-24: probes of the DMA kernel
    else launch_gemm_bf16x3_probe(gx, kernel - 10, c->stream);      // 11 / 12 / 13: timing probes
#endif
  };
  run();
  // VX_BENCH_CLOCK=1: sample the shader clock on a second stream while the timed launches run (power / clock ceiling check)
  const char* want_clock = getenv("VX_BENCH_CLOCK");
  hipStream_t s2 = nullptr;
  unsigned long long* d_clk = nullptr;
  if (clock_mhz) *clock_mhz = 0.0;
  if (clock_mhz || (want_clock && want_clock[0] == '1')) {
    TRY(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    TRY(hipMalloc((void**)&d_clk, 16));
    SYNC();
  }
  hipEvent_t e0, e1;
  TRY(hipEventCreate(&e0));
  TRY(hipEventCreate(&e1));
  TRY(hipEventRecord(e0, c->stream));
  run();                                                           // the probe starts once the device is busy
  if (s2) hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, s2, d_clk, 100ull * 2000ull);   // 2 ms at 100 MHz
  for (int r = 1; r < reps; ++r) run();
  TRY(hipEventRecord(e1, c->stream));
  TRY(hipEventSynchronize(e1));
  float ms = 0;
  TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_us = (double)ms * 1e3 / reps;
  if (s2) {
    unsigned long long hclk[2] = {0, 0};
    TRY(hipStreamSynchronize(s2));
    D2H(hclk, d_clk, 16); SYNC();
    const double mhz = hclk[1] ? (double)hclk[0] / ((double)hclk[1] / 100.0) : 0.0;
    if (clock_mhz) *clock_mhz = mhz;
    else
      fprintf(stderr, "[vx_bench_gemm] kernel %d M=%d N=%d K=%d: shader clock while running = %.0f MHz (%llu ticks in %.3f ms)\n", kernel,
              M, N, K, mhz, hclk[0], (double)hclk[1] / 1e5);
    (void)hipFree(d_clk);
    (void)hipStreamDestroy(s2);
  }
  // every output row against the register-staged fp32 kernel (round 6: interior tiles, the XCD remap and the tile-walk grouping are
  // compared too, not only the first and the last 256 rows), in chunks through the pinned ring
  double md = 0;
  {
    const int chunk = std::max(1, std::min(M, (int)((16u << 20) / ((size_t)N * 4))));
    std::vector<float> h0((size_t)chunk * N), h1((size_t)chunk * N);
    for (int r0 = 0; r0 < M; r0 += chunk) {
      const size_t n = (size_t)std::min(chunk, M - r0) * N;
      D2H(h0.data(), C0 + (size_t)r0 * N, n * 4);
      D2H(h1.data(), C1 + (size_t)r0 * N, n * 4);
      SYNC();
      for (size_t i = 0; i < n; ++i) {
        const double d = (double)fabsf(h0[i] - h1[i]);
        md = d > md || d != d ? (d != d ? 1e30 : d) : md;        // a NaN on either side is a mismatch, not a silent pass
      }
    }
  }
  *max_abs_diff = md;
#undef TRY
  cleanup();
  HIPCHK(hipGetLastError());
  return VX_OK;
}

int vx_bench_gemm(vx_ctx* c, int32_t M, int32_t N, int32_t K, int32_t kernel, int32_t reps, double* avg_us,
                  double* max_abs_diff) {
  return bench_gemm_impl(c, M, N, K, kernel, reps, avg_us, max_abs_diff, nullptr);
}

int vx_bench_gemm_clock(vx_ctx* c, int32_t M, int32_t N, int32_t K, int32_t kernel, int32_t reps, double* avg_us,
                        double* max_abs_diff, double* clock_mhz) {
  if (!clock_mhz) return VX_EINVAL;
  return bench_gemm_impl(c, M, N, K, kernel, reps, avg_us, max_abs_diff, clock_mhz);
}

// kernel-development aid: time attn_full (variant 0) or one of its probes (1 no staging, 2 no MFMA, 3 no softmax) on
// random q|k|v for `batch` sequences of length `len`, unmasked (NAR) or prefix-LM with prefix = len/3 (causal != 0).
int vx_bench_attn(vx_ctx* c, int32_t batch, int32_t len, int32_t causal, int32_t variant, int32_t reps, double* avg_us,
                  double* max_diff) {
  // variant: 0 fp32 kernel, 1-3 its probes; 10 bf16x3 kernel, 11-13 its probes.  max_diff (optional) = max |out - out of
  // the fp32 kernel| for the product variants (0 / 10), -1 for probes.
  if (!c || batch <= 0 || len <= 0 || reps <= 0 || !avg_us) return VX_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  const long M = (long)batch * len;
  float *qkv = nullptr, *out = nullptr, *ref = nullptr;
  int* meta = nullptr;
  auto cleanup = [&]() { for (void* p : {(void*)qkv, (void*)out, (void*)ref, (void*)meta}) if (p) (void)hipFree(p); };
  hipError_t he;
#define TRY(x) if ((he = (x)) != hipSuccess) { cleanup(); c->err = std::string(#x) + ": " + hipGetErrorString(he); return VX_EHIP; }
  TRY(hipMalloc((void**)&qkv, (size_t)M * 3 * D_MODEL * 4));
  TRY(hipMalloc((void**)&out, (size_t)M * D_MODEL * 4));
  TRY(hipMalloc((void**)&ref, (size_t)M * D_MODEL * 4));
  TRY(hipMalloc((void**)&meta, (size_t)3 * batch * 4));
  {
    std::vector<float> h((size_t)M * 3 * D_MODEL);
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    for (auto& v : h) { st = st * 6364136223846793005ull + 1442695040888963407ull; v = (float)((st >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; }
    // Q columns x4: scores of a few units instead of ~0.3, so the softmax is not nearly uniform
    for (long r = 0; r < M; ++r) for (int k = 0; k < D_MODEL; ++k) h[(size_t)r * 3 * D_MODEL + k] *= 4.0f;
    H2D(qkv, h.data(), h.size() * 4);
    std::vector<int> m(3 * batch);
    for (int i = 0; i < batch; ++i) { m[i] = i * len; m[batch + i] = len; m[2 * batch + i] = len / 3; }
    H2D(meta, m.data(), m.size() * 4);
  }
  const int* pre = causal ? meta + 2 * batch : nullptr;
  auto run = [&]() {
    if (variant == 0) launch_attn_full(qkv, out, meta, meta + batch, pre, batch, len, c->stream);
    else if (variant == 8 || variant == 9)          // 8 / 9: the fp32 kernel with two / three LDS buffers forced (round-6 A/B)
      launch_attn_full(qkv, out, meta, meta + batch, pre, batch, len, c->stream, nullptr, nullptr, variant - 6);
#ifdef VX_DEV_PROBES
    else if (variant < 10) launch_attn_full_probe(qkv, out, meta, meta + batch, pre, batch, len, variant, c->stream);
#else
    else if (variant < 10) return;
#endif
    else if (variant >= 20 && variant <= 23)      // 20: the product's f16x2 kernel; 21-23: its wave-priority variants (tools builds)
      launch_attn_full_h2(qkv, out, meta, meta + batch, pre, batch, len, c->stream, nullptr, 0, nullptr, variant - 20);
    else launch_attn_full_x3(qkv, out, meta, meta + batch, pre, batch, len, variant - 10, c->stream);
  };
  run();
  hipEvent_t e0, e1;
  TRY(hipEventCreate(&e0));
  TRY(hipEventCreate(&e1));
  TRY(hipEventRecord(e0, c->stream));
  for (int r = 0; r < reps; ++r) run();
  TRY(hipEventRecord(e1, c->stream));
  TRY(hipEventSynchronize(e1));
  float ms = 0;
  TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_us = (double)ms * 1e3 / reps;
  if (max_diff) {
    *max_diff = -1.0;
    if (variant == 0 || variant == 8 || variant == 9 || variant == 10 || (variant >= 20 && variant <= 23)) {
      launch_attn_full(qkv, ref, meta, meta + batch, pre, batch, len, c->stream, nullptr, nullptr, 2);      // reference: the two-buffer fp32 kernel
      SYNC();
      std::vector<float> ho((size_t)M * D_MODEL), hr((size_t)M * D_MODEL);
      D2H(ho.data(), out, ho.size() * 4); SYNC();
      D2H(hr.data(), ref, hr.size() * 4); SYNC();
      double md = 0;
      for (size_t i = 0; i < ho.size(); ++i) {
        const double d = std::fabs((double)ho[i] - (double)hr[i]);
        md = (d > md || d != d) ? (d != d ? 1e30 : d) : md;
      }
      *max_diff = md;
    }
  }
#undef TRY
  cleanup();
  HIPCHK(hipGetLastError());
  return VX_OK;
}


// Epilogue cross-check of the two f16x2 GEMM kernels (round 6, ADVICE): the four-wave 128 x 128 kernel (gemm_f16x2_w128_kernel, every
// 256 x 256 tile of the product) against the eight-wave kernel on the SAME operand planes, with the epilogue branches the plain
// micro-benchmark never reaches:  mode 0 = bias + ReLU + out_planes (linear1: the result leaves as the next GEMM's A planes);
// mode 1 = bias + residual read through resid_rows (the trimmed out_proj of the last NAR layer) on whatever ragged M was asked for;
// mode 2 = bias + residual, rows in place (out_proj / linear2).  Both kernels accumulate an output element in the same order, so
// every fp32 result and every plane half-word must be IDENTICAL: returns the number of differing 32-bit words of C (modes 1, 2) or
// 16-bit words of the planes (mode 0), and how many words were compared.
int vx_bench_gemm_epilogue(vx_ctx* c, int32_t M, int32_t N, int32_t K, int32_t mode, int64_t* differing, int64_t* compared) {
  // mode + 10: the eight-wave template on 128 x 128 tiles (the prefill / trimmed-layer instance) against the four-wave kernel
  const bool small_tiles = mode >= 10;
  if (small_tiles) mode -= 10;
  if (!c || M <= 0 || N <= 0 || K < 64 || K % 32 || N % 256 || mode < 0 || mode > 2 || !differing || !compared) return VX_EINVAL;
  HIPCHK(hipSetDevice(c->dev));
  const long a_pl = h2_plane(M, K, H2_TILE_A), w_pl = h2_plane(N, K, H2_TILE_W), o_pl = h2_plane(M, N, H2_TILE_A);
  float *A = nullptr, *Wt = nullptr, *bias = nullptr, *resid = nullptr, *C0 = nullptr, *C1 = nullptr;
  unsigned short *A3 = nullptr, *W3 = nullptr, *P0 = nullptr, *P1 = nullptr;
  int *rows = nullptr, *flag = nullptr;
  const int RR = M + 77;                                    // rows of the residual source (mode 1 reads it through a row map)
  auto cleanup = [&]() {
    for (void* p : {(void*)A, (void*)Wt, (void*)bias, (void*)resid, (void*)C0, (void*)C1, (void*)A3, (void*)W3, (void*)P0, (void*)P1, (void*)rows,
                    (void*)flag})
      if (p) (void)hipFree(p);
  };
  hipError_t he;
#define TRY(x) if ((he = (x)) != hipSuccess) { cleanup(); c->err = std::string(#x) + ": " + hipGetErrorString(he); return VX_EHIP; }
  TRY(hipMalloc((void**)&A, (size_t)M * K * 4));
  TRY(hipMalloc((void**)&Wt, (size_t)N * K * 4));
  TRY(hipMalloc((void**)&bias, (size_t)N * 4));
  TRY(hipMalloc((void**)&resid, (size_t)RR * N * 4));
  TRY(hipMalloc((void**)&C0, (size_t)M * N * 4));
  TRY(hipMalloc((void**)&C1, (size_t)M * N * 4));
  TRY(hipMalloc((void**)&A3, (size_t)2 * a_pl * 2));
  TRY(hipMalloc((void**)&W3, (size_t)2 * w_pl * 2));
  TRY(hipMalloc((void**)&P0, (size_t)2 * o_pl * 2));
  TRY(hipMalloc((void**)&P1, (size_t)2 * o_pl * 2));
  TRY(hipMalloc((void**)&rows, (size_t)M * 4));
  TRY(hipMalloc((void**)&flag, 4));
  TRY(hipMemsetAsync(A3, 0, (size_t)2 * a_pl * 2, c->stream));
  TRY(hipMemsetAsync(W3, 0, (size_t)2 * w_pl * 2, c->stream));
  TRY(hipMemsetAsync(P0, 0, (size_t)2 * o_pl * 2, c->stream));
  TRY(hipMemsetAsync(P1, 0, (size_t)2 * o_pl * 2, c->stream));
  TRY(hipMemsetAsync(C0, 0, (size_t)M * N * 4, c->stream));
  TRY(hipMemsetAsync(C1, 0, (size_t)M * N * 4, c->stream));
  TRY(hipMemsetAsync(flag, 0, 4, c->stream));
  {
    unsigned long long st = 0x2545F4914F6CDD1Dull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; };
    std::vector<float> h((size_t)std::max<long>((long)std::max(M, N) * K, (long)RR * N));
    for (size_t i = 0; i < (size_t)M * K; ++i) h[i] = rnd();
    H2D(A, h.data(), (size_t)M * K * 4);
    for (size_t i = 0; i < (size_t)N * K; ++i) h[i] = rnd();
    H2D(Wt, h.data(), (size_t)N * K * 4);
    for (size_t i = 0; i < (size_t)N; ++i) h[i] = rnd();
    H2D(bias, h.data(), (size_t)N * 4);
    for (size_t i = 0; i < (size_t)RR * N; ++i) h[i] = 3.0f * rnd();
    H2D(resid, h.data(), (size_t)RR * N * 4);
    std::vector<int> rm(M);
    for (int i = 0; i < M; ++i) rm[i] = (int)(((long)i * 7919 + 13) % RR);       // a scattered, non-monotonic row map
    H2D(rows, rm.data(), (size_t)M * 4);
  }
  launch_split2h(A, K, M, K, nullptr, A3, a_pl, H2_TILE_A, nullptr, H2_ACT_SCALE, c->stream);
  launch_split2h(Wt, K, N, K, nullptr, W3, w_pl, H2_TILE_W, nullptr, 16384.0f, c->stream);
  GemmX3Args g{};
  g.A = A3; g.a_plane = a_pl; g.W = W3; g.w_plane = w_pl; g.bias = bias; g.M = M; g.N = N; g.K = K; g.ldc = N;
  g.descale = ldexpf(1.0f, -(H2_ACT_SHIFT + 14)); g.range_flag = flag;
  if (mode == 0) { g.act = ACT_RELU; g.out_plane = o_pl; }
  else { g.act = ACT_NONE; g.resid = resid; g.ldr = N; if (mode == 1) g.resid_rows = rows; }
  GemmX3Args g0 = g, g1 = g;
  if (mode == 0) { g0.out_planes = P0; g1.out_planes = P1; } else { g0.C = C0; g1.C = C1; }
  launch_gemm_f16x2(g0, c->stream, small_tiles ? -128 : 256);        // eight-wave template: 128 x 128 tiles / 256 x 256 tiles (64 x 128 per wave)
  launch_gemm_f16x2(g1, c->stream, 257);        // four waves of 128 x 128
  SYNC();
  TRY(hipGetLastError());
  int64_t bad = 0, total = 0;
  if (mode == 0) {
    // the planes are tile-major with pad rows inside the last row tile: compare the whole allocation (pads were zeroed on both sides)
    const size_t words = (size_t)2 * o_pl, chunk = (size_t)8 << 20;
    std::vector<unsigned short> h0(chunk), h1(chunk);
    for (size_t o = 0; o < words; o += chunk) {
      const size_t n = std::min(chunk, words - o);
      D2H(h0.data(), P0 + o, n * 2);
      D2H(h1.data(), P1 + o, n * 2);
      SYNC();
      for (size_t i = 0; i < n; ++i) bad += h0[i] != h1[i];
      total += (int64_t)n;
    }
  } else {
    const size_t words = (size_t)M * N, chunk = (size_t)4 << 20;
    std::vector<unsigned> h0(chunk), h1(chunk);
    for (size_t o = 0; o < words; o += chunk) {
      const size_t n = std::min(chunk, words - o);
      D2H(h0.data(), C0 + o, n * 4);
      D2H(h1.data(), C1 + o, n * 4);
      SYNC();
      for (size_t i = 0; i < n; ++i) bad += h0[i] != h1[i];
      total += (int64_t)n;
    }
  }
#undef TRY
  cleanup();
  *differing = bad;
  *compared = total;
  return VX_OK;
}

}  // extern "C"
