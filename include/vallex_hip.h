/* vallex_hip.h -- C ABI of the MI355X-native VALL-E X inference hot path (libvallex_hip.so).
 *
 * The reference (Plachtaa/VALL-E-X) has no FFI or plugin interface: its drop-in boundary is the Python API
 * (utils/generation.py:50,92,155 and models/vallex.py:458).  The package `vall-e-x_amd/` keeps those Python
 * signatures and binds THIS header through ctypes; each entry point below cites the reference code it replaces
 * (paths relative to the reference repository root).
 *
 * Conventions
 *   - plain pointers and sizes only; host pointers unless stated; the library owns all device memory inside an
 *     opaque context (weights, KV arena, activations) and one HIP stream per context;
 *   - every function returns 0 on success or a negative VX_E* code; vx_last_error() gives the message;
 *   - a context is not thread-safe; different contexts (one per GPU / per host thread) are independent;
 *   - token ids are int32 on the way in (max id 2047) and int64 on the way out, like the reference's LongTensor.
 */
#ifndef VALLEX_HIP_H
#define VALLEX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VX_OK 0
#define VX_EINVAL (-1)   /* bad argument / shape (reference: AssertionError, models/vallex.py:488-493) */
#define VX_EHIP (-2)     /* HIP runtime failure (no GPU, OOM, launch error) */
#define VX_ESTATE (-3)   /* call order violated (e.g. infer before vx_finalize_weights) */
#define VX_ENOTFOUND (-4)/* unknown tensor name (reference: load_state_dict(strict=True) unexpected key) */

typedef struct vx_ctx vx_ctx;

/* ABI guard.  Every descriptor struct starts with `struct_size` = sizeof(that struct) as the CALLER compiled it; the library
 * rejects a mismatch with VX_EINVAL instead of reading past the end of a shorter (older) struct.  vx_abi_version() returns
 * VX_ABI_VERSION of the library that was actually loaded, so a binding can check it before the first call. */
#define VX_ABI_VERSION 5
int32_t vx_abi_version(void);

/* Model/arena geometry.  d_model=1024, 16 heads, FFN 4096, 8 codebooks are fixed by the kernels
 * (macros.py:1-6, utils/generation.py:67-78); the layer count is configurable so tests can run reduced stacks. */
typedef struct vx_config {
  uint32_t struct_size;    /* = sizeof(vx_config) */
  int32_t num_layers;      /* 12 for the shipped checkpoint */
  int32_t max_batch;       /* rows per vx_infer call; AR runs in micro-batches of <= 32 rows */
  int32_t max_text;        /* max text ids per row (prompt text + text), S */
  int32_t max_prompt;      /* max prompt frames per row, Tp */
  int32_t max_new;         /* max generated frames per row (the reference cap is 16*S, models/vallex.py:577) */
  int32_t use_graph;       /* 1: replay the AR step as a hipGraph */
  int32_t with_vocos;      /* 1: allocate the Vocos head */
  int32_t debug_taps;      /* 1: keep per-layer activations for vx_read_tap */
  int32_t with_encodec;    /* 1: allocate the EnCodec SEANet decoder arena (needs the "encodec.*" tensors) */
  uint32_t cu_mask[8];     /* all zero: the context's stream may use every CU.  Otherwise bit i of word i/32 enables CU i
                              (hipExtStreamCreateWithCUMask): contexts that SHARE one GPU get disjoint CU sets, so the
                              latency-bound decode of one batch runs beside the matrix-bound NAR stages of another */
  int32_t arith;           /* arithmetic of the full-sequence projections and attention (AR prefill, NAR stages; the cached decode
                              step is exact fp32 always): 0 = default (f16x2 unless the environment says otherwise:
                              VX_GEMM_X3 / VX_GEMM_F32 / VX_ATTN_X3 / VX_ATTN_F32 = 1), VX_ARITH_F16X2, VX_ARITH_BF16X3,
                              VX_ARITH_F32 (the reference's own arithmetic).  See vx_arith_mode / vx_last_fallbacks. */
} vx_config;
#define VX_ARITH_DEFAULT 0
#define VX_ARITH_F16X2 1   /* operands split into fp16 head + tail (22 significant bits), fp32 accumulate, f16 MFMA */
#define VX_ARITH_BF16X3 2  /* operands split into three bf16 terms (24 bits), fp32 accumulate, bf16 MFMA */
#define VX_ARITH_F32 3     /* fp32 operands, fp32 MFMA */

/* ---- lifetime -------------------------------------------------------------------------------------------
 * replaces: model construction + .to(device) in preload_models(), utils/generation.py:67-89 */
int vx_create(int device_id, const vx_config* cfg, vx_ctx** out);
void vx_destroy(vx_ctx* ctx);
const char* vx_last_error(const vx_ctx* ctx);   /* ctx may be NULL: last error of a failed vx_create */
int vx_synchronize(vx_ctx* ctx);

/* ---- weights --------------------------------------------------------------------------------------------
 * replaces: VALLE.load_state_dict(checkpoint["model"], strict=True), utils/generation.py:79-83, and
 * Vocos.from_pretrained, utils/generation.py:89.  `name` is the reference state-dict key,
 * Vocos tensors are prefixed "vocos." + their key in the vocos package.  fp32, C-contiguous, copied. */
int vx_load_tensor(vx_ctx* ctx, const char* name, const float* data, const int64_t* shape, int32_t ndim);
/* checks that all 374 (for 12 layers) keys arrived, builds packed decode images, the positional table
 * (modules/embedding.py:75-91) and the per-stage AdaLN projections (modules/transformer.py:96-100). */
int vx_finalize_weights(vx_ctx* ctx);

/* ---- batch descriptor -------------------------------------------------------------------------------------
 * One row = one utterance = one reference `VALLE.inference(x, x_lens, y, enroll_x_lens, ...)` call. */
typedef struct vx_batch {
  uint32_t struct_size;         /* = sizeof(vx_batch) */
  int32_t batch;
  const int32_t* text_ids;      /* [batch][text_stride]   x: prompt text ids ++ text ids (utils/generation.py:133) */
  const int32_t* text_lang;     /* [batch][text_stride]   per-token MODEL language id en0/zh1/ja2 (models/vallex.py:439-443,
                                   499-505): first enroll_len entries = prompt_language, rest = text_language;
                                   -1 = add no language embedding to that token (VALLE.continual, models/vallex.py:716-729) */
  int32_t text_stride;
  const int32_t* text_lens;     /* [batch]  x_lens */
  const int32_t* prompt_codes;  /* [batch][prompt_stride][8]   y (audio prompt), values 0..1023 */
  int32_t prompt_stride;
  const int32_t* prompt_lens;   /* [batch]  y.shape[1] */
} vx_batch;

/* topk_sampling arguments (models/vallex.py:836-853) + reproducibility hooks */
typedef struct vx_sampling {
  uint32_t struct_size;         /* = sizeof(vx_sampling) */
  int32_t top_k;                /* <= 0: no filtering (API default -100); 1: greedy */
  float temperature;            /* > 0 */
  const float* uniforms;        /* optional [uniforms_steps][batch] in [0,1): inverse-CDF draws replacing
                                   torch.multinomial (models/vallex.py:850); NULL -> counter-based RNG from `seed` */
  int32_t uniforms_steps;
  uint64_t seed;
  int32_t force_eos_at;         /* >= 0: the (n+1)-th sample is forced to EOS (benchmark stand-in for a trained
                                   model's termination; -1 = off) */
  int32_t sync_every;           /* host polls the device EOS flags every n steps (reference: every step,
                                   models/vallex.py:574-578); <= 0 -> 8 */
  int32_t best_of;              /* <= 1: off.  N > 1 (batch must be 1): N beams of the one utterance sampled independently,
                                   the beam with the best sum(logp)/len^length_penalty goes on to the NAR stages
                                   (models/vallex.py:525-527,572,583-594); uniforms, if given, are [steps][best_of] */
  float length_penalty;         /* models/vallex.py:584 */
  int32_t return_worst;         /* models/vallex.py:590-591 */
} vx_sampling;

/* ---- the hot path ---------------------------------------------------------------------------------------- */
/* replaces: VALLE.inference AR loop + 7 NAR stages, models/vallex.py:458-686.
 * out_codes [batch][out_stride][8] int64 (row b valid for out_lens[b] frames), out_lens [batch]. */
int vx_infer(vx_ctx* ctx, const vx_batch* b, const vx_sampling* s, int64_t* out_codes, int32_t out_stride,
             int32_t* out_lens);

/* replaces: vocos.codes_to_features + vocos.decode(features, bandwidth_id), utils/generation.py:148-150.
 * codes [batch][codes_stride][8] int64, lens [batch] frames; audio [batch][audio_stride] fp32, 320*len samples each. */
int vx_vocos_decode(vx_ctx* ctx, const int64_t* codes, int32_t codes_stride, const int32_t* lens, int32_t batch,
                    int32_t bandwidth_id, float* audio, int64_t audio_stride);

/* replaces: AudioTokenizer.decode(frames) -> codec.decode (EnCodec 24 kHz SEANet decoder: RVQ sum, Conv1d, 2-layer LSTM,
 * 4 x [ELU, ConvTranspose1d, ResnetBlock], ELU, Conv1d), data/tokenizer.py:95-96 -- the legacy vocoder the reference keeps
 * beside Vocos (README.md:29-30).  Tensors are loaded as "encodec." + {quantizer.{q}.embed, decoder.{i}.weight|bias,
 * decoder.{i}.block1|block3|shortcut.weight|bias, decoder.1.lstm.*} with weight-norm already folded.
 * codes [batch][codes_stride][8] int64, lens [batch]; audio [batch][audio_stride] fp32, 320*len samples each. */
int vx_encodec_decode(vx_ctx* ctx, const int64_t* codes, int32_t codes_stride, const int32_t* lens, int32_t batch,
                      float* audio, int64_t audio_stride);

/* replaces: AudioTokenizer.encode(wav) -> codec.encode (EnCodec 24 kHz SEANet encoder + residual vector quantiser at 6 kbps,
 * 8 codebooks), data/tokenizer.py:92-111 -- the prompt-enrolment path (tokenize_audio, utils/prompt_making.py:57-84).
 * Needs the decoder tensors plus "encodec.encoder." + {0,3,6,9,12,15}.{weight,bias}, {1,4,7,10}.{block1,block3,shortcut}.{weight,
 * bias}, 13.lstm.* (weight-norm folded).  wav [batch][wav_stride] fp32 mono 24 kHz, lens [batch] samples;
 * codes [batch][codes_stride][8] int64, out_lens [batch] = ceil(len / 320) frames. */
int vx_encodec_encode(vx_ctx* ctx, const float* wav, int64_t wav_stride, const int32_t* lens, int32_t batch, int64_t* codes,
                      int32_t codes_stride, int32_t* out_lens);

/* ---- step-level entries (kernel-level parity tests; same kernels as vx_infer) ---------------------------- */
/* first ar_decoder.infer call (models/vallex.py:528-562): embeds, runs the prefix-LM prefill, fills the KV arena,
 * leaves the logits of the last row available.  batch <= 32. */
int vx_ar_prefill(vx_ctx* ctx, const vx_batch* b);
/* logits of the newest position, [batch][1025] (ar_predict_layer, models/vallex.py:568) */
int vx_ar_logits(vx_ctx* ctx, float* out);
/* teacher-forced decode step: append tokens[b] (as if sampled) and run one cached step (models/vallex.py:552-562) */
int vx_ar_step(vx_ctx* ctx, const int32_t* tokens);
/* NAR stages only (models/vallex.py:600-686): codes0 [batch][codes0_stride] first-codebook ids, lens [batch] */
int vx_nar(vx_ctx* ctx, const vx_batch* b, const int32_t* codes0, int32_t codes0_stride, const int32_t* lens,
           int64_t* out_codes, int32_t out_stride);
/* copy a named debug buffer (needs cfg.debug_taps) -- or, when no tap has that name, a tensor exactly as vx_load_tensor stored it
 * (read-back check of an upload) -- to the host; returns the number of floats copied or < 0 */
int64_t vx_read_tap(vx_ctx* ctx, const char* name, float* dst, int64_t max_floats);

/* counters of the last vx_infer: AR steps run, generated frames, AR / NAR wall milliseconds (stream-synchronised) */
int vx_last_stats(vx_ctx* ctx, int64_t* ar_steps, int64_t* frames, double* ar_ms, double* nar_ms);

/* number of rows of the last vx_infer whose generation was cut by the ARENA (cfg.max_new frames) before the reference's own stop
 * rule -- EOS, or more than 16 * text_len frames (models/vallex.py:575-578) -- would have ended it.  0 = every row is what the
 * reference would have produced; > 0: create the context with a larger max_new. */
int vx_last_truncated(vx_ctx* ctx, int32_t* rows);

/* The f16x2 kernels need |activation|, |q|/8, |k|, |v| < 2047 (fp16 range at their fixed scale); the reference's fp32 path has
 * no such bound (modules/transformer.py:371-373, modules/activation.py:144-166).  A phase (AR prefill of a micro-batch / the 7
 * NAR stages of a micro-batch) whose operands leave that range is detected on the device and RE-RUN on the exact-fp32 kernels
 * automatically; the call succeeds with the fp32 result.  This reports how many phases of the last vx_infer / vx_ar_prefill /
 * vx_nar took that path, and the count since vx_create (any pointer may be NULL). */
int vx_last_fallbacks(vx_ctx* ctx, int32_t* prefill_phases, int32_t* nar_phases, int64_t* lifetime_phases);

/* Sticky fallback (ABI 5).  After two CONSECUTIVE raises of a phase kind (a clean f16x2 pass of the kind resets the count) the
 * context runs that kind on the exact-fp32 kernels directly instead of paying an f16x2 pass and an fp32 pass per call; every 32nd
 * phase of the kind is tried on f16x2 again and a clean pass leaves sticky mode.  vx_fallback_state reports whether the AR prefill /
 * the NAR stages are in sticky mode right now and how many times the context entered it since vx_create (any pointer may be NULL);
 * vx_fallback_reset leaves it at once (e.g. after a batch of known outlier inputs).  No reference counterpart: the reference is
 * fp32 throughout. */
int vx_fallback_state(vx_ctx* ctx, int32_t* sticky_prefill, int32_t* sticky_nar, int64_t* times_engaged);
int vx_fallback_reset(vx_ctx* ctx);

/* the arithmetic the context actually runs (after vx_finalize_weights): gemm_mode / attn_mode = 0 f16x2, 1 bf16x3, 2 fp32 */
int vx_arith_mode(vx_ctx* ctx, int32_t* gemm_mode, int32_t* attn_mode);

#ifdef __cplusplus
}
#endif
#endif /* VALLEX_HIP_H */
