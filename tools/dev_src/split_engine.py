"""One-off used in round 4 to cut engine.hip (2 262 lines) into engine_ctx.h + engine.hip + weights.hip + vocoders.hip +
bench_harness.hip by line ranges of commit bac7fd3.  Kept for the record only; not part of any build."""
import os
import sys

csrc = sys.argv[1]
src = open(os.path.join(csrc, "engine.hip")).read().split("\n")
L = lambda a, b: "\n".join(src[a - 1:b])   # noqa: E731  inclusive, 1-indexed

hdr = []
hdr.append("""// Internal declarations shared by the engine's translation units (engine.hip: context, AR / NAR drivers and the hot-path ABI;
// weights.hip: ingest of the reference state-dict; vocoders.hip: Vocos head, EnCodec decoder / encoder drivers;
// bench_harness.hip: the measurement entries of include/vallex_hip_dev.h).  Not part of the public C ABI.
#pragma once
""")
hdr.append(L(4, 18))            # includes + using namespace vx
hdr.append("")
hdr.append("namespace vxe {\n")
hdr.append(L(24, 44))           # Tensor, LayerW, ProfClass, SK_*, PRED_NPAD
hdr.append("\n}  // namespace vxe\nusing namespace vxe;\n")
hdr.append(L(48, 143))          # struct vx_ctx
hdr.append("")
hdr.append(L(147, 164))         # HIPCHK, FAIL
hdr.append("\nnamespace vxe {\n")
hdr.append(L(166, 174))         # dev_alloc
hdr.append("""
const float* W(vx_ctx* c, const std::string& name);
""")
hdr.append(L(181, 204))         # ProfScope
hdr.append("""
int upload_meta(vx_ctx* c);
""")
hdr.append(L(215, 226))         # MetaBuilder
hdr.append("""
int tap_store(vx_ctx* c, const std::string& name, const float* src, size_t n);
// C = resid + colscale * act(A W^T + bias) on the fp32 MFMA (cls: profiling class, 2 = transformer projections, 4 = vocoders)
void gemm(vx_ctx* c, const float* A, int lda, const float* Wt, int ldw, const float* bias, const float* resid, int ldr,
          const float* colscale, float* C, int ldc, long M, int N, int K, int act, const int* gather = nullptr, int cls = 4);
bool range_guarded(const vx_ctx* c);
int ensure_f32_buffers(vx_ctx* c);
int take_range_flag(vx_ctx* c, bool* raised);
int check_batch(vx_ctx* c, const vx_batch* b, int max_rows);
SampleArgs make_sample_args(vx_ctx* c, const vx_sampling* s, int commit, float* logits_out);
void ar_step_launches(vx_ctx* c, const SampleArgs* sa);
""")
hdr.append(L(331, 337))         # F32Scope
hdr.append("\n}  // namespace vxe")
open(os.path.join(csrc, "engine_ctx.h"), "w").write("\n".join(hdr) + "\n")

core = []
core.append(L(1, 3))
core.append('#include "engine_ctx.h"\n')
core.append("namespace {\nstd::string g_create_err;\n}  // namespace\n")
core.append("namespace vxe {\n")
core.append(L(176, 179))        # W
core.append("")
core.append(L(206, 213))        # upload_meta
core.append("")
core.append(L(228, 330))        # tap_store .. range_guarded
core.append(L(338, 804))        # ensure_f32_buffers .. nar_generate  (F32Scope 331-337 lives in the header)
core.append("}  // namespace vxe\n")
core.append(L(832, 904))        # ABI banner .. vx_synchronize
core.append(L(1442, 1594))      # vx_ar_prefill .. vx_infer
core.append(L(1882, 1892))      # vx_read_tap
core.append(L(2231, 2262))      # vx_last_*, close extern "C"
open(os.path.join(csrc, "engine_new.hip"), "w").write("\n".join(core) + "\n")

w = []
w.append("""// Weight ingest: the reference state-dict (374 keys for 12 layers, SURVEY.md A.4) arrives tensor by tensor through
// vx_load_tensor; vx_finalize_weights checks presence and shapes (load_state_dict(strict=True), utils/generation.py:79-83),
// allocates the arenas and derives every device image the kernels read: f16x2 / bf16x3 operand planes, packed decode images,
// the positional table (modules/embedding.py:75-91), the AdaLN stage projections (modules/transformer.py:96-100), the Vocos
// head's matrices (Vocos.from_pretrained, utils/generation.py:89) and the EnCodec decoder / encoder images.
#include "engine_ctx.h"

namespace {
""")
w.append(L(805, 816))           # need, pack
w.append("\n}  // namespace\n\nextern \"C\" {\n")
w.append(L(905, 1441))          # vx_load_tensor, vx_finalize_weights
w.append("}  // extern \"C\"")
open(os.path.join(csrc, "weights.hip"), "w").write("\n".join(w) + "\n")

v = []
v.append("""// Vocoder drivers: the Vocos head (vocos.codes_to_features + vocos.decode, utils/generation.py:148-150) and the EnCodec 24 kHz
// SEANet decoder / encoder + RVQ (AudioTokenizer.decode / .encode, data/tokenizer.py:92-96); kernels in vocos.hip, encodec.hip,
// gemm_f32.hip and the skinny MFMA GEMM of decode.hip (LSTM recurrence).
#include "engine_ctx.h"

extern "C" {
""")
v.append(L(1595, 1881))
v.append("}  // extern \"C\"")
open(os.path.join(csrc, "vocoders.hip"), "w").write("\n".join(v) + "\n")

b = []
b.append("""// Measurement and kernel-development entries (include/vallex_hip_dev.h): per-class HIP-event profiling, back-to-back kernel
// replays on the live decode state, stand-alone GEMM / attention micro-benchmarks.  Never on the product path.
#include "engine_ctx.h"

namespace {
""")
b.append(L(818, 829))           # clock probe kernel
b.append("\n}  // namespace\n\nextern \"C\" {\n")
b.append(L(1893, 2230))
b.append("}  // extern \"C\"")
open(os.path.join(csrc, "bench_harness.hip"), "w").write("\n".join(b) + "\n")
