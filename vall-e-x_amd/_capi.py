"""ctypes binding of include/vallex_hip.h (the drop-in boundary) and include/vallex_hip_dev.h (measurement entries) -- the only
way the Python layer reaches the GPU.

There is deliberately NO CPU fallback: if libvallex_hip.so is missing or no HIP device is present the calls
raise (VallexHipError / OSError); nothing here imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvallex_hip.so")
LIB_OVERRIDDEN = False
if os.environ.get("VX_LIB"):
    # kernel development only: an experiment build of the SAME sources (_build.py --variant=...) for A/B runs.  Honoured only together
    # with VX_DEV=1, announced on stderr, and the library's ABI version is checked before any other symbol is bound (load_library).
    if os.environ.get("VX_DEV") == "1":
        LIB_PATH = os.path.abspath(os.environ["VX_LIB"])
        LIB_OVERRIDDEN = True
    else:
        import warnings
        warnings.warn("VX_LIB is set but VX_DEV=1 is not: ignoring it and loading the in-tree library (VX_LIB is a kernel-development "
                      "hook, not a deployment option)", RuntimeWarning)

VX_OK, VX_EINVAL, VX_EHIP, VX_ESTATE, VX_ENOTFOUND = 0, -1, -2, -3, -4


class VallexHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"vallex_hip error {code}: {msg}")
        self.code = code


class vx_config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("num_layers", C.c_int32), ("max_batch", C.c_int32), ("max_text", C.c_int32),
                ("max_prompt", C.c_int32), ("max_new", C.c_int32), ("use_graph", C.c_int32),
                ("with_vocos", C.c_int32), ("debug_taps", C.c_int32), ("with_encodec", C.c_int32),
                ("cu_mask", C.c_uint32 * 8), ("arith", C.c_int32)]


class vx_batch(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("batch", C.c_int32), ("text_ids", C.POINTER(C.c_int32)), ("text_lang", C.POINTER(C.c_int32)),
                ("text_stride", C.c_int32), ("text_lens", C.POINTER(C.c_int32)),
                ("prompt_codes", C.POINTER(C.c_int32)), ("prompt_stride", C.c_int32),
                ("prompt_lens", C.POINTER(C.c_int32))]


class vx_sampling(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("top_k", C.c_int32), ("temperature", C.c_float), ("uniforms", C.POINTER(C.c_float)),
                ("uniforms_steps", C.c_int32), ("seed", C.c_uint64), ("force_eos_at", C.c_int32),
                ("sync_every", C.c_int32), ("best_of", C.c_int32), ("length_penalty", C.c_float),
                ("return_worst", C.c_int32)]


# every symbol include/vallex_hip.h declares (tests/test_abi.py checks the library exports exactly these)
ABI_VERSION = 5       # VX_ABI_VERSION of include/vallex_hip.h this binding was written against

SYMBOLS = ["vx_abi_version", "vx_create", "vx_destroy", "vx_last_error", "vx_synchronize", "vx_load_tensor", "vx_finalize_weights",
           "vx_infer", "vx_vocos_decode", "vx_encodec_decode", "vx_encodec_encode", "vx_ar_prefill", "vx_ar_logits", "vx_ar_step",
           "vx_nar", "vx_read_tap", "vx_last_stats", "vx_last_truncated", "vx_last_fallbacks", "vx_fallback_state",
           "vx_fallback_reset", "vx_arith_mode"]
# ... and include/vallex_hip_dev.h: measurement / kernel development, never called by the mirrors of the reference API
DEV_SYMBOLS = ["vx_prof_enable", "vx_prof_get", "vx_prof_reset", "vx_bench_kernel", "vx_bench_gemm", "vx_bench_attn",
               "vx_bench_gemm_clock", "vx_bench_gemm_epilogue"]

_lib = None


def load_library() -> C.CDLL:
    """dlopen the in-tree library (built by __graft_entry__.build() / vall-e-x_amd/_build.py).  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # ONE HIP runtime per process.  torch bundles its own libamdhip64 / libhsa-runtime64 (same soname as /opt/rocm's): dlopen'ing
    # this library BEFORE torch binds it to the system runtime and the later `import torch` maps a second copy of the HIP runtime
    # and of ROCr into the process (two runtimes driving one GPU address space).  The API mirrors import torch anyway (the
    # reference's signatures hand tensors in and out), so import it here, first: the library then resolves libamdhip64.so.7 to
    # the copy that is already loaded.  A C client (examples/*.c) has no torch and runs on the system runtime alone.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    if LIB_OVERRIDDEN:
        import sys
        print(f"[vallex_amd] VX_DEV=1 VX_LIB: loading the experiment library {LIB_PATH} instead of the in-tree build", file=sys.stderr)
    lib.vx_abi_version.argtypes = []
    lib.vx_abi_version.restype = C.c_int32
    if lib.vx_abi_version() != ABI_VERSION:
        raise OSError(f"{LIB_PATH} speaks ABI version {lib.vx_abi_version()}, this binding {ABI_VERSION}: rebuild the library")
    P = C.POINTER
    ctx = C.c_void_p
    lib.vx_create.argtypes = [C.c_int, P(vx_config), P(ctx)]
    lib.vx_destroy.argtypes = [ctx]
    lib.vx_destroy.restype = None
    lib.vx_last_error.argtypes = [ctx]
    lib.vx_last_error.restype = C.c_char_p
    lib.vx_synchronize.argtypes = [ctx]
    lib.vx_load_tensor.argtypes = [ctx, C.c_char_p, P(C.c_float), P(C.c_int64), C.c_int32]
    lib.vx_finalize_weights.argtypes = [ctx]
    lib.vx_infer.argtypes = [ctx, P(vx_batch), P(vx_sampling), P(C.c_int64), C.c_int32, P(C.c_int32)]
    lib.vx_vocos_decode.argtypes = [ctx, P(C.c_int64), C.c_int32, P(C.c_int32), C.c_int32, C.c_int32, P(C.c_float),
                                    C.c_int64]
    lib.vx_encodec_decode.argtypes = [ctx, P(C.c_int64), C.c_int32, P(C.c_int32), C.c_int32, P(C.c_float), C.c_int64]
    lib.vx_encodec_encode.argtypes = [ctx, P(C.c_float), C.c_int64, P(C.c_int32), C.c_int32, P(C.c_int64), C.c_int32,
                                      P(C.c_int32)]
    lib.vx_ar_prefill.argtypes = [ctx, P(vx_batch)]
    lib.vx_ar_logits.argtypes = [ctx, P(C.c_float)]
    lib.vx_ar_step.argtypes = [ctx, P(C.c_int32)]
    lib.vx_nar.argtypes = [ctx, P(vx_batch), P(C.c_int32), C.c_int32, P(C.c_int32), P(C.c_int64), C.c_int32]
    lib.vx_read_tap.argtypes = [ctx, C.c_char_p, P(C.c_float), C.c_int64]
    lib.vx_read_tap.restype = C.c_int64
    lib.vx_prof_enable.argtypes = [ctx, C.c_int32]
    lib.vx_prof_get.argtypes = [ctx, C.c_int32, P(C.c_double), P(C.c_int64), P(C.c_double)]
    lib.vx_prof_reset.argtypes = [ctx]
    lib.vx_bench_kernel.argtypes = [ctx, C.c_int32, C.c_int32, C.c_int32, P(C.c_double), P(C.c_double)]
    lib.vx_bench_gemm.argtypes = [ctx, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P(C.c_double), P(C.c_double)]
    lib.vx_bench_attn.argtypes = [ctx, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P(C.c_double), P(C.c_double)]
    lib.vx_bench_gemm_epilogue.argtypes = [ctx, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P(C.c_int64), P(C.c_int64)]
    lib.vx_bench_gemm_clock.argtypes = [ctx, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P(C.c_double), P(C.c_double),
                                        P(C.c_double)]
    lib.vx_last_stats.argtypes = [ctx, P(C.c_int64), P(C.c_int64), P(C.c_double), P(C.c_double)]
    lib.vx_last_truncated.argtypes = [ctx, P(C.c_int32)]
    lib.vx_last_fallbacks.argtypes = [ctx, P(C.c_int32), P(C.c_int32), P(C.c_int64)]
    lib.vx_fallback_state.argtypes = [ctx, P(C.c_int32), P(C.c_int32), P(C.c_int64)]
    lib.vx_fallback_reset.argtypes = [ctx]
    lib.vx_arith_mode.argtypes = [ctx, P(C.c_int32), P(C.c_int32)]
    for name in SYMBOLS + DEV_SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("vx_destroy", "vx_last_error", "vx_read_tap", "vx_abi_version"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def _ptr(a: np.ndarray, ty):
    return a.ctypes.data_as(C.POINTER(ty))


class Batch:
    """Host-side batch descriptor: one row per utterance (= one reference VALLE.inference call)."""

    def __init__(self, texts: Sequence[np.ndarray], text_langs: Sequence[np.ndarray], prompts: Sequence[np.ndarray]):
        n = len(texts)
        assert n == len(text_langs) == len(prompts) and n > 0
        self.n = n
        self.text_lens = np.array([len(t) for t in texts], np.int32)
        self.prompt_lens = np.array([p.shape[0] for p in prompts], np.int32)
        ts, ps = max(1, int(self.text_lens.max())), max(1, int(self.prompt_lens.max()))
        self.text_ids = np.zeros((n, ts), np.int32)
        self.text_lang = np.zeros((n, ts), np.int32)
        self.prompt_codes = np.zeros((n, ps, 8), np.int32)
        for i in range(n):
            self.text_ids[i, : len(texts[i])] = texts[i]
            self.text_lang[i, : len(texts[i])] = text_langs[i]
            if prompts[i].shape[0]:
                self.prompt_codes[i, : prompts[i].shape[0]] = prompts[i]
        self.c = vx_batch(C.sizeof(vx_batch), n, _ptr(self.text_ids, C.c_int32), _ptr(self.text_lang, C.c_int32), ts,
                          _ptr(self.text_lens, C.c_int32), _ptr(self.prompt_codes, C.c_int32), ps,
                          _ptr(self.prompt_lens, C.c_int32))


ARITH = {"default": 0, "f16x2": 1, "bf16x3": 2, "f32": 3}      # vx_config.arith


def cu_partition(n: int, total: int = 256):
    """CU masks of `n` contexts that share one GPU: contiguous, disjoint blocks of total // n CUs each."""
    per = total // max(1, n)
    return [((1 << per) - 1) << (per * i) for i in range(n)] if n > 1 else [0]


class Engine:
    """Owns one vx_ctx.  Not thread-safe: one host thread per context; contexts are independent.  `cu_mask` (int, bit i = CU i;
    0 = all) confines the context's stream to a CU subset, for several contexts that share one GPU (`cu_partition`)."""

    def __init__(self, device_id: int = 0, num_layers: int = 12, max_batch: int = 32, max_text: int = 512,
                 max_prompt: int = 2048, max_new: int = 2048, use_graph: bool = True, with_vocos: bool = True,
                 debug_taps: bool = False, with_encodec: bool = False, cu_mask: int = 0, arith: int = 0):
        """arith: 0 default (f16x2 unless VX_GEMM_* / VX_ATTN_* say otherwise), 1 f16x2, 2 bf16x3, 3 fp32 (ARITH)."""
        self.lib = load_library()
        words = (C.c_uint32 * 8)(*[(int(cu_mask) >> (32 * w)) & 0xFFFFFFFF for w in range(8)])
        self.cfg = vx_config(C.sizeof(vx_config), num_layers, max_batch, max_text, max_prompt, max_new, int(use_graph), int(with_vocos),
                             int(debug_taps), int(with_encodec), words, int(arith))
        self.ctx = C.c_void_p()
        rc = self.lib.vx_create(device_id, C.byref(self.cfg), C.byref(self.ctx))
        if rc != VX_OK:
            msg = self.lib.vx_last_error(None).decode()
            self.ctx = None
            raise VallexHipError(rc, msg)
        self.max_new = max_new
        self.max_batch = max_batch
        self.finalized = False

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.vx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int):
        if rc < 0:
            raise VallexHipError(rc, self.lib.vx_last_error(self.ctx).decode())
        return rc

    def load_tensor(self, name: str, arr: np.ndarray):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        shape = (C.c_int64 * max(1, a.ndim))(*a.shape)
        self._chk(self.lib.vx_load_tensor(self.ctx, name.encode(), _ptr(a, C.c_float), shape, a.ndim))

    def finalize(self):
        self._chk(self.lib.vx_finalize_weights(self.ctx))
        self.finalized = True

    def synchronize(self):
        self._chk(self.lib.vx_synchronize(self.ctx))

    @staticmethod
    def _sampling(n, top_k, temperature, uniforms, seed, force_eos_at, sync_every, best_of=1, length_penalty=1.0,
                  return_worst=False):
        u = None
        s = vx_sampling(C.sizeof(vx_sampling), int(top_k), float(temperature), None, 0, int(seed), -1 if force_eos_at is None else int(force_eos_at),
                        int(sync_every), int(best_of), float(length_penalty), int(bool(return_worst)))
        if best_of > 1:
            n = int(best_of)
        if uniforms is not None:
            u = np.ascontiguousarray(uniforms, np.float32)
            if u.ndim == 1:
                u = u[:, None]
            assert u.shape[1] == n, "uniforms must be [steps][batch]"
            s.uniforms = _ptr(u, C.c_float)
            s.uniforms_steps = u.shape[0]
        return s, u

    def infer(self, batch: Batch, top_k=-100, temperature=1.0, uniforms=None, seed=0, force_eos_at=None,
              sync_every=8, best_of=1, length_penalty=1.0, return_worst=False):
        s, _keep = self._sampling(batch.n, top_k, temperature, uniforms, seed, force_eos_at, sync_every, best_of,
                                  length_penalty, return_worst)
        out = np.zeros((batch.n, self.max_new, 8), np.int64)
        lens = np.zeros(batch.n, np.int32)
        self._chk(self.lib.vx_infer(self.ctx, C.byref(batch.c), C.byref(s), _ptr(out, C.c_int64), self.max_new,
                                    _ptr(lens, C.c_int32)))
        cut = C.c_int32()
        self._chk(self.lib.vx_last_truncated(self.ctx, C.byref(cut)))
        if cut.value:
            import warnings
            warnings.warn(f"{cut.value} row(s) were cut at the engine's max_new = {self.max_new} frames before the reference's stop "
                          "rule (EOS or 16 x text length, models/vallex.py:575-578): create the engine with a larger max_new",
                          RuntimeWarning, stacklevel=2)
        self._warn_fallbacks()
        return [out[i, : lens[i]].copy() for i in range(batch.n)]

    def _warn_fallbacks(self):
        """Warnings when the f16x2 range guard re-ran a phase in fp32 (vx_last_fallbacks): once per engine at the first raise, and
        again EVERY time the context enters sticky mode (vx_fallback_state: the affected phase kind now runs on the ~3x slower
        fp32 kernels directly until a probe pass comes back clean).  Results are the reference's either way."""
        import warnings
        fb = self.last_fallbacks()
        if fb["lifetime"] and not getattr(self, "_fb_warned", False):
            self._fb_warned = True
            warnings.warn(f"activations of this model left the fp16 range of the f16x2 kernels: {fb['prefill']} prefill / {fb['nar']} NAR "
                          "phase(s) of this call were re-run on the exact-fp32 kernels (results are exact; after two consecutive such "
                          "phases the engine goes to fp32 directly).  If this model does it regularly, create the engine with arith='f32'.",
                          RuntimeWarning, stacklevel=3)
        if fb["lifetime"]:
            st = self.fallback_state()
            if st["times_engaged"] > getattr(self, "_sticky_seen", 0):
                self._sticky_seen = st["times_engaged"]
                kinds = [k for k in ("prefill", "nar") if st[k]]
                warnings.warn(f"sticky fp32 fallback ENGAGED for {' + '.join(kinds) or 'a phase kind'}: two consecutive phases left the fp16 "
                              "range, so this kind now runs on the exact-fp32 kernels directly (~3x slower on these phases); every 32nd "
                              "phase is tried on f16x2 again and a clean pass leaves the mode, Engine.fallback_reset() leaves it at once",
                              RuntimeWarning, stacklevel=3)

    def vocos_decode(self, codes: Sequence[np.ndarray], bandwidth_id: int = 2):
        n = len(codes)
        lens = np.array([c.shape[0] for c in codes], np.int32)
        stride = max(1, int(lens.max()))
        buf = np.zeros((n, stride, 8), np.int64)
        for i, c in enumerate(codes):
            buf[i, : c.shape[0]] = c
        # no zero fill and no second copy of the 24.6 MB a 32 x 8 s batch brings back: the library writes samples [0, 320 T_i) of row i
        # and the rows are returned as views of this one fresh array (which they keep alive)
        audio = np.empty((n, stride * 320), np.float32)
        self._chk(self.lib.vx_vocos_decode(self.ctx, _ptr(buf, C.c_int64), stride, _ptr(lens, C.c_int32), n,
                                           int(bandwidth_id), _ptr(audio, C.c_float), stride * 320))
        return [audio[i, : lens[i] * 320] for i in range(n)]

    def encodec_decode(self, codes: Sequence[np.ndarray]):
        n = len(codes)
        lens = np.array([c.shape[0] for c in codes], np.int32)
        stride = max(1, int(lens.max()))
        buf = np.zeros((n, stride, 8), np.int64)
        for i, c in enumerate(codes):
            buf[i, : c.shape[0]] = c
        audio = np.zeros((n, stride * 320), np.float32)
        self._chk(self.lib.vx_encodec_decode(self.ctx, _ptr(buf, C.c_int64), stride, _ptr(lens, C.c_int32), n,
                                             _ptr(audio, C.c_float), stride * 320))
        return [audio[i, : lens[i] * 320].copy() for i in range(n)]

    def encodec_encode(self, wavs: Sequence[np.ndarray]):
        """mono 24 kHz fp32 waveforms (L_i,) -> codes (ceil(L_i / 320), 8) int64 per row (EnCodec encoder + RVQ, 6 kbps)."""
        n = len(wavs)
        lens = np.array([len(w) for w in wavs], np.int32)
        stride = max(1, int(lens.max()))
        buf = np.zeros((n, stride), np.float32)
        for i, w in enumerate(wavs):
            buf[i, : len(w)] = np.asarray(w, np.float32)
        cstride = (stride + 319) // 320
        codes = np.zeros((n, cstride, 8), np.int64)
        out_lens = np.zeros(n, np.int32)
        self._chk(self.lib.vx_encodec_encode(self.ctx, _ptr(buf, C.c_float), stride, _ptr(lens, C.c_int32), n,
                                             _ptr(codes, C.c_int64), cstride, _ptr(out_lens, C.c_int32)))
        return [codes[i, : out_lens[i]].copy() for i in range(n)]

    # ---- step-level (tests) ----
    def ar_prefill(self, batch: Batch):
        self._chk(self.lib.vx_ar_prefill(self.ctx, C.byref(batch.c)))
        self._nb = batch.n

    def ar_logits(self) -> np.ndarray:
        out = np.zeros((self._nb, 1025), np.float32)
        self._chk(self.lib.vx_ar_logits(self.ctx, _ptr(out, C.c_float)))
        return out

    def ar_step(self, tokens):
        t = np.ascontiguousarray(tokens, np.int32)
        self._chk(self.lib.vx_ar_step(self.ctx, _ptr(t, C.c_int32)))

    def nar(self, batch: Batch, codes0: Sequence[np.ndarray]):
        lens = np.array([len(c) for c in codes0], np.int32)
        stride = max(1, int(lens.max()))
        c0 = np.zeros((batch.n, stride), np.int32)
        for i, c in enumerate(codes0):
            c0[i, : len(c)] = c
        out = np.zeros((batch.n, stride, 8), np.int64)
        self._chk(self.lib.vx_nar(self.ctx, C.byref(batch.c), _ptr(c0, C.c_int32), stride, _ptr(lens, C.c_int32),
                                  _ptr(out, C.c_int64), stride))
        self._warn_fallbacks()
        return [out[i, : lens[i]].copy() for i in range(batch.n)]

    def read_tap(self, name: str, n: int) -> np.ndarray:
        out = np.zeros(n, np.float32)
        got = self.lib.vx_read_tap(self.ctx, name.encode(), _ptr(out, C.c_float), n)
        self._chk(int(got))
        return out[: int(got)]

    # ---- measurement ----
    def prof_enable(self, level):
        """0 off; 1 per-launch events on every class (AR step runs eagerly); 2 full-sequence classes only."""
        self._chk(self.lib.vx_prof_enable(self.ctx, int(level)))

    def prof_reset(self):
        self._chk(self.lib.vx_prof_reset(self.ctx))

    def prof_get(self, which: int):
        ms, n, by = C.c_double(), C.c_int64(), C.c_double()
        self._chk(self.lib.vx_prof_get(self.ctx, which, C.byref(ms), C.byref(n), C.byref(by)))
        return ms.value, n.value, by.value

    def bench_kernel(self, which: int, reps: int, gen_offset: int = 0):
        us, by = C.c_double(), C.c_double()
        self._chk(self.lib.vx_bench_kernel(self.ctx, which, reps, gen_offset, C.byref(us), C.byref(by)))
        return us.value, by.value

    def bench_gemm(self, M: int, N: int, K: int, kernel: int, reps: int = 10):
        us, md = C.c_double(), C.c_double()
        self._chk(self.lib.vx_bench_gemm(self.ctx, M, N, K, kernel, reps, C.byref(us), C.byref(md)))
        return us.value, md.value

    def bench_gemm_epilogue(self, M: int, N: int, K: int, mode: int):
        """(differing words, compared words) of the four-wave f16x2 kernel against the eight-wave one under epilogue `mode`
        (0 bias + ReLU + out_planes, 1 bias + residual through resid_rows, 2 bias + residual) -- include/vallex_hip_dev.h"""
        bad, tot = C.c_int64(), C.c_int64()
        self._chk(self.lib.vx_bench_gemm_epilogue(self.ctx, M, N, K, mode, C.byref(bad), C.byref(tot)))
        return bad.value, tot.value

    def bench_gemm_clock(self, M: int, N: int, K: int, kernel: int = 6, reps: int = 8):
        """(avg_us, max |diff| to the fp32 kernel, shader clock in MHz held WHILE the kernel runs) -- include/vallex_hip_dev.h"""
        us, md, mhz = C.c_double(), C.c_double(), C.c_double()
        self._chk(self.lib.vx_bench_gemm_clock(self.ctx, M, N, K, kernel, reps, C.byref(us), C.byref(md), C.byref(mhz)))
        return us.value, md.value, mhz.value

    def bench_attn(self, batch: int, length: int, causal: bool, variant: int, reps: int = 5):
        """variant 0 fp32 kernel / 10 bf16x3 kernel (+1..3: timing probes); returns (avg_us, max |out - fp32 out| or -1)"""
        us, md = C.c_double(), C.c_double()
        self._chk(self.lib.vx_bench_attn(self.ctx, batch, length, int(causal), variant, reps, C.byref(us), C.byref(md)))
        return us.value, md.value

    def last_fallbacks(self):
        """phases of the last call that left the fp16 range of the f16x2 kernels and were re-run in fp32 (+ lifetime count)"""
        p, n, t = C.c_int32(), C.c_int32(), C.c_int64()
        self._chk(self.lib.vx_last_fallbacks(self.ctx, C.byref(p), C.byref(n), C.byref(t)))
        return dict(prefill=p.value, nar=n.value, lifetime=t.value)

    def fallback_state(self):
        """sticky fp32 fallback: is the AR prefill / the NAR phase in sticky mode right now, and how often the context entered it"""
        p, n, t = C.c_int32(), C.c_int32(), C.c_int64()
        self._chk(self.lib.vx_fallback_state(self.ctx, C.byref(p), C.byref(n), C.byref(t)))
        return dict(prefill=bool(p.value), nar=bool(n.value), times_engaged=t.value)

    def fallback_reset(self):
        """leave sticky mode and forget the consecutive-raise counts (e.g. after a batch of known outlier inputs)"""
        self._chk(self.lib.vx_fallback_reset(self.ctx))

    def arith_mode(self):
        """(gemm, attention) arithmetic of the full-sequence path: 'f16x2' | 'bf16x3' | 'f32' each"""
        g, a = C.c_int32(), C.c_int32()
        self._chk(self.lib.vx_arith_mode(self.ctx, C.byref(g), C.byref(a)))
        names = ("f16x2", "bf16x3", "f32")
        return names[g.value], names[a.value]

    def last_stats(self):
        a, f, am, nm = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
        self._chk(self.lib.vx_last_stats(self.ctx, C.byref(a), C.byref(f), C.byref(am), C.byref(nm)))
        return dict(ar_steps=a.value, frames=f.value, ar_ms=am.value, nar_ms=nm.value)
