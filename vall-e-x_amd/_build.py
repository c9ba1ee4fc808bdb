"""Build libvallex_hip.so in-tree with hipcc for gfx950 (no torch involved: the product is a plain C-ABI library)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libvallex_hip.so")
PREFLIGHT = os.path.join(CSRC, "vx_preflight.bin")      # library-free box check (csrc/preflight.hip, vall-e-x_amd/_preflight.py)
SOURCES = ["gemm_f32.hip", "gemm_f16x2.hip", "gemm_bf16x3.hip", "gemm_bf16x3_dma.hip", "rows.hip", "attn_full.hip", "attn_full_x3.hip", "attn_full_h2.hip", "decode.hip", "vocos.hip", "encodec.hip", "engine.hip", "weights.hip", "vocoders.hip", "bench_harness.hip"]
HEADERS = ["vx_common.h"]                    # every translation unit
# the engine's translation units (host code: context, drivers, C ABI) also see the internal context header and the public ABI
ENGINE_TUS = ("engine.hip", "weights.hip", "vocoders.hip", "bench_harness.hip")
ENGINE_HEADERS = ["engine_ctx.h", os.path.join("..", "..", "include", "vallex_hip.h"),
                  os.path.join("..", "..", "include", "vallex_hip_dev.h")]
# kernarg preload: the first kernel arguments arrive in SGPRs with the wave instead of through an s_load round trip at the
# head of every launch (the compiler keeps a compatibility prologue for firmware without the feature)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result",
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def _digest(paths, extra=()) -> str:
    import hashlib
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    for e in extra:
        h.update(str(e).encode())
    return h.hexdigest()


def _stale(target: str, deps, extra=()) -> bool:
    """Content-keyed, not mtime-keyed: an object is reused only if (a) the digest of its source, the headers and the flags
    equals the one recorded when it was compiled and (b) the digest of the object ITSELF equals the one recorded right after
    the compiler wrote it (`<target>.sha256` holds both) -- a stale or foreign .o/.so that travelled with a snapshot is rebuilt
    even if somebody copied a matching source stamp next to it.  VX_FORCE_BUILD=1 rebuilds everything."""
    stamp = target + ".sha256"
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    rec = open(stamp).read().split()
    return len(rec) != 2 or rec[0] != _digest(deps, extra) or rec[1] != _digest([target])


def _stamp(target: str, deps, extra=()):
    with open(target + ".sha256", "w") as f:
        f.write(_digest(deps, extra) + " " + _digest([target]))


# per-source extra flags (device code generation choices that were measured per kernel)
EXTRA_FLAGS = {}


def build_library(force: bool = False, verbose: bool = False, dev: bool = False, variant: str = "") -> str:
    """dev=True: a SEPARATE library with the kernels' timing probes compiled in (-DVX_DEV_PROBES), for tools/gemm_bench.py and
    tools/attn_bench.py only (tools/dev/libvallex_hip.so); the product library never contains them."""
    force = force or os.environ.get("VX_FORCE_BUILD", "") == "1"
    hdrs_common = [os.path.join(CSRC, h) for h in HEADERS]
    hdrs_engine = hdrs_common + [os.path.join(CSRC, h) for h in ENGINE_HEADERS]
    objs, jobs = [], []
    out_dir = os.path.join(os.path.dirname(HERE), "tools", "dev") if dev else CSRC
    extra = dict(EXTRA_FLAGS)
    if variant:      # experiment builds: tools/devx_<variant>/libvallex_hip.so, `variant` = "name:file1.hip,file2.hip:flag flag ..."
        name, files, fl = variant.split(":", 2)
        out_dir = os.path.join(os.path.dirname(HERE), "tools", "devx_" + name)
        for f in files.split(","):
            extra[f] = extra.get(f, []) + fl.split()
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libvallex_hip.so")
    flags = FLAGS + (["-DVX_DEV_PROBES"] if dev else [])
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    if dev:      # the research template of the f16x2 GEMM with its timing probes: a tools-only translation unit
        srcs.append(os.path.join(os.path.dirname(HERE), "tools", "dev_src", "gemm_f16x2_probes.hip"))
    for src in srcs:
        s = os.path.basename(src)
        obj = os.path.join(out_dir, s.replace(".hip", ".o"))
        objs.append(obj)
        fl = flags + extra.get(s, [])
        hdrs = hdrs_engine if s in ENGINE_TUS else hdrs_common
        if force or _stale(obj, [src] + hdrs, fl):
            jobs.append(([_hipcc()] + fl + ["-c", src, "-o", obj], obj, [src] + hdrs, fl))

    def run(job):
        cmd, target, deps = job[:3]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        _stamp(target, deps, (job[3] if len(job) > 3 else flags) if target.endswith(".o") else ())

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    relinked = False
    if force or jobs or _stale(lib, objs):
        run(([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, lib, objs))
        relinked = True
    print(f"[build] {len(jobs)} of {len(srcs)} sources compiled for gfx950, library {'linked' if relinked else 'up to date (source AND object digests match)'}",
          flush=True)
    return lib


def build_preflight(force: bool = False) -> str:
    """The stand-alone pre-flight executable (no libvallex, no torch): one translation unit, linked against the system HIP runtime."""
    force = force or os.environ.get("VX_FORCE_BUILD", "") == "1"
    src = os.path.join(CSRC, "preflight.hip")
    fl = ["--offload-arch=gfx950", "-O2", "-std=c++17"]
    if force or _stale(PREFLIGHT, [src], fl):
        cmd = [_hipcc()] + fl + [src, "-o", PREFLIGHT]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        _stamp(PREFLIGHT, [src], fl)
        print("[build] vx_preflight.bin compiled for gfx950", flush=True)
    return PREFLIGHT


if __name__ == "__main__":
    var = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--variant=")]
    print(build_library(force="--force" in sys.argv, verbose=True, dev="--dev" in sys.argv, variant=var[0] if var else ""))
    if "--dev" not in sys.argv and not var:
        print(build_preflight(force="--force" in sys.argv))
