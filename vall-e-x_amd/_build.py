"""Build libvallex_hip.so in-tree with hipcc for gfx950 (no torch involved: the product is a plain C-ABI library)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libvallex_hip.so")
SOURCES = ["gemm_f32.hip", "gemm_bf16x3.hip", "gemm_bf16x3_dma.hip", "gemm_bf16x3_ring.hip", "gemm_bf16x3_pipe.hip", "gemm_f16x2_dma.hip", "rows.hip", "attn_full.hip", "attn_full_x3.hip", "decode.hip", "vocos.hip", "encodec.hip", "engine.hip"]
HEADERS = ["vx_common.h", os.path.join("..", "..", "include", "vallex_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([_hipcc()] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
