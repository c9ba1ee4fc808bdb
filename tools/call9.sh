#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out
timeout 200 python tools/gemm_bench.py > gpurun_out/c9_gemm_bench.log 2>&1; echo "gemm_bench rc=$?"
python - <<'PY'
import re
for ln in open("gpurun_out/c9_gemm_bench.log"):
    cols = ln.strip().split("  |  ")
    if len(cols) < 3: print(ln.strip()); continue
    keep = [c for c in cols[1:] if any(k in c for k in ("h2-256x256:", "b2b", "spread", "noDMA"))]
    print(cols[0], " | ".join(re.sub(r" TF diff.*", "", re.sub(r"\s+", " ", c)) for c in keep))
PY
