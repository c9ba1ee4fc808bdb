#!/bin/bash
# One gpurun call that re-establishes the baseline of a new round and times the prepared kernel probes (about 4 GPU-minutes):
#   /usr/local/graft/bin/gpurun --timeout 420 -- 'bash tools/first_call.sh'
# Writes gpurun_out/fc_*.{log,json}.  Needs the tools-only library for the probes: run `python vall-e-x_amd/_build.py --dev`
# in the build container first (tools/dev/ travels with the snapshot).
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
timeout 200 python -m pytest tests -m gpu -x -q > gpurun_out/fc_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 gpurun_out/fc_gpu_tests.log
timeout 120 python bench.py > gpurun_out/fc_bench.json 2> gpurun_out/fc_bench.err; echo "bench rc=$?"; head -c 400 gpurun_out/fc_bench.json; echo
if [ -f tools/dev/libvallex_hip.so ]; then
  # columns h2-256x256 (product tile) vs h2-256x256-dma-early / -dma-spread (probes 14 / 15, DESIGN.md section 6 "Next levers" (4))
  timeout 90 python tools/gemm_bench.py > gpurun_out/fc_gemm_bench.log 2>&1; echo "gemm_bench rc=$?"; cat gpurun_out/fc_gemm_bench.log
fi
