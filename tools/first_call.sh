#!/bin/bash
# One gpurun call that re-establishes the baseline of a round and times the prepared kernel probes:
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/first_call.sh'
# Writes gpurun_out/fc_*.{log,json}.  Needs the tools-only library for the probes: run `python vall-e-x_amd/_build.py --dev`
# in the build container first (tools/dev/ travels with the snapshot).
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
python -c "import torch; print('devices visible:', torch.cuda.device_count())" 2>/dev/null; /opt/rocm/bin/rocminfo 2>/dev/null | grep -c "gfx950" | sed 's/^/rocminfo gfx950 lines: /'
timeout 600 python -m pytest tests -m gpu -q -rf --durations=15 > gpurun_out/fc_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -25 gpurun_out/fc_gpu_tests.log
timeout 200 python bench.py > gpurun_out/fc_bench.json 2> gpurun_out/fc_bench.err; echo "bench rc=$?"; head -c 600 gpurun_out/fc_bench.json; echo
if [ -f tools/dev/libvallex_hip.so ]; then
  timeout 120 python tools/gemm_bench.py > gpurun_out/fc_gemm_bench.log 2>&1; echo "gemm_bench rc=$?"; cat gpurun_out/fc_gemm_bench.log
fi
timeout 420 python tools/logit_error.py gpurun_out/r03_logit_error.json > gpurun_out/fc_logit_error.log 2>&1; echo "logit_error rc=$?"; tail -40 gpurun_out/fc_logit_error.log
