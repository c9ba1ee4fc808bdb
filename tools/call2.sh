#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -rf --capture=sys --durations=12 > gpurun_out/c2_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -30 gpurun_out/c2_gpu_tests.log
timeout 200 python bench.py --rows 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/c2_bench_b1.json 2> gpurun_out/c2_bench_b1.err; echo "bench b1 rc=$?"; head -c 900 gpurun_out/c2_bench_b1.json; echo
VX_SB_FUSE=0 timeout 200 python bench.py --rows 1 --steps 5 --warmup 2 --no-cpu-baseline --no-profile > gpurun_out/c2_bench_b1_nofuse.json 2> gpurun_out/c2_bench_b1_nofuse.err; echo "bench b1 nofuse rc=$?"; head -c 500 gpurun_out/c2_bench_b1_nofuse.json; echo
timeout 120 python tools/gemm_clock.py > gpurun_out/c2_gemm_clock.log 2>&1; echo "gemm_clock rc=$?"; cat gpurun_out/c2_gemm_clock.log
