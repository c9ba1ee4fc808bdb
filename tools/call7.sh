#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_range_fallback.py tests/test_gpu_trained_like.py -m gpu -q -rf --capture=sys > gpurun_out/c7_tests.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/c7_tests.log
timeout 200 python bench.py --gpus 2 --rows 4 --frames 64 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/c7_two_ranks.json 2> gpurun_out/c7_two_ranks.err; echo "2 ranks rc=$?"; head -c 1500 gpurun_out/c7_two_ranks.json; echo
