#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -rf --capture=sys > gpurun_out/c8_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/c8_gpu_tests.log
timeout 300 python bench.py --long-text --steps 2 --warmup 1 > gpurun_out/c8_bench_longtext.json 2> gpurun_out/c8_bench_longtext.err; echo "bench longtext rc=$?"; head -c 260 gpurun_out/c8_bench_longtext.json; echo
VX_SB_FUSE=0 timeout 300 python bench.py --long-text --steps 2 --warmup 1 > gpurun_out/c8_bench_longtext_nofuse.json 2> /dev/null; echo "bench longtext nofuse rc=$?"; head -c 260 gpurun_out/c8_bench_longtext_nofuse.json; echo
timeout 200 python bench.py > gpurun_out/c8_bench.json 2> gpurun_out/c8_bench.err; echo "bench rc=$?"; head -c 260 gpurun_out/c8_bench.json; echo
bash tools/evidence.sh 03
