#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -rf --capture=sys > gpurun_out/c4_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -6 gpurun_out/c4_gpu_tests.log
timeout 200 python bench.py > gpurun_out/c4_bench.json 2> gpurun_out/c4_bench.err; echo "bench rc=$?"; head -c 300 gpurun_out/c4_bench.json; echo
timeout 200 python bench.py --rows 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/c4_bench_b1.json 2> gpurun_out/c4_bench_b1.err; echo "bench b1 rc=$?"; head -c 300 gpurun_out/c4_bench_b1.json; echo
timeout 300 python bench.py --long-text --steps 2 --warmup 1 > gpurun_out/c4_bench_longtext.json 2> gpurun_out/c4_bench_longtext.err; echo "bench longtext rc=$?"; head -c 300 gpurun_out/c4_bench_longtext.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_b1" -o b1 -- python "$R/bench.py" --rows 1 --steps 2 --warmup 1 --no-cpu-baseline --no-profile > "$R/gpurun_out/c4_prof_b1.log" 2>&1; echo "rocprof b1 rc=$?"
DB=$(find "$R/gpurun_out/prof_b1" -name '*.db' | head -1); [ -n "$DB" ] && python "$R/tools/rocpd_summary.py" "$DB" > "$R/gpurun_out/c4_b1_kernel_stats.csv" && head -9 "$R/gpurun_out/c4_b1_kernel_stats.csv"
rm -rf "$R/gpurun_out/prof_b1"
