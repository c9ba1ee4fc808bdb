#!/bin/bash
# round-3 validation call: suite on the final kernels, headline + fp32-arithmetic bench lines, GEMM A/B, batch-1 kernel trace
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out
python -c "import torch; print('devices visible:', torch.cuda.device_count())" 2>/dev/null; /opt/rocm/bin/rocminfo 2>/dev/null | grep -c "gfx950" | sed 's/^/rocminfo gfx950 agent lines: /'
timeout 900 python -m pytest tests -m gpu -q -rf --capture=sys --durations=8 > gpurun_out/c3_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -16 gpurun_out/c3_gpu_tests.log
timeout 120 python -m pytest tests/test_encodec.py -m gpu -q -s 2>&1 | grep -E "bit-identical|passed|failed" | tee gpurun_out/c3_rvq_ties.log
timeout 200 python bench.py > gpurun_out/c3_bench.json 2> gpurun_out/c3_bench.err; echo "bench rc=$?"; head -c 300 gpurun_out/c3_bench.json; echo
timeout 300 python bench.py --arith f32 --no-cpu-baseline > gpurun_out/c3_bench_f32.json 2> gpurun_out/c3_bench_f32.err; echo "bench f32 rc=$?"; head -c 300 gpurun_out/c3_bench_f32.json; echo
timeout 120 python tools/gemm_bench.py > gpurun_out/c3_gemm_bench.log 2>&1; echo "gemm_bench rc=$?"; cut -c1-1200 gpurun_out/c3_gemm_bench.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_b1" -o b1 -- python "$R/bench.py" --rows 1 --steps 2 --warmup 1 --no-cpu-baseline --no-profile > "$R/gpurun_out/c3_prof_b1.log" 2>&1; echo "rocprof b1 rc=$?"
DB=$(find "$R/gpurun_out/prof_b1" -name '*.db' | head -1); [ -n "$DB" ] && python "$R/tools/rocpd_summary.py" "$DB" > "$R/gpurun_out/c3_b1_kernel_stats.csv" && head -20 "$R/gpurun_out/c3_b1_kernel_stats.csv"
rm -rf "$R/gpurun_out/prof_b1"
