#!/bin/bash
# Round 4, seventh GPU call: the fused small-batch attention with its requests in order of need and LDS-only barriers: goldens + A/B.
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/c7
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_length.py tests/test_gpu_fuzz.py tests/test_gpu_long_context.py tests/test_gpu_trained_like.py tests/test_gpu_properties.py -m gpu -q -x > ${O}_tests.log 2>&1; echo "tests rc=$?"; tail -3 ${O}_tests.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], 'ms', d['ms_per_step'], 'ar', d['ar_ms_per_step'], 'nar', d['nar_ms_per_step'])"; }
BQ="--steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-ref-arith"
for rows in 1 2 4; do
  for i in 1 2; do
    timeout 200 python bench.py --rows $rows $BQ 2>/dev/null | line "rows $rows" | tee -a ${O}_ab.log
  done
done
for ns in 16 4; do
  VX_SB_QKV_NSPLIT=$ns timeout 200 python bench.py --rows 1 $BQ 2>/dev/null | line "rows 1 nsplit $ns" | tee -a ${O}_ab.log
done
VX_SB_QKV_NSPLIT=4 timeout 200 python bench.py --rows 2 $BQ 2>/dev/null | line "rows 2 nsplit 4" | tee -a ${O}_ab.log
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/prof_b1"
timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_b1" -o b1 -- python "$R/bench.py" --rows 1 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-ref-arith > "$R/gpurun_out/c7_b1prof.log" 2>&1
DB=$(find "$R/gpurun_out/prof_b1" -name '*.db' | head -1); [ -n "$DB" ] && python "$R/tools/rocpd_gaps.py" "$DB" --window dec_sample_kernel > "$R/gpurun_out/c7_gaps_b1.csv" && head -8 "$R/gpurun_out/c7_gaps_b1.csv" && tail -1 "$R/gpurun_out/c7_gaps_b1.csv"; rm -rf "$R/gpurun_out/prof_b1"
