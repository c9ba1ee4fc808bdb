#!/bin/bash
# Round 4, call 13: dec_attn with everything its first requests need in the preloaded kernel arguments (no s_load round trip at the
# head of the kernel) on top of call 12; decode goldens, then A/B against tools/devx_late (first tile behind the record).
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/c13b
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_length.py tests/test_gpu_batch32_golden.py tests/test_gpu_long_context.py tests/test_gpu_properties.py -m gpu -q -x > ${O}_tests.log 2>&1; echo "tests rc=$?"; tail -3 ${O}_tests.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], 'ms', d['ms_per_step'], 'ar', d['ar_ms_per_step'], 'nar', d['nar_ms_per_step'])"; }
BQ="--steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-ref-arith"
for i in 1 2 3; do
  VX_LIB=tools/devx_late/libvallex_hip.so timeout 200 python bench.py $BQ 2>/dev/null | line "first tile behind the record (args preloaded)" | tee -a ${O}_early_ab.log
  timeout 200 python bench.py $BQ 2>/dev/null | line "first tile ahead (product, args preloaded)   " | tee -a ${O}_early_ab.log
done
