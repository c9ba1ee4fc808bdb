#!/bin/bash
# Round 4, eighth GPU call: timeline inside dec_attn_qkv (dev build); A/B of the LDS-only epilogue barriers of dec_attn (product vs
# tools/devx_sync = -DVX_DEC_ATTN_SYNC=1), goldens on the product.
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/c8
timeout 200 python tools/step_timeline_b1.py 1 > ${O}_timeline_b1.log 2>&1; echo "timeline rc=$?"; cat ${O}_timeline_b1.log | tail -12
timeout 200 python tools/step_timeline_b1.py 4 > ${O}_timeline_b4.log 2>&1; tail -8 ${O}_timeline_b4.log
timeout 300 python -m pytest tests/test_gpu_batch32_golden.py tests/test_gpu_full_length.py tests/test_gpu_properties.py -m gpu -q -x > ${O}_tests.log 2>&1; echo "tests rc=$?"; tail -3 ${O}_tests.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], 'ms', d['ms_per_step'], 'ar', d['ar_ms_per_step'], 'nar', d['nar_ms_per_step'])"; }
BQ="--steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-ref-arith"
for i in 1 2 3; do
  VX_LIB=tools/devx_sync/libvallex_hip.so timeout 200 python bench.py $BQ 2>/dev/null | line "fenced barriers (__syncthreads)" | tee -a ${O}_barrier_ab.log
  timeout 200 python bench.py $BQ 2>/dev/null | line "LDS-only barriers (product)   " | tee -a ${O}_barrier_ab.log
done
