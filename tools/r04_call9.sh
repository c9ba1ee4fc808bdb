#!/bin/bash
# Round 4, ninth GPU call: the fused small-batch attention without the slot-record round trip in front of its prologue: goldens, A/B, timeline.
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/c9
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_length.py tests/test_gpu_fuzz.py tests/test_gpu_long_context.py tests/test_gpu_trained_like.py tests/test_gpu_properties.py -m gpu -q -x > ${O}_tests.log 2>&1; echo "tests rc=$?"; tail -3 ${O}_tests.log
timeout 200 python tools/step_timeline_b1.py 1 > ${O}_timeline_b1.log 2>&1; tail -8 ${O}_timeline_b1.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], 'ms', d['ms_per_step'], 'ar', d['ar_ms_per_step'], 'nar', d['nar_ms_per_step'])"; }
BQ="--steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-ref-arith"
for rows in 1 2 4; do
  for i in 1 2; do
    timeout 200 python bench.py --rows $rows $BQ 2>/dev/null | line "rows $rows" | tee -a ${O}_ab.log
  done
done
VX_SB_QKV_NSPLIT=16 timeout 200 python bench.py --rows 1 $BQ 2>/dev/null | line "rows 1 nsplit 16" | tee -a ${O}_ab.log
