#!/bin/bash
# Round 4, fifth GPU call: linear2 of the small-batch chain prefetching the next layer's q rows into L2 (VX_SB_PREFETCH): goldens + A/B.
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/c5
VX_SB_PREFETCH=1 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_length.py tests/test_gpu_fuzz.py -m gpu -q -x > ${O}_pf_tests.log 2>&1; echo "prefetch tests rc=$?"; tail -3 ${O}_pf_tests.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], 'ms', d['ms_per_step'], 'ar', d['ar_ms_per_step'], 'nar', d['nar_ms_per_step'])"; }
BQ="--steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-ref-arith"
for rows in 1 2 4; do
  for sw in VX_SB_PREFETCH=0 VX_SB_PREFETCH=1 VX_SB_PREFETCH=0 VX_SB_PREFETCH=1; do
    env $sw timeout 200 python bench.py --rows $rows $BQ 2>/dev/null | line "rows $rows $sw" | tee -a ${O}_pf_ab.log
  done
done
