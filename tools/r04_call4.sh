#!/bin/bash
# Round 4, fourth GPU call: the staggered wave-group schedule of gemm_f16x2 (VX_GEMM_STG / kernel id 14): isolated A/B, goldens, bench A/B.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/r04_call4.sh'
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/c4
timeout 150 python tools/gemm_stg_ab.py 3 > ${O}_stg_ab.log 2>&1; echo "stg_ab rc=$?"; cat ${O}_stg_ab.log
SUB="tests/test_gpu_full_length.py tests/test_gpu_batch32_golden.py tests/test_gpu_trained_like.py tests/test_gpu_long_context.py tests/test_gpu_range_fallback.py"
VX_GEMM_STG=1 timeout 420 python -m pytest $SUB -m gpu -q -x > ${O}_stg_tests.log 2>&1; echo "stg tests rc=$?"; tail -3 ${O}_stg_tests.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], 'ms', d['ms_per_step'], 'ar', d['ar_ms_per_step'], 'nar', d['nar_ms_per_step'])"; }
for sw in VX_GEMM_STG=0 VX_GEMM_STG=1 VX_GEMM_STG=0 VX_GEMM_STG=1; do
  env $sw timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-ref-arith 2>/dev/null | line "$sw" | tee -a ${O}_stg_bench.log
done
