#!/bin/bash
# Round 4, second GPU call: (1) K / V operand planes from the QKV GEMM's epilogue (VX_KV_PLANES=1): goldens + A/B; (2) norm1 + QKV
# folded into the split attention launch for <= 4 rows (VX_SB_QKV=n): goldens + A/B at 1, 2 and 4 rows; (3) kernel traces with
# durations AND gaps of the 8-row chain (config 5) and of the one-row chain, to see where those steps spend their time.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r04_call2.sh'
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/c2
SUB="tests/test_gpu_full_length.py tests/test_gpu_batch32_golden.py tests/test_gpu_trained_like.py tests/test_gpu_long_context.py tests/test_gpu_range_fallback.py"
VX_KV_PLANES=1 timeout 420 python -m pytest $SUB -m gpu -q -x > ${O}_kvp_tests.log 2>&1; echo "kv-planes tests rc=$?"; tail -3 ${O}_kvp_tests.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-ref-arith"
for sw in VX_KV_PLANES=0 VX_KV_PLANES=1 VX_KV_PLANES=0 VX_KV_PLANES=1; do
  env $sw timeout 200 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$sw', d['value'], 'ms', d['ms_per_step'], 'ar', d['ar_ms_per_step'], 'nar', d['nar_ms_per_step'])" | tee -a ${O}_kvp_ab.log
done
SUB2="tests/test_gpu_parity.py tests/test_gpu_full_length.py tests/test_gpu_fuzz.py tests/test_gpu_long_context.py tests/test_gpu_properties.py tests/test_gpu_trained_like.py"
VX_SB_QKV=4 timeout 500 python -m pytest $SUB2 -m gpu -q -x > ${O}_sbqkv_tests.log 2>&1; echo "sb-qkv tests rc=$?"; tail -3 ${O}_sbqkv_tests.log
for rows in 1 2 4; do
  for sw in VX_SB_QKV=0 VX_SB_QKV=4 VX_SB_QKV=0 VX_SB_QKV=4; do
    env $sw timeout 200 python bench.py --rows $rows --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-ref-arith 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('rows $rows $sw', d['value'], 'ms', d['ms_per_step'], 'ar', d['ar_ms_per_step'], 'nar', d['nar_ms_per_step'])" | tee -a ${O}_sbqkv_ab.log
  done
done
VX_SB_QKV=1 VX_SB_QKV_NSPLIT=16 timeout 200 python bench.py --rows 1 --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-ref-arith 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('rows 1 nsplit16', d['ms_per_step'], 'ar', d['ar_ms_per_step'])" | tee -a ${O}_sbqkv_ab.log
VX_SB_QKV=1 VX_SB_QKV_NSPLIT=8 timeout 200 python bench.py --rows 1 --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-ref-arith 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('rows 1 nsplit8', d['ms_per_step'], 'ar', d['ar_ms_per_step'])" | tee -a ${O}_sbqkv_ab.log
cd /tmp && export TMPDIR=/tmp
trace() {   # name, env, bench args
  rm -rf "$R/gpurun_out/prof_$1"
  env $2 timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_$1" -o t -- python "$R/bench.py" $3 --no-cpu-baseline --no-profile --no-ref-arith > "$R/gpurun_out/c2_trace_$1.log" 2>&1
  DB=$(find "$R/gpurun_out/prof_$1" -name '*.db' | head -1)
  [ -n "$DB" ] && python "$R/tools/rocpd_gaps.py" "$DB" --window dec_sample_kernel > "$R/gpurun_out/c2_gaps_$1.csv" && head -16 "$R/gpurun_out/c2_gaps_$1.csv" && tail -1 "$R/gpurun_out/c2_gaps_$1.csv"
  rm -rf "$R/gpurun_out/prof_$1"
}
echo "== 8-row chain (config 5, one chunk pass), general"; trace lt0 VX_MID_FUSE=0 "--long-text --steps 1 --warmup 0"
echo "== 8-row chain, out_proj with the per-head combine"; trace lt1 VX_MID_FUSE=1 "--long-text --steps 1 --warmup 0"
echo "== one row, small-batch chain"; trace b1a VX_SB_QKV=0 "--rows 1 --steps 2 --warmup 1"
echo "== one row, norm1 + QKV inside the attention launch"; trace b1b VX_SB_QKV=1 "--rows 1 --steps 2 --warmup 1"
