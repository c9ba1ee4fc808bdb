#!/bin/bash
# Round 4, third GPU call: the GPU suite on the new defaults (fused small-batch attention v2, 4 decode steps per graph launch, NAR row
# trimming), A/Bs of each, class-level timing of the K / V planes variant, the mid-batch out_proj prologue with all loads up front,
# and gap traces of the 32-row and the one-row chains.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r04_call3.sh'
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/c3
timeout 600 python -m pytest tests -m gpu -q -rf --durations=5 > ${O}_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -12 ${O}_gpu_tests.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], 'ms', d['ms_per_step'], 'ar', d['ar_ms_per_step'], 'nar', d['nar_ms_per_step'])"; }
BQ="--steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-ref-arith"
for rows in 1 2 4; do
  for sw in VX_SB_QKV=0 VX_SB_QKV=4 VX_SB_QKV=0 VX_SB_QKV=4; do
    env $sw timeout 200 python bench.py --rows $rows $BQ 2>/dev/null | line "rows $rows $sw" | tee -a ${O}_sbqkv_ab.log
  done
done
for ns in 16 8 4; do
  VX_SB_QKV_NSPLIT=$ns timeout 200 python bench.py --rows 1 $BQ 2>/dev/null | line "rows 1 nsplit $ns" | tee -a ${O}_sbqkv_ab.log
  VX_SB_QKV_NSPLIT=$ns timeout 200 python bench.py --rows 2 $BQ 2>/dev/null | line "rows 2 nsplit $ns" | tee -a ${O}_sbqkv_ab.log
done
for sw in VX_GRAPH_MULTI=0 VX_GRAPH_MULTI=1 VX_GRAPH_MULTI=0 VX_GRAPH_MULTI=1; do
  env $sw timeout 200 python bench.py --rows 1 $BQ 2>/dev/null | line "rows 1 $sw" | tee -a ${O}_graph_ab.log
  env $sw timeout 200 python bench.py $BQ 2>/dev/null | line "rows 32 $sw" | tee -a ${O}_graph_ab.log
done
for sw in VX_KV_PLANES=0 VX_KV_PLANES=1; do
  env $sw timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ref-arith 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; o=dict(r['others']); o[r['kernel']]=r
print('$sw', d['value'], 'nar', d['nar_ms_per_step'], 'gemm ms', o['gemm_f16x2']['ms_per_step'], 'attn ms', o['attn_full_h2']['ms_per_step'], 'attn TF', o['attn_full_h2']['achieved'])" | tee -a ${O}_kvp_classes.log
done
for sw in VX_MID_FUSE=0 VX_MID_FUSE=1 VX_MID_FUSE=0 VX_MID_FUSE=1; do
  env $sw timeout 200 python bench.py --long-text --steps 2 --warmup 1 2>/dev/null | line "long-text $sw" | tee -a ${O}_mid_ab.log
done
for rows in 8 16; do
  for sw in VX_MID_FUSE=0 VX_MID_FUSE=1; do
    env $sw timeout 200 python bench.py --rows $rows $BQ 2>/dev/null | line "rows $rows $sw" | tee -a ${O}_mid_ab.log
  done
done
cd /tmp && export TMPDIR=/tmp
trace() {   # name, env, bench args
  rm -rf "$R/gpurun_out/prof_$1"
  env $2 timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_$1" -o t -- python "$R/bench.py" $3 --no-cpu-baseline --no-profile --no-ref-arith > "$R/gpurun_out/c3_trace_$1.log" 2>&1
  DB=$(find "$R/gpurun_out/prof_$1" -name '*.db' | head -1)
  [ -n "$DB" ] && python "$R/tools/rocpd_gaps.py" "$DB" --window dec_sample_kernel > "$R/gpurun_out/c3_gaps_$1.csv" && head -14 "$R/gpurun_out/c3_gaps_$1.csv" && tail -1 "$R/gpurun_out/c3_gaps_$1.csv"
  rm -rf "$R/gpurun_out/prof_$1"
}
echo "== 32-row chain"; trace b32 VX_X=0 "--steps 1 --warmup 0"
echo "== one row, fused attention v2"; trace b1c VX_X=0 "--rows 1 --steps 2 --warmup 1"
