#!/bin/bash
# Round 4, call 10: do any of the HIP runtime's launch-path knobs move the decode chain?  (74 dependent launches per step at 32 rows,
# 50 at one row; every kernel boundary is 1.3-1.5 us.)  One short bench per setting, 32 rows and one row, alternating with the default.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/env_sweep.sh'
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/c10_env_sweep.log; : > $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', 'rows', d['config']['rows_per_gpu'], 'ms', d['ms_per_step'], 'ar', d['ar_ms_per_step'], 'nar', d['nar_ms_per_step'])"; }
BQ="--warmup 1 --no-cpu-baseline --no-profile --no-ref-arith"
run() {  # label, env assignment(s)
  env $2 timeout 120 python bench.py --rows 32 --steps 2 $BQ 2>/dev/null | line "$1" | tee -a $O
  env $2 timeout 120 python bench.py --rows 1 --steps 3 $BQ 2>/dev/null | line "$1" | tee -a $O
}
run default X=0
run dev_kernarg=0 HIP_FORCE_DEV_KERNARG=0
run dev_kernarg=1 HIP_FORCE_DEV_KERNARG=1
run graph_packet_capture=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run graph_packet_capture=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run opt_flush=0 AMD_OPT_FLUSH=0
run fgs_kernarg=0 ROC_USE_FGS_KERNARG=0
run kernarg_copy_opt=0 DEBUG_HIP_KERNARG_COPY_OPT=0
run default X=0
# what the dispatch headers of the decode step look like (barrier bit, acquire / release scopes)
AMD_LOG_LEVEL=4 timeout 120 python bench.py --rows 1 --frames 12 --steps 1 --warmup 0 $BQ 2>&1 >/dev/null | grep -o "Dispatch Header = 0x[0-9a-f]* (type=[0-9]*, barrier=[0-9]*, acquire=[0-9]*, release=[0-9]*)" | sort | uniq -c | sort -rn | head -8 | tee -a $O
