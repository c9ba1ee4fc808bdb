#!/bin/bash
# Round 4, last GPU minutes: the whole hot path from the torch-free C99 client (examples/c_bench.c) on an MI355X.
#   /usr/local/graft/bin/gpurun --timeout 300 -- 'bash tools/r04_cbench.sh'
# Most important first (the call may be cut by the round's GPU budget): headline geometry + property checks, one utterance,
# the reference's arithmetic, then the result-preserving switches (ids digest printed by every run).
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out
O=gpurun_out/c_bench
gcc -std=c99 -O2 -Wall -Wextra -Werror -pedantic -Iinclude examples/c_bench.c -Lvall-e-x_amd/csrc -lvallex_hip \
    -Wl,-rpath,"$R/vall-e-x_amd/csrc" -lm -o /tmp/c_bench 2> ${O}_build.log || { cat ${O}_build.log; exit 1; }
: > ${O}.jsonl; : > ${O}.log
run() {  # label, env..., -- args
  local label="$1"; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  echo "== $label: ${envs[*]} c_bench $*" | tee -a ${O}.log
  local t0=$(date +%s%N)
  env "${envs[@]}" timeout 120 /tmp/c_bench "$@" >> ${O}.jsonl 2>> ${O}.log
  echo "   rc $? wall $(( ($(date +%s%N) - t0) / 1000000 )) ms" | tee -a ${O}.log
}
run headline -- --steps 3 --warmup 1 --check
run one_utterance -- --rows 1 --steps 3 --warmup 1
run reference_arithmetic -- --arith 3 --steps 2 --warmup 1
run eight_rows -- --rows 8 --steps 2 --warmup 1
run fuse_out_off VX_FUSE_OUT=0 -- --steps 2 --warmup 1
run balance_rows_off VX_BALANCE_ROWS=0 -- --steps 2 --warmup 1
run nar_trim_off VX_NAR_TRIM=0 -- --steps 2 --warmup 1
tail -n 40 ${O}.log
cat ${O}.jsonl
