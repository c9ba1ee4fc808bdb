#!/bin/bash
# Round 4, call 12: the KV arena indexed by launch slot + dec_attn's first tile requested ahead of the slot record.
# Full GPU suite on the product, then A/B against tools/devx_late (-DVX_DEC_ATTN_LATE_TILE=1: same arena, tile behind the record).
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/c12
timeout 600 python -m pytest tests -m gpu -q -x > ${O}_tests.log 2>&1; echo "tests rc=$?"; tail -4 ${O}_tests.log
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], 'ms', d['ms_per_step'], 'ar', d['ar_ms_per_step'], 'nar', d['nar_ms_per_step'])"; }
BQ="--steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-ref-arith"
for i in 1 2 3; do
  VX_LIB=tools/devx_late/libvallex_hip.so timeout 200 python bench.py $BQ 2>/dev/null | line "first tile behind the record" | tee -a ${O}_early_ab.log
  timeout 200 python bench.py $BQ 2>/dev/null | line "first tile ahead (product)  " | tee -a ${O}_early_ab.log
done
for rows in 16 8; do
  VX_LIB=tools/devx_late/libvallex_hip.so timeout 200 python bench.py --rows $rows $BQ 2>/dev/null | line "rows $rows late " | tee -a ${O}_early_ab.log
  timeout 200 python bench.py --rows $rows $BQ 2>/dev/null | line "rows $rows early" | tee -a ${O}_early_ab.log
done
