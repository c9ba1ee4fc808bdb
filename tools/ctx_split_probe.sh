#!/bin/bash
# Experiment (round 6): the 32 rows of a GPU as TWO 16-row half-batches in flight (two contexts, two host threads), so the
# latency-bound GEMM chain of one half's decode step runs beside the HBM-bound attention of the other.  Four points: CU partition
# (2 x 128) or shared CUs, both halves in phase (stagger 0) or 0.35 s apart.  One JSON line each -> gpurun_out/<tag>_ctxsplit.jsonl
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
TAG="${1:-r06}"; ROWS="${2:-16}"; N="${3:-2}"
O=gpurun_out/${TAG}_ctxsplit.jsonl
for nomask in 0 1; do
  for stag in 0 0.35; do
    echo "== contexts $N x rows $ROWS nomask=$nomask stagger=$stag"
    VX_BENCH_CTX_NOMASK=$nomask VX_BENCH_CTX_STAGGER=$stag timeout 600 python bench.py --contexts $N --rows $ROWS --steps 4 --warmup 1 2> gpurun_out/${TAG}_ctxsplit.err | tee -a $O | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  value', d['value'], 'ms per pass', d['ms_per_step'], 'ar in ctx', d['ar_ms_per_pass_in_context'], 'nar in ctx', d['nar_ms_per_pass_in_context'])"
  done
done
