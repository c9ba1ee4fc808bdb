#!/bin/bash
# Counter sweep over the bf16x3 GEMM kernels (next step in DESIGN.md section 6: why do all four generations sit at ~50 % of
# the sustained MFMA rate?).  ONE counter per rocprofv3 pass, every pass under `timeout`: a request for more counters than
# the hardware can collect makes rocprofv3 abort and then hang until killed (it cost 10 GPU-minutes in round 1).
#   on the GPU box:  bash tools/gemm_pmc_sweep.sh            -> gpurun_out/pmc_gemm_<COUNTER>.csv  (+ .log on failure)
# Reading guide: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES = matrix-pipe utilisation; SQ_WAIT_INST_LDS, SQ_WAIT_ANY = where the
# waves wait; SQ_LDS_BANK_CONFLICT / SQ_LDS_DATA_FIFO_FULL = LDS back-pressure; TCP_PENDING_STALL_CYCLES_sum, TA_TA_BUSY_sum =
# vector-memory / DMA path; TCC_HIT_sum, TCC_MISS_sum, FETCH_SIZE = L2 behaviour and HBM-side traffic.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COUNTERS="${*:-SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum FETCH_SIZE}"
for C in $COUNTERS; do
  D=/tmp/pmc_$C
  rm -rf "$D"
  if timeout 120 rocprofv3 --kernel-trace --pmc "$C" -d "$D" -o g -- python "$ROOT/tools/gemm_pmc_case.py" > "/tmp/pmc_$C.log" 2>&1; then
    DB=$(find "$D" -name "*.db" | head -1)
    timeout 60 python "$ROOT/tools/rocpd_pmc_summary.py" "$DB" gemm > "$OUT/pmc_gemm_$C.csv" && echo "$C ok"
  else
    echo "$C FAILED (rc $?)"; tail -5 "/tmp/pmc_$C.log" > "$OUT/pmc_gemm_$C.log"
  fi
done
