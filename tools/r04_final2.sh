#!/bin/bash
# Round 4, end-of-round evidence, part 2: every runtime switch on the golden subset, a seeded soak of random ragged batches against the
# oracle (all three decode chains), MFMA-busy counter passes.
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/r04_final2.sh'
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out; : > gpurun_out/switches.log
SUB="tests/test_gpu_parity.py tests/test_gpu_full_length.py tests/test_gpu_batch32_golden.py tests/test_gpu_long_context.py"
for sw in VX_SB_QKV=0 VX_SB_FUSE=0 VX_NAR_TRIM=0 VX_GRAPH_MULTI=0 VX_FUSE_OUT=0 VX_BALANCE_ROWS=0 VX_GEMM_X3=1 VX_ATTN_X3=1 VX_GEMM_F32=1 VX_ATTN_F32=1; do
  echo "== $sw" | tee -a gpurun_out/switches.log
  env $sw timeout 400 python -m pytest $SUB -m gpu -q -x 2>&1 | tail -2 | tee -a gpurun_out/switches.log
done
timeout 480 python tools/fuzz_soak.py --seconds 400 > gpurun_out/r04_fuzz_soak.log 2>&1; echo "soak rc=$?"; tail -6 gpurun_out/r04_fuzz_soak.log
bash tools/pmc_mfma.sh 2>&1 | tail -30
python tools/pmc_mfma_summary.py > gpurun_out/r04_mfma_busy.json 2>/dev/null; head -40 gpurun_out/r04_mfma_busy.json
