#!/bin/bash
# Round 4, end-of-round validation on the FINAL tree (decode.hip changed after tools/r04_final2.sh ran): the decode-chain switches on the
# golden subset, then a seeded soak of random ragged batches against the oracle (all three decode chains, new seed).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r04_final3.sh'
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out; : > gpurun_out/switches3.log
SUB="tests/test_gpu_parity.py tests/test_gpu_full_length.py tests/test_gpu_batch32_golden.py tests/test_gpu_long_context.py"
for sw in VX_SB_QKV=0 VX_SB_FUSE=0 VX_GRAPH_MULTI=0 VX_FUSE_OUT=0 VX_BALANCE_ROWS=0; do
  echo "== $sw" | tee -a gpurun_out/switches3.log
  env $sw timeout 400 python -m pytest $SUB -m gpu -q -x 2>&1 | tail -2 | tee -a gpurun_out/switches3.log
done
timeout 380 python tools/fuzz_soak.py --seconds 300 --first-trial 400 > gpurun_out/r04_fuzz_soak3.log 2>&1; echo "soak rc=$?"; tail -6 gpurun_out/r04_fuzz_soak3.log
