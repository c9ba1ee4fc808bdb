#!/bin/bash
# Round 4, LAST evidence call on the final tree (after the slot-indexed KV arena / early first tile of dec_attn): everything of
# tools/r04_final1.sh, then the two decode switches that change how the arena is walked on the golden subset.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r04_final4.sh'
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
bash tools/r04_final1.sh
: > gpurun_out/switches4.log
SUB="tests/test_gpu_parity.py tests/test_gpu_full_length.py tests/test_gpu_batch32_golden.py tests/test_gpu_long_context.py"
for sw in VX_FUSE_OUT=0 VX_BALANCE_ROWS=0; do
  echo "== $sw" | tee -a gpurun_out/switches4.log
  env $sw timeout 400 python -m pytest $SUB -m gpu -q -x 2>&1 | tail -2 | tee -a gpurun_out/switches4.log
done
