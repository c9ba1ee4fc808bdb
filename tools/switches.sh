#!/bin/bash
# every runtime switch once more on the final kernels (golden parity subset), C client, smoke:
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/switches.sh'
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out; : > gpurun_out/switches.log
SUB="tests/test_gpu_parity.py tests/test_gpu_full_length.py tests/test_gpu_batch32_golden.py tests/test_gpu_long_context.py"
for sw in VX_SB_FUSE=0 VX_FUSE_OUT=0 VX_BALANCE_ROWS=0 VX_GEMM_X3=1 VX_ATTN_X3=1 VX_GEMM_F32=1 VX_ATTN_F32=1; do
  echo "== $sw" | tee -a gpurun_out/switches.log
  env $sw timeout 400 python -m pytest $SUB -m gpu -q -x 2>&1 | tail -2 | tee -a gpurun_out/switches.log
done
gcc -std=c99 -Wall -Wextra -Werror -pedantic -Iinclude examples/c_client.c -Lvall-e-x_amd/csrc -lvallex_hip -Wl,-rpath,$R/vall-e-x_amd/csrc -o /tmp/c_client && /tmp/c_client --run 2>&1 | tail -4 | tee -a gpurun_out/switches.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/switches.log
