#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out
VX_LN_FUSE=1 timeout 400 python -m pytest tests/test_gpu_batch32_golden.py tests/test_gpu_properties.py tests/test_gpu_trained_like.py -m gpu -q --capture=sys 2>&1 | tail -4
for i in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --no-profile --steps 4 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('plain  ', d['value'], d['ar_ms_per_step'], d['ar_tokens_per_s'])"
  VX_LN_FUSE=1 timeout 200 python bench.py --no-cpu-baseline --no-profile --steps 4 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ln_fuse', d['value'], d['ar_ms_per_step'], d['ar_tokens_per_s'])"
done
