#!/bin/bash
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out
timeout 200 python tools/gemm_short_rows.py > gpurun_out/c13_gemm_tiles.log 2>&1; awk '{print $1,$2,$3, "auto", $11, "best", $14, $15, $16, $17}' gpurun_out/c13_gemm_tiles.log | head -40
timeout 600 python -m pytest tests -m gpu -q --capture=sys 2>&1 | tail -3
timeout 200 python bench.py > gpurun_out/c13_bench.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/c13_bench.json')); print('default', d['value'], d['ms_per_step'], d['ar_ms_per_step'], d['nar_ms_per_step'], d['roofline']['others']['gemm_f16x2']['frac'], d['roofline']['others']['gemm_f16x2']['traffic_source'])"
timeout 200 python bench.py --rows 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/c13_bench_b1.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/c13_bench_b1.json')); print('b1', d['ms_per_step'], d['ar_ms_per_step'], d['nar_ms_per_step'])"
timeout 300 python bench.py --long-text --steps 2 --warmup 1 > gpurun_out/c13_bench_longtext.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/c13_bench_longtext.json')); print('longtext', d['value'], d['ar_ms_per_step'], d['nar_ms_per_step'])"
bash tools/evidence.sh 03 2>&1 | grep -E "rc=|gemm_f16x2|dec_attn_kernel<true|profiles/r03"
