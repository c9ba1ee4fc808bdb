#!/bin/bash
# end-of-round refresh in one call: evidence (kernel stats + FETCH_SIZE) of the headline, batch-1 kernel stats, bench lines
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out
bash tools/evidence.sh "${1:-03}" 2>&1 | grep -E "rc=|gemm_f16x2|dec_attn_kernel<true"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_b1" -o b1 -- python "$R/bench.py" --rows 1 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-ref-arith > "$R/gpurun_out/b1prof.log" 2>&1
DB=$(find "$R/gpurun_out/prof_b1" -name '*.db' | head -1); [ -n "$DB" ] && python "$R/tools/rocpd_summary.py" "$DB" > "$R/gpurun_out/ev_b1_kernel_stats.csv"; rm -rf "$R/gpurun_out/prof_b1"
cd "$R"
timeout 200 python bench.py > gpurun_out/ev_bench.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/ev_bench.json')); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_source'], d['roofline']['others']['gemm_f16x2']['traffic_source'])"
timeout 200 python bench.py --rows 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ev_bench_b1.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/ev_bench_b1.json')); print('b1', d['ms_per_step'], d['ar_ms_per_step'], d['nar_ms_per_step'])"
