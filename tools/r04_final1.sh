#!/bin/bash
# Round 4, end-of-round evidence, part 1: GPU suite, kernel stats + FETCH_SIZE pass of the headline, one-row kernel stats, the bench
# lines of configs 3 (default), 2 (--rows 1) and 5 (--long-text), smoke + C client.
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/r04_final1.sh'
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -rf --durations=5 > gpurun_out/r04_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -10 gpurun_out/r04_gpu_tests.log
bash tools/evidence.sh 04 2>&1 | grep -E "rc=|gemm_f16x2|dec_attn_kernel<true|attn_full_h2|r04_pmc"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_b1" -o b1 -- python "$R/bench.py" --rows 1 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-ref-arith > "$R/gpurun_out/b1prof.log" 2>&1
DB=$(find "$R/gpurun_out/prof_b1" -name '*.db' | head -1); [ -n "$DB" ] && python "$R/tools/rocpd_summary.py" "$DB" > "$R/gpurun_out/ev_b1_kernel_stats.csv"; rm -rf "$R/gpurun_out/prof_b1"
cd "$R"
timeout 400 python bench.py > gpurun_out/ev_bench.json 2> gpurun_out/ev_bench.err; echo "bench rc=$?"
python -c "import json; d=json.load(open('gpurun_out/ev_bench.json')); r=d['roofline']; print('default', d['value'], d['ms_per_step'], 'ar', d['ar_ms_per_step'], 'nar', d['nar_ms_per_step'], 'roof', r['kernel'], r['frac'], r['traffic_source'], '| gemm', r['others']['gemm_f16x2']['frac'], r['others']['gemm_f16x2'].get('clock_held_mhz'), r['others']['gemm_f16x2']['traffic_source'], '| ref_arith', d['ref_arith']['value'], '| ar_step', r['others']['ar_step']['frac'])"
timeout 200 python bench.py --rows 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ev_bench_b1.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/ev_bench_b1.json')); print('b1', d['ms_per_step'], d['ar_ms_per_step'], d['nar_ms_per_step'], d.get('real_time_factor'))"
timeout 300 python bench.py --long-text --steps 2 --warmup 1 > gpurun_out/ev_bench_lt.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/ev_bench_lt.json')); print('long-text', d['value'], d['ms_per_step'], d['ar_ms_per_step'], d['nar_ms_per_step'])"
gcc -std=c99 -Wall -Wextra -Werror -pedantic -Iinclude examples/c_client.c -Lvall-e-x_amd/csrc -lvallex_hip -Wl,-rpath,$R/vall-e-x_amd/csrc -o /tmp/c_client && /tmp/c_client --run 2>&1 | tail -4 | tee gpurun_out/ev_c_client.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/ev_c_client.log
