#!/bin/bash
# Round 4: bench.py (the driver's contract, Python + ctypes) and examples/c_bench.c (C99) on the SAME box, the C client before and
# after -- calibrates the torch-free client as a measurement instrument (same library, same geometry, different weights generator).
#   /usr/local/graft/bin/gpurun --timeout 260 -- 'bash tools/r04_cbench3.sh'
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out
O=gpurun_out/c_bench3
gcc -std=c99 -O2 -Wall -Wextra -Werror -pedantic -Iinclude examples/c_bench.c -Lvall-e-x_amd/csrc -lvallex_hip \
    -Wl,-rpath,"$R/vall-e-x_amd/csrc" -lm -o /tmp/c_bench 2> ${O}_build.log || { cat ${O}_build.log; exit 1; }
: > ${O}.jsonl; : > ${O}.log
timeout 60 /tmp/c_bench --steps 3 --warmup 1 >> ${O}.jsonl 2>> ${O}.log
t0=$(date +%s)
timeout 230 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ref-arith --no-throughput-leg --no-profile > ${O}_py.json 2> ${O}_py.log
echo "bench.py rc $? wall $(( $(date +%s) - t0 )) s" >> ${O}.log
timeout 60 /tmp/c_bench --steps 3 --warmup 1 >> ${O}.jsonl 2>> ${O}.log
grep -v "^\[c_bench\] \(step\|warmup\)" ${O}.log; tail -n 12 ${O}_py.log; cat ${O}.jsonl; cat ${O}_py.json
