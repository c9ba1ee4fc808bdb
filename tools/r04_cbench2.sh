#!/bin/bash
# Round 4: property checks of the C client at every decode-chain size (<= 4 rows: fused small-batch chain, 5..31: context-split
# chain, 32: fused out_proj chain), and the long contexts of BASELINE config 5 (8 rows prompted by 640 frames).
#   /usr/local/graft/bin/gpurun --timeout 200 -- 'bash tools/r04_cbench2.sh'
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out
O=gpurun_out/c_bench2
gcc -std=c99 -O2 -Wall -Wextra -Werror -pedantic -Iinclude examples/c_bench.c -Lvall-e-x_amd/csrc -lvallex_hip \
    -Wl,-rpath,"$R/vall-e-x_amd/csrc" -lm -o /tmp/c_bench 2> ${O}_build.log || { cat ${O}_build.log; exit 1; }
: > ${O}.jsonl; : > ${O}.log
for rows in 2 3 4 5 8 16 31; do
  echo "== rows $rows --check" >> ${O}.log
  timeout 100 /tmp/c_bench --rows $rows --steps 1 --warmup 1 --check >> ${O}.jsonl 2>> ${O}.log; echo "   rc $?" >> ${O}.log
done
echo "== config-5 contexts: 8 rows, Tp 640, 563 frames" >> ${O}.log
timeout 100 /tmp/c_bench --rows 8 --tp 640 --frames 563 --steps 2 --warmup 1 --check >> ${O}.jsonl 2>> ${O}.log; echo "   rc $?" >> ${O}.log
echo "== cap: 4 rows x 1024 frames" >> ${O}.log
timeout 100 /tmp/c_bench --rows 4 --frames 1024 --steps 1 --warmup 0 --check >> ${O}.jsonl 2>> ${O}.log; echo "   rc $?" >> ${O}.log
grep -v "^\[c_bench\] \(step\|warmup\)" ${O}.log
cat ${O}.jsonl
