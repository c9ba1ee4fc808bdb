#!/bin/bash
# First GPU call of round 5 (prepared at the end of round 4, when the budget was spent): the three measurements the plan in
# DESIGN.md section 6 ("where round 5 starts") hangs on, about a minute of box time together.
#   /usr/local/graft/bin/gpurun --timeout 240 -- 'bash tools/r05_first_call.sh'
# 1. is gemm_f16x2 bound by power or by its schedule?   tools/c_gemm.c: the same launch on random / zero-tail / zero / constant operands
# 2. does the NAR phase sit at the board's power cap?    tools/power_watch.c beside the C client (if the box exposes amdgpu hwmon)
# 3. the baseline of the round on this box               examples/c_bench.c, headline geometry, twice
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r05_first
CF="-std=c99 -O2 -Wall -Wextra -Werror -pedantic -Iinclude"
LF="-Lvall-e-x_amd/csrc -lvallex_hip -Wl,-rpath,$R/vall-e-x_amd/csrc"
gcc $CF examples/c_bench.c $LF -lm -o /tmp/c_bench && gcc $CF tools/c_gemm.c $LF -o /tmp/c_gemm && gcc $CF tools/power_watch.c -o /tmp/power_watch \
  || { echo "compile failed"; exit 1; }
{
  echo "== 1. operand patterns, QKV shape of the NAR stages (M 31616, N 3072, K 1024), product tile choice"
  timeout 60 /tmp/c_gemm 31616 3072 1024 6 20
  echo "== 1b. linear2 shape (N 1024, K 4096)"
  timeout 60 /tmp/c_gemm 31616 1024 4096 6 20
} > ${O}_gemm_operands.txt 2>&1
timeout 30 /tmp/power_watch 9000 5 > ${O}_power.csv 2> ${O}_power.log &
PW=$!
timeout 60 /tmp/c_bench --steps 4 --warmup 1 > ${O}_c_bench.jsonl 2> ${O}_c_bench.log
timeout 60 /tmp/c_bench --steps 4 --warmup 1 >> ${O}_c_bench.jsonl 2>> ${O}_c_bench.log
wait $PW
cat ${O}_gemm_operands.txt; cat ${O}_power.log; wc -l ${O}_power.csv
python3 - <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open("gpurun_out/r05_first_power.csv"))]
if rows:
    w = [float(r["watts"]) for r in rows if float(r["watts"]) > 0]
    f = [float(r["sclk_mhz"]) for r in rows if float(r["sclk_mhz"]) > 0]
    if w:
        w.sort()
        print(f"power: n {len(w)}  median {w[len(w)//2]:.0f} W  p90 {w[int(len(w)*0.9)]:.0f} W  max {w[-1]:.0f} W  cap {rows[0]['cap_watts']} W")
    if f:
        f.sort()
        print(f"sclk:  median {f[len(f)//2]:.0f} MHz  p10 {f[int(len(f)*0.1)]:.0f}  max {f[-1]:.0f}")
PY
cat ${O}_c_bench.jsonl
