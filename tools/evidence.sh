#!/bin/bash
# rocprofv3 evidence of a round (kernel trace + separate PMC pass of the SAME command), summaries into gpurun_out/ev_*:
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/evidence.sh 03'
RND="${1:-03}"
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out
CMD="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-ref-arith"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_kt" -o kt -- $CMD > "$R/gpurun_out/ev_kt.log" 2>&1; echo "kernel-trace rc=$?"
DB=$(find "$R/gpurun_out/prof_kt" -name '*.db' | head -1)
[ -n "$DB" ] && python "$R/tools/rocpd_summary.py" "$DB" > "$R/gpurun_out/ev_kernel_stats.csv" && head -14 "$R/gpurun_out/ev_kernel_stats.csv"
find "$R/gpurun_out/prof_kt" -name '*stats*.csv' | head -3 | while read f; do cp "$f" "$R/gpurun_out/ev_rocprofv3_$(basename $f)"; done
grep -h '"metric"' "$R/gpurun_out/ev_kt.log" | head -1 > "$R/gpurun_out/ev_bench_under_profiler.json"
rm -rf "$R/gpurun_out/prof_kt"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$R/gpurun_out/prof_pmc" -o pmc -- $CMD > "$R/gpurun_out/ev_pmc.log" 2>&1; echo "pmc rc=$?"
DB=$(find "$R/gpurun_out/prof_pmc" -name '*.db' | head -1)
[ -n "$DB" ] && python "$R/tools/rocpd_pmc_summary.py" "$DB" > "$R/gpurun_out/ev_pmc_fetch.csv" && head -12 "$R/gpurun_out/ev_pmc_fetch.csv"
rm -rf "$R/gpurun_out/prof_pmc"
cd "$R" && mkdir -p gpurun_out/ev_profiles && python - <<PY
import sys, os, shutil
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tools")
import pmc_traffic
pmc_traffic.main("$R/gpurun_out/ev_pmc_fetch.csv", "$RND")
for f in os.listdir("$R/profiles"):
    if f.startswith("r$RND" + "_pmc_") and f.endswith(".json"):
        shutil.copy(os.path.join("$R/profiles", f), "$R/gpurun_out/ev_profiles/" + f)
PY
ls gpurun_out/ev_profiles
