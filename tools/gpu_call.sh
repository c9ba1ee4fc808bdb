#!/bin/bash
# ONE parametrised script for every metered GPU call (replaces the per-call scripts of rounds 3-4).  Each argument is one step,
# "name arg arg ..."; steps run in order, every step under its own timeout, outputs under gpurun_out/<tag>_*:
#
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_call.sh "tag c1" devices gemm_operands "env_ab VX_QKV_BALANCED 3" "evidence 05 f32"'
#
# steps
#   tag NAME                       prefix of the output files of the following steps (default "call")
#   devices                        GPUs visible to torch / rocminfo
#   suite                          the whole GPU test suite (what the driver runs at round end)
#   golden [ENV=V ...]             the golden parity subset (parity, full-length, batch-32, long-context, trained-like) under env switches
#   switches                       `golden` under every runtime switch of the library, one by one, then the C client and smoke()
#   smoke                          __graft_entry__.smoke()
#   bench LABEL [bench.py args]    one bench.py line -> <tag>_bench_LABEL.json (+ .err)
#   cbench LABEL [ENV=V ...] -- [c_bench args]      one run of the torch-free C client (examples/c_bench.c) -> <tag>_cbench.jsonl
#   env_ab VAR ROUNDS [c_bench args]                the C client with VAR=0 / VAR=1 alternating (ONE library, a runtime switch)
#   lib_ab base,v1,v2 ROUNDS [c_bench args]         the C client on complete library variants (tools/devx_<name>/, _build.py --variant)
#   gemm_operands                  is gemm_f16x2 power- or schedule-bound: the same launch on random / zero-tail / zero / constant operands
#   gemm_f32_ab [ROUNDS]           the fp32-MFMA GEMM kernels (register-staged / LDS-DMA 256 x 128 / LDS-DMA 128 x 128) on the four NAR shapes
#   walk [REPS]                    tile-order sweep of gemm_f16x2_w128 / gemm_f32_dma<256,256> (tools/gemm_walk_sweep.py): us per shape and walk,
#                                  then FETCH_SIZE per (walk, shape) from one rocprofv3 --pmc pass each -> <tag>_walk_time.txt, <tag>_walk_fetch.csv
#   sb_sweep                       small-batch split counts against the context length (1 row: VX_SB_QKV_NSPLIT x prompt length; 8 rows:
#                                  VX_ATT_NSPLIT, VX_QKV_BALANCED at context ~1300) with the C client -> <tag>_cbench.txt
#   power_bench                    board power + sclk from sysfs (tools/power_watch.c) beside two runs of the C client
#   evidence RND [ARITH]           rocprofv3 --kernel-trace summary + a separate --pmc FETCH_SIZE pass of `bench.py --steps 1 [--arith ARITH]`
#   mfma RND [ARITH]               three more separate --pmc passes (MFMA / VALU busy, issue stalls) of the same command
#   gaps LABEL [bench.py args]     kernel trace of a short bench run -> per-kernel duration + gap to the next launch (tools/rocpd_gaps.py)
#   by_shape LABEL [ENV=V ...]    kernel trace of one default bench pass, per-kernel stats split by grid size (tools/rocpd_by_grid.py): the launches
#                                  of one GEMM kernel over different shapes come out as separate rows -> <tag>_by_shape_LABEL.csv
#   py SCRIPT [args]               python tools/SCRIPT args  (the kernel micro-benchmarks: gemm_ab.py, attn_ab.py, step_timeline.py ...)
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out
TAG=call
CF="-std=c99 -O2 -Wall -Wextra -Werror -pedantic -Iinclude"
LF="-Lvall-e-x_amd/csrc -lvallex_hip -Wl,-rpath,$R/vall-e-x_amd/csrc"
GOLDEN="tests/test_gpu_parity.py tests/test_gpu_full_length.py tests/test_gpu_batch32_golden.py tests/test_gpu_long_context.py tests/test_gpu_trained_like.py"
BQ="--no-cpu-baseline --no-profile --no-ref-arith"

need_cbench() { [ -x /tmp/c_bench ] || gcc $CF examples/c_bench.c $LF -lm -o /tmp/c_bench || { echo "c_bench: compile failed"; return 1; }; }
cline() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-22s %8.2f %8.2f %8.2f %8.2f  %s' % ('$1', d['ms_per_step'], d['ar_ms_per_step'], d['nar_ms_per_step'], d['value'], d['ids_fnv1a']))"; }

step_tag() { TAG="$1"; }
step_devices() {
  python -c "import torch; print('devices visible:', torch.cuda.device_count(), torch.cuda.get_device_name(0))" 2>/dev/null
  /opt/rocm/bin/rocminfo 2>/dev/null | grep -c "gfx950" | sed 's/^/rocminfo gfx950 lines: /'
}
step_suite() {
  timeout 900 python -m pytest tests -m gpu -q -rf --durations=8 > gpurun_out/${TAG}_gpu_tests.log 2>&1; echo "gpu suite rc=$?"
  tail -14 gpurun_out/${TAG}_gpu_tests.log
}
step_golden() {
  echo "== golden subset under: ${*:-defaults}" | tee -a gpurun_out/${TAG}_golden.log
  env "$@" timeout 600 python -m pytest $GOLDEN -m gpu -q -x 2>&1 | tail -3 | tee -a gpurun_out/${TAG}_golden.log
}
step_switches() {
  for sw in VX_SB_QKV=0 VX_SB_FUSE=0 VX_FUSE_OUT=0 VX_FUSE_SPLIT=0 VX_BALANCE_ROWS=0 VX_NAR_TRIM=0 VX_GEMM_X3=1 VX_ATTN_X3=1 VX_GEMM_F32=1 VX_ATTN_F32=1; do
    step_golden $sw
  done
  gcc -std=c99 -Wall -Wextra -Werror -pedantic -Iinclude examples/c_client.c $LF -o /tmp/c_client && /tmp/c_client --run 2>&1 | tail -4 | tee -a gpurun_out/${TAG}_golden.log
  step_smoke
}
step_smoke() { timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/${TAG}_smoke.log; }
step_bench() {
  local label="$1"; shift
  timeout 900 python bench.py "$@" > gpurun_out/${TAG}_bench_${label}.json 2> gpurun_out/${TAG}_bench_${label}.err; echo "bench $label rc=$?"
  python3 - <<PY
import json
try:
    d = json.load(open("gpurun_out/${TAG}_bench_${label}.json"))
except Exception as e:
    print("no JSON line:", e); raise SystemExit
print("$label", d["value"], "ms", d["ms_per_step"], "ar", d.get("ar_ms_per_step"), "nar", d.get("nar_ms_per_step"), "|", d["dtype"][:40])
for k in ("ref_arith", "exact_operand_arith", "cpu_baseline"):
    if d.get(k):
        print(" ", k, json.dumps(d[k])[:700])
r = d.get("roofline")
if r:
    print("  roofline", r["kernel"], r["frac"], "others:", {k: v.get("frac") for k, v in r["others"].items()})
PY
  tail -3 gpurun_out/${TAG}_bench_${label}.err | cut -c1-300
}
step_cbench() {
  need_cbench || return 1
  local label="$1"; shift
  local envs=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
  [ "$1" = "--" ] && shift
  out=$(env "${envs[@]}" timeout 200 /tmp/c_bench "$@" 2>> gpurun_out/${TAG}_cbench.log) || { echo "$label: rc $?"; return; }
  echo "$out" >> gpurun_out/${TAG}_cbench.jsonl
  echo "$out" | cline "$label" | tee -a gpurun_out/${TAG}_cbench.txt
}
step_env_ab() {
  local var="$1" rounds="$2"; shift 2
  [ $# -eq 0 ] && set -- --steps 3 --warmup 1
  printf "%-22s %8s %8s %8s %8s  %s\n" "$var" ms_step ar_ms nar_ms audio_s digest | tee -a gpurun_out/${TAG}_cbench.txt
  for r in $(seq 1 "$rounds"); do for v in 0 1; do step_cbench "$var=$v" "$var=$v" -- "$@"; done; done
}
step_lib_ab() {
  need_cbench || return 1
  local variants="${1//,/ }" rounds="$2"; shift 2      # comma-separated (a step's arguments are split on blanks): base,v1,v2
  [ $# -eq 0 ] && set -- --steps 3 --warmup 1
  printf "%-22s %8s %8s %8s %8s  %s\n" variant ms_step ar_ms nar_ms audio_s digest | tee -a gpurun_out/${TAG}_cbench.txt
  for r in $(seq 1 "$rounds"); do
    for v in $variants; do
      if [ "$v" = base ]; then lib="$R/vall-e-x_amd/csrc"; else lib="$R/tools/devx_$v"; fi
      [ -f "$lib/libvallex_hip.so" ] || { echo "$v: no library"; continue; }
      out=$(LD_LIBRARY_PATH="$lib:$LD_LIBRARY_PATH" timeout 200 /tmp/c_bench "$@" 2>> gpurun_out/${TAG}_cbench.log) || { echo "$v: rc $?"; continue; }
      echo "$out" | cline "$v" | tee -a gpurun_out/${TAG}_cbench.txt
    done
  done
}
step_gemm_operands() {
  gcc $CF tools/c_gemm.c $LF -o /tmp/c_gemm || { echo "c_gemm: compile failed"; return 1; }
  {
    echo "== operand patterns, QKV shape of the NAR stages (M 31616, N 3072, K 1024), product tile choice"
    timeout 90 /tmp/c_gemm 31616 3072 1024 6 20
    echo "== linear2 shape (N 1024, K 4096)"
    timeout 90 /tmp/c_gemm 31616 1024 4096 6 20
    echo "== fp32 MFMA kernel, QKV shape (the reference-arithmetic leg's GEMM)"
    timeout 90 /tmp/c_gemm 31616 3072 1024 0 6
  } 2>&1 | tee gpurun_out/${TAG}_gemm_operands.txt
}
step_gemm_f32_ab() {
  gcc $CF tools/c_gemm.c $LF -o /tmp/c_gemm || { echo "c_gemm: compile failed"; return 1; }
  local rounds="${1:-2}"
  {
    for shape in "31616 3072 1024" "31616 1024 1024" "31616 4096 1024" "31616 1024 4096" "12288 3072 1024"; do
      for r in $(seq 1 "$rounds"); do
        for k in ${GEMM_F32_KERNELS:-3 4 5 14}; do
          VX_C_GEMM_MODES=random timeout 90 /tmp/c_gemm $shape $k 6 | tail -1 | sed "s/^/shape $shape kernel $k: /"
        done
      done
    done
  } 2>&1 | tee gpurun_out/${TAG}_gemm_f32_ab.txt
}
# tile-order sweep (round 6): time table, then one FETCH_SIZE pass per walk for the two dominant GEMM kernels, split by grid (= shape)
step_walk() {
  local reps="${1:-8}"
  timeout 900 python tools/gemm_walk_sweep.py time "$reps" 2>&1 | tee gpurun_out/${TAG}_walk_time.txt
  : > gpurun_out/${TAG}_walk_fetch.csv
  for k in 15 14; do
    for w in ${WALKS:-default 2 4 8 16 2,c 3,c 4,c}; do
      ( cd /tmp && export TMPDIR=/tmp
        rm -rf "$R/gpurun_out/prof_walk"
        if [ "$w" = default ]; then unset VX_GEMM_WALK; else export VX_GEMM_WALK="$w"; fi
        timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$R/gpurun_out/prof_walk" -o pmc -- python "$R/tools/gemm_walk_sweep.py" one $k 2 > "$R/gpurun_out/${TAG}_walk_pmc.log" 2>&1
        DB=$(find "$R/gpurun_out/prof_walk" -name '*.db' | head -1)
        [ -n "$DB" ] && python "$R/tools/rocpd_pmc_summary.py" "$DB" --by-grid gemm_f16x2_w128 "gemm_f32_dma_kernel<256, 256>" | sed "s/^/$k,\"$w\",/" >> "$R/gpurun_out/${TAG}_walk_fetch.csv"
        rm -rf "$R/gpurun_out/prof_walk" )
    done
  done
  cat gpurun_out/${TAG}_walk_fetch.csv | cut -c1-200
}
# small batches (round 6): the split counts that are constants today, swept against the context length with the C client
#   1 row:  VX_SB_QKV_NSPLIT 4 / 8 / 16 (fused norm1 + QKV + attention launch) x prompt 100 / 500 / 1000 frames (mean context ~400 / 800 / 1300)
#   8 rows: VX_ATT_NSPLIT 1 / 2 / 4 / 8 (context splits of dec_attn on the general chain) and VX_QKV_BALANCED 0 / 1, prompt 1000 frames
step_sb_sweep() {
  need_cbench || return 1
  printf "%-34s %8s %8s %8s %8s  %s\n" point ms_step ar_ms nar_ms audio_s digest | tee -a gpurun_out/${TAG}_cbench.txt
  for tp in 100 500 1000; do
    for ns in 4 8 16; do
      for r in 1 2; do step_cbench "rows1 tp$tp SB_QKV_NSPLIT=$ns" VX_SB_QKV_NSPLIT=$ns -- --rows 1 --tp $tp --frames 300 --steps 3 --warmup 1 --no-vocos; done
    done
    step_cbench "rows1 tp$tp SB_QKV=0 (5 launches)" VX_SB_QKV=0 -- --rows 1 --tp $tp --frames 300 --steps 3 --warmup 1 --no-vocos
  done
  for ns in 1 2 4 8; do
    for r in 1 2; do step_cbench "rows8 tp1000 ATT_NSPLIT=$ns" VX_ATT_NSPLIT=$ns -- --rows 8 --tp 1000 --frames 300 --steps 3 --warmup 1 --no-vocos; done
  done
  for v in 0 1; do
    for r in 1 2; do step_cbench "rows8 tp1000 QKV_BALANCED=$v" VX_QKV_BALANCED=$v -- --rows 8 --tp 1000 --frames 300 --steps 3 --warmup 1 --no-vocos; done
  done
}
# mid-size batches: context splits of dec_attn (VX_ATT_NSPLIT; 0 = the engine's rule) for 5 / 8 / 12 / 16 rows at short and long contexts
step_sb_sweep2() {
  need_cbench || return 1
  printf "%-34s %8s %8s %8s %8s  %s\n" point ms_step ar_ms nar_ms audio_s digest | tee -a gpurun_out/${TAG}_cbench.txt
  for rows in ${SB2_ROWS:-5 8 12 16}; do
    for tp in 200 1000; do
      for ns in 0 1 2 3 4 6; do
        step_cbench "rows$rows tp$tp ATT_NSPLIT=$ns" VX_ATT_NSPLIT=$ns -- --rows $rows --tp $tp --frames 300 --steps 3 --warmup 1 --no-vocos
      done
    done
  done
}
step_power_bench() {
  need_cbench || return 1
  gcc $CF tools/power_watch.c -o /tmp/power_watch || { echo "power_watch: compile failed"; return 1; }
  timeout 30 /tmp/power_watch 9000 5 > gpurun_out/${TAG}_power.csv 2> gpurun_out/${TAG}_power.log &
  local pw=$!
  step_cbench baseline_a -- --steps 4 --warmup 1
  step_cbench baseline_b -- --steps 4 --warmup 1
  wait $pw
  cat gpurun_out/${TAG}_power.log; wc -l gpurun_out/${TAG}_power.csv
  python3 - <<PY
import csv
rows = [r for r in csv.DictReader(open("gpurun_out/${TAG}_power.csv"))]
w = sorted(float(r["watts"]) for r in rows if float(r["watts"]) > 0)
f = sorted(float(r["sclk_mhz"]) for r in rows if float(r["sclk_mhz"]) > 0)
if w:
    print(f"power: n {len(w)}  median {w[len(w)//2]:.0f} W  p90 {w[int(len(w)*0.9)]:.0f} W  max {w[-1]:.0f} W  cap {rows[0]['cap_watts']} W")
if f:
    print(f"sclk:  median {f[len(f)//2]:.0f} MHz  p10 {f[int(len(f)*0.1)]:.0f}  max {f[-1]:.0f}")
PY
}
prof_cmd() { echo "python $R/bench.py --steps 1 --warmup 0 $BQ ${1:+--arith $1}"; }
step_evidence() {
  local rnd="$1" arith="$2" sfx="${2:+_$2}"
  local cmd; cmd=$(prof_cmd "$arith")
  ( cd /tmp && export TMPDIR=/tmp
    timeout 500 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_kt" -o kt -- $cmd > "$R/gpurun_out/${TAG}_ev${sfx}_kt.log" 2>&1; echo "kernel-trace rc=$?"
    DB=$(find "$R/gpurun_out/prof_kt" -name '*.db' | head -1)
    [ -n "$DB" ] && python "$R/tools/rocpd_summary.py" "$DB" > "$R/gpurun_out/${TAG}_ev${sfx}_kernel_stats.csv" && head -12 "$R/gpurun_out/${TAG}_ev${sfx}_kernel_stats.csv"
    grep -h '"metric"' "$R/gpurun_out/${TAG}_ev${sfx}_kt.log" | head -1 > "$R/gpurun_out/${TAG}_ev${sfx}_bench_under_profiler.json"
    rm -rf "$R/gpurun_out/prof_kt"
    timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$R/gpurun_out/prof_pmc" -o pmc -- $cmd > "$R/gpurun_out/${TAG}_ev${sfx}_pmc.log" 2>&1; echo "pmc FETCH_SIZE rc=$?"
    DB=$(find "$R/gpurun_out/prof_pmc" -name '*.db' | head -1)
    [ -n "$DB" ] && python "$R/tools/rocpd_pmc_summary.py" "$DB" > "$R/gpurun_out/${TAG}_ev${sfx}_pmc_fetch.csv" && head -12 "$R/gpurun_out/${TAG}_ev${sfx}_pmc_fetch.csv"
    rm -rf "$R/gpurun_out/prof_pmc" )
  mkdir -p gpurun_out/${TAG}_profiles
  python - <<PY
import sys, os, shutil
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tools")
import pmc_traffic
pmc_traffic.main("$R/gpurun_out/${TAG}_ev${sfx}_pmc_fetch.csv", "$rnd", arith="$arith")
for f in os.listdir("$R/profiles"):
    if f.startswith("r$rnd" + "_pmc_") and f.endswith(".json"):
        shutil.copy(os.path.join("$R/profiles", f), "$R/gpurun_out/${TAG}_profiles/" + f)
PY
}
step_mfma() {
  local rnd="$1" arith="$2" sfx="${2:+_$2}" i=0
  local cmd; cmd=$(prof_cmd "$arith")
  for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    ( cd /tmp && export TMPDIR=/tmp
      timeout 500 rocprofv3 --kernel-trace --pmc $set -d "$R/gpurun_out/prof_m$i" -o m$i -- $cmd > "$R/gpurun_out/${TAG}_pm${sfx}_$i.log" 2>&1; echo "pass $i ($set) rc=$?"
      DB=$(find "$R/gpurun_out/prof_m$i" -name '*.db' | head -1)
      [ -n "$DB" ] && python "$R/tools/rocpd_pmc_summary.py" "$DB" > "$R/gpurun_out/${TAG}_pm${sfx}_$i.csv"
      rm -rf "$R/gpurun_out/prof_m$i" )
  done
  python tools/pmc_mfma_summary.py gpurun_out/${TAG}_pm${sfx}_1.csv gpurun_out/${TAG}_pm${sfx}_2.csv gpurun_out/${TAG}_pm${sfx}_3.csv \
    > gpurun_out/${TAG}_mfma_busy${sfx}.json && cat gpurun_out/${TAG}_mfma_busy${sfx}.json
}
step_gaps() {
  local label="$1"; shift
  ( cd /tmp && export TMPDIR=/tmp
    timeout 400 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_g" -o g -- python "$R/bench.py" --steps 1 --warmup 1 $BQ "$@" > "$R/gpurun_out/${TAG}_gaps_${label}.log" 2>&1; echo "gaps trace rc=$?"
    DB=$(find "$R/gpurun_out/prof_g" -name '*.db' | head -1)
    [ -n "$DB" ] && python "$R/tools/rocpd_gaps.py" "$DB" > "$R/gpurun_out/${TAG}_gaps_${label}.csv" && head -30 "$R/gpurun_out/${TAG}_gaps_${label}.csv"
    [ -n "$DB" ] && python "$R/tools/rocpd_summary.py" "$DB" > "$R/gpurun_out/${TAG}_gaps_${label}_kernel_stats.csv"
    rm -rf "$R/gpurun_out/prof_g" )
}
step_by_shape() {
  local label="$1"; shift
  ( cd /tmp && export TMPDIR=/tmp
    env "$@" timeout 400 rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_s" -o s -- python "$R/bench.py" --steps 1 --warmup 0 $BQ > "$R/gpurun_out/${TAG}_by_shape_${label}.log" 2>&1; echo "trace rc=$?"
    DB=$(find "$R/gpurun_out/prof_s" -name '*.db' | head -1)
    [ -n "$DB" ] && python "$R/tools/rocpd_by_grid.py" "$DB" gemm_ attn_full > "$R/gpurun_out/${TAG}_by_shape_${label}.csv" && cat "$R/gpurun_out/${TAG}_by_shape_${label}.csv"
    rm -rf "$R/gpurun_out/prof_s" )
}
step_py() { local s="$1"; shift; timeout 600 python tools/$s "$@" 2>&1 | tee gpurun_out/${TAG}_$(basename $s .py).log | tail -60; }

for spec in "$@"; do
  set -- $spec
  name="$1"; shift
  echo "#### step: $name $* ($(date +%H:%M:%S))"
  if declare -f "step_$name" > /dev/null; then "step_$name" "$@"; else echo "unknown step $name"; fi
done
echo "#### done ($(date +%H:%M:%S))"
