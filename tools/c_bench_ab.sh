#!/bin/bash
# A/B of library variants with the torch-free C client: every variant is a complete libvallex_hip.so (tools/devx_<name>/, built by
# `python vall-e-x_amd/_build.py --variant=<name>:<files>:<flags>`), each run is its own process (weights in, warm-up, timed batches:
# ~3 s), variants interleaved over ROUNDS rounds.  The ids digest of every run is printed: a result-preserving variant prints the
# product's.
#   /usr/local/graft/bin/gpurun --timeout 200 -- 'bash tools/c_bench_ab.sh "base maxilp maxmem" 2 --steps 3 --warmup 1'
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
VARIANTS="${1:-base}"; ROUNDS="${2:-2}"; shift 2
mkdir -p gpurun_out
O=gpurun_out/c_bench_ab
gcc -std=c99 -O2 -Wall -Wextra -Werror -pedantic -Iinclude examples/c_bench.c -Lvall-e-x_amd/csrc -lvallex_hip -lm -o /tmp/c_bench_ab \
    2> ${O}_build.log || { cat ${O}_build.log; exit 1; }
: > ${O}.log
printf "%-14s %8s %8s %8s %8s  %s\n" variant ms_step ar_ms nar_ms audio_s digest | tee ${O}.txt
for r in $(seq 1 "$ROUNDS"); do
  for v in $VARIANTS; do
    if [ "$v" = base ]; then lib="$R/vall-e-x_amd/csrc"; else lib="$R/tools/devx_$v"; fi
    [ -f "$lib/libvallex_hip.so" ] || { echo "$v: no library" | tee -a ${O}.txt; continue; }
    out=$(LD_LIBRARY_PATH="$lib:$LD_LIBRARY_PATH" timeout 120 /tmp/c_bench_ab "$@" 2>> ${O}.log) || { echo "$v: rc $?" | tee -a ${O}.txt; continue; }
    echo "$out" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-14s %8.2f %8.2f %8.2f %8.2f  %s' % ('$v', d['ms_per_step'], d['ar_ms_per_step'], d['nar_ms_per_step'], d['value'], d['ids_fnv1a']))" | tee -a ${O}.txt
  done
done
