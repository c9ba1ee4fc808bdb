#!/bin/bash
# Round-6 sweep behind the fused out_proj with context splits (engine.hip: split_fused): AR ms per batch of the C client for 5 .. 10 rows x
# VX_ATT_NSPLIT 2 / 3 / 4 x {dec_attn | combine | out_proj, fused}; VX_FUSE_SPLIT=2 fuses whenever the forced split count is 2 .. 4.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/fuse_split_sweep.sh | tee gpurun_out/fuse_split_sweep.txt'
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
gcc -std=c99 -O2 -Iinclude examples/c_bench.c -Lvall-e-x_amd/csrc -lvallex_hip -Wl,-rpath,$R/vall-e-x_amd/csrc -lm -o /tmp/c_bench || exit 1
run() { # rows tp ns fuse
  out=$(VX_ATT_NSPLIT=$3 VX_FUSE_SPLIT=$(( $4 * 2 )) timeout 120 /tmp/c_bench --rows $1 ${2:+--tp $2} --steps 3 --warmup 1 2>/dev/null) || { echo "rows $1 ns $3 fuse $4: rc $?"; return; }
  echo "$out" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('rows %2d tp %4s ns %d fuse %d  ar %8.2f  total %8.2f  %s' % ($1, '${2:-def}', $3, $4, d['ar_ms_per_step'], d['ms_per_step'], d['ids_fnv1a']))"
}
for rep in 1 2; do
for rows in 5 6 8 10; do for ns in 2 3 4; do for f in 0 1; do run $rows "" $ns $f; done; done; done
for ns in 2 3 4; do for f in 0 1; do run 8 1000 $ns $f; done; done
done
