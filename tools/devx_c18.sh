R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"; mkdir -p gpurun_out
CMD="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-ref-arith"
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export VX_GEMM_W128=$v
  timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_w$v -o t -- $CMD > $R/gpurun_out/c18_w$v.log 2>&1; echo "trace w128=$v rc=$?"
  DB=$(find $R/gpurun_out/prof_w$v -name '*.db' | head -1)
  python $R/tools/rocpd_by_grid.py $DB gemm_f16x2 > $R/gpurun_out/c18_by_grid_w$v.csv; cat $R/gpurun_out/c18_by_grid_w$v.csv
  rm -rf $R/gpurun_out/prof_w$v
done
