#!/bin/bash
# MFMA-busy evidence for the matrix kernels: separate rocprofv3 --pmc passes (kernel-trace only, each under its own timeout)
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/pmc_mfma.sh'
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out
CMD="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-ref-arith"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 420 rocprofv3 --kernel-trace --pmc $set -d "$R/gpurun_out/prof_m$i" -o m$i -- $CMD > "$R/gpurun_out/pm_$i.log" 2>&1; echo "pass $i ($set) rc=$?"
  DB=$(find "$R/gpurun_out/prof_m$i" -name '*.db' | head -1)
  [ -n "$DB" ] && python "$R/tools/rocpd_pmc_summary.py" "$DB" > "$R/gpurun_out/pm_$i.csv" && grep -E "gemm_f16x2_kernel<256, 256|attn_full_h2|dec_attn_kernel<true|skinny_gemm_kernel|^kernel" "$R/gpurun_out/pm_$i.csv"
  tail -3 "$R/gpurun_out/pm_$i.log" | cut -c1-300
  rm -rf "$R/gpurun_out/prof_m$i"
done
