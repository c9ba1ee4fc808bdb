cd $GRAFT_REPO_ROOT
for sw in VX_GEMM_X3=1 VX_GEMM_F32=1 VX_ATTN_F32=1 VX_FUSE_OUT=0 VX_BALANCE_ROWS=0; do
  echo "== $sw"; env $sw python -m pytest tests/test_gpu_full_length.py tests/test_gpu_batch32_golden.py tests/test_gpu_parity.py -q --timeout 900 -p no:cacheprovider -k "not vocos_long" 2>&1 | tail -2
done > gpurun_out/r2_switches.log 2>&1
cat gpurun_out/r2_switches.log
python bench.py --gpus 2 --steps 1 --warmup 1 --rows 4 --frames 64 --no-cpu-baseline --no-profile > gpurun_out/r2_two_ranks.json 2> gpurun_out/r2_two_ranks.err; tail -3 gpurun_out/r2_two_ranks.err; cat gpurun_out/r2_two_ranks.json | cut -c1-1500
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
