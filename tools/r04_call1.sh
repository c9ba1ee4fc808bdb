#!/bin/bash
# Round 4, first GPU call: (1) the whole GPU suite on the split engine, (2) wave-priority A/B of the two NAR MFMA kernels, (3) the
# NAR last-layer row trimming and (4) the mid-batch out_proj prologue -- each validated on the goldens with its switch on and A/B'd
# in the bench --, (5) the default bench line with the new legs (reference arithmetic, measured clock, vocoder rooflines).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r04_call1.sh'
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/c1
python -c "import torch; print('devices visible:', torch.cuda.device_count())" 2>/dev/null
timeout 600 python -m pytest tests -m gpu -q -rf --durations=8 > ${O}_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -14 ${O}_gpu_tests.log
if [ -f tools/dev/libvallex_hip.so ]; then
  timeout 200 python tools/prio_ab.py 3 > ${O}_prio_ab.log 2>&1; echo "prio_ab rc=$?"; cat ${O}_prio_ab.log
fi
SUB="tests/test_gpu_full_length.py tests/test_gpu_batch32_golden.py tests/test_gpu_trained_like.py tests/test_gpu_long_context.py"
VX_NAR_TRIM=1 timeout 420 python -m pytest $SUB -m gpu -q -x > ${O}_trim_tests.log 2>&1; echo "trim tests rc=$?"; tail -3 ${O}_trim_tests.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-ref-arith"
for sw in VX_NAR_TRIM=0 VX_NAR_TRIM=1 VX_NAR_TRIM=0 VX_NAR_TRIM=1; do
  env $sw timeout 200 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$sw', d['value'], 'ms', d['ms_per_step'], 'ar', d['ar_ms_per_step'], 'nar', d['nar_ms_per_step'])" | tee -a ${O}_trim_ab.log
done
SUB2="tests/test_gpu_fuzz.py tests/test_gpu_long_context.py tests/test_gpu_properties.py tests/test_gpu_parity.py"
VX_MID_FUSE=1 timeout 420 python -m pytest $SUB2 -m gpu -q -x > ${O}_mid_tests.log 2>&1; echo "mid-fuse tests rc=$?"; tail -3 ${O}_mid_tests.log
for sw in VX_MID_FUSE=0 VX_MID_FUSE=1 VX_MID_FUSE=0 VX_MID_FUSE=1; do
  env $sw timeout 200 python bench.py --long-text --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$sw long-text', d['value'], 'ms', d['ms_per_step'], 'ar', d['ar_ms_per_step'], 'nar', d['nar_ms_per_step'])" | tee -a ${O}_mid_ab.log
done
for sw in VX_MID_FUSE=0 VX_MID_FUSE=1; do
  env $sw timeout 200 python bench.py --rows 16 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-ref-arith 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$sw rows16', d['value'], 'ms', d['ms_per_step'], 'ar', d['ar_ms_per_step'])" | tee -a ${O}_mid_ab.log
done
timeout 400 python bench.py > ${O}_bench.json 2> ${O}_bench.err; echo "bench rc=$?"; python - <<PY
import json
d = json.load(open("${O}_bench.json"))
print("default", d["value"], "ms", d["ms_per_step"], "ar", d["ar_ms_per_step"], "nar", d["nar_ms_per_step"])
print("ref_arith", json.dumps(d.get("ref_arith"))[:600])
r = d["roofline"]
print("roofline", r["kernel"], r["frac"], "others:", {k: (v.get("frac"), v.get("clock_held_mhz")) for k, v in r["others"].items()})
print("vocos", json.dumps(r["others"].get("vocos_head"))[:500])
print("encodec", json.dumps(r["others"].get("encodec_decode"))[:700])
PY
tail -5 ${O}_bench.err
