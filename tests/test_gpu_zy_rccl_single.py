"""GPU: the N > 1 code path of bench.py on the one GPU a box has -- `--rccl-single` takes every collective call of the multi-GPU
path (RCCL process group bound to the device, barriers, MAX / SUM all_reduce on device tensors, all_gather_object of the per-rank
records, vallex_amd.sharding.gather_rows of the results) with a world of ONE rank.  Not a scaling point (BASELINE config 4 needs 8
GPUs); it keeps the branch the driver's 1/2/4/8 run depends on from being code that has never executed on hardware.  (File name: runs
near the end of the suite, so that a box-level RCCL problem cannot hide the parity tests behind `-x`.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_runs_its_collectives_over_rccl_with_a_world_of_one():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rccl-single", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-ref-arith", "--no-profile"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # stdout carries exactly ONE line, the JSON
    out = json.loads(lines[0])
    assert out["config"]["collective_backend"] == "nccl" and out["n_gpus"] == 1
    assert out["rows_gathered"] == 32 and len(out["per_rank"]) == 1 and out["per_rank"][0]["rows"] == [0, 32]
    assert out["value"] > 0 and out["scaling"] == "weak"
