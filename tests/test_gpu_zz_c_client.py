"""GPU: the whole hot path from a C99 process (examples/c_bench.c over include/vallex_hip.h -- no Python, torch or HIP headers on
the caller's side): 374 + 81 tensors of the checkpoint's shapes in through vx_load_tensor, two utterances of the benchmark geometry
through vx_infer + vx_vocos_decode, and the properties that hold on ANY weights (`--check`): the same call twice gives the same ids,
a row decoded alone equals itself inside the batch, every row has the forced length and holds codes only.  The configuration is the
one of profiles/r04_c_bench_chain_sizes.jsonl (2 rows, 12 layers, 600 frames).  Runs last in the suite (file name)."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vall-e-x_amd", "csrc")


@pytest.mark.gpu
def test_c99_client_runs_the_whole_path_and_its_property_checks(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    import __graft_entry__ as g
    g.build()
    exe = str(tmp_path / "c_bench")
    cc = subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
                         os.path.join(ROOT, "examples", "c_bench.c"), "-L" + CSRC, "-lvallex_hip", "-Wl,-rpath," + CSRC, "-lm",
                         "-o", exe], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    r = subprocess.run([exe, "--rows", "2", "--steps", "1", "--warmup", "1", "--check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["check"] == "ok" and "[check] ok (0 failures)" in r.stderr
    assert out["rows"] == 2 and out["frames"] == 600 and out["layers"] == 12
    assert out["phases_rerun_in_f32"] == 0 and out["rows_truncated"] == 0
    assert out["value"] > 0 and out["ar_ms_per_step"] > 0 and out["nar_ms_per_step"] > 0
