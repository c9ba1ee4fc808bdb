"""CPU: the development tools stay loadable -- every Python tool parses, every shell script passes `bash -n`, and the experiments
kept as patches under tools/experiments/ apply to the commit they name (so a measured dead end can be re-measured)."""
import ast
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_tools_parse():
    files = glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "oracle", "*.py"))
    assert len(files) > 15
    for f in files:
        ast.parse(open(f).read(), filename=f)


def test_shell_tools_are_valid_bash():
    if shutil.which("bash") is None:
        pytest.skip("no bash")
    for f in glob.glob(os.path.join(ROOT, "tools", "*.sh")):
        r = subprocess.run(["bash", "-n", f], capture_output=True, text=True)
        assert r.returncode == 0, (f, r.stderr)


def test_experiment_patches_apply_to_their_base_commit(tmp_path):
    """every patch names the commit it was measured on (`# base: <sha>` in its first line) and applies to THAT tree -- the
    product sources move on, a measured dead end stays re-measurable"""
    if shutil.which("git") is None or not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("not a git checkout (the GPU box runs from a snapshot)")
    patches = glob.glob(os.path.join(ROOT, "tools", "experiments", "*.patch"))
    assert patches
    for p in patches:
        first = open(p).readline()
        assert first.startswith("# base: "), (os.path.basename(p), "first line must be `# base: <commit>`")
        base = first.split()[2]
        files = [ln.split(" b/", 1)[1].strip() for ln in open(p) if ln.startswith("diff --git ")]
        d = tmp_path / os.path.basename(p)
        d.mkdir()
        tar = subprocess.run(["git", "archive", base] + files, cwd=ROOT, capture_output=True)
        assert tar.returncode == 0, tar.stderr
        subprocess.run(["tar", "-x", "-C", str(d)], input=tar.stdout, check=True)
        r = subprocess.run(["git", "apply", "--check", p], cwd=str(d), capture_output=True, text=True)
        assert r.returncode == 0, (os.path.basename(p), r.stderr)


def test_committed_resource_report_shows_no_spills():
    """profiles/rNN_isa_resources.txt (tools/isa_resources.sh, compiler's own figures): every kernel of the library, none with
    scratch, and the occupancies the design counts on"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_isa_resources.txt")))
    assert files
    rows = {}
    for line in open(files[-1]).read().splitlines()[1:]:
        name, rest = line[:58].strip(), line[58:].split()
        rows[name] = dict(zip(("vgpr", "agpr", "scratch", "lds", "occ"), (int(v) for v in rest)))
    assert len(rows) >= 50
    assert all(r["scratch"] == 0 for r in rows.values()), [n for n, r in rows.items() if r["scratch"]]
    assert rows["gemm_f16x2_kernel<256, 256, 2, 0> (gemm_f16x2)"]["lds"] == 131072       # one 8-wave workgroup per CU
    assert rows["attn_full_h2_kernel<0> (attn_full_h2)"]["occ"] == 2 and rows["dec_attn_kernel<true, 4, false> (decode)"]["occ"] == 4 and rows["dec_attn_kernel<true, 84, true> (decode)"]["occ"] == 4
    # the fused small-batch attention holds ONE 8-wave workgroup per CU (its grid is sized for one round, engine.hip)
    assert rows["dec_attn_qkv_kernel<8, 8> (decode)"]["occ"] == 2


def _synthetic_checkpoint(tmp_path, layers=2):
    import torch
    from oracle import synth
    sd = synth.vallex_state_dict(layers, 3, 1.0)
    torch.save({"model": {k: torch.from_numpy(v) for k, v in sd.items()}, "epoch": 1}, tmp_path / "vallex-checkpoint.pt")
    torch.save({k: torch.from_numpy(v) for k, v in synth.vocos_state_dict(2).items()}, tmp_path / "vocos.bin")
    return str(tmp_path / "vallex-checkpoint.pt"), str(tmp_path / "vocos.bin")


def _run_verifier(args):
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verify_checkpoint.py")] + args, cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-2000:]
    return json.loads(r.stdout[r.stdout.index("{"):])


def test_checkpoint_verifier_cpu_legs(tmp_path):
    """tools/verify_checkpoint.py --no-gpu on a checkpoint FILE of the reference's layout: strict load through torch.load(...)
    ["model"], operand headroom against the f16x2 limit -- what a user with the real vallex-checkpoint.pt runs first"""
    ck, _ = _synthetic_checkpoint(tmp_path)
    rep = _run_verifier(["--ckpt", ck, "--presets", os.path.join(ROOT, "tests", "golden", "presets"), "--max-presets", "1",
                         "--n-text", "10", "--frames", "12", "--no-gpu"])
    assert rep["ok"] and rep["load"]["strict"] and rep["load"]["keys"] == rep["load"]["expected_keys"] == 94
    assert rep["load"]["num_layers"] == 2 and rep["load"]["checkpoint_top_level"] == ["epoch", "model"]
    assert set(rep["headroom"]["max_abs_operand"]) == {"ln", "q8", "k", "v", "att", "ffn"} and rep["headroom"]["worst"] < 2047.0


@pytest.mark.gpu
def test_checkpoint_verifier_on_the_gpu(tmp_path):
    """the GPU legs: ids of every preset job equal between the f16x2 and the fp32 arithmetic, no fallback, Vocos head within 1e-4
    RMS of the CPU restatement -- the report a user gets from the real files"""
    ck, vo = _synthetic_checkpoint(tmp_path)
    rep = _run_verifier(["--ckpt", ck, "--vocos", vo, "--presets", os.path.join(ROOT, "tests", "golden", "presets"), "--max-presets",
                         "2", "--n-text", "12", "--frames", "24"])
    assert rep["ok"]
    assert all(v["equal"] for v in rep["parity"]["cross_arith"].values()) and len(rep["parity"]["cross_arith"]) == 4
    assert all(j["fallbacks"]["lifetime"] == 0 for j in rep["parity"]["f16x2"].values())
    assert rep["vocos"]["rms_vs_cpu_restatement"] <= 1e-4 and "pip_vocos" in rep["vocos"]


def test_dev_build_has_no_duplicate_kernel_symbols():
    """the tools-only dev library (vall-e-x_amd/_build.py --dev) links the product translation units PLUS tools/dev_src/*.hip into one
    shared object: a kernel promoted from the probe file to the product must be renamed in the probe file, or the dev link fails with a
    duplicate symbol (it did, silently, for a round).  Static check of the non-template __global__ names (templates may repeat)."""
    import re
    pat = re.compile(r"^(template\s*<[^>]*>\s*)?__global__[^;{]*?\bvoid\s+(\w+)\s*\(", re.M | re.S)

    def kernels(path):
        return {m.group(2) for m in pat.finditer(open(path).read()) if not m.group(1)}
    product = {}
    for f in glob.glob(os.path.join(ROOT, "vall-e-x_amd", "csrc", "*.hip")):
        if os.path.basename(f) == "preflight.hip":      # its own executable
            continue
        for k in kernels(f):
            assert k not in product, (k, f, product[k])
            product[k] = f
    assert len(product) > 30
    for f in glob.glob(os.path.join(ROOT, "tools", "dev_src", "*.hip")):
        dup = kernels(f) & set(product)
        assert not dup, (os.path.basename(f), sorted(dup))
