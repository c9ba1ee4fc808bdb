"""CPU: the development tools stay loadable -- every Python tool parses, every shell script passes `bash -n`, and the experiments
kept as patches under tools/experiments/ still apply to the product sources (so a measured dead end can be re-measured)."""
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


def test_experiment_patches_still_apply():
    if shutil.which("git") is None or not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("not a git checkout (the GPU box runs from a snapshot)")
    patches = glob.glob(os.path.join(ROOT, "tools", "experiments", "*.patch"))
    assert patches
    for p in patches:
        r = subprocess.run(["git", "apply", "--check", p], cwd=ROOT, capture_output=True, text=True)
        assert r.returncode == 0, (os.path.basename(p), r.stderr)
