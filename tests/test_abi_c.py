"""CPU: include/vallex_hip.h is valid, warning-free C99 and the layout a C compiler gives its three descriptor structs is the
layout of the ctypes binding, field by field (offsetof / sizeof printed by examples/c_client.c, which is compiled against the
header and linked against the in-tree libvallex_hip.so -- no HIP, torch or C++ on the caller's side of the boundary)."""
import ctypes as C
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vall-e-x_amd", "csrc")


@pytest.fixture(scope="module")
def client(tmp_path_factory):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    import __graft_entry__ as g
    g.build()
    exe = str(tmp_path_factory.mktemp("cabi") / "c_client")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "c_client.c"), "-L" + CSRC, "-lvallex_hip", "-Wl,-rpath," + CSRC, "-o", exe],
                   check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout.splitlines()


def test_c_layout_equals_ctypes_binding(client):
    from vallex_amd import _capi
    sizes = {ln.split()[1]: int(ln.split()[2]) for ln in client if ln.startswith("sizeof ")}
    fields = {}
    for ln in client:
        if ln.startswith("vx_") and "." in ln.split()[0]:
            name, off, size = ln.split()
            st, f = name.split(".")
            fields.setdefault(st, []).append((f, int(off), int(size)))
    for st in ("vx_config", "vx_batch", "vx_sampling"):
        cls = getattr(_capi, st)
        assert sizes[st] == C.sizeof(cls), st
        mine = [(n, getattr(cls, n).offset, getattr(cls, n).size) for n, *_ in cls._fields_]
        assert fields[st] == mine, (st, fields[st], mine)


def test_c_caller_sees_abi_version_and_struct_size_guard(client):
    from vallex_amd._capi import ABI_VERSION
    assert client[0] == f"abi {ABI_VERSION} header {ABI_VERSION}"
    guard = [ln for ln in client if ln.startswith("short_struct")][0]
    assert guard.startswith("short_struct rc -1 ") and "struct_size" in guard


# ---- examples/c_bench.c: the whole path from C99 (weights in through vx_load_tensor, vx_infer, vx_vocos_decode) -------------
@pytest.fixture(scope="module")
def c_bench_plan(tmp_path_factory):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    import __graft_entry__ as g
    g.build()
    exe = str(tmp_path_factory.mktemp("cbench") / "c_bench")
    subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "c_bench.c"), "-L" + CSRC, "-lvallex_hip", "-Wl,-rpath," + CSRC, "-lm",
                    "-o", exe], check=True)
    r = subprocess.run([exe, "--plan-fill"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r2 = subprocess.run([exe, "--plan-fill"], capture_output=True, text=True)
    return exe, r.stdout.splitlines(), r2.stdout.splitlines()


def test_c_bench_loads_the_reference_state_dict_layout(c_bench_plan):
    """every key and shape the C client sends through vx_load_tensor is a key / shape of the reference state-dict (the oracle's
    synthetic dict loads strict=True in the live reference, oracle/make_golden.py) or of the Vocos head; ties included"""
    from oracle import synth
    _, lines, _ = c_bench_plan
    want = {k: tuple(v.shape) for k, v in synth.vallex_state_dict(12, 0).items()}
    want.update({"vocos." + k: tuple(v.shape) for k, v in synth.vocos_state_dict(2).items()})
    got, ties = {}, {}
    for ln in lines:
        if ln.startswith("tensor "):
            parts = ln.split()
            if "=" in parts:
                ties[parts[1]] = parts[parts.index("=") + 1]
                parts = parts[: parts.index("=")]
            got[parts[1]] = tuple(int(v) for v in parts[2:])
    assert got == want
    assert sum(1 for k in got if not k.startswith("vocos.")) == 374
    assert ties == {f"nar_predict_layers.{j}.weight": f"nar_audio_embeddings.{j + 2}.word_embeddings.weight" for j in range(6)}


def test_c_bench_runs_the_geometry_of_the_headline_workload(c_bench_plan):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import c_bench_rows
    _, lines, again = c_bench_plan
    rows = [tuple(int(v) for v in ln.split()[3::2]) for ln in lines if ln.startswith("row ")]
    assert rows == c_bench_rows.table(32)
    # the generator is deterministic (same digest twice) and draws what it says (sampled second moment of all tensors)
    assert lines[-1].split()[3:] == again[-1].split()[3:] and lines[-1].startswith("fill ")


def test_c_bench_fails_loudly_without_a_gpu(c_bench_plan):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe, _, _ = c_bench_plan
    r = subprocess.run([exe, "--rows", "1", "--frames", "8", "--layers", "2"], capture_output=True, text=True)
    assert r.returncode == 10 and "vx_create" in r.stderr and r.stdout == ""
