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
