"""CPU: the C-ABI library builds/loads, exports every symbol include/vallex_hip.h and include/vallex_hip_dev.h declare, and the product path
fails loudly (no CPU fallback, no oracle import) when there is no GPU."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    import vallex_amd
    return vallex_amd.load_library()


def test_header_symbols_exported(lib):
    from vallex_amd._capi import DEV_SYMBOLS, SYMBOLS
    for header, syms in (("vallex_hip.h", SYMBOLS), ("vallex_hip_dev.h", DEV_SYMBOLS)):
        hdr = open(os.path.join(ROOT, "include", header)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        declared = set(re.findall(r"\b(vx_[a-z_]+)\s*\(", hdr))
        assert declared == set(syms), header
        for s in declared:
            assert hasattr(lib, s), s
    # the drop-in boundary carries no measurement entry
    assert not [s for s in SYMBOLS if s.startswith(("vx_bench", "vx_prof"))]


def test_struct_layout_matches_header():
    import ctypes as C
    from vallex_amd._capi import vx_batch, vx_config, vx_sampling
    # every descriptor starts with struct_size (ABI guard, include/vallex_hip.h)
    assert vx_config.struct_size.offset == vx_batch.struct_size.offset == vx_sampling.struct_size.offset == 0
    assert vx_config.cu_mask.offset == 10 * 4 and vx_config.arith.offset == 18 * 4 and C.sizeof(vx_config) == 19 * 4
    assert vx_batch.text_ids.offset == 8 and vx_batch.text_lens.offset == 32 and C.sizeof(vx_batch) == 64
    assert vx_sampling.uniforms.offset == 16 and vx_sampling.seed.offset == 32 and vx_sampling.best_of.offset == 48
    assert C.sizeof(vx_sampling) == 64


def test_header_struct_fields_match_binding():
    """field names and order of the three descriptor structs in include/vallex_hip.h == the ctypes binding"""
    from vallex_amd._capi import vx_batch, vx_config, vx_sampling
    hdr = open(os.path.join(ROOT, "include", "vallex_hip.h")).read()
    for st in (vx_config, vx_batch, vx_sampling):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (st.__name__, st.__name__), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = re.findall(r"(\w+)\s*(?:\[\d+\])?\s*;", body)
        assert names == [f[0] for f in st._fields_], (st.__name__, names)


def test_abi_version_and_struct_size_guard(lib):
    """a caller compiled against an older / shorter struct is rejected with VX_EINVAL instead of being read past its end"""
    import ctypes as C
    from vallex_amd._capi import ABI_VERSION, VX_EINVAL, vx_config
    hdr = open(os.path.join(ROOT, "include", "vallex_hip.h")).read()
    assert int(re.search(r"#define VX_ABI_VERSION (\d+)", hdr).group(1)) == ABI_VERSION == lib.vx_abi_version()
    ctx = C.c_void_p()
    short = vx_config(C.sizeof(vx_config) - 4, 2, 1, 8, 8, 8, 1, 0, 0, 0)          # e.g. the ABI-3 struct without `arith`
    assert lib.vx_create(0, C.byref(short), C.byref(ctx)) == VX_EINVAL
    assert b"struct_size" in lib.vx_last_error(None)


def test_integration_md_stub_matches_binding():
    """the ctypes structs a maintainer would paste from INTEGRATION.md have the binding's exact layout"""
    import ctypes as C
    from vallex_amd import _capi
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = max(re.findall(r"```python\n(.*?)```", md, re.S), key=len)
    classes = re.findall(r"(class vx_\w+\(C\.Structure\):.*?)(?=\nclass |\n\ndef |\ndef )", block, re.S)
    assert len(classes) == 3, len(classes)
    ns = {"C": C}
    exec("\n".join(classes), ns)
    for name in ("vx_config", "vx_batch", "vx_sampling"):
        mine, theirs = getattr(_capi, name), ns[name]
        assert C.sizeof(mine) == C.sizeof(theirs), name
        assert [(f[0], getattr(mine, f[0]).offset) for f in mine._fields_] == \
               [(f[0], getattr(theirs, f[0]).offset) for f in theirs._fields_], name


def test_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import vallex_amd
    with pytest.raises(vallex_amd.VallexHipError):
        vallex_amd.Engine(num_layers=2)


def test_product_never_imports_oracle():
    code = ("import sys, vallex_amd\nfrom vallex_amd.utils import generation\nfrom vallex_amd.models import vallex\n"
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'\n")
    subprocess.run([sys.executable, "-c", code], cwd=ROOT, check=True)
    for dp, _, fs in os.walk(os.path.join(ROOT, "vall-e-x_amd")):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def test_library_sources_never_abort_the_host_process():
    """a C-ABI library reports VX_E* and a message; it must not contain abort() / exit() / assert() in host code paths (round 3
    had five abort() calls in launchers: a split count that was not compiled in killed the caller)"""
    import glob
    bad = []
    for f in glob.glob(os.path.join(ROOT, "vall-e-x_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "vall-e-x_amd", "csrc", "*.h")):
        src = re.sub(r"//.*", "", open(f).read())
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"(?<![A-Za-z_])(abort|exit|_Exit|quick_exit|assert)\s*\(", src):
            if m.group(1) == "assert" and "static_assert" in src[max(0, m.start() - 7):m.end()]:
                continue
            bad.append((os.path.basename(f), m.group(0)))
    assert not bad, bad
