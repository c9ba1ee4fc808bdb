"""CPU: the C-ABI library builds/loads, exports every symbol include/vallex_hip.h declares, and the product path
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
    hdr = open(os.path.join(ROOT, "include", "vallex_hip.h")).read()
    declared = set(re.findall(r"\b(vx_[a-z_]+)\s*\(", hdr))
    from vallex_amd._capi import SYMBOLS
    assert declared == set(SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def test_struct_layout_matches_header():
    import ctypes as C
    from vallex_amd._capi import vx_batch, vx_config, vx_sampling
    assert C.sizeof(vx_config) == 9 * 4
    assert vx_batch.text_lens.offset == 32 and C.sizeof(vx_batch) == 64
    assert vx_sampling.seed.offset == 24 and vx_sampling.best_of.offset == 40 and C.sizeof(vx_sampling) == 56


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
