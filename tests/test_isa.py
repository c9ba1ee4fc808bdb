"""CPU (cross-compiles, no GPU): properties of the generated gfx950 code that the design counts on and that a source edit can lose
silently.  dec_attn (the dominant kernel, 7 200 launches per batch) must reach its first memory requests without a dependent round
trip: every operand of those requests lies in the 16 preloaded kernarg dwords (vall-e-x_amd/_build.py), the slot record is
requested first and the first K / V tile right behind it, and the first wait lets the tile stay in flight."""
import importlib.util
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _product_flags(src):
    spec = importlib.util.spec_from_file_location("_vx_build", os.path.join(ROOT, "vall-e-x_amd", "_build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m._hipcc(), m.FLAGS + m.EXTRA_FLAGS.get(src, [])


def test_fused_dec_attn_reaches_its_first_requests_without_a_round_trip(tmp_path):
    hipcc, flags = _product_flags("decode.hip")
    if shutil.which(hipcc) is None and not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = tmp_path / "decode.s"
    r = subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", "-o", str(out),
                                          os.path.join(ROOT, "vall-e-x_amd", "csrc", "decode.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = out.read_text().splitlines()
    start = next(i for i, ln in enumerate(lines) if re.match(r"_ZN2vx15dec_attn_kernelILb1ELi4EEE\w*:", ln))
    body = []
    for ln in lines[start + 1:]:
        if ln.startswith("\ts_endpgm") or ln.startswith(".Lfunc_end"):
            break
        ln = ln.split(";")[0].rstrip()
        if ln.strip():
            body.append(ln.strip())
    # with kernarg preload the hardware enters 256 bytes behind the symbol: skip the compatibility header (s_load ... s_branch)
    entry = next(i for i, ln in enumerate(body) if ln.startswith("s_branch")) + 1
    head = body[entry:]
    first_wait = next(i for i, ln in enumerate(head) if ln.startswith("s_waitcnt") and "vmcnt" in ln)
    before = head[:first_wait]
    assert not [ln for ln in before if ln.startswith("s_load")], "an argument of the head-of-kernel requests is not preloaded"
    loads = [ln for ln in before if ln.startswith("global_load_dwordx4")]
    assert len(loads) == 9 and " nt" not in loads[0] and all(" nt" in ln for ln in loads[1:]), loads     # record, then 4 K + 4 V rows
    assert re.search(r"vmcnt\(8\)", head[first_wait]), head[first_wait]                                   # waits for the record only
