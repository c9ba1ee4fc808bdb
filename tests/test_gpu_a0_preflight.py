"""First GPU test the suite runs (conftest.py puts it in front): a library-free box check in a SUBPROCESS.  A lease that cannot move
8 MiB through a pinned buffer and run a one-line kernel fails HERE, with a message that says so and the strings that identify the
box, instead of aborting pytest inside the first weight upload of whatever test happens to come first (round 5's driver record:
rc 134, 0 passed, the fault inside vx_load_tensor before any kernel of the library had run)."""
import pytest

import vallex_amd  # noqa: F401
from vallex_amd import _preflight

pytestmark = pytest.mark.gpu


def test_box_can_copy_and_launch_through_pinned_memory():
    res = _preflight.run("pinned")
    assert res["ok"], _preflight.describe(res)
    d = res["detail"]
    assert d["arch"].startswith("gfx950"), f"not an MI355X: {d}"
    assert d["mismatches"] == 0 and d["bytes"] == 8 << 20
    print("preflight:", d, _preflight.box_facts())


def test_pageable_copy_is_reported_not_required():
    """the runtime's pin-on-the-fly path for pageable host memory -- the library does not use it any more (every transfer is staged
    through the context's pinned ring, csrc/engine.hip xfer_*); a box where it faults is worth a line in the log, not a red suite"""
    res = _preflight.run("pageable")
    if not res["ok"]:
        pytest.xfail("box-level, not used by the library: " + _preflight.describe(res))


def test_library_transfers_are_staged_through_the_pinned_ring():
    """round trip through the product's own path: a tensor far larger than one chunk and than the whole ring (VX_PIN_MB) goes up
    through vx_load_tensor in chunks and comes back through vx_read_tap bit for bit"""
    import numpy as np
    from vallex_amd import Engine
    eng = Engine(num_layers=1, max_batch=1, max_text=8, max_prompt=8, max_new=8, with_vocos=False)
    rng = np.random.default_rng(5)
    big = rng.standard_normal((5000, 4099)).astype(np.float32)            # 82 MB: > the 64 MiB ring, odd sizes
    eng.load_tensor("roundtrip.big", big)
    small = rng.standard_normal((3,)).astype(np.float32)
    eng.load_tensor("roundtrip.small", small)
    back = eng.read_tap("roundtrip.big", big.size).reshape(big.shape)
    assert np.array_equal(back, big)
    assert np.array_equal(eng.read_tap("roundtrip.small", 3), small)
