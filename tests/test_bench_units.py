"""CPU: the pieces of bench.py that decide what the JSON line CLAIMS -- the arithmetic label and whether a committed counter
file may be quoted as `roofline.traffic` -- without a GPU."""
import json
import os
import shutil

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dtype_label_follows_the_engine_mode():
    assert bench.dtype_label("f32", "f32") == "f32"                       # the reference's arithmetic end to end
    d = bench.dtype_label("f16x2", "f16x2")
    assert d.startswith("f32 (cached AR decode step) + f16x2") and "22 significant bits" in d and "fp32 accumulate" in d
    m = bench.dtype_label("bf16x3", "f32")
    assert "bf16x3" in m and "projections" in m and m.endswith("f32 full-sequence attention")
    assert bench.dtype_label("stub", "stub") == "none (stub)"


def test_counter_file_is_refused_when_the_kernel_source_changed(tmp_path, monkeypatch):
    """tools/pmc_traffic.py stamps sha256(kernel translation unit + vx_common.h) into profiles/rNN_pmc_<kernel>.json; bench.py
    quotes `traffic` only from a file whose stamp equals the digest of the source in THIS tree (the quoted file went
    stale the moment a kernel changed)."""
    fake = tmp_path / "repo"
    csrc = fake / "vall-e-x_amd" / "csrc"
    csrc.mkdir(parents=True)
    (fake / "profiles").mkdir()
    for f in ("decode.hip", "vx_common.h"):
        shutil.copy(os.path.join(ROOT, "vall-e-x_amd", "csrc", f), csrc / f)
    monkeypatch.setattr(bench, "ROOT", str(fake))
    digest = bench.kernel_source_digest("dec_attn")
    assert digest and len(digest) == 64 and bench.kernel_source_digest("no_such_kernel") is None
    json.dump({"traffic_bytes_per_launch": 123, "source_sha256": digest}, open(fake / "profiles" / "r07_pmc_dec_attn.json", "w"))
    json.dump({"traffic_bytes_per_launch": 1}, open(fake / "profiles" / "r02_pmc_dec_attn.json", "w"))     # older round, no stamp
    pj, src, stale = bench.newest_pmc("dec_attn")
    assert (pj["traffic_bytes_per_launch"], src, stale) == (123, os.path.join("profiles", "r07_pmc_dec_attn.json"), False)
    with open(csrc / "decode.hip", "a") as fh:                             # the kernel changes ...
        fh.write("\n// edited\n")
    assert bench.newest_pmc("dec_attn")[2] is True                         # ... and the committed counters are refused
    assert bench.newest_pmc("gemm_f16x2") == (None, None, False)           # no file at all


def test_committed_counter_files_carry_a_stamp():
    """every counter file of round 3 on names the source it was measured on (older rounds predate the stamp)"""
    prof = os.path.join(ROOT, "profiles")
    new = [f for f in os.listdir(prof) if f.startswith("r0") and "_pmc_" in f and f.endswith(".json") and f[:3] >= "r03"]
    assert new, "no round-3 counter files"
    for f in new:
        pj = json.load(open(os.path.join(prof, f)))
        assert len(pj.get("source_sha256", "")) == 64 and pj["traffic_bytes_per_launch"] > 0, f


def test_config_3_rows_follow_the_reference_presets():
    """BASELINE config 3: 32 rows cycling the reference's 41 preset shapes -- prompts of 161-758 frames, zh / ja / en prompt languages
    from the .npz lang_code (macros.py:15-19), text language cycling en / zh / ja, 100 text ids after the prompt's own"""
    import numpy as np
    from oracle.make_golden import CODE2LANG, PRESET_SHAPES
    rows = bench.make_preset_rows(0, 32)
    assert len(rows) == 32
    for g, r in enumerate(rows):
        name, tp, sp, code = PRESET_SHAPES[g]
        assert r["prompt"].shape == (tp, 8) and r["enroll"] == sp and len(r["text"]) == sp + 100
        assert r["prompt_language"] == CODE2LANG[code] and r["text_language"] == ("en", "zh", "ja")[g % 3]
        assert 0 <= r["prompt"].min() and r["prompt"].max() < 1024
    tps = [r["prompt"].shape[0] for r in rows]
    assert min(tps) == 161 and max(tps) == 758 and {r["prompt_language"] for r in rows} == {"zh", "ja", "en"}
    assert max(len(r["text"]) for r in rows) == 260                     # alan: 160 prompt ids + 100 (the engine is built with max_text 320)
    again = bench.make_preset_rows(8, 4)                                # seeded by the GLOBAL row index: any sharding sees the same job
    assert all(np.array_equal(a["text"], b["text"]) and np.array_equal(a["prompt"], b["prompt"]) for a, b in zip(again, rows[8:12]))


def test_loopback_world_reassembles_a_job_like_ranks_would():
    import numpy as np
    from vallex_amd.sharding import gather_rows, shard_range
    job = [np.full((3, 8), g, np.int64) for g in range(37)]               # ragged: 37 rows over 8 shards
    shards = [shard_range(37, 8, v) for v in range(8)]
    parts = [(a, job[a:b]) for a, b in shards]
    for a, mine in parts:
        got = gather_rows(a, mine, 37, bench.LoopbackWorld(parts))
        assert bench.ids_digest(got) == bench.ids_digest(job)
    assert bench.ids_digest(job[:5]) != bench.ids_digest(job[1:6]) and len(bench.ids_digest(job)) == 16


def test_newest_pmc_files_were_measured_on_the_kernel_sources_in_this_tree():
    """bench.py quotes `roofline.traffic` from the newest profiles/rNN_pmc_<kernel>.json and refuses a file whose source digest is not
    the digest of the kernel's translation unit (+ vx_common.h) in this tree.  For the kernel classes the default line and its fp32 leg
    report, the committed files must be fresh -- otherwise the driver's BENCH line would carry `traffic: null`."""
    import bench
    for k in ("dec_attn", "gemm_f16x2", "attn_full_h2", "skinny_gemm", "skinny16", "gemm_f32", "attn_full"):
        pj, src, stale = bench.newest_pmc(k)
        assert pj is not None and not stale, (k, src, "re-run tools/gpu_call.sh evidence NN [f32] after changing a kernel source")
        assert pj["traffic_bytes_per_launch"] > 0 and pj.get("by_symbol"), (k, src)
