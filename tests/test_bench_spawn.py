"""CPU: `python bench.py --gpus 2` starts two ranks BY ITSELF (no launcher), they rendezvous over gloo, shard the job's rows
contiguously, time with barrier + max-over-ranks and gather the results after the timed region -- the N > 1 code path of
bench.py with a stub engine (`--stub`: no GPU, nothing measured).  Also the launcher form the driver uses."""
import json
import os
import pytest
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(out):
    # the contract: stdout is EXACTLY one line, the JSON (gloo's "[Gloo] Rank 0 is connected ..." and any other native
    # chatter on fd 1 must have been routed to stderr)
    lines = out.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), out
    return json.loads(lines[0])


def _check(j, n):
    assert j["n_gpus"] == n and j["scaling"] == "weak"
    assert j["config"]["rows_total"] == 4 * n and j["config"]["rows_per_gpu"] == 4
    pr = sorted(j["per_rank"], key=lambda d: d["rank"])
    assert [d["rank"] for d in pr] == list(range(n))
    assert [d["rows"] for d in pr] == [[4 * r, 4 * r + 4] for r in range(n)]      # row r -> rank r // rows_per_gpu
    frames_all = 2 * 4 * n * 10                                                   # steps x rows x frames
    assert abs(j["value"] * (j["ms_per_step"] * 2 / 1e3) - frames_all / 75.0) < 1e-2 * frames_all / 75.0


def test_bench_gpus2_spawns_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "4",
                        "--frames", "10", "--stub"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    _check(_json_line(r.stdout), 2)


def test_bench_under_torchrun_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--rows", "4", "--frames", "10", "--stub"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    _check(_json_line(r.stdout), 2)


@pytest.mark.parametrize("carry", ["always", "coin"])
def test_bench_long_text_mode_stub(carry):
    """`bench.py --long-text` (BASELINE config 5: 8 rows x 8 chunks; prompt carried over every chunk, or the reference's
    per-chunk coin with `--carry coin`) runs its chunk loop and reports the concatenated frames; stub engine, no GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--long-text", "--carry", carry, "--steps", "1",
                        "--warmup", "0", "--stub"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["config"]["rows_per_gpu"] == 8 and j["config"]["frames"] == 563
    assert abs(j["value"] * j["ms_per_step"] / 1e3 - 8 * 8 * 563 / 75.0) < 1.0          # 8 rows x 8 chunks x 563 frames of audio


def test_bench_contexts_mode_stub():
    """`bench.py --contexts 2` (throughput mode: two contexts share a GPU, each with its own batch in flight) runs both
    worker threads and counts every pass; stub engine, no GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--contexts", "2", "--steps", "2", "--warmup", "1",
                        "--rows", "4", "--frames", "10", "--stub"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["steps"] == 4 and j["config"]["contexts_per_gpu"] == 2 and j["config"]["rows_in_flight_per_gpu"] == 8
    assert abs(j["value"] * j["ms_per_step"] * 4 / 1e3 - 2 * 2 * 4 * 10 / 75.0) < 1e-2        # every pass of every context counted
    assert "not the headline" in j["note"]


def test_bench_gpus8_stub_is_baseline_config_4():
    """the driver's 8-GPU call shape (BASELINE config 4: 256 rows, 32 per GPU): eight self-started ranks, ONE JSON line, eight
    per_rank entries, rows 0..255 sharded contiguously and all gathered after the timed region; stub engine over gloo"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--frames", "10",
                        "--stub"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 8 and j["scaling"] == "weak" and j["config"]["rows_per_gpu"] == 32 and j["config"]["rows_total"] == 256
    pr = sorted(j["per_rank"], key=lambda d: d["rank"])
    assert [d["rows"] for d in pr] == [[32 * r, 32 * r + 32] for r in range(8)]
    assert j["rows_gathered"] == 256
    assert "parallelism" in j["config"] and "replicas x8" in j["config"]["parallelism"]
    # eight ranks on one host do not oversubscribe it: every rank caps its host threads at cores // world
    assert "host threads per rank" in r.stderr


def test_bench_refuses_ranks_without_a_gpu_each():
    """--gpus N with fewer than N visible devices is not a scaling point: fail fast and say why (unless --allow-shared-gpu)"""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this box has a GPU per rank")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "one rank per GPU is required" in r.stderr and "--allow-shared-gpu" in r.stderr


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_config_4_is_one_256_row_job_whatever_n_is(gpus):
    """`bench.py --config 4` (BASELINE config 4): ONE job of 256 rows.  N = 1: eight sequential 32-row shards in one process, put back
    together by sharding.gather_rows over the loopback world; N = 2: two ranks x 128 rows (four engine calls each) gathered over gloo.
    Either way 256 rows arrive and every shard's ids digest is the same before and after the gather.  Stub engine, no GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "4", "--gpus", str(gpus), "--steps", "1", "--warmup", "0",
                        "--frames", "6", "--stub"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["scaling"] == "strong" and j["config"]["rows_total"] == 256 and j["config"]["rows_per_gpu"] == 256 // gpus
    assert j["config"]["baseline_config"] == 4 and j["rows_gathered"] == 256
    sc = j["shard_check"]
    assert sc["match"] is True and sc["rows_gathered"] == 256 and sc["shards"] == (8 if gpus == 1 else gpus)
    assert len(set(sc["shard_digests_before_gather"])) == sc["shards"]          # shards really hold different rows
    assert abs(j["value"] * j["ms_per_step"] / 1e3 - 256 * 6 / 75.0) < 1e-2 * 256 * 6 / 75.0


def test_config_4_job_digest_does_not_depend_on_the_sharding():
    """the 256-row job gives the same ids digest on 1 rank (8 loopback shards) and on 2 ranks (gloo): row g is seeded by its GLOBAL index"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    dig = []
    for gpus in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "4", "--gpus", str(gpus), "--steps", "1", "--warmup",
                            "0", "--frames", "3", "--stub"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        dig.append(_json_line(r.stdout)["shard_check"]["ids_digest_job"])
    assert dig[0] == dig[1]
