"""CPU, world_size 2, gloo: the N>1 path of the replicas design (row sharding + the one result gather)."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

import vallex_amd  # noqa: F401
from vallex_amd.sharding import infer_sharded, shard_range


def test_shard_range_partitions_rows():
    for n in (0, 1, 7, 32, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake_infer(rows):
    # stands in for VALLE.inference_batch: a deterministic per-row result of ragged length
    return [np.arange(int(r["n"]), dtype=np.int64) * int(r["k"]) for r in rows]


def _worker(rank, world, port, n_rows, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = [dict(n=3 + (i * 7) % 11, k=i + 1) for i in range(n_rows)]
    calls = []

    def fn(shard):
        calls.append(len(shard))
        return _fake_infer(shard)

    out = infer_sharded(rows, fn, dist)
    ok = len(out) == n_rows and all(np.array_equal(o, e) for o, e in zip(out, _fake_infer(rows)))
    # bench.py's timing reduction: max over ranks
    import torch
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, ok, calls, t.item()))
    dist.destroy_process_group()


def test_infer_sharded_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_rows = 9
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_rows, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert res[0][2] == [5] and res[1][2] == [4]          # each rank ran only its own contiguous shard
    assert res[0][3] == res[1][3] == 2.0
