"""Multi-GPU = replicas.  Utterances never interact (no cross-row dependency in AR, NAR or Vocos), so the
only "parallelism" is a contiguous split of the batch rows over ranks, one process per GPU with a full weight copy, and a
result gather outside the compute path.  `torch.distributed` (backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests) is
used for exactly one collective per call: `all_gather_object` of the per-rank results (ids: <= 1 MB per GPU)."""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple


def shard_range(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Balanced contiguous split: row r -> rank floor(r / ceil-ish); the first n_rows % world ranks get one extra row."""
    assert world >= 1 and 0 <= rank < world and n_rows >= 0
    q, r = divmod(n_rows, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_rows(lo: int, mine: Sequence, n_rows: int, dist) -> List:
    """The one collective of the multi-GPU path: every rank contributes the results of its contiguous shard [lo, lo+len(mine))
    and gets the results of all `n_rows` rows back in job order (`all_gather_object`; RCCL on GPUs, gloo in the CPU tests)."""
    world = dist.get_world_size()
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, list(mine)))
    out: List = [None] * n_rows
    for lo_r, part in gathered:
        out[lo_r: lo_r + len(part)] = part
    return out


def infer_sharded(rows: Sequence, infer_fn: Callable[[Sequence], List], dist=None) -> List:
    """Run `infer_fn` on this rank's shard and return the results of ALL rows, in the original order, on every rank.
    `dist` is an initialised torch.distributed module (or None for a single process)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(infer_fn(rows))
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(len(rows), world, rank)
    mine = list(infer_fn(rows[lo:hi])) if hi > lo else []
    assert len(mine) == hi - lo
    return gather_rows(lo, mine, len(rows), dist)
