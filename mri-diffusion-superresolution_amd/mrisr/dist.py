"""One-process-per-GPU helpers.  Inference shards over independent slices (reference ``log_validation`` handles one
slice at a time, ``src/adapters/res_srdiff.py:42-43``): NO data-path collective - ``torch.distributed`` (RCCL on GPUs,
gloo in CPU tests) only carries the barrier, the max-over-ranks wall time and the gather of per-rank counts."""
from __future__ import annotations

from typing import List, Tuple

import torch


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) of ``n_items`` slices for ``rank`` (first ``n_items % world`` ranks get one more)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad world/rank {world}/{rank}")
    q, r = divmod(n_items, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def shard_indices(n_items: int, world: int, rank: int) -> List[int]:
    """Round-robin assignment i -> rank i mod world: slice order is preserved per rank."""
    return list(range(rank, n_items, world))


def max_over_ranks(value: float, device=None) -> float:
    """Wall time of the slowest rank (bench contract).  No-op without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def all_reduce_sum_(flat: torch.Tensor, group=None) -> int:
    """The ONE exchange step of the data-parallel fine-tune: in-place SUM over ranks of the flat adapter
    gradient bucket (RCCL ``ncclAllReduce`` on GPUs; gloo in the CPU tests).  Returns the world size, which the optimiser
    folds in as ``grad_scale = 1 / world``.  Without a process group it is the identity (world 1)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return dist.get_world_size(group)
