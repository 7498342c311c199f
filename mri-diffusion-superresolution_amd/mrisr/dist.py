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


class BucketedReducer:
    """The large-bucket exchange of the data-parallel fine-tune (BASELINE configs[2]: 935 MB of T2I-Adapter
    gradients per step).  The flat f32 gradient vector is cut into contiguous BUCKETS, given in the order in which their
    gradients become final during the backward (the adapter's top level first); ``reduce(i)`` is called right after the part
    of the backward that finalises bucket ``i`` has been ENQUEUED and launches its collective asynchronously - on RCCL the
    collective runs on the process group's own stream behind an event on the compute stream, so it overlaps the rest of the
    backward instead of following it.

    ``mode="all_reduce"``: every rank ends up with the summed bucket (then runs the whole optimiser).
    ``mode="reduce_scatter"``: rank r ends up with the summed slice ``shard(i)`` of every bucket only (ZeRO-1 style): it
    updates just those parameters - 1/world of the AdamW work and moments - and ``all_gather_params`` circulates the updated
    slices.  Same bytes on the xGMI links as one all-reduce (a ring all-reduce IS a reduce-scatter + an all-gather), but the
    second half moves parameters after the optimiser and can overlap the next step's forward.

    Without an initialised process group everything is the identity (world 1)."""

    def __init__(self, flat: torch.Tensor, ranges, group=None, mode: str = "all_reduce"):
        import torch.distributed as dist
        if mode not in ("all_reduce", "reduce_scatter"):
            raise ValueError(f"unknown mode {mode!r}")
        self.flat, self.group, self.mode = flat, group, mode
        self.ranges = [(int(lo), int(hi)) for lo, hi in ranges]
        covered = sorted(self.ranges)
        if covered[0][0] != 0 or covered[-1][1] != flat.numel() or any(a[1] != b[0] for a, b in zip(covered, covered[1:])):
            raise ValueError("buckets must tile the flat vector")
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self._pending = []

    def shard(self, i: int):
        """[lo, hi) of bucket ``i`` owned by this rank in reduce_scatter mode (balanced; the whole bucket at world 1)."""
        lo, hi = self.ranges[i]
        b, e = shard_range(hi - lo, self.world, self.rank)
        return lo + b, lo + e

    def reduce(self, i: int):
        if not self.active:
            return
        import torch.distributed as dist
        lo, hi = self.ranges[i]
        bucket = self.flat[lo:hi]
        n = hi - lo
        native_rs = self.mode == "reduce_scatter" and n % self.world == 0 and dist.get_backend(self.group) != "gloo"
        if native_rs:  # RCCL: the summed slice lands in place in this rank's shard of the bucket
            s_lo, s_hi = self.shard(i)
            self._pending.append(dist.reduce_scatter_tensor(self.flat[s_lo:s_hi], bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:          # all-reduce (also the functional stand-in for reduce-scatter on gloo / ragged buckets: the shard is a view)
            self._pending.append(dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self) -> int:
        """Block the current stream (RCCL) / the host (gloo) until every launched bucket has been reduced; returns the world size."""
        for w in self._pending:
            w.wait()
        self._pending = []
        return self.world

    def all_gather_params(self, theta: torch.Tensor):
        """reduce_scatter mode, after the optimiser updated this rank's shards of ``theta`` (same layout as the gradients):
        every rank receives every other rank's updated slices."""
        if not self.active or self.mode != "reduce_scatter":
            return
        import torch.distributed as dist
        for i, (lo, hi) in enumerate(self.ranges):
            n = hi - lo
            if n % self.world == 0 and dist.get_backend(self.group) != "gloo":
                s_lo, s_hi = self.shard(i)
                dist.all_gather_into_tensor(theta[lo:hi], theta[s_lo:s_hi].clone(), group=self.group)
            else:  # ragged bucket / gloo: broadcast each owner's slice
                for r in range(self.world):
                    b, e = shard_range(n, self.world, r)
                    if e > b:
                        dist.broadcast(theta[lo + b:lo + e], src=dist.get_global_rank(self.group, r) if self.group is not None else r, group=self.group)
