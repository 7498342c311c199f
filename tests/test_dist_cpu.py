"""CPU: the N>1 path (slice sharding + barrier + max/sum reductions) under gloo with world_size 2."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "mri-diffusion-superresolution_amd"))
    from mrisr import dist as md
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = md.shard_range(n_items, world, rank)
    idx = md.shard_indices(n_items, world, rank)
    dist.barrier()
    elapsed = 1.0 + rank  # rank 1 is the slow one
    mx = md.max_over_ranks(elapsed)
    total = md.sum_over_ranks(float(e - b))
    gathered = [None] * world
    dist.all_gather_object(gathered, idx)
    if rank == 0:
        out.put({"max": mx, "total": total, "gathered": gathered, "range0": (b, e)})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_reductions():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_items, world = 65, 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res["max"] == 2.0  # slowest rank
    assert res["total"] == n_items  # every slice processed exactly once
    allidx = sorted(i for part in res["gathered"] for i in part)
    assert allidx == list(range(n_items))  # disjoint cover, no collective on the data path
    assert res["range0"] == (0, 33)


def test_shard_helpers_single_process():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "mri-diffusion-superresolution_amd"))
    from mrisr import dist as md
    for n in (0, 1, 7, 64, 257):
        for w in (1, 2, 4, 8):
            cover = []
            for r in range(w):
                b, e = md.shard_range(n, w, r)
                cover += list(range(b, e))
                assert abs((e - b) - n / w) < 1
            assert cover == list(range(n))
            assert sorted(i for r in range(w) for i in md.shard_indices(n, w, r)) == list(range(n))
    assert md.max_over_ranks(3.5) == 3.5 and md.sum_over_ranks(2.0) == 2.0
