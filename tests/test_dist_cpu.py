"""CPU: the N>1 path (slice sharding + barrier + max/sum reductions) under gloo with world_size 2."""
import os
import socket

import pytest
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


def _dp_worker(rank, world, port, out):
    """Data-parallel fine-tune exchange: per-rank micro-batch gradients of the adapters, one flat f32
    bucket, SUM all-reduce, 1/world - must equal the single-process gradient of the concatenated batch."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "mri-diffusion-superresolution_amd"))
    torch.set_num_threads(2)
    from mrisr import dist as md
    from oracle import unet as ou
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = ou.MNIST
    up = ou.init_unet_params(cfg, seed=7)       # identical seeds for the weights on every rank
    lora = ou.init_lora_params(up, rank=4, seed=8)
    keys = sorted(lora)
    g = torch.Generator().manual_seed(99)       # the GLOBAL batch; each rank takes its contiguous share
    Bg = 2 * world
    x = torch.randn((Bg, 1, 8, 8), generator=g)
    ctx = torch.randn((Bg, 4, cfg.cross_attention_dim), generator=g)
    tgt = torch.randn((Bg, 1, 8, 8), generator=g)
    t = torch.randint(0, 1000, (Bg,), generator=g)

    def flat_grad(sl):
        lp = {k: v.clone().requires_grad_(True) for k, v in lora.items()}
        pred = ou.unet_forward({**up, **lp}, cfg, x[sl], t[sl], ctx[sl])
        torch.nn.functional.mse_loss(pred, tgt[sl]).backward()
        return torch.cat([lp[k].grad.reshape(-1) for k in keys])

    b, e = md.shard_range(Bg, world, rank)
    bucket = flat_grad(slice(b, e)).contiguous()
    w = md.all_reduce_sum_(bucket)
    bucket /= w
    norm = float(bucket.norm())
    norms = [None] * world
    dist.all_gather_object(norms, norm)
    if rank == 0:
        full = flat_grad(slice(0, Bg))
        out.put({"world": w, "err": float((bucket - full).norm() / full.norm()), "norms": norms, "n": bucket.numel()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_large_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res["world"] == 2 and res["n"] > 0
    assert res["err"] < 1e-5                      # mean of per-rank means == mean over the global batch
    assert res["norms"][0] == res["norms"][1]     # every rank clips with the same global norm: no second collective


def test_all_reduce_is_identity_without_process_group():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "mri-diffusion-superresolution_amd"))
    from mrisr import dist as md
    from mrisr.train import cosine_lr
    v = torch.arange(5, dtype=torch.float32)
    assert md.all_reduce_sum_(v) == 1 and torch.equal(v, torch.arange(5, dtype=torch.float32))
    # diffusers get_cosine_schedule_with_warmup: linear warm-up, half cosine to 0
    assert cosine_lr(0, 1e-4, 500, 10000) == 0.0
    assert abs(cosine_lr(250, 1e-4, 500, 10000) - 5e-5) < 1e-12
    assert abs(cosine_lr(500, 1e-4, 500, 10000) - 1e-4) < 1e-12
    assert abs(cosine_lr(5250, 1e-4, 500, 10000) - 5e-5) < 1e-9
    assert cosine_lr(10000, 1e-4, 500, 10000) < 1e-12
    # like diffusers the progress is not clamped: one full period later the multiplier is back at 1
    assert abs(cosine_lr(500 + 2 * 9500, 1e-4, 500, 10000) - 1e-4) < 1e-12
    assert cosine_lr(12000, 1e-4, 500, 10000) > 0.0


def test_lora_checkpoint_key_forms_and_ema_schedule():
    """peft writes adapters without the adapter name under "base_model.model.", diffusers under "unet."; both load back to
    the in-memory "<module>.lora_A.default.weight" keys.  EMA decay follows diffusers EMAModel.get_decay."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "mri-diffusion-superresolution_amd"))
    from mrisr.train import _FlatAdamW, lora_keys_from_disk, lora_keys_to_disk
    mem = {"down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.lora_A.default.weight": torch.zeros(4, 8),
           "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_out.0.lora_B.default.weight": torch.ones(8, 4)}
    peft = lora_keys_to_disk(mem, "peft")
    assert sorted(peft) == ["base_model.model.down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_out.0.lora_B.weight",
                            "base_model.model.down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.lora_A.weight"]
    dif = lora_keys_to_disk(mem, "diffusers")
    assert all(k.startswith("unet.") and ".default." not in k for k in dif)
    for disk in (peft, dif, lora_keys_to_disk(mem, "memory"), mem):
        back = lora_keys_from_disk(disk)
        assert set(back) == set(mem) and all(torch.equal(back[k], mem[k]) for k in mem)
    assert lora_keys_from_disk({"conv_in.weight": 1}) == {"conv_in.weight": 1}
    with pytest.raises(ValueError):
        lora_keys_to_disk(mem, "hf")
    d = _FlatAdamW.ema_decay_at
    assert d(1) == 0.0 and abs(d(2) - 2 / 11) < 1e-15 and abs(d(101) - 101 / 110) < 1e-15 and d(10 ** 7) == 0.9999
    assert d(5, update_after_step=10) == 0.0 and d(12, update_after_step=10) == 2 / 11
    assert abs(d(11, use_ema_warmup=True, inv_gamma=1.0, power=2 / 3) - (1 - 11 ** (-2 / 3))) < 1e-15
    assert d(3, min_decay=0.5) == 0.5


def _bucket_worker(rank, world, port, out):
    """SURVEY.md 8e large-bucket exchange: the flat gradient vector reduced bucket by bucket in backward order (async) equals
    ONE flat all-reduce; in reduce_scatter mode a rank owns the summed slice of every bucket, updates only those parameters,
    and the all-gather of the updated slices reproduces the full update on every rank."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "mri-diffusion-superresolution_amd"))
    from mrisr import dist as md
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1000 + 37                                                      # ragged on purpose
    ranges = [(700, n), (300, 700), (0, 300)]                          # backward order: the top of the vector first
    g = torch.Generator().manual_seed(1234 + rank)
    grad = torch.randn(n, generator=g)
    flat = grad.clone()
    md.all_reduce_sum_(flat)                                           # the round-1 exchange: one flat all-reduce
    res = {}
    buck = grad.clone()
    red = md.BucketedReducer(buck, ranges, mode="all_reduce")
    for i in range(len(ranges)):
        red.reduce(i)
    res["world"] = red.wait()
    res["bucketed_equals_flat"] = bool(torch.equal(buck, flat))
    # sharded optimiser: theta' = theta - 0.1 * mean gradient, computed only on this rank's shards, then all-gathered
    theta0 = torch.arange(n, dtype=torch.float32) / n
    sh = grad.clone()
    red2 = md.BucketedReducer(sh, ranges, mode="reduce_scatter")
    for i in range(len(ranges)):
        red2.reduce(i)
    red2.wait()
    theta = theta0.clone()
    own = 0
    for i in range(len(ranges)):
        lo, hi = red2.shard(i)
        theta[lo:hi] -= 0.1 * sh[lo:hi] / world
        own += hi - lo
    red2.all_gather_params(theta)
    want = theta0 - 0.1 * flat / world
    res["sharded_update_ok"] = bool(torch.allclose(theta, want, rtol=0, atol=1e-7))
    res["own"] = own
    shards = [None] * world
    dist.all_gather_object(shards, [red2.shard(i) for i in range(len(ranges))])
    if rank == 0:
        res["shards"] = shards
        out.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bucketed_exchange_equals_flat_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res["world"] == 2 and res["bucketed_equals_flat"] and res["sharded_update_ok"]
    # the two ranks' shards tile every bucket
    for i, (lo, hi) in enumerate([(700, 1037), (300, 700), (0, 300)]):
        a, b = res["shards"][0][i], res["shards"][1][i]
        assert a[0] == lo and a[1] == b[0] and b[1] == hi and abs((a[1] - a[0]) - (b[1] - b[0])) <= 1
    assert abs(res["own"] - 1037 / 2) <= 2


def test_bucketed_reducer_validates_and_is_identity_without_group():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "mri-diffusion-superresolution_amd"))
    from mrisr import dist as md
    v = torch.arange(10, dtype=torch.float32)
    r = md.BucketedReducer(v, [(6, 10), (0, 6)])
    r.reduce(0); r.reduce(1)
    assert r.wait() == 1 and torch.equal(v, torch.arange(10, dtype=torch.float32)) and r.shard(0) == (6, 10)
    r.all_gather_params(v)
    with pytest.raises(ValueError):
        md.BucketedReducer(v, [(0, 4), (5, 10)])   # gap
    with pytest.raises(ValueError):
        md.BucketedReducer(v, [(0, 10)], mode="ring")


def test_bench_self_launches_its_ranks_without_torchrun():
    """`python bench.py --gpus 2` with no launcher (WORLD_SIZE unset - how the driver invokes the bench): the parent, which makes no
    GPU call, starts two children with torchrun's environment, relays rank 0's single JSON line and exits with the worst child's
    code.  CPU stand-in for the model (`--launcher-selftest`: gloo rendezvous + the contract's barrier / max-over-ranks)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launcher-selftest"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]  # (gloo itself prints a connection note on stdout)
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["slowest"] == 2.0
    # a launcher/flag mismatch is a clean error, not an assert trace
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launcher-selftest"],
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_controlnet_bucket_ranges_tile_the_flat_vector_in_backward_order():
    """The 1.45 GB ControlNet gradient bucket of SURVEY.md 8e, cut for `BucketedReducer`: the ranges follow the order in which the
    backward finalises them (zero convs, mid block, down blocks 3..0, conv_in, condition embedding, time embedding) and tile the flat
    vector exactly; at SD-1.5 size the total is 361,279,120 parameters.  (Layout = sorted state-dict keys: what the library uses.)"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "mri-diffusion-superresolution_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from mrisr import dist as md
    from mrisr import params as P
    from mrisr.train import controlnet_bucket_ranges
    import mrisr
    shapes = {k: tuple(shp) for k, shp, _ in P.controlnet_param_shapes(mrisr.UNetConfig())}   # SD-1.5 size, names only: no GPU needed
    off, lay = 0, []
    for k in sorted(shapes):
        n = 1
        for d in shapes[k]:
            n *= d
        lay.append((k, off, n))
        off += n
    ranges = controlnet_bucket_ranges(lay)
    names = [g for g, _, _ in ranges]
    assert names[:2] == ["zero_convs", "mid_block"] and names[-3:] == ["conv_in", "controlnet_cond_embedding", "time_embedding"]
    assert [g for g in names if g.startswith("down_blocks.")] == sorted((g for g in names if g.startswith("down_blocks.")), reverse=True)
    covered = sorted((lo, hi) for _, lo, hi in ranges)
    assert covered[0][0] == 0 and covered[-1][1] == off and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    assert off == 361_279_120   # the ControlNet's parameter count
    # the reducer accepts exactly these buckets (identity without a process group); 1.45 GB of f32: an empty strided stand-in
    r = md.BucketedReducer(torch.zeros(off, dtype=torch.float32), [(lo, hi) for _, lo, hi in ranges])
    for i in range(len(ranges)):
        r.reduce(i)
    assert r.wait() == 1
