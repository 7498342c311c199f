#!/usr/bin/env python3
"""Throughput of the LoRA fine-tuning step (BASELINE config 3 without the adapter's own training):
SD-1.5 UNet + r=4 LoRA, [B,4,32,32] latents, forward + MSE + backward + all-reduce(flat LoRA grads) + clip + AdamW.
Not the headline metric (bench.py is) - a secondary line for DESIGN.md.

  python tools/bench_train.py [--batch 32] [--steps 5] [--warmup 2] [--dtype bf16] [--profile]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py --gpus N
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))


def _adapter_shapes(channels=(320, 640, 1280, 1280), nums_rb=3, cin=192):
    """(key, shape, fan_in) of Adapter_XL(sk=True, use_conv=True, ksize=3) - reference src/adapters/modules.py:114-157."""
    def conv(n, ci, co, k):
        yield n + ".weight", (co, ci, k, k), ci * k * k
        yield n + ".bias", (co,), ci * k * k
    yield from conv("conv_in", cin, channels[0], 3)
    k = 0
    for i, ch in enumerate(channels):
        for j in range(nums_rb):
            n = f"body.{k}"
            if i > 0 and j == 0:
                yield from conv(n + ".down_opt.op", channels[i - 1], channels[i - 1], 3)
                if channels[i - 1] != ch:
                    yield from conv(n + ".in_conv", channels[i - 1], ch, 3)  # ksize, not 1x1 (modules.py:84-87)
            yield from conv(n + ".block1", ch, ch, 3)
            yield from conv(n + ".block2", ch, ch, 3)
            k += 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--adapter-features", action="store_true", help="add constant T2I-Adapter features (cfg 3 shape)")
    ap.add_argument("--train-adapter", action="store_true", help="BASELINE config 3 in full: Adapter_XL(sk=True, cin=192) on [B,3,256,256] "
                                                                  "runs AND trains every step (233.7 M more parameters, 935 MB gradient bucket)")
    ap.add_argument("--profile", action="store_true", help="per-kernel-class HIP-event profile of one step")
    ap.add_argument("--train-controlnet", action="store_true", help="the ControlNet configuration: a full ControlNet (361,279,120 trainable parameters) feeding a "
                                                                     "FROZEN UNet; condition images 8x the latent size; all-reduce of the 1.45 GB bucket")
    ap.add_argument("--fp8", action="store_true", help="BASELINE configs[4] training leg: forward through the fp8 projections + fp8 attention "
                                                       "(fp8_train), backward in bf16")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    import mrisr
    from mrisr import _lib as L
    from mrisr import params as P
    cfg = mrisr.UNetConfig()
    sd = P.random_state_dict(P.unet_param_shapes(cfg), 20260501, dev)  # same weights on every rank
    sd.update(P.random_state_dict(P.lora_param_shapes(cfg, 4), 20260504, dev))
    f8 = dict(fp8=True, fp8_attention=True, fp8_train=True) if args.fp8 else {}
    if args.train_controlnet:
        sd = P.random_state_dict(P.unet_param_shapes(cfg), 20260501, dev)
        unet = mrisr.UNet2DConditionModel(cfg, compute_dtype=args.dtype)
    else:
        unet = mrisr.UNet2DConditionModel(cfg, compute_dtype=args.dtype, lora_rank=4, lora_alpha=4, lora_fused=True, **f8)
    unet.load_state_dict(sd)
    tr = mrisr.LoRATrainer(unet, lr=1e-4)
    B, h = args.batch, args.latent
    g = torch.Generator(device=dev).manual_seed(1234 + rank)  # rank-offset seeds for data / noise / timesteps
    x = torch.randn((B, 4, h, h), generator=g, device=dev)
    ctx = torch.randn((B, 77, cfg.cross_attention_dim), generator=g, device=dev)
    noise = torch.randn((B, 4, h, h), generator=g, device=dev)
    t = torch.randint(0, 1000, (B,), generator=g, device=dev)
    intra = None
    if args.adapter_features:
        intra = [0.1 * torch.randn((B, c, h >> i, h >> i), generator=g, device=dev) for i, c in enumerate(cfg.block_out_channels)]

    atr = cond = None
    if args.train_adapter:
        ad = mrisr.Adapter_XL(compute_dtype=args.dtype)
        ad.load_state_dict(P.random_state_dict(P.adapter_param_shapes() if hasattr(P, "adapter_param_shapes") else _adapter_shapes(), 20260506, dev))
        atr = mrisr.AdapterTrainer(ad, lr=1e-4)
        assert atr.num_trainable == 233_743_360, atr.num_trainable  # Adapter_XL(sk=True), measured on the reference
        cond = torch.randn((B, 3, 8 * h, 8 * h), generator=g, device=dev)
        intra = None

    ctr = None
    if args.train_controlnet:
        cnet = mrisr.ControlNetModel(cfg, compute_dtype=args.dtype)
        cnet.load_state_dict(P.random_state_dict(P.controlnet_param_shapes(cfg), 20260507, dev))
        ctr = mrisr.ControlNetTrainer(cnet, lr=1e-5)
        assert ctr.num_trainable == 361_279_120, ctr.num_trainable
        cond = torch.randn((B, 3, 8 * h, 8 * h), generator=g, device=dev)

    def one_step():
        if ctr is not None:
            return ctr.step(tr, x, t, ctx, noise, cond)
        if atr is not None:
            return mrisr.joint_step(tr, atr, x, t, ctx, noise, cond)
        return tr.step(x, t, ctx, noise, intra)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    losses = []
    for i in range(args.warmup):
        losses.append(float(one_step()))
        print(f"[bench_train] warmup {i} loss {losses[-1]:.5f}", file=sys.stderr, flush=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    fence()
    el = time.perf_counter() - t0
    from mrisr import dist as md
    el = md.max_over_ranks(el, dev)
    losses.append(float(loss))
    out = {"metric": "lora_finetune_samples_per_s", "value": round(world * B * args.steps / el, 3), "unit": "samples/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 3),
           "dtype": args.dtype + (" (fp8 e4m3 forward: K=320 projections + attention; bf16 backward)" if args.fp8 else ""), "data": "synthetic", "losses": [round(v, 5) for v in losses],
           "workspace_GiB": round(unet.workspace_bytes / 2**30, 2),
           "config": {"workload": f"SD-1.5 UNet + LoRA r=4 fine-tune step, [{B},4,{h},{h}] per GPU, all-reduce of "
                                  f"{tr.num_trainable} f32 grads" + (f" + trainable Adapter_XL ({atr.num_trainable} f32 grads)" if atr else "") + (f"; TRAINABLE ControlNet ({ctr.num_trainable} f32 grads), UNet frozen" if ctr else ""),
                      "adapter_features": bool(intra), "adapter_trained": atr is not None}}
    if args.profile and rank == 0:
        lib = L.lib()
        lib.mrisr_prof_reset()
        lib.mrisr_prof_enable(1)
        one_step()
        torch.cuda.synchronize()
        lib.mrisr_prof_enable(0)
        import ctypes as C
        buf = C.create_string_buffer(1 << 20)
        n = lib.mrisr_prof_report(buf, len(buf))
        classes = json.loads(buf.value[:n].decode()) if n > 0 else {}
        tot = sum(v["ms"] for v in classes.values()) or 1.0
        rows = sorted(classes.items(), key=lambda kv: -kv[1]["ms"])
        out["profile_ms"] = {k: {"ms": round(v["ms"], 3), "launches": v["launches"],
                                 "TFLOPs": round(v["flops"] / v["ms"] / 1e9, 1) if v["ms"] > 0 else 0} for k, v in rows}
        out["profile_total_ms"] = round(tot, 3)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
