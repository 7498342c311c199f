#!/usr/bin/env python3
"""Secondary configurations of BASELINE.json on ONE GPU (inference legs): not the headline metric, reported in DESIGN.md.
  cfg4: 512x512 slices -> 4x64x64 latents, UNet + ControlNet (full parallel encoder), 50-step DDIM, B=16
  cfg3i: 256x256 slices, UNet + T2I-Adapter features (Adapter_XL run once per slice), 50-step DDIM, B=32
usage: bench_configs.py [cfg4|cfg3i] [ddim_steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import mrisr  # noqa: E402
from mrisr import params as P  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda", 0)
cfg = mrisr.UNetConfig()
t0 = time.perf_counter()
sd = P.random_state_dict(P.unet_param_shapes(cfg), bench.SEED, dev)
sd.update(P.random_state_dict(P.lora_param_shapes(cfg, 4), bench.SEED + 3, dev))
unet = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16", lora_rank=4, lora_alpha=4)
unet.load_state_dict(sd)
sched = mrisr.DDIMScheduler(timestep_spacing="leading", steps_offset=1)
sched.set_timesteps(nsteps)
g = torch.Generator(device=dev).manual_seed(bench.SEED + 5)
if which == "cfg4":
    B, h = 16, 64
    csd = P.random_state_dict(P.controlnet_param_shapes(cfg), bench.SEED + 1, dev)
    for k in csd:  # zero-init in diffusers; small normals keep the injection path numerically alive
        if k.startswith(("controlnet_down_blocks", "controlnet_mid_block", "controlnet_cond_embedding.conv_out")):
            csd[k] = torch.randn(csd[k].shape, device=dev, generator=g) * 0.02
    cnet = mrisr.ControlNetModel(cfg, compute_dtype="bf16")
    cnet.load_state_dict(csd)
    cond = torch.randn((B, 3, 8 * h, 8 * h), device=dev, generator=g).clamp(-1, 1)
    feats = None
else:
    B, h = 32, 32
    cnet, cond = None, None
    ad = mrisr.Adapter_XL(compute_dtype="bf16")
    from mrisr.params import random_state_dict
    shapes = []
    ch = (320, 640, 1280, 1280)
    k = 0
    for i, c in enumerate(ch):
        for j in range(3):
            ic = ch[i - 1] if (i > 0 and j == 0) else c
            if ic != c:
                shapes += [(f"body.{k}.in_conv.weight", (c, ic, 3, 3), ic * 9), (f"body.{k}.in_conv.bias", (c,), ic * 9)]
            shapes += [(f"body.{k}.block1.weight", (c, c, 3, 3), c * 9), (f"body.{k}.block1.bias", (c,), c * 9),
                       (f"body.{k}.block2.weight", (c, c, 3, 3), c * 9), (f"body.{k}.block2.bias", (c,), c * 9)]
            if i > 0 and j == 0:
                shapes += [(f"body.{k}.down_opt.op.weight", (ic, ic, 3, 3), ic * 9), (f"body.{k}.down_opt.op.bias", (ic,), ic * 9)]
            k += 1
    shapes += [("conv_in.weight", (320, 192, 3, 3), 192 * 9), ("conv_in.bias", (320,), 192 * 9)]
    ad.load_state_dict(random_state_dict(shapes, bench.SEED + 9, dev))
    lr_px = torch.randn((B, 3, 256, 256), device=dev, generator=g).clamp(-1, 1)
    torch.cuda.synchronize()
    ta = time.perf_counter()
    feats = ad(lr_px)
    torch.cuda.synchronize()
    ta1 = time.perf_counter()
    feats = ad(lr_px)
    torch.cuda.synchronize()
    print(f"adapter forward (B={B}, once per slice): {1e3 * (time.perf_counter() - ta1):.2f} ms (first call {1e3 * (ta1 - ta):.0f} ms incl. autotune)")
ctx = torch.randn((B, 77, 768), device=dev, generator=g)
x_T = torch.randn((B, 4, h, h), device=dev, generator=g)
smp = mrisr.Sampler(unet, sched, cnet, kind="ddim")
lat = x_T.clone()
smp.run(lat, ctx, controlnet_cond=cond, adapter_features=feats)  # warm-up: plans workspaces, autotunes, captures
torch.cuda.synchronize()
print(f"setup + warm-up {time.perf_counter() - t0:.1f} s; UNet workspace {unet.workspace_bytes / 2**30:.2f} GiB"
      + (f", ControlNet workspace {cnet.workspace_bytes / 2**30:.2f} GiB" if cnet else ""))
reps = 2
t1 = time.perf_counter()
for _ in range(reps):
    lat.copy_(x_T)
    smp.run(lat, ctx, controlnet_cond=cond, adapter_features=feats)
torch.cuda.synchronize()
dt = (time.perf_counter() - t1) / reps
print(f"{which}: B={B} latents {h}x{h}, {nsteps}-step DDIM: {dt * 1e3:.1f} ms per batch = {B / dt:.2f} slices/s, "
      f"{dt * 1e3 / nsteps:.2f} ms per denoising step, finite={bool(torch.isfinite(lat).all())}")
if "--profile" in sys.argv:  # per-shape kernel classes of one eager denoising step (MRISR_PROF_SHAPES=1 for shapes)
    from mrisr import _lib as L
    lib = L.lib()
    smp.set_range(0, 1)
    lat.copy_(x_T)
    smp.run(lat, ctx, controlnet_cond=cond, adapter_features=feats, use_graph=False)
    torch.cuda.synchronize()
    lib.mrisr_prof_reset()
    lib.mrisr_prof_enable(1)
    smp.run(lat, ctx, controlnet_cond=cond, adapter_features=feats, use_graph=False)
    torch.cuda.synchronize()
    lib.mrisr_prof_enable(0)
    cls = bench.prof_report(lib)
    tot = sum(v["ms"] for v in cls.values())
    print(f"eager profile: total {tot:.3f} ms/step")
    for k, v in sorted(cls.items(), key=lambda kv: -kv[1]["ms"])[:45]:
        tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
        print(f"{k:76s} {v['launches']:4d} {v['ms']:8.4f} {100 * v['ms'] / tot:6.2f} {tf:7.0f}")
