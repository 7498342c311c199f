#!/usr/bin/env python3
"""Per-shape in-model kernel timing (one eager denoising step at the bench geometry, per-launch HIP events):
    MRISR_PROF_SHAPES=1 python tools/shape_profile.py"""
import os
import sys

os.environ.setdefault("MRISR_PROF_SHAPES", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import mrisr  # noqa: E402
from mrisr import _lib as L  # noqa: E402
from mrisr import params as P  # noqa: E402

dev = torch.device("cuda", 0)
cfg = mrisr.UNetConfig()
sd = P.random_state_dict(P.unet_param_shapes(cfg), bench.SEED, dev)
sd.update(P.random_state_dict(P.lora_param_shapes(cfg, 4), bench.SEED + 3, dev))
merged = "--lora-merged" in sys.argv
fp8 = os.environ.get("MRISR_FP8", "0") == "1"  # BASELINE configs[4]: fp8 projections
BATCH = int(os.environ.get("MRISR_BATCH", "32"))
unet = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16", lora_rank=4, lora_alpha=4, lora_fused=not merged, fp8=fp8)
unet.load_state_dict(sd)
sched = mrisr.DDIMScheduler(timestep_spacing="leading", steps_offset=1)
sched.set_timesteps(50)
lr_lat, ctx, noise, _hr = bench.synthetic_batch(BATCH, dev, 0)
lat = (lr_lat + noise).contiguous()
smp = mrisr.Sampler(unet, sched, kind="ddim")
smp.set_range(0, 1)
smp.run(lat, ctx, use_graph=False)  # warm-up (plans the workspace)
torch.cuda.synchronize()
lib = L.lib()
lib.mrisr_prof_reset()
lib.mrisr_prof_enable(1)
smp.run(lat, ctx, use_graph=False)
torch.cuda.synchronize()
lib.mrisr_prof_enable(0)
cls = bench.prof_report(lib)
tot = sum(v["ms"] for v in cls.values())
print(f"total {tot:.3f} ms/step")
print(f"{'class':72s} {'n':>4s} {'ms':>8s} {'%':>6s} {'TF/s':>7s}")
for k, v in sorted(cls.items(), key=lambda kv: -kv[1]["ms"]):
    tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
    print(f"{k:72s} {v['launches']:4d} {v['ms']:8.4f} {100 * v['ms'] / tot:6.2f} {tf:7.0f}")
