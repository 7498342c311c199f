#!/usr/bin/env python3
"""Experiment: does splitting the B=32 batch into two B=16 halves sampled concurrently on two HIP streams (two handles,
two step graphs) beat one B=32 graph?  (Latency-floor overlap between the halves.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))
import torch
import bench, mrisr
from mrisr import params as P
dev = torch.device("cuda", 0)
cfg = mrisr.UNetConfig()
sd = P.random_state_dict(P.unet_param_shapes(cfg), bench.SEED, dev)
sd.update(P.random_state_dict(P.lora_param_shapes(cfg, 4), bench.SEED + 3, dev))
sched = mrisr.DDIMScheduler(timestep_spacing="leading", steps_offset=1); sched.set_timesteps(50)
def mk():
    u = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16", lora_rank=4, lora_alpha=4)
    u.load_state_dict(sd)
    return u, mrisr.Sampler(u, sched, kind="ddim")
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 32
lr_lat, ctx, noise = bench.synthetic_batch(B, dev, 0)
x = (lr_lat + noise).contiguous()
models = [mk() for _ in range(nsplit)]
streams = [torch.cuda.Stream() for _ in range(nsplit)]
hb = B // nsplit
lats = [x[i * hb:(i + 1) * hb].clone() for i in range(nsplit)]
ctxs = [ctx[i * hb:(i + 1) * hb].contiguous() for i in range(nsplit)]
def run_all():
    for (u, s), st, l, c in zip(models, streams, lats, ctxs):
        with torch.cuda.stream(st):
            s.run(l, c, use_graph=True)
for _ in range(2):
    run_all(); torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for _ in range(n):
    run_all()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"nsplit={nsplit}: {B * n / el:.2f} slices/s  ({1e3 * el / n / 50:.3f} ms per denoising step of the full batch)")
