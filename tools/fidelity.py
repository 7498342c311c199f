#!/usr/bin/env python3
"""bf16 "fast" mode vs f32 "parity" mode over the full 50-step DDIM sample at SD-1.5 size (same seeds/weights):
PSNR / relative error of the final latents, and of the [0,1]-mapped stub-decoded slices (north_star: within 0.05 dB
of the reference path is about PSNR-vs-ground-truth; here we report the direct bf16-vs-f32 distance)."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import mrisr  # noqa: E402
from mrisr import params as P  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
cfg = mrisr.UNetConfig()
sd = P.random_state_dict(P.unet_param_shapes(cfg), bench.SEED, dev)
sd.update(P.random_state_dict(P.lora_param_shapes(cfg, 4), bench.SEED + 3, dev))
lr_lat, ctx, noise = bench.synthetic_batch(B, dev, 0)
sched = mrisr.DDIMScheduler(timestep_spacing="leading", steps_offset=1)
sched.set_timesteps(50)
a_T = float(sched.alphas_cumprod[int(sched.timesteps[0])])
x_T = (lr_lat + (1 - a_T) ** 0.5 * noise).contiguous()
outs = {}
for dt in ("f32", "bf16"):
    net = mrisr.UNet2DConditionModel(cfg, compute_dtype=dt, lora_rank=4, lora_alpha=4)
    net.load_state_dict(sd)
    lat = x_T.clone()
    mrisr.Sampler(net, sched, kind="ddim").run(lat, ctx)
    torch.cuda.synchronize()
    outs[dt] = lat.double().cpu()
    del net
ref, got = outs["f32"], outs["bf16"]
mse = float(((ref - got) ** 2).mean())
rng = float(ref.max() - ref.min())
print(f"latents: rel-l2 {float((ref - got).norm() / ref.norm()):.3e}  PSNR(bf16 vs f32, range {rng:.3f}) = {10 * math.log10(rng * rng / mse):.2f} dB")
img = lambda z: ((z / 0.18215).mean(1, keepdim=True) / 2 + 0.5).clamp(0, 1)
mse_i = float(((img(ref) - img(got)) ** 2).mean())
print(f"stub-decoded slices in [0,1]: PSNR(bf16 vs f32) = {10 * math.log10(1.0 / max(mse_i, 1e-20)):.2f} dB; lr anchor PSNR vs f32 result = "
      f"{10 * math.log10(1.0 / float(((img(ref) - img(lr_lat.double().cpu())) ** 2).mean())):.2f} dB vs bf16 result = "
      f"{10 * math.log10(1.0 / float(((img(got) - img(lr_lat.double().cpu())) ** 2).mean())):.2f} dB")

# ---- image domain: decode both results with the device VAE (SD-1.5 size, f32) and score them against the same ground truth
# with the device metrics (the reference's evaluator: PSNR / SSIM / HFEN / NMSE on 8-bit grayscale) ----
vae = mrisr.AutoencoderKL(mrisr.VAEConfig(), compute_dtype="f32")
vae.load_state_dict(P.random_state_dict(mrisr.vae_param_shapes(mrisr.VAEConfig()), bench.SEED + 5, dev))
ev = mrisr.MRIEvaluator()
to_u8 = lambda im: ((im.mean(1, keepdim=True) / 2 + 0.5).clamp(0, 1) * 255).round() / 255  # noqa: E731  gray, 8-bit like the PNGs
dec = {k: to_u8(vae.decode((v.float().to(dev)) / 0.18215).sample) for k, v in outs.items()}
gt = to_u8(vae.decode(lr_lat.float() / 0.18215).sample)  # a fixed target: the decoded low-field anchor
m32, m16 = ev.evaluate(dec["f32"], gt), ev.evaluate(dec["bf16"], gt)
d = ev.evaluate(dec["bf16"], dec["f32"])
print("decoded 256x256 slices (8-bit gray), mean over the batch:")
for k in ("PSNR", "SSIM", "HFEN", "NMSE"):
    print(f"  {k}: f32 vs target {float(m32[k].mean()):.4f} | bf16 vs target {float(m16[k].mean()):.4f} | "
          f"|difference| max over slices {float((m32[k] - m16[k]).abs().max()):.4f} | bf16 vs f32 directly {float(d[k].mean()):.4f}")
