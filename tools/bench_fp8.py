#!/usr/bin/env python3
"""BASELINE configs[4] on ONE GPU, inference leg: 256x256 slices -> 4x32x32 latents, SD-1.5-size UNet + rank-4 LoRA, 50-step DDIM,
bs=64 per GPU, the K = 320 / 640 projections of the transformer blocks (incl. the LoRA targets to_q/k/v, to_out) with OCP e4m3
operands on the fp8 MFMA (`fp8=True`).  Reports slices/s for the fp8 and the bf16 engine on the same inputs, the fp8 kernels'
achieved rate (per-launch HIP events) against the 5 PF fp8 peak and against the 2.5 PF rate of the un-scaled fp8 MFMA they use,
and the fidelity of both engines against the f32 engine on two slices (latents rel-L2, stub-decoded PSNR).
usage: bench_fp8.py [batch] [ddim_steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import mrisr  # noqa: E402
from mrisr import _lib as L  # noqa: E402
from mrisr import params as P  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda", 0)
cfg = mrisr.UNetConfig()
sd = P.random_state_dict(P.unet_param_shapes(cfg), bench.SEED, dev)
sd.update(P.random_state_dict(P.lora_param_shapes(cfg, 4), bench.SEED + 3, dev))
sched = mrisr.DDIMScheduler(timestep_spacing="leading", steps_offset=1)
sched.set_timesteps(nsteps)
lr_lat, ctx, noise, hr = bench.synthetic_batch(B, dev, 0)
a_T = float(sched.alphas_cumprod[int(sched.timesteps[0])])
x_T = (lr_lat + (1 - a_T) ** 0.5 * noise).contiguous()
lib = L.lib()
out = {"config": f"configs[4] inference leg: 256^2, SD-1.5 UNet + rank-4 LoRA, {nsteps}-step DDIM, bs={B}, 1 GPU", "engines": {}}
finals = {}
# "fp8": the K = 320 projections (round 2); "fp8_attn": those + every attention's Q K^T / P V in e4m3 (round 3: the whole of configs[4]'s inference side)
for name, kw in (("bf16", {}), ("fp8", {"fp8": True}), ("fp8_attn", {"fp8": True, "fp8_attention": True})):
    unet = mrisr.UNet2DConditionModel(cfg, compute_dtype="bf16", lora_rank=4, lora_alpha=4, **kw)
    unet.load_state_dict(sd)
    smp = mrisr.Sampler(unet, sched, kind="ddim")
    lat = x_T.clone()
    smp.run(lat, ctx)  # warm-up: workspace, tile choice, graph capture
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        lat.copy_(x_T)
        smp.run(lat, ctx)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    finals[name] = lat[:2].float().cpu()
    # per-class profile of two eager denoising steps
    s2 = mrisr.Sampler(unet, sched, kind="ddim")
    s2.set_range(0, 2)
    l2 = x_T.clone()
    lib.mrisr_prof_reset(); lib.mrisr_prof_enable(1)
    s2.run(l2, ctx, use_graph=False)
    torch.cuda.synchronize(); lib.mrisr_prof_enable(0)
    cls = bench.prof_report(lib)
    lib.mrisr_prof_reset()
    rp = {k: v for k, v in cls.items() if "_rp" in k}
    fl, ms = sum(v["flops"] for v in rp.values()), sum(v["ms"] for v in rp.values())
    att = {k: v for k, v in cls.items() if k.startswith(("flash_attention", "attention_quant"))}
    afl, ams = sum(v["flops"] for v in att.values()), sum(v["ms"] for v in att.values())
    e = {"slices_per_s": B / dt, "ms_per_denoising_step": dt * 1e3 / nsteps, "finite": bool(torch.isfinite(lat).all()),
         "row_panel_kernels": {"classes": sorted(rp), "ms_per_step": ms / 2, "achieved_tflops": fl / (ms * 1e-3) / 1e12 if ms else 0.0},
         "attention_kernels": {"classes": {k: round(v["ms"] / 2, 4) for k, v in att.items()}, "ms_per_step": ams / 2,
                               "achieved_tflops": afl / (ams * 1e-3) / 1e12 if ams else 0.0}}
    if name == "fp8_attn":
        e["attention_kernels"]["frac_of_fp8_peak_5pf"] = e["attention_kernels"]["achieved_tflops"] / 5000.0
    if name.startswith("fp8"):
        t = e["row_panel_kernels"]["achieved_tflops"]
        e["row_panel_kernels"]["frac_of_fp8_peak_5pf"] = t / 5000.0
        e["row_panel_kernels"]["frac_of_unscaled_fp8_mfma_rate_2p5pf"] = t / 2500.0
    out["engines"][name] = e
    del unet, smp, s2
# fidelity on two slices against the f32 engine
u32 = mrisr.UNet2DConditionModel(cfg, compute_dtype="f32", lora_rank=4, lora_alpha=4)
u32.load_state_dict(sd)
l32 = x_T[:2].clone().contiguous()
mrisr.Sampler(u32, sched, kind="ddim").run(l32, ctx[:2], use_graph=False)
torch.cuda.synchronize()
ref = l32.float().cpu()
hr01 = (hr[:2].float().cpu() / 2 + 0.5).clamp(0, 1)
ref_img = bench.stub_decode(ref)
for name, z in finals.items():
    img = bench.stub_decode(z)
    out["engines"][name]["fidelity_vs_f32_engine"] = {"rel_l2_latents": float((z - ref).norm() / ref.norm()), "psnr_vs_f32_db": bench.psnr(img, ref_img),
                                                      "psnr_vs_hr_db": bench.psnr(img, hr01)}
out["f32_psnr_vs_hr_db"] = bench.psnr(ref_img, hr01)
for name in ("fp8", "fp8_attn"):
    out[f"psnr_diff_{name}_vs_bf16_db"] = abs(out["engines"][name]["fidelity_vs_f32_engine"]["psnr_vs_hr_db"] - out["engines"]["bf16"]["fidelity_vs_f32_engine"]["psnr_vs_hr_db"])
    out[f"speedup_{name}_over_bf16"] = out["engines"][name]["slices_per_s"] / out["engines"]["bf16"]["slices_per_s"]
print(json.dumps(out))
