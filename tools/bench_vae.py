#!/usr/bin/env python3
"""AutoencoderKL (SD-1.5 size) encode / decode timing on one MI355X: B slices of 256x256 (latents 4x32x32), bf16.
Secondary line for DESIGN.md (the VAE runs once before / once after the 50-step loop)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))
import torch
import mrisr
from mrisr import _lib as L
from mrisr import params as P

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--profile", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = mrisr.VAEConfig()
sd = P.random_state_dict(mrisr.vae_param_shapes(cfg), 20260505, dev)
vae = mrisr.AutoencoderKL(cfg, compute_dtype=a.dtype)
vae.load_state_dict(sd)
g = torch.Generator(device=dev).manual_seed(1)
x = (torch.rand((a.batch, 3, a.size, a.size), generator=g, device=dev) * 2 - 1)
z = torch.randn((a.batch, 4, a.size // 8, a.size // 8), generator=g, device=dev)
def timed(fn):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.iters * 1e3
enc_ms = timed(lambda: vae.encode(x))
dec_ms = timed(lambda: vae.decode(z))
out = {"workload": f"AutoencoderKL SD-1.5 size, B={a.batch}, {a.size}x{a.size}, {a.dtype}", "encode_ms": round(enc_ms, 3), "decode_ms": round(dec_ms, 3),
       "encode_slices_per_s": round(a.batch / enc_ms * 1e3, 1), "decode_slices_per_s": round(a.batch / dec_ms * 1e3, 1),
       "finite": bool(torch.isfinite(vae.decode(z).sample).all())}
if a.profile:
    import ctypes as C
    lib = L.lib()
    for name, fn in (("encode", lambda: vae.encode(x)), ("decode", lambda: vae.decode(z))):
        lib.mrisr_prof_reset(); lib.mrisr_prof_enable(1)
        fn(); torch.cuda.synchronize()
        lib.mrisr_prof_enable(0)
        buf = C.create_string_buffer(1 << 20)
        n = lib.mrisr_prof_report(buf, len(buf))
        cls = json.loads(buf.value[:n].decode()) if n > 0 else {}
        fl = sum(v["flops"] for v in cls.values()); ms = sum(v["ms"] for v in cls.values())
        out[name + "_profile"] = {"sum_ms": round(ms, 3), "gflop": round(fl / 1e9, 1), "TFLOPs": round(fl / ms / 1e9, 1) if ms else 0,
                                  "top": {k: round(v["ms"], 3) for k, v in sorted(cls.items(), key=lambda kv: -kv[1]["ms"])[:8]}}
print(json.dumps(out))
