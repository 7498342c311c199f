#!/usr/bin/env python3
"""Flash attention micro-benchmark at the UNet's self/cross-attention shapes (B=32, 8 heads)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))
import torch
from mrisr import ops
B, H = 32, 8
for (N, Nk, C) in [(1024, 1024, 320), (256, 256, 640), (64, 64, 1280), (1024, 77, 320), (256, 77, 640)]:
    q = torch.randn(B, N, C, device="cuda").bfloat16(); k = torch.randn(B, Nk, C, device="cuda").bfloat16(); v = torch.randn(B, Nk, C, device="cuda").bfloat16()
    import ctypes as Ct, json
    from mrisr import _lib as L
    lib = L.lib()
    for _ in range(3): ops.attention(q, k, v, H, flash=True)
    lib.mrisr_prof_reset(); lib.mrisr_prof_enable(1)
    n = 10
    for _ in range(n): ops.attention(q, k, v, H, flash=True)
    torch.cuda.synchronize(); lib.mrisr_prof_enable(0)
    buf = Ct.create_string_buffer(1 << 16); m = lib.mrisr_prof_report(buf, len(buf))
    ms = json.loads(buf.value[:m].decode())["flash_attention"]["ms"] / n
    fl = 4.0 * B * N * Nk * C
    print(f"N={N:5d} Nk={Nk:5d} C={C:5d}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:.0f} TF", flush=True)
