#!/usr/bin/env python3
"""Per-shape GEMM micro-benchmark over the SD-1.5 UNet's contraction shapes (B=32, 32^2 latents):
forces each tile configuration / split-K through mrisr_bench_gemm and prints TFLOP/s (random bf16 operands)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))
from mrisr import _lib as L  # noqa: E402

lib = L.lib()
B = 32
# (name, conv, M, N, K, (B,H,W,stride,ups), c1)
SHAPES = [
    ("conv 320->320 @32", 1, B * 1024, 320, 2880, (B, 32, 32, 1, 0), 0),
    ("conv 640->320 @32 (cat)", 1, B * 1024, 320, 5760, (B, 32, 32, 1, 0), 320),
    ("conv 640->640 @32 up", 1, B * 1024, 640, 5760, (B, 16, 16, 1, 1), 0),
    ("conv 640->640 @16", 1, B * 256, 640, 5760, (B, 16, 16, 1, 0), 0),
    ("conv 1280->640 @16 (cat)", 1, B * 256, 640, 11520, (B, 16, 16, 1, 0), 640),
    ("conv 1280->1280 @16 up", 1, B * 256, 1280, 11520, (B, 8, 8, 1, 1), 0),
    ("conv 1280->1280 @8", 1, B * 64, 1280, 11520, (B, 8, 8, 1, 0), 0),
    ("conv 2560->1280 @8 (cat)", 1, B * 64, 1280, 23040, (B, 8, 8, 1, 0), 1280),
    ("conv 1280->1280 @4", 1, B * 16, 1280, 11520, (B, 4, 4, 1, 0), 0),
    ("conv 2560->1280 @4 (cat)", 1, B * 16, 1280, 23040, (B, 4, 4, 1, 0), 1280),
    ("conv 320->320 s2", 1, B * 256, 320, 2880, (B, 32, 32, 2, 0), 0),
    ("lin 320->320 (+lora) M32k", 0, B * 1024, 320, 384, None, 64),
    ("qkv 320->960 (+lora)", 0, B * 1024, 960, 384, None, 64),
    ("ff1 320->2560", 0, B * 1024, 2560, 320, None, 0),
    ("ff2 1280->320", 0, B * 1024, 320, 1280, None, 0),
    ("qkv 640->1920 (+lora) M8k", 0, B * 256, 1920, 704, None, 64),
    ("ff1 640->5120", 0, B * 256, 5120, 640, None, 0),
    ("ff2 2560->640", 0, B * 256, 640, 2560, None, 0),
    ("lin 640->640 (+lora)", 0, B * 256, 640, 704, None, 64),
    ("qkv 1280->3840 (+lora) M2k", 0, B * 64, 3840, 1344, None, 64),
    ("ff1 1280->10240", 0, B * 64, 10240, 1280, None, 0),
    ("ff2 5120->1280", 0, B * 64, 1280, 5120, None, 0),
    ("lin 1280->1280 (+lora) M2k", 0, B * 64, 1280, 1344, None, 64),
    ("shortcut 1920->640 M8k", 0, B * 256, 640, 1920, None, 640),
]
TILES = {14: "bl128x128", 25: "bl128x160", 27: "bl160x160", 28: "bl128x192", 29: "bl192x128", 30: "bl160x128", 31: "bl96x160"}


def run(shape, tile, splitk, iters=20):
    name, conv, M, N, K, geo, c1 = shape
    b, h, w, stride, ups = geo if geo else (0, 0, 0, 1, 0)
    ms = C.c_float()
    rc = lib.mrisr_bench_gemm(M, N, K, conv, b, h, w, stride, ups, c1, tile, splitk, iters, C.byref(ms))
    if rc:
        return None
    return ms.value


if __name__ == "__main__":
    only = sys.argv[1] if len(sys.argv) > 1 else None
    print(f"{'shape':34s} {'GFLOP':>7s} | " + " | ".join(f"{v:>14s}" for v in TILES.values()) + " | auto")
    for sh in SHAPES:
        if only and only not in sh[0]:
            continue
        gf = 2.0 * sh[2] * sh[3] * sh[4] / 1e9
        cells = []
        for t in TILES:
            best = None
            for s in (1, 2, 4, 8, 16):
                if s > 1 and sh[4] // 64 // s < 6:
                    break
                ms = run(sh, t, s, 10)
                if ms and (best is None or ms < best[0]):
                    best = (ms, s)
            cells.append(f"{gf / best[0]:7.0f}TF s{best[1]:<2d}" if best else "      -      ")
        ms = run(sh, 0, 0, 10)
        print(f"{sh[0]:34s} {gf:7.1f} | " + " | ".join(f"{c:>14s}" for c in cells) + f" | {gf / ms:5.0f}TF", flush=True)
