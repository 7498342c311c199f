#!/usr/bin/env python3
"""2-stage tiles vs the counted-ring tiles (3 / 4 LDS stages, raw barriers) on the latency- / fill-bound GEMM shapes of the step
(B = 32, 32^2 latents): isolated launches on random operands through mrisr_bench_gemm, us per launch and TFLOP/s."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))
from mrisr import _lib as L  # noqa: E402

lib = L.lib()
SHAPES = [(2048, 1280, 1280), (8192, 640, 640), (8192, 640, 2560), (2048, 1280, 5120), (2048, 10240, 1280), (2048, 3840, 1280),
          (8192, 5120, 640), (8192, 1920, 640), (32768, 320, 1280), (512, 1280, 1280), (2048, 1280, 2560), (32768, 320, 640),
          (8192, 640, 1280), (8192, 640, 1920), (4096, 4096, 4096)]
TILES = {17: "64x64", 26: "64x160", 14: "128x128", 25: "128x160", 33: "64x64d3", 32: "64x64d4", 34: "64x128d3", 35: "128x64d3",
         39: "64x160d3", 36: "128x128d3", 38: "128x128d4", 37: "128x160d4"}
print(f"{'M,N,K':22s} | " + " | ".join(f"{v:>10s}" for v in TILES.values()))
for M, N, K in SHAPES:
    gf = 2.0 * M * N * K / 1e9
    cells = []
    for t in TILES:
        ms = C.c_float()
        rc = lib.mrisr_bench_gemm(M, N, K, 0, 0, 0, 0, 1, 0, 0, t, 1, 20, C.byref(ms))
        cells.append(f"{ms.value * 1e3:5.1f}/{gf / ms.value:4.0f}" if rc == 0 and ms.value > 0 else "    -     ")
    print(f"{M:6d},{N:6d},{K:6d}   | " + " | ".join(f"{c:>10s}" for c in cells), flush=True)
