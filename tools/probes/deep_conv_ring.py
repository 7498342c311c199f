"""Would the M = 512 convs (4 x 4 maps, K = 11,520 / 23,040: 22-45 K tiles per split at two stages, latency bound) run faster as a plain GEMM
over an explicit im2col (11.8 / 23.6 MB) with the 3- / 4-stage ring variants of the tiled kernel (which cannot gather: an all-padding DMA
piece retires out of order and breaks their counted waits)?  Isolated launches, random operands: mrisr_bench_gemm (includes the split-K reduce)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "mri-diffusion-superresolution_amd"))
import torch  # noqa: F401  (device init)
from mrisr import _lib as L

lib = L.lib()
lib.mrisr_bench_gemm.argtypes = [C.c_int] * 13 + [C.POINTER(C.c_float)]


def run(M, N, K, conv, geo, c1, tile, split, iters=20):
    b, h, w, stride, ups = geo if geo else (0, 0, 0, 1, 0)
    ms = C.c_float()
    rc = lib.mrisr_bench_gemm(M, N, K, conv, b, h, w, stride, ups, c1, tile, split, iters, C.byref(ms))
    return None if rc else ms.value * 1e3


names = {26: "64x160", 39: "64x160 x3", 25: "128x160", 37: "128x160 x4", 14: "128x128", 38: "128x128 x4", 36: "128x128 x3", 17: "64x64", 32: "64x64 x4", 33: "64x64 x3", 34: "64x128 x3", 35: "128x64 x3"}
for K, c1 in ((11520, 0), (23040, 1280)):
    for M, hw in ((512, 4),):
        print(f"M = {M}, N = 1280, K = {K}")
        for s in (2, 4, 8, 16):
            t = run(M, 1280, K, 1, (32, hw, hw, 1, 0), c1, 26, s)
            t2 = run(M, 1280, K, 1, (32, hw, hw, 1, 0), c1, 43, s)
            print(f"   implicit conv  64x160 (2 stages) split {s:2d}: {t and round(t, 1)} us    halo 64x160: {t2 and round(t2, 1)} us")
        for tile in (26, 39, 25, 37, 14, 36, 38, 17, 33, 32):
            cells = []
            for s in (2, 4, 8, 16):
                t = run(M, 1280, K, 0, None, 0, tile, s)
                cells.append(f"s{s}: {t:6.1f}" if t else f"s{s}:    -  ")
            print(f"   plain GEMM {names[tile]:12s} " + "  ".join(cells))
