#!/usr/bin/env python3
"""bl vs halo conv kernels on the stride-1 3x3 shapes of the two high-resolution levels (micro-benchmark)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_sweep as G
B = 32
SH = [
    ("conv 320->320 @32", 1, B * 1024, 320, 2880, (B, 32, 32, 1, 0), 0),
    ("conv 640->320 @32 (cat)", 1, B * 1024, 320, 5760, (B, 32, 32, 1, 0), 320),
    ("conv 960->320 @32 (cat)", 1, B * 1024, 320, 8640, (B, 32, 32, 1, 0), 320),
    ("conv 640->640 @16", 1, B * 256, 640, 5760, (B, 16, 16, 1, 0), 0),
    ("conv 1280->640 @16 (cat)", 1, B * 256, 640, 11520, (B, 16, 16, 1, 0), 640),
    ("conv 320->640 @16", 1, B * 256, 640, 2880, (B, 16, 16, 1, 0), 0),
]
tiles = [25, 14, 28, 26, 41, 42, 43, 44, 45]
print(f"{'shape':28s} " + " ".join(f"{t:>9d}" for t in tiles))
for sh in SH:
    gf = 2.0 * sh[2] * sh[3] * sh[4] / 1e9
    cells = []
    for t in tiles:
        best = None
        for s in (1, 2):
            ms = G.run(sh, t, s, 10)
            if ms and (best is None or ms < best):
                best = ms
        cells.append(f"{gf / best:7.0f}TF" if best else "     -   ")
    print(f"{sh[0]:28s} " + " ".join(cells), flush=True)
