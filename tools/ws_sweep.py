#!/usr/bin/env python3
"""Weight-stationary vs tiled kernels on the short-K projections (micro-benchmark; K = 320 / 640)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_sweep as G
B = 32
SH = [("lin 320->320 M32k", 0, B * 1024, 320, 320, None, 0), ("qkv 320->960 M32k", 0, B * 1024, 960, 320, None, 0),
      ("ff1 320->2560 M32k", 0, B * 1024, 2560, 320, None, 0), ("lin 640->640 M8k", 0, B * 256, 640, 640, None, 0),
      ("qkv 640->1920 M8k", 0, B * 256, 1920, 640, None, 0), ("ff1 640->5120 M8k", 0, B * 256, 5120, 640, None, 0),
      ("lin 1280->1280 M2k", 0, B * 64, 1280, 1280, None, 0), ("ff2 1280->320 M32k", 0, B * 1024, 320, 1280, None, 0),
      ("ff2 2560->640 M8k", 0, B * 256, 640, 2560, None, 0), ("ff1 1280->10240 M2k", 0, B * 64, 10240, 1280, None, 0)]
tiles = [25, 14, 26, 16, 17, 18, 50, 51, 52]
print(f"{'shape':24s} " + " ".join(f"{t:>9d}" for t in tiles))
for sh in SH:
    gf = 2.0 * sh[2] * sh[3] * sh[4] / 1e9
    cells = []
    for t in tiles:
        ms = G.run(sh, t, 1, 20)
        cells.append(f"{ms * 1e3:6.1f}us" if ms else "     -   ")
    print(f"{sh[0]:24s} " + " ".join(f"{c:>9s}" for c in cells), flush=True)
