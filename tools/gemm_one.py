#!/usr/bin/env python3
"""Run ONE shape of tools/gemm_sweep.py with a forced tile/split for profiling:  gemm_one.py <shape-substring> <tile> <splitk> [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_sweep as G
sub, tile, split = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
for sh in G.SHAPES:
    if sub in sh[0]:
        ms = G.run(sh, tile, split, iters)
        gf = 2.0 * sh[2] * sh[3] * sh[4] / 1e9
        print(f"{sh[0]} tile={tile} split={split}: {ms:.4f} ms  {gf / ms:.0f} TFLOP/s")
        break
