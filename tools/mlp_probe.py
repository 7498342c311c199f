#!/usr/bin/env python3
"""Time the fused feed-forward kernel alone (M rows of width 320, hidden 1280): python tools/mlp_probe.py [M]
MRISR_MLP_DBG=<bits> selects the probe build (gemm.hip MlpDev.dbg)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))
import torch  # noqa: E402

from mrisr import _lib as L  # noqa: E402
import ctypes as C  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
H = 1280
ms = C.c_float(0)
L.check(L.lib().mrisr_bench_mlp(M, H, 50, C.byref(ms)))
fl = 2.0 * M * (2 * H * 320 + H * 320)
print(f"M={M} H={H} dbg={os.environ.get('MRISR_MLP_DBG', '0')}: {ms.value * 1e3:8.2f} us  {fl / ms.value / 1e9:7.1f} TF/s")
