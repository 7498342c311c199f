#!/usr/bin/env python3
"""Micro-benchmark of one GEMM signature with a forced tile:  python tools/rp_probe.py M N K tile [iters]
(env knobs are read by the library at load: MRISR_GEMM_FLAGS, MRISR_RP_YSPLIT, MRISR_BENCH_NOSTORE)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mri-diffusion-superresolution_amd"))
import torch  # noqa: E402

from mrisr import _lib as L  # noqa: E402

M, N, K, tile = (int(a) for a in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
torch.zeros(1).cuda()
lib = L.lib()
ms = C.c_float()
L.check(lib.mrisr_bench_gemm(M, N, K, 0, 0, 0, 0, 1, 0, 0, tile, 1, iters, C.byref(ms)))
fl = 2.0 * M * N * K
by = 2.0 * (M * K + N * K + M * N)
print(f"M={M} N={N} K={K} tile={tile} flags={os.environ.get('MRISR_GEMM_FLAGS', '0')} ysplit={os.environ.get('MRISR_RP_YSPLIT', 'auto')}: "
      f"{ms.value * 1e3:8.2f} us  {fl / ms.value / 1e9:7.1f} TF/s  {by / ms.value / 1e6:7.1f} GB/s")
