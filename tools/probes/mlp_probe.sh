#!/bin/bash
# fused feed-forward kernel: where the time goes (probe build; MlpDev.dbg bits: 1 no DMA, 2 no GEGLU, 4 no FF1 MFMAs, 8 no FF2 MFMAs)
for d in 0 1 2 4 8 12 14 15; do MRISR_MLP_DBG=$d python tools/mlp_probe.py 32768 2>&1 | grep -v amdgpu.ids; done
MRISR_MLP_DBG=0 python tools/mlp_probe.py 16384 2>&1 | grep -v amdgpu.ids
MRISR_MLP_DBG=0 python tools/mlp_probe.py 65536 2>&1 | grep -v amdgpu.ids
