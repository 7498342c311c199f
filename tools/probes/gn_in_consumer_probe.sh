#!/bin/bash
# VERDICT r1 item 6, measured: the cost of applying GroupNorm + SiLU in the CONSUMER conv (one LDS -> VALU -> LDS pass over the staged patch per 64-channel
# chunk; probe flag 1024, results wrong) on the convs that follow a GroupNorm, against the one-pass GroupNorm kernel it would replace
cd $GRAFT_REPO_ROOT
P="timeout -k 5 60 python tools/conv_probe.py"
{
echo "# level-0 conv 320->320 (M=32768), halo 128x160: plain | with the in-LDS apply pass"
$P 32 32 320 320 41; MRISR_GEMM_FLAGS=1024 $P 32 32 320 320 41
echo "# level-0 up-path conv 640->320 / 960->320"
$P 32 32 640 320 41 1 320; MRISR_GEMM_FLAGS=1024 $P 32 32 640 320 41 1 320
echo "# level-1 conv 640->640 (M=8192), halo 64x160"
$P 32 16 640 640 43; MRISR_GEMM_FLAGS=1024 $P 32 16 640 640 43
echo "# level-2 conv 1280->1280 (M=2048), halo 64x160 split 2"
$P 32 8 1280 1280 43 2; MRISR_GEMM_FLAGS=1024 $P 32 8 1280 1280 43 2
} 2>&1 | grep -E "^conv|^#"
