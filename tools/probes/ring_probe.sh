#!/bin/bash
cd $GRAFT_REPO_ROOT
P="timeout -k 5 60 python tools/rp_probe.py"
{
for t in 17 33 32; do $P 2048 1280 1280 $t; done
for t in 18 34 16 35 14 36; do $P 2048 1280 1280 $t; done
for t in 14 36 18 34; do $P 2048 3840 1280 $t; $P 2048 10240 1280 $t; done
for t in 17 32 26; do $P 2048 1280 5120 $t; $P 8192 640 640 $t; $P 8192 640 2560 $t; done
for t in 32 34 36; do $P 8192 640 2560 $t; $P 32768 320 1280 $t; done
$P 32768 320 1280 25
} > gpurun_out/r02i_ring_probe.log 2>&1
grep "^M=\|rror" gpurun_out/r02i_ring_probe.log
