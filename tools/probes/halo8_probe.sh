#!/bin/bash
# level-0 / level-1 convs: the 128-row halo kernel (41, two workgroups per CU) against the 256-row eight-wave kernel with 2 / 3 / 4 weight stages (46 / 47 / 48)
cd $GRAFT_REPO_ROOT
P="timeout -k 5 60 python tools/conv_probe.py"
{
for t in 41 46 47 48; do $P 32 32 320 320 $t; done
for t in 41 46 47 48; do $P 32 32 640 320 $t 1 320; done
for t in 41 46 47 48; do $P 32 32 960 320 $t 1 320; done
for t in 41 43; do $P 32 16 640 640 $t 1; $P 32 16 640 640 $t 2; done
for t in 46 47 48; do $P 32 16 640 640 $t 1; $P 32 16 640 640 $t 2; done
for t in 41 48; do $P 32 16 1280 640 $t 2 640; $P 32 16 1920 640 $t 2 640;  done
} 2>&1 | grep "^conv"
