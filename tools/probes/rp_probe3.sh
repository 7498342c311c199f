#!/bin/bash
cd $GRAFT_REPO_ROOT
P="timeout -k 5 60 python tools/rp_probe.py"
{
for t in 26 61 64; do $P 8192 640 640 $t; done
for t in 14 61 64; do $P 8192 1920 640 $t; $P 8192 5120 640 $t; done
for t in 16 60 65; do $P 32768 320 320 $t; $P 32768 960 320 $t; done
} > gpurun_out/r02h_probe3.log 2>&1
grep "^M=" gpurun_out/r02h_probe3.log
