#!/bin/bash
# row-panel kernel, reads-first scheduling barrier on (default build) - compare with a -DRP_READS_FIRST=0 build
cd $GRAFT_REPO_ROOT
P="timeout -k 5 60 python tools/rp_probe.py"
{
for t in 61 64; do $P 8192 640 640 $t; $P 8192 1920 640 $t; $P 8192 5120 640 $t; done
for t in 60 65; do $P 32768 320 320 $t; $P 32768 960 320 $t; $P 32768 2560 320 $t; done
} 2>&1 | grep "^M="
