#!/bin/bash
# round 3, batch 26: is the submitting thread ahead of the device?  wall-clock of an evaluation next to the host time of its enqueues
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b26
mkdir -p $O
{
date
timeout 300 python scripts/host_submit.py 1024 2048 4096 8192 16384 32768
date
} > $O/log.txt 2>&1
cat $O/log.txt
