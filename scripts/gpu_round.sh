#!/bin/bash
# round 3, batch 29: determinism / potf2 stress on the final tree (after the NEG-field change of the update kernels)
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b29
mkdir -p $O
{
date
timeout 600 python scripts/stress_determinism.py
timeout 600 python scripts/stress_nan.py 3000 50000
timeout 600 python scripts/stress_nan.py 2000 30000
date
} > $O/log.txt 2>&1
cat $O/log.txt | tail -20
