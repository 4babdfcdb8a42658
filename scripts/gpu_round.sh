#!/bin/bash
# round 2, session 2, batch 22: config 4 (N = 131072) through the block-column driver at world size 1 with the new update kernel
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
mkdir -p $O
{
date
timeout 400 python bench.py --distributed --workload c4 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>$O/dist_c4.err | tail -1 > $O/dist_c4.json; cut -c1-1500 $O/dist_c4.json
date
} > $O/round22.log 2>&1
cat $O/round22.log
