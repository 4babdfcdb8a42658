#!/bin/bash
# round 2, session 2, batch 25: staged potf2_sync variant under the stress that exposed the function-form failures (12 in 25000)
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
mkdir -p $O
{
date
timeout 40 python scripts/stress_nan.py 3000 18000 potf2_sync=1 lookahead=0 2>&1 | grep -v "Warning\|msg +=" | tail -3 | cut -c1-160
date
} > $O/round25.log 2>&1
cat $O/round25.log
