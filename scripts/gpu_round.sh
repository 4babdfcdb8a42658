#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
{
timeout 200 python scripts/dbg_dist_host.py 16384 2>&1 | grep -v Warn | tail -9
TGP_DIST_SELF_BROADCAST=1 timeout 200 python scripts/dbg_dist_host.py 16384 2>&1 | grep -v Warn | tail -9
} > gpurun_out/round.log 2>&1
cat gpurun_out/round.log
