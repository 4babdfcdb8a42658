#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
{
timeout 600 python scripts/stress_determinism.py
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
