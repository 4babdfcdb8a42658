#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
{
timeout 600 python scripts/time_paths.py 16384 4096
timeout 300 python scripts/time_paths.py 4096 1024
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
