#!/bin/bash
# BASELINE config 5's matrix (N = 262144, fp32: a 256 GiB factor) on ONE MI355X
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
{
timeout 300 python bench.py --workload n262144f32 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-1600
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
