#!/bin/bash
# round 3, batch 12: config 5 at full size (N = 262144 fp32, M = 4096) against the fp64 LAPACK values
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b12
mkdir -p $O
export TMPDIR=/tmp
{
date
timeout 900 python -m pytest tests/test_gpu_gp.py -m gpu -q -k "n262144 or n32768 or m4096" --durations=3 2>&1 | tail -8
date
} > $O/log.txt 2>&1
cat $O/log.txt
