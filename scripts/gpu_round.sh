#!/bin/bash
# round 3, batch 24: BASELINE config 5's matrix at full size (N = 262 144, fp32, 256 GiB) through bench.py
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b24
mkdir -p $O
export TMPDIR=/tmp
{
date
timeout 900 python bench.py --workload n262144f32 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | tee $O/line.json | cut -c1-1200
date
} > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-1300
