#!/bin/bash
# round 2, batch 29: one-pass gradient sums for leaf / amp*leaf programs (kgrad_fast_kernel)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
{
timeout 900 python -m pytest tests -m gpu -q -k "grad" 2>&1 | grep -E "passed|failed|rror" | head
timeout 200 python scripts/time_paths.py 16384 4096 | grep grad
timeout 200 python scripts/time_paths.py 4096 1024 | grep grad
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_grad -o bench -- python $R/scripts/time_paths.py 16384 4096 > /dev/null 2>&1
cd $R
python scripts/prof_top.py $(ls $O/prof_grad/*.db | head -1) 40 | grep -E "kgrad|sum_partials|noise_grad"
date
} > $O/round.log 2>&1
tail -30 $O/round.log
