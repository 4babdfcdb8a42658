#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
{
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
for sp in 0 4 5 6 7; do for w in c2 n4096 n8192; do echo "## first_split=$sp $w"; TGP_HIP_OPTIONS=first_split=$sp bash scripts/bench_variants.sh "--workload $w" | tail -1 | cut -c1-70; done; done
TGP_HIP_OPTIONS=first_split=6 bash scripts/bench_variants.sh "--workload n32768 --steps 3 --warmup 1" | tail -1 | cut -c1-90
TGP_HIP_OPTIONS=first_split=0 bash scripts/bench_variants.sh "--workload n32768 --steps 3 --warmup 1" | tail -1 | cut -c1-90
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
