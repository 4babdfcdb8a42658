#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
{
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
for w in c2 n4096 n8192; do bash scripts/bench_variants.sh "--workload $w" | tail -1 | cut -c1-200; done
for t in 1100 2200 4400; do echo "## first_small_tiles=$t n32768"; TGP_HIP_OPTIONS=first_small_tiles=$t bash scripts/bench_variants.sh "--workload n32768 --steps 3 --warmup 1" | tail -1 | cut -c1-90; done
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
