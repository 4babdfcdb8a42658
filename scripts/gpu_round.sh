#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
{
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
for w in c2 n4096 n8192 n2048; do echo "## $w"; bash scripts/bench_variants.sh "--workload $w" | tail -1 | cut -c1-70; done
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
