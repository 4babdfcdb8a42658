#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
{
for w in n512 n1024 n2048 n4096 n8192 c2; do echo "## $w"; bash scripts/bench_variants.sh "--workload $w --steps 20 --warmup 5" | tail -1 | cut -c1-90; done
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
