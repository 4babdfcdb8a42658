#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
{
timeout 60 ./scripts/probe_potf2 | tail -9
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gp.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert|Mismatch" | tail -5
for w in c2 n4096; do bash scripts/bench_variants.sh "--workload $w" | tail -1 | cut -c1-70; done
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
