#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
{
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
for w in c2 n4096 n8192 n32768; do bash scripts/bench_variants.sh "--workload $w" | tail -1; done
bash scripts/bench_variants.sh "--workload n65536 --steps 2 --warmup 1" | tail -1
bash scripts/bench_variants.sh "--workload c3 --steps 2 --warmup 1" | tail -1
bash scripts/bench_variants.sh "--workload n16384f32" | tail -1
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
