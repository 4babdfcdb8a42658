#!/bin/bash
# full GPU suite + default bench after the dry-run instrumentation of the schedule
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
{
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-260
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
