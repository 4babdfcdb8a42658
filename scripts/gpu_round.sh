#!/bin/bash
# round 3, batch 1: barrier-ordered potf2 -- full GPU suite, 1e5-evaluation stress per look-ahead mode, bench line, timings
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b01
mkdir -p $O
export TMPDIR=/tmp
{
date
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
date
} > $O/pytest.log 2>&1
{
date
timeout 400 python scripts/stress_nan.py 3000 100000 lookahead=0 2>&1 | grep -v "Warning\|msg +=" | tail -5 | cut -c1-200
date
timeout 400 python scripts/stress_nan.py 3000 100000 lookahead=1 2>&1 | grep -v "Warning\|msg +=" | tail -5 | cut -c1-200
date
timeout 200 python scripts/stress_nan.py 3000 30000 lookahead=1 fused_step=1 2>&1 | grep -v "Warning\|msg +=" | tail -5 | cut -c1-200
date
} > $O/stress.log 2>&1
{
timeout 300 python bench.py 2>&1 | tail -3
for n in 2048 4096 8192; do timeout 120 python bench.py --workload n$n --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c1-400; done
} > $O/bench.log 2>&1
tail -5 $O/pytest.log; cat $O/stress.log; cut -c1-600 $O/bench.log
