#!/bin/bash
# final check of the committed state: full GPU suite, smoke, default bench
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
{
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default.json; cat gpurun_out/bench_default.json
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
