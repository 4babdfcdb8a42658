#!/bin/bash
# round 2, session 2, batch 23: the GPU suite and the smoke check on the final tree
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
mkdir -p $O
{
date
timeout 140 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke
date
} > $O/round23.log 2>&1
cat $O/round23.log
