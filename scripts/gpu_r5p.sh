#!/bin/bash
# round 5, batch p: repeated evaluations on the final tree -- bit-identical factors and log-likelihoods (the band order, the
# wall-clock-bounded pollers); stamped chain timeline.
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/${1:-r5p}
mkdir -p $O
export TMPDIR=/tmp
{
echo "== tree: $(cat $R/.tree_sha 2>/dev/null)"; date
timeout 600 python scripts/stress_determinism.py 2>&1 | tail -12
echo "== chain timeline"; date
timeout 200 python scripts/chain_timeline.py 1024 4096 > $O/chain_timeline.txt 2>&1; grep -B2 -A10 "^launch col" $O/chain_timeline.txt | head -40
date
} > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-220
