#!/bin/bash
# round 5, batch d: the new distributed tests first (multi-RHS backward solve, gradient on the block-column path), then the
# WHOLE GPU suite in driver order with -x -q on this tree.
R=$GRAFT_REPO_ROOT
cd $R
TAG=${1:-r5d}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
{
echo "== tree: $(cat $R/.tree_sha 2>/dev/null)"; date
echo "== new distributed tests"
timeout 900 python -m pytest tests/test_gpu_5_distributed.py tests/test_gpu_6_multirank_one_gpu.py -x -q -m gpu -p no:cacheprovider --durations=6 -k "gradient or resident or peers" 2>&1 | tail -40
echo "== pytest tests -x -q -m gpu (driver order)"; date
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=8 2>&1 | tail -25
echo "== smoke"; date
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
date
} > $O/log.txt 2>&1
tail -90 $O/log.txt | cut -c1-500
