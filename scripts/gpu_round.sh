#!/bin/bash
# round 3, batch 35: the round-end sequence on the final tree (streaming solves with two workgroups per block)
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b35
mkdir -p $O
export TMPDIR=/tmp
{
echo "== pytest -m gpu"; date
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
echo "== smoke"; date
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench default"; date
timeout 600 python bench.py 2>/dev/null | tail -1 | tee $O/bench_default.json | cut -c1-400
date
} > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-500
