#!/bin/bash
# round 5, batch g: block-column timings again (the backward update as ONE product), then the WHOLE GPU suite in driver order,
# smoke and the default bench line on this tree.
R=$GRAFT_REPO_ROOT
cd $R
TAG=${1:-r5g}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
{
echo "== tree: $(cat $R/.tree_sha 2>/dev/null)"; date
echo "== timings, N = 16384"
timeout 600 python scripts/dist_timing.py 16384 1024 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -12
echo "== timings, N = 65536"; date
timeout 900 python scripts/dist_timing.py 65536 1024 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -12
echo "== pytest tests -x -q -m gpu (driver order)"; date
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --durations=8 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -22
echo "== smoke"; date
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== bench default (the driver's line)"; date
timeout 900 python bench.py 2>/dev/null | tail -1 | tee $O/bench_default.json | cut -c1-1200
date
} > $O/log.txt 2>&1
tail -90 $O/log.txt | cut -c1-400
