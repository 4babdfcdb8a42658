#!/bin/bash
# BASELINE config 4's matrix (N = 131072, fp64: a 137 GB factor) on ONE MI355X
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
{
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "total\|used" | head -4
timeout 240 python bench.py --workload n131072 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-1500
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
