#!/bin/bash
# round 2, session 2, batch 1: the trailing-update kernel with LDS-direct operand staging and the
# C read spread over the k-loop -- parity tests, K sweep of the kernel alone, bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/s2b1.log
: > $L
echo "== pytest -m gpu" >> $L; date >> $L
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 >> $L
echo "== gemm alone (K sweep, M = 16384)" >> $L; date >> $L
timeout 300 python scripts/gemm_bench.py f64 16384 >> $L 2>&1
echo "== bench lines" >> $L; date >> $L
for w in c2 n4096 n8192 n32768 n65536; do
  timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-900 >> $L
done
date >> $L
tail -60 $L
