#!/bin/bash
# round 2, batch 25: block-column driver with three streams of its own (assembly deferred on the main stream, forward steps on the update stream)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
B="--no-cpu-baseline --no-secondary"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline') or {}
print('$1', 'evals/s %.2f ms %.2f  update %.1f TF (%.3f)' % (d['value'], d['ms_per_step'], r.get('achieved',0), r.get('frac',0)))"; }
{
for w in c2 n8192; do
  timeout 120 python bench.py $B --workload $w --steps 10 --warmup 3 2>/dev/null | tail -1 | line "single $w"
  for sb in 0 1; do
  TGP_DIST_SELF_BROADCAST=$sb timeout 300 python bench.py $B --distributed --workload $w --steps 10 --warmup 3 2>/dev/null | tail -1 | line "dist $w [self_broadcast=$sb]"
  done
done
for sb in 0 1; do
TGP_DIST_SELF_BROADCAST=$sb timeout 300 python bench.py $B --distributed --workload n65536 --steps 2 --warmup 1 2>/dev/null | tail -1 | line "dist n65536 [self_broadcast=$sb]"
done
timeout 600 python bench.py $B --distributed --workload c4 --steps 1 --warmup 1 2>/dev/null | tail -1 | line "dist c4"
timeout 900 python -m pytest tests/test_gpu_distributed.py -m gpu -q -x 2>&1 | grep -E "passed|failed"
TGP_DIST_SELF_BROADCAST=1 timeout 900 python -m pytest tests/test_gpu_distributed.py -m gpu -q -x 2>&1 | grep -E "passed|failed"
date
} > $O/round.log 2>&1
tail -60 $O/round.log
