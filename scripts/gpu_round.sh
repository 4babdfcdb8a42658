#!/bin/bash
# round 2, batch 21: defaults after "solve on the update stream"; block-column driver with the same; full GPU suite
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
B="--no-cpu-baseline --no-secondary"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline') or {}
print('$1', 'evals/s %.2f ms %.2f  update %.1f TF (%.3f)  potrf %.2f ms' % (d['value'], d['ms_per_step'], r.get('achieved',0), r.get('frac',0), (d.get('stage_ms') or {}).get('potrf',0)))"; }
{
echo "== pytest -m gpu"; date
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|rror" $O/pytest_gpu.log | head -5
for w in c2 c1 n2048 n4096 n8192; do
  timeout 120 python bench.py $B --workload $w --steps 10 --warmup 3 2>/dev/null | tail -1 | line "$w"
done
for w in c2 n8192; do for v in 0 1; do
  TGP_HIP_OPTIONS="solve_on_update=$v" timeout 300 python bench.py $B --distributed --workload $w --steps 10 --warmup 3 2>/dev/null | tail -1 | line "dist $w [solve_on_update=$v]"
done; done
echo "== determinism stress"; date
timeout 400 python scripts/stress_determinism.py 2>&1 | tail -6
date
} > $O/round.log 2>&1
tail -60 $O/round.log
