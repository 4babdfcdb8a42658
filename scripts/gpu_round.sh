#!/bin/bash
# round 2, batch 27: the big update of a step launched after the first blocks of the next chain (head_blocks / head_rows)
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
for opts in "head_blocks=0" "head_blocks=1,head_rows=100000" "head_blocks=1,head_rows=11000" "head_blocks=1,head_rows=8000" "head_blocks=2,head_rows=100000" "head_blocks=2,head_rows=11000" "head_blocks=2,head_rows=8000" "head_blocks=3,head_rows=9000" "head_blocks=4,head_rows=7000"; do
  TGP_HIP_OPTIONS="$opts" timeout 120 python bench.py $B --steps 10 --warmup 3 2>/dev/null | tail -1 | line "c2 [$opts]"
done
for w in n8192 n32768; do for opts in "head_blocks=0" "head_blocks=1,head_rows=11000" "head_blocks=2,head_rows=11000"; do
  TGP_HIP_OPTIONS="$opts" timeout 120 python bench.py $B --workload $w --steps 5 --warmup 2 2>/dev/null | tail -1 | line "$w [$opts]"
done; done
TGP_HIP_OPTIONS="head_blocks=2,head_rows=11000" timeout 900 python -m pytest tests/test_gpu_gp.py tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | grep -E "passed|failed"
date
} > $O/round.log 2>&1
tail -60 $O/round.log
