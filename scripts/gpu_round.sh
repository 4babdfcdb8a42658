#!/bin/bash
# round 2, batch 11: does a fifth ACTIVE queue cost the chain?  far updates on their own stream vs on the solve stream,
# fused vs unfused evaluations
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
for w in c2 n4096 n8192; do
for fu in "" "--unfused"; do
for opts in "inpanel_near=0" "inpanel_near=1" "inpanel_near=1,far_shares_solve=1" "inpanel_near=2,far_shares_solve=1"; do
  TGP_HIP_OPTIONS="$opts" timeout 120 python bench.py $B --workload $w $fu --steps 10 --warmup 3 2>/dev/null | tail -1 | line "$w $fu [$opts]"
done; done; done
date
} > $O/round.log 2>&1
tail -150 $O/round.log
