#!/bin/bash
# round 3, batch 22: forward substitution underneath the factorisation (2 x N/128 small launches) or as one streaming launch behind it
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b22
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary"
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['config']['workload'][:14], round(d['value'],3), round(d['ms_per_step'],3), 'update TF', round(r['achieved'],2), 'frac', round(r['frac'],4))"; }
{
date
TGP_HIP_OPTIONS=fused_solve=0 timeout 600 python -m pytest tests/test_gpu_gp.py -m gpu -x -q -k "logp or log_prob or config2" 2>&1 | tail -2
for rep in 1 2; do
for fs in 1 0; do
echo "== fused_solve=$fs"
for wl in n2048 n4096 n8192 c2 n32768; do
TGP_HIP_OPTIONS=fused_solve=$fs timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 $B 2>/dev/null | tail -1 | line
done
done
done
for fs in 1 0; do
TGP_HIP_OPTIONS=fused_solve=$fs timeout 300 python bench.py --workload n65536 --steps 3 --warmup 1 $B 2>/dev/null | tail -1 | line
done
date
} > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-200
