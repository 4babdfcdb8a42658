#!/bin/bash
# round 2, batch 12: panels 2 nb_outer wide while the trailing update is the bound (nb_wide_rows)
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
echo "== parity with wide panels"; date
TGP_HIP_OPTIONS="nb_wide_rows=3072" timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gp.py -m gpu -q -x 2>&1 | tail -3
for r in 0 15000 13000 11000 9000 7000 4096; do
  TGP_HIP_OPTIONS="nb_wide_rows=$r" timeout 120 python bench.py $B --steps 10 --warmup 3 2>/dev/null | tail -1 | line "c2 [nb_wide_rows=$r]"
done
for r in 0 20000 12000 4096; do
  TGP_HIP_OPTIONS="nb_wide_rows=$r" timeout 120 python bench.py $B --workload n32768 --steps 4 --warmup 1 2>/dev/null | tail -1 | line "n32768 [nb_wide_rows=$r]"
done
for r in 0 30000 12000; do
  TGP_HIP_OPTIONS="nb_wide_rows=$r" timeout 200 python bench.py $B --workload n65536 --steps 2 --warmup 1 2>/dev/null | tail -1 | line "n65536 [nb_wide_rows=$r]"
done
date
} > $O/round.log 2>&1
tail -150 $O/round.log
