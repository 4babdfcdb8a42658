#!/bin/bash
# round 2, session 2, batch 19: 64x64-tile kernel with C fetched behind the first operand tile (A/B by option)
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
mkdir -p $O
B="--no-cpu-baseline --no-secondary"
{
date
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
for w in c2 n8192 n4096; do for o in "small_prefetch_c=1" "small_prefetch_c=0"; do echo "# $w $o"; TGP_HIP_OPTIONS="$o" timeout 300 python bench.py $B --workload $w --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d.get('roofline') or {}
    print(json.dumps({'evals_s':round(d['value'],3),'ms':round(d['ms_per_step'],3),'syrk_TF':round(r.get('achieved',0),2),'chol_TF':round(d.get('cholesky_tflops',0),2)}))
"; done; done
timeout 100 python scripts/stress_nan.py 3000 6000 2>&1 | tail -1
date
} > $O/round19.log 2>&1
cat $O/round19.log | cut -c1-200
