#!/bin/bash
# round 2, session 2, batch 20: tuning knobs re-checked with the new trailing-update kernel (options only)
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
mkdir -p $O
B="--no-cpu-baseline --no-secondary"
run() { echo "# $1 $2"; TGP_HIP_OPTIONS="$2" timeout 300 python bench.py $B --workload $1 --steps $3 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d.get('roofline') or {}
    print(json.dumps({'evals_s':round(d['value'],3),'ms':round(d['ms_per_step'],3),'syrk_TF':round(r.get('achieved',0),2),'chol_TF':round(d.get('cholesky_tflops',0),2)}))
"; }
{
date
run c2 "" 10
run c2 "first_small_tiles=600" 10
run c2 "first_small_tiles=2000" 10
run c2 "first_split=4" 10
run c2 "first_split=6" 10
run c2 "nb_outer=768" 10
run c2 "nb_outer=512" 10
run n8192 "" 10
run n8192 "nb_outer=512" 10
run n8192 "first_small_tiles=400" 10
run n32768 "" 3
run n32768 "nb_wide_rows=20000" 3
run n32768 "first_small_tiles=2500" 3
run n65536 "nb_wide_rows=0" 2
run n65536 "nb_wide_rows=20000" 2
date
} > $O/round20.log 2>&1
cat $O/round20.log | cut -c1-200
