#!/bin/bash
# round 2, session 2, batch 9: the last partial round of a trailing update on small tiles
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/s2b9.log
: > $L
run() { # workload options
  echo "# $1 $2" >> $L
  TGP_HIP_OPTIONS="$2" timeout 600 python bench.py --workload $1 --steps $3 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    r=d.get('roofline') or {}
    print(json.dumps({'evals_s':round(d['value'],3),'ms':round(d['ms_per_step'],3),'syrk_TF':round(r.get('achieved',0),2),'chol_TF':round(d.get('cholesky_tflops',0),2)}))
" >> $L
}
echo "== pytest -m gpu" >> $L; date >> $L
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 >> $L
date >> $L
for t in 0 192 448; do
  run c2 "tail_small=$t" 8
  run n8192 "tail_small=$t" 8
  run n32768 "tail_small=$t" 4
done
run n65536 "tail_small=0" 2
run n65536 "tail_small=448" 2
run c2 "tail_small=448,chain_reserve=0" 8
echo "== gemm alone lower (tail 448)" >> $L
timeout 200 python scripts/gemm_bench.py f64 16384 2>&1 | grep "lower=1" >> $L
date >> $L
tail -40 $L
