#!/bin/bash
# round 2, session 2, batch 7: split gate + rows fetched before the flag (fused chain), against the unfused chain
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/s2b7.log
: > $L
run() { # workload options
  echo "# $1 $2" >> $L
  TGP_HIP_OPTIONS="$2" timeout 600 python bench.py --workload $1 --steps 8 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    r=d.get('roofline') or {}
    print(json.dumps({'evals_s':round(d['value'],3),'ms':round(d['ms_per_step'],3),'syrk_TF':round(r.get('achieved',0),2),'chol_TF':round(d.get('cholesky_tflops',0),2)}))
" >> $L
}
echo "== pytest -m gpu" >> $L; date >> $L
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 >> $L
date >> $L
for w in c2 n8192 n4096 n2048 n32768; do
  run $w "fused_step=0,chain_reserve=0"
  run $w "fused_step=0,chain_reserve=128"
  run $w "fused_step=1,gate_split=0,chain_reserve=0"
  run $w "fused_step=1,gate_split=1,chain_reserve=0"
  run $w "fused_step=1,gate_split=1,chain_reserve=128"
done
echo "== determinism stress (defaults)" >> $L; date >> $L
timeout 300 python scripts/stress_determinism.py 2>&1 | tail -6 >> $L
date >> $L
tail -80 $L
