#!/bin/bash
# round 2, session 2, batch 2: the fused panel step (one launch per 128-column block) -- parity,
# determinism, bench lines with the option on and off
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/s2b2.log
: > $L
echo "== quick check: N = 4096 factor through the fused step" >> $L; date >> $L
timeout 120 python bench.py --workload n4096 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -2 | cut -c1-300 >> $L
echo "== pytest -m gpu" >> $L; date >> $L
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 >> $L
echo "== bench lines, fused_step = 1 / 0" >> $L; date >> $L
for w in n1024 n2048 n4096 n8192 c2 n32768; do
  for f in 1 0; do
    echo "# $w fused_step=$f" >> $L
    TGP_HIP_OPTIONS="fused_step=$f" timeout 600 python bench.py --workload $w --steps 8 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    r=d.get('roofline') or {}
    print(json.dumps({'evals_s':round(d['value'],3),'ms':round(d['ms_per_step'],3),'syrk_TF':round(r.get('achieved',0),2),'chol_TF':round(d.get('cholesky_tflops',0),2)}))
" >> $L
  done
done
echo "== determinism stress" >> $L; date >> $L
timeout 300 python scripts/stress_determinism.py 2>&1 | tail -8 >> $L
date >> $L
tail -70 $L
