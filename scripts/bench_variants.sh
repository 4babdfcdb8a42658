#!/bin/bash
# bench.py at several tunings; one JSON line each -> gpurun_out/variants.log
mkdir -p gpurun_out
for args in "$@"; do
  echo "### $args" >> gpurun_out/variants.log
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $args 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    r=d.get('roofline') or {}
    print(json.dumps({'evals_s':round(d['value'],2),'ms':round(d['ms_per_step'],2),'syrk_TF':round(r.get('achieved',0),2),'chol_TF':round(d.get('cholesky_tflops',0),2),'stage':d.get('stage_ms')}))
" >> gpurun_out/variants.log
done
cat gpurun_out/variants.log
