#!/bin/bash
# round 2, batch 5: device-scope vs system-scope release of the inter-stream events
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
DESEL=""
python -c "import numpy as np,sys; sys.exit(0 if 'c3_n65536__logp' in np.load('tests/golden/large.npz').files else 1)" 2>/dev/null || DESEL="--deselect tests/test_gpu_gp.py::test_config3_n65536_full_size"
B="--no-cpu-baseline --no-secondary"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d.get('roofline') or {}
print('$1', 'evals/s %.2f ms %.2f  update %.1f TF (%.3f)  potrf %.2f ms' % (d['value'], d['ms_per_step'], r.get('achieved',0), r.get('frac',0), (d.get('stage_ms') or {}).get('potrf',0)))"; }
{
echo "== pytest -m gpu (device-scope events = new default)"; date
timeout 1200 python -m pytest tests -m gpu -q -x $DESEL 2>&1 | tail -6
echo "== event scope A/B"; date
for rep in 1 2; do for sc in device system; do
  TGP_EVENT_SCOPE=$sc timeout 120 python bench.py $B --steps 10 --warmup 3 2>/dev/null | tail -1 | line "c2 [events=$sc]"
done; done
for w in n2048 n4096 n8192 n32768; do for sc in device system; do
  TGP_EVENT_SCOPE=$sc timeout 120 python bench.py $B --workload $w --steps 8 --warmup 3 2>/dev/null | tail -1 | line "$w [events=$sc]"
done; done
for sc in device system; do
TGP_EVENT_SCOPE=$sc timeout 300 python bench.py --distributed --workload c2 --steps 10 --warmup 3 2>/dev/null | tail -1 | line "dist c2 [events=$sc]"
done
echo "== determinism under device-scope events"; date
timeout 300 python scripts/stress_determinism.py 2>&1 | tail -4
date
} > $O/round.log 2>&1
tail -60 $O/round.log
