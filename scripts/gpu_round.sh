#!/bin/bash
# round 3, batch 6: fp32 trailing-update kernel rebuilt (32x32x2 MFMA, LDS-direct 3-stage) -- full parity suite, fp32 sizes,
# block-column driver at world size 1 after the reserve fix
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b06
mkdir -p $O
export TMPDIR=/tmp
{
date
timeout 1200 python -m pytest tests -m gpu -x -q -k "not n131072 and not n262144" 2>&1 | tail -12
date
} > $O/pytest.log 2>&1
B="--no-cpu-baseline --no-secondary"
{
date
for w in n16384f32 n65536f32 n131072f32; do timeout 400 python bench.py --workload $w --steps 2 --warmup 1 $B 2>/dev/null | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d.get('roofline') or {}
    print(json.dumps({'n':d['config']['n'],'ms':round(d['ms_per_step'],2),'chol_TF':round(d.get('cholesky_tflops',0),2),'update_TF':round(r.get('achieved',0),2),'frac':round(r.get('frac',0),3),'avg_launch_ms':r.get('avg_launch_ms')}))
"; done
for w in c2 n65536; do timeout 300 python bench.py --distributed --workload $w --steps 3 --warmup 1 $B 2>/dev/null | tail -1 | cut -c1-330; done
date
} > $O/bench.log 2>&1
cat $O/pytest.log; cat $O/bench.log
