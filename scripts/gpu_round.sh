#!/bin/bash
# round 2, batch 4: CU holders again, with EVERY tile kernel padded (the 64x64-tile workgroups were the squatters)
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
echo "== pytest -m gpu (defaults)"; date
timeout 1200 python -m pytest tests -m gpu -q -x $DESEL 2>&1 | tail -6
echo "== reserve_cus sweep"; date
for r in 0 2 4 8 16 32; do
  TGP_HIP_OPTIONS="reserve_cus=$r" timeout 120 python bench.py $B --steps 10 --warmup 3 2>/dev/null | tail -1 | line "c2 [reserve_cus=$r]"
done
for w in n4096 n8192 n32768; do for r in 0 8; do
  TGP_HIP_OPTIONS="reserve_cus=$r" timeout 120 python bench.py $B --workload $w --steps 5 --warmup 2 2>/dev/null | tail -1 | line "$w [reserve_cus=$r]"
done; done
TGP_HIP_OPTIONS="reserve_cus=8" timeout 200 python bench.py $B --workload n65536 --steps 1 --warmup 1 2>/dev/null | tail -1 | line "n65536 [reserve_cus=8]"
echo "== parity with holders parked"; date
TGP_HIP_OPTIONS="reserve_cus=8" timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gp.py -m gpu -q -x $DESEL -k "potrf or ragged or config2 or config1 or determin or mid_sizes or golden" 2>&1 | tail -4
echo "== kernel stats with holders"; date
cd /tmp
TGP_HIP_OPTIONS="reserve_cus=8" timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_c2_hold -o bench -- python $R/bench.py --steps 2 --warmup 1 $B > /dev/null 2>&1
cd $R
python scripts/prof_top.py $(ls $O/prof_c2_hold/*.db | head -1) 12
python scripts/timeline.py $(ls $O/prof_c2_hold/*.db | head -1) /tmp/tl_hold.csv 2500 > /dev/null; python scripts/timeline_panels.py /tmp/tl_hold.csv | tail -16
date
} > $O/round.log 2>&1
tail -80 $O/round.log
