#!/bin/bash
# round 3, batch 19: the negation out of the k-loops (f64: the MFMA's NEG field; f32: -C in, -acc out): parity + rates
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b19
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary"
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['config']['workload'][:14], d['value'], d['ms_per_step'], 'update TF', round(r['achieved'],2), 'frac', round(r['frac'],4))"; }
{
date
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gp.py tests/test_gpu_grad.py -m gpu -x -q -k "not full_size and not stress" 2>&1 | tail -4
for rep in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 $B 2>/dev/null | tail -1 | line
done
timeout 300 python bench.py --workload n4096 --steps 20 --warmup 5 $B 2>/dev/null | tail -1 | line
timeout 300 python bench.py --workload n65536 --steps 3 --warmup 1 $B 2>/dev/null | tail -1 | line
timeout 300 python bench.py --workload n65536f32 --steps 3 --warmup 1 $B 2>/dev/null | tail -1 | line
timeout 300 python bench.py --workload n131072f32 --steps 2 --warmup 1 $B 2>/dev/null | tail -1 | line
timeout 300 python scripts/grad_profile.py 16384 4
date
} > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-300
