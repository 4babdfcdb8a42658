#!/bin/bash
# round 3, batch 32: forward streaming solve with two workgroups per block row (each streams every second tile)
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b32
mkdir -p $O
B="--no-cpu-baseline"
sec() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:12], round(d['value'],4), round(d['ms_per_step'],3))
for r in d.get('roofline_secondary', [])[1:]: print('   ', r['kernel'][:40], round(r['achieved'],1), r['unit'], 'frac', round(r['frac'],3), 'ms', round(r.get('ms'),4))"; }
{
date
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gp.py tests/test_gpu_grad.py -m gpu -x -q -k "not full_size and not stress" 2>&1 | tail -3
for wl in n1024 n4096 c2 n32768 n65536 n65536f32; do
timeout 300 python bench.py --workload $wl --steps 2 --warmup 1 $B 2>/dev/null | tail -1 | sec
done
timeout 300 python scripts/time_paths.py 16384 4096 2>&1 | head -4
date
} > $O/log.txt 2>&1
cat $O/log.txt
