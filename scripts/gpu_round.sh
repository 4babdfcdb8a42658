#!/bin/bash
# round 3, batch 30: forward streaming solve with four staggered polls in flight on the critical hand-off
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b30
mkdir -p $O
B="--no-cpu-baseline"
sec() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:12], round(d['value'],4), round(d['ms_per_step'],3))
for r in d.get('roofline_secondary', [])[1:]: print('   ', r['kernel'][:40], round(r['achieved'],1), r['unit'], 'ms', round(r.get('ms'),4))"; }
{
date
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gp.py -m gpu -x -q -k "trsv or solve or logp or log_prob" 2>&1 | tail -2
for rep in 1 2; do
for wl in n4096 c2 n65536; do
timeout 300 python bench.py --workload $wl --steps 2 --warmup 1 $B 2>/dev/null | tail -1 | sec
done
done
date
} > $O/log.txt 2>&1
cat $O/log.txt
