#!/bin/bash
# round 3, batch 25: last share of the block-column update + first potf2 on the chain's stream (option gate_on_chain)
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b25
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary"
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['config']['workload'][:14], round(d['value'],3), round(d['ms_per_step'],3), 'update TF', round(r['achieved'] or 0,2))"; }
{
date
TGP_HIP_OPTIONS=gate_on_chain=1 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gp.py -m gpu -x -q -k "potrf or logp or log_prob or config2 or n32768 or determin" 2>&1 | tail -2
for rep in 1 2; do
for g in 0 1; do
echo "== gate_on_chain=$g"
for wl in n4096 n8192 c2 n32768; do
TGP_HIP_OPTIONS=gate_on_chain=$g timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 $B 2>/dev/null | tail -1 | line
done
done
done
for g in 0 1; do
TGP_HIP_OPTIONS=gate_on_chain=$g timeout 300 python bench.py --workload n65536 --steps 3 --warmup 1 $B 2>/dev/null | tail -1 | line
done
timeout 300 rocprofv3 --kernel-trace -d $O/kt -o g -- env TGP_HIP_OPTIONS=gate_on_chain=1 python bench.py --steps 3 --warmup 2 $B --no-profile > /dev/null 2>&1
python scripts/timeline.py $(ls $O/kt/*.db | head -1) $O/tl_c2_goc.csv 900 | tail -1
gzip -f $O/tl_c2_goc.csv; rm -rf $O/kt
date
} > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-200
