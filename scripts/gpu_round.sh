#!/bin/bash
# round 3, batch 36: PMC traffic of the trailing update re-stamped on the final gemm.hip (comments changed its hash), default bench line
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b36
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary"
{
date
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm or potrf or fp32" 2>&1 | tail -1
for cn in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --pmc $cn --kernel-trace -d $O/pmc_$cn -o bench -- python bench.py --steps 2 --warmup 1 $B --no-profile > /dev/null 2>&1
echo "-- c2 $cn"; python scripts/pmc_summary.py $(ls $O/pmc_$cn/*.db | head -1) $cn | head -3
done
python scripts/pmc_to_bench.py $(ls $O/pmc_FETCH_SIZE/*.db | head -1) $(ls $O/pmc_WRITE_SIZE/*.db | head -1) profiles/r03_k_final_evidence.md | cut -c1-300
rm -rf $O/pmc_*
timeout 300 python bench.py --steps 10 --warmup 3 $B 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline']['frac'])"
date
} > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-300
