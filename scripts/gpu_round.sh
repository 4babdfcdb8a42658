#!/bin/bash
# round 3, batch 13: pytree inputs on the device path; PMC traffic of the trailing update re-stamped for the final gemm.hip
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/b13
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary"
{
date
timeout 600 python -m pytest tests/test_gpu_gp.py tests/test_gpu_kernels.py -m gpu -x -q -k "pytree or beyond or gemm or potrf_vs or fp32 or host_evaluated" 2>&1 | tail -4
for cn in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --pmc $cn --kernel-trace -d $O/pmc_$cn -o bench -- python $R/bench.py --steps 2 --warmup 1 $B --no-profile > /dev/null 2>&1
echo "-- c2 $cn"; python scripts/pmc_summary.py $(ls $O/pmc_$cn/*.db | head -1) $cn | head -4
done
python scripts/pmc_to_bench.py $(ls $O/pmc_FETCH_SIZE/*.db | head -1) $(ls $O/pmc_WRITE_SIZE/*.db | head -1) profiles/r03_h_final_evidence.md | cut -c1-300
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:700])"
rm -rf $O/pmc_*
date
} > $O/log.txt 2>&1
cat $O/log.txt | cut -c1-700
