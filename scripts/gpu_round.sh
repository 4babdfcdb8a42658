#!/bin/bash
# final-state evidence: gpu tests, default bench, kernel stats, PMC traffic passes, timeline, distributed driver at world 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
{
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default.json; cat gpurun_out/bench_default.json
timeout 300 python bench.py --distributed --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_distributed_w1.json; cat gpurun_out/bench_distributed_w1.json
cd /tmp
rm -rf /tmp/p0; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p0 -o st -- python $R/bench.py --no-cpu-baseline 2>&1 | grep '"metric"' | cut -c1-200
python $R/scripts/prof_top.py $(find /tmp/p0 -name "*.db" | head -1) 14 | tee $R/gpurun_out/kernel_stats.txt
for C in FETCH_SIZE WRITE_SIZE; do
rm -rf /tmp/p1; timeout 600 rocprofv3 --pmc $C --kernel-trace -d /tmp/p1 -o pm -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile 2>&1 | grep '"metric"' | cut -c1-100
python $R/scripts/pmc_summary.py $(find /tmp/p1 -name "*.db" | head -1) $C | tee $R/gpurun_out/pmc_$C.txt
done
rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof -o tl -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile 2>&1 | grep '"metric"' | cut -c1-120
python $R/scripts/timeline.py $(find /tmp/prof -name "*.db" | head -1) $R/gpurun_out/timeline_c2.csv 2000 | tail -1
python $R/scripts/timeline_panels.py $R/gpurun_out/timeline_c2.csv | tee $R/gpurun_out/panels.txt
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
