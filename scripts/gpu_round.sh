#!/bin/bash
# final-state evidence: gpu tests, default bench, kernel stats, timeline, determinism stress
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
{
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default.json; cat gpurun_out/bench_default.json
timeout 300 python scripts/stress_determinism.py | tail -5
for w in n512 n1024 n2048 n4096 n8192 n32768; do S=""; [ $w = n32768 ] && S="--steps 3 --warmup 1"; bash scripts/bench_variants.sh "--workload $w $S" | tail -1 | cut -c1-100; done
cd /tmp
rm -rf /tmp/p0; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p0 -o st -- python $R/bench.py --no-cpu-baseline 2>&1 | grep '"metric"' | cut -c1-200
python $R/scripts/prof_top.py $(find /tmp/p0 -name "*.db" | head -1) 14 | tee $R/gpurun_out/kernel_stats.txt
rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof -o tl -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile 2>&1 | grep '"metric"' | cut -c1-120
python $R/scripts/timeline.py $(find /tmp/prof -name "*.db" | head -1) $R/gpurun_out/timeline_c2.csv 2000 | tail -1
python $R/scripts/timeline_panels.py $R/gpurun_out/timeline_c2.csv | tee $R/gpurun_out/panels.txt
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
