#!/bin/bash
# kernel stats + panel timeline of the final commit
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
{
rm -rf /tmp/p0; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p0 -o st -- python $R/bench.py --no-cpu-baseline 2>&1 | grep '"metric"' | cut -c1-400
python $R/scripts/prof_top.py $(find /tmp/p0 -name "*.db" | head -1) 14 | tee $R/gpurun_out/kernel_stats.txt
python $R/scripts/timeline.py $(find /tmp/p0 -name "*.db" | head -1) $R/gpurun_out/timeline_c2.csv 2000 | tail -1
python $R/scripts/timeline_panels.py $R/gpurun_out/timeline_c2.csv | tee $R/gpurun_out/panels.txt
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
