#!/bin/bash
# round 2, session 2, batch 18: bench lines of every size at the final defaults + determinism stress
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
mkdir -p $O
B="--no-cpu-baseline --no-secondary"
{
date
for w in c1 n2048 n4096 n8192 c2 n32768 n65536 ref2000 ref10000 ref20000; do timeout 400 python bench.py $B --workload $w --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/final_$w.json; cut -c1-200 $O/final_$w.json; done
timeout 300 python bench.py --distributed --workload c2 --steps 10 --warmup 3 $B 2>/dev/null | tail -1 > $O/final_dist_c2.json; cut -c1-200 $O/final_dist_c2.json
echo "== determinism stress"; date
timeout 400 python scripts/stress_determinism.py 2>&1 | tail -6
timeout 100 python scripts/stress_nan.py 3000 8000 lookahead=0 2>&1 | tail -1
date
} > $O/round18.log 2>&1
tail -30 $O/round18.log | cut -c1-220
