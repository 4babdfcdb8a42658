#!/bin/bash
# round 2, batch 14: straight-line evaluators (kmat_fast_kernel, kmat_gemv_fast_kernel)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
B="--no-cpu-baseline"
{
echo "== pytest -m gpu"; date
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error" | head
echo "== bench c2 with secondary rooflines"; date
timeout 600 python bench.py $B 2>/dev/null | tail -1 > $O/bench_c2.json
python -c "
import json; d=json.load(open('$O/bench_c2.json')); print(d['value'], d['ms_per_step'], d.get('stage_ms')); print(json.dumps(d.get('roofline_secondary'))[:900])"
echo "== n65536 / c3 assembly stage"; date
for w in n65536 c3; do timeout 300 python bench.py $B --no-secondary --workload $w --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$w', d['value'], d['ms_per_step'], d.get('stage_ms'))"; done
echo "== adjacent paths"; date
timeout 200 python scripts/time_paths.py 16384 4096
echo "== kernel times"; date
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_kmat -o bench -- python $R/scripts/time_paths.py 16384 4096 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_kmat65 -o bench -- python $R/bench.py --workload n65536 --steps 1 --warmup 0 $B --no-secondary > /dev/null 2>&1
cd $R
python scripts/prof_top.py $(ls $O/prof_kmat/*.db | head -1) 40 | grep -E "kmat|kgrad|kdiag"
python scripts/prof_top.py $(ls $O/prof_kmat65/*.db | head -1) 40 | grep -E "kmat"
date
} > $O/round.log 2>&1
tail -150 $O/round.log
