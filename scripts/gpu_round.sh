#!/bin/bash
# final check of the committed state: full GPU suite, smoke, default bench, size table, determinism
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
{
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default.json; cat gpurun_out/bench_default.json
timeout 300 python scripts/stress_determinism.py | tail -5
for w in n512 n1024 n2048 n4096 n8192; do bash scripts/bench_variants.sh "--workload $w --steps 20 --warmup 5" | tail -1 | cut -c1-100; done
timeout 60 ./scripts/probe_potf2 | tail -9
} > $R/gpurun_out/round.log 2>&1
cat $R/gpurun_out/round.log
